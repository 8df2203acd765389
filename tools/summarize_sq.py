"""Fold a rocprofv3 --pmc SQ_* pass (--kernel-trace --output-format csv) into per-(kernel, grid) figures:
    mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)   (SQ_BUSY_CYCLES sums over 32 shader engines' SQs)
    lds_busy    = SQ_ACTIVE_INST_LDS / (SQ_BUSY_CYCLES / 32 x 256 CUs)
    lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS
    valu / lds instructions and waves per launch
  python tools/summarize_sq.py <counter_collection.csv> [substring of kernel names to keep] > out.json"""
import collections, csv, json, re, sys

keep = sys.argv[2] if len(sys.argv) > 2 else 'sg::'
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    if keep not in row['Kernel_Name']:
        continue
    name = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', row['Kernel_Name']))
    key = (re.sub(r'\(.*$', '', name), int(row['Grid_Size']))
    acc[key][row['Counter_Name']] += float(row['Counter_Value'])
    launches[key].add(row['Dispatch_Id'])
out = []
for key in sorted(acc):
    c, n = acc[key], len(launches[key])
    cycles = c['SQ_BUSY_CYCLES'] / 32.0
    out.append(dict(kernel=key[0], grid=key[1], launches=n,
                    mfma_busy=round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cycles * 1024, 1), 3),
                    lds_busy=round(c['SQ_ACTIVE_INST_LDS'] / max(cycles * 256, 1), 3),
                    lds_conflict_frac=round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_ACTIVE_INST_LDS'], 1), 3),
                    valu_insts_per_launch=round(c['SQ_INSTS_VALU'] / n), lds_insts_per_launch=round(c['SQ_INSTS_LDS'] / n),
                    waves_per_launch=round(c['SQ_WAVES'] / n)))
json.dump(dict(note=__doc__.split('\n  python')[0], kernels=out), sys.stdout, indent=1)
