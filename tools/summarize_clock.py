"""Effective clock of each kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv pass:
    clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration   (the counter is summed over the 8 XCDs)
The chip lowers its clock under matrix-instruction load, so a kernel's share of the nominal 2.4 GHz peak is its
share of the peak at the clock it actually ran at, times clock / 2.4 GHz.  Dispatches shorter than 0.3 ms read high.
  python tools/summarize_clock.py <directory of the pass> [substring of kernel names to keep] > out.json"""
import collections, csv, glob, json, os, re, sys

root, keep = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'sg::')
find = lambda pat: (glob.glob(os.path.join(root, '*' + pat)) + glob.glob(os.path.join(root, '*', '*' + pat)))[0]
dur = {}
for row in csv.DictReader(open(find('kernel_trace.csv'))):
    dur[row['Dispatch_Id']] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for row in csv.DictReader(open(find('counter_collection.csv'))):
    if keep not in row['Kernel_Name'] or row['Counter_Name'] != 'GRBM_GUI_ACTIVE' or row['Dispatch_Id'] not in dur:
        continue
    name = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', row['Kernel_Name']))
    a = acc[re.sub(r'\(.*$', '', name)]
    a[0] += float(row['Counter_Value']); a[1] += dur[row['Dispatch_Id']]; a[2] += 1
out = [dict(kernel=k, launches=n, avg_ms=round(t / n * 1e3, 4), clock_ghz=round(c / 8 / t * 1e-9, 3))
       for k, (c, t, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]) if t > 0]
json.dump(dict(note=__doc__.split('\n  python')[0], kernels=out), sys.stdout, indent=1)
