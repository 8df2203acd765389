"""Fold two rocprofv3 counter runs of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) into
profiles/<name>: HBM-side bytes per launch for every kernel of this library.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py ...
    python tools/summarize_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_traffic.json

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes for wide
(16 bytes per lane) streaming reads, which is how every kernel here reads its tensors: it is doubled.
"""
import collections, csv, glob, json, re, sys


def fold(directory, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(directory + '/*counter_collection.csv'):
        for row in csv.DictReader(open(path)):
            if row['Counter_Name'] != counter or 'sg::' not in row['Kernel_Name']:
                continue
            name = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', row['Kernel_Name']))
            name = re.sub(r'^sg::', '', re.sub(r'\(.*$', '', name))
            entry = per[name]
            entry[0] += float(row['Counter_Value']) * 1024.0
            entry[1] += 1
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = fold(fetch_dir, 'FETCH_SIZE'), fold(write_dir, 'WRITE_SIZE')
    result = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
        result[name] = dict(launches=max(f[1], w[1]),
                            fetch_bytes_per_launch=round(2.0 * f[0] / max(f[1], 1)),
                            write_bytes_per_launch=round(w[0] / max(w[1], 1)))
    json.dump(dict(note='HBM-side bytes per launch from rocprofv3 FETCH_SIZE (x2, gfx950 correction for wide reads) and '
                        'WRITE_SIZE, separate passes of bench.py --config <workload> --steps 4 --warmup 1 --also none (tools/prof_bench.sh)', kernels=result), open(out, 'w'), indent=1)
    for name, v in sorted(result.items(), key=lambda kv: -kv[1]['launches'] * (kv[1]['fetch_bytes_per_launch'] + kv[1]['write_bytes_per_launch'])):
        print(f"{v['launches']:6d}  {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB read  {v['write_bytes_per_launch'] / 1e6:9.1f} MB written  {name}")


if __name__ == '__main__':
    main()
