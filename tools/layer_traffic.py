"""HBM-side bytes per launch, per (kernel, grid), from two rocprofv3 counter passes over one command (--pmc FETCH_SIZE
and --pmc WRITE_SIZE, each with --kernel-trace --output-format csv), as JSON on stdout.

    python tools/layer_traffic.py <fetch dir> <write dir> [substring of the kernel names to keep]

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 bytes per lane)
streaming reads as 64 bytes, which is how the kernels here read their tensors: it is doubled."""
import collections, csv, glob, json, re, sys


def fold(directory, counter, keep):
    per = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(directory + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(path)):
            if row['Counter_Name'] != counter or keep not in row['Kernel_Name']:
                continue
            name = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', row['Kernel_Name']))
            name = re.sub(r'\(.*$', '', name)
            entry = per[(name, int(row['Grid_Size']))]
            entry[0] += float(row['Counter_Value']) * 1024.0
            entry[1] += 1
    return per


fetch_dir, write_dir = sys.argv[1:3]
keep = sys.argv[3] if len(sys.argv) > 3 else 'sg::'
fetch, write = fold(fetch_dir, 'FETCH_SIZE', keep), fold(write_dir, 'WRITE_SIZE', keep)
out = []
for key in sorted(set(fetch) | set(write)):
    f, w = fetch.get(key, [0.0, 0]), write.get(key, [0.0, 0])
    out.append(dict(kernel=key[0], grid=key[1], launches=max(f[1], w[1]), fetch_bytes_per_launch=round(2.0 * f[0] / max(f[1], 1)),
                    write_bytes_per_launch=round(w[0] / max(w[1], 1))))
json.dump(dict(note='HBM-side bytes per launch: rocprofv3 FETCH_SIZE (x2: gfx950 tallies the 128-byte requests of wide reads as 64 bytes) '
                    'and WRITE_SIZE, separate passes of the same command', kernels=out), sys.stdout, indent=1)
