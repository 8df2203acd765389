"""Per-kernel statistics from a rocprofv3 database (rocpd SQLite output of `rocprofv3 --kernel-trace`):
calls, total / average / minimum duration grouped by kernel symbol and launch grid, as CSV on stdout.
    python tools/rocpd_stats.py results.db [min_avg_us]
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
groups = {}
for name, s, e, gx, wx, lds, vg, ag in db.execute(
        'select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from kernels order by start'):
    short = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', name))
    short = re.sub(r'\(.*', '', short.replace('at::native::', ''))[:100]
    key = (short, gx, wx, lds, vg, ag)
    g = groups.setdefault(key, [0, 0.0, 1e30, len(groups)])
    g[0] += 1
    g[1] += (e - s) / 1e3
    g[2] = min(g[2], (e - s) / 1e3)
total = sum(g[1] for g in groups.values())
print('kernel,grid,workgroup,lds_bytes,vgprs,agprs,calls,total_us,avg_us,min_us,share')
for key, g in sorted(groups.items(), key=lambda kv: kv[1][3]):
    if g[1] / g[0] < floor:
        continue
    print(f'"{key[0]}",{key[1]},{key[2]},{key[3]},{key[4]},{key[5]},{g[0]},{g[1]:.1f},{g[1] / g[0]:.1f},{g[2]:.1f},{g[1] / total:.4f}')
