"""The last launches of a rocprofv3 kernel trace, one per line: gap before the launch (us), duration (us), kernel name -
what one iteration of a launch-bound configuration looks like.
Usage: python scripts/trace_tail.py <dir with *kernel_trace.csv> [number of launches, default 120] [skip from the end, default 0]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) - count - skip:len(rows) - skip]
prev = None
tot = gap_tot = 0.0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    name = re.sub(r'\(.*', '', r['Kernel_Name'])[:60]
    grid = r.get('Grid_Size', r.get('Grid_Size_X', ''))
    print(f'{gap:8.1f} {(e - s) / 1e3:9.1f}  {name}  [{grid}]')
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0.0)
    prev = e
print(f'kernels {tot / 1e3:.3f} ms, gaps {gap_tot / 1e3:.3f} ms over {len(rows)} launches')
