"""GPU busy fraction and the gaps between launches from a rocprofv3 kernel trace (steady-state part of a run).
Usage: python scripts/trace_gaps.py <dir with *kernel_trace.csv> [fraction of the run to skip at the start]"""
import csv
import glob
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
f = glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = int(len(rows) * skip)
t0, t1 = int(rows[first]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
busy, prev, gaps = 0, None, []
for r in rows[first:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if prev is not None:
        gaps.append(((s - prev) / 1e3, r['Kernel_Name'][:48]))
    prev = max(prev or e, e)
print(f'span {(t1 - t0) / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms = {busy / (t1 - t0):.3f} of it, {len(rows) - first} launches')
gaps.sort(reverse=True)
print('largest gaps before (us):', [(round(g, 1), k) for g, k in gaps[:10]])
print(f'gaps > 30 us: {sum(g for g, _ in gaps if g > 30) / 1e3:.2f} ms in {sum(1 for g, _ in gaps if g > 30)}; '
      f'gaps <= 30 us: {sum(g for g, _ in gaps if 0 < g <= 30) / 1e3:.2f} ms in {sum(1 for g, _ in gaps if 0 < g <= 30)}')
