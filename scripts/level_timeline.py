#!/usr/bin/env python3
"""Kernel timeline of BFS levels from a rocprofv3 kernel trace CSV: three consecutive mid-build
levels kernel by kernel, then the whole level loop of the last build in the trace summed up (kernel
time, gaps between consecutive kernels, wall time per level)."""
import csv
import glob
import re
import sys

f = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*", "", k).split("::")[-1].strip()
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
commits = [i for i, r in enumerate(rows) if r[2] == "k_level_resolve"]  # (the last kernel of a level)
# builds are separated by the index kernels; take the last build's commits
last = [i for i in commits if i > max([j for j, r in enumerate(rows) if r[2] == "k_bounds"] or [0])]
mid = last[len(last) // 2]
i0, i1 = mid + 1, last[len(last) // 2 + 3]
t0 = rows[i0][0]
prev_end = rows[mid][1]
print("three mid-build levels (us from the first kernel; duration; gap to the previous kernel's end)")
for s, e, k in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:6.1f}  {k}")
    prev_end = max(prev_end, e)
a, b = last[0], last[-1]
kern = sum(e - s for s, e, _ in rows[a:b + 1])
wall = rows[b][1] - rows[a][0]
gaps = sum(max(0, rows[i + 1][0] - rows[i][1]) for i in range(a, b))
print(f"\nlevel loop of the last build: {len(last)} levels, {b - a + 1} kernels, wall {wall / 1e6:.2f} ms "
      f"({wall / 1e3 / len(last):.1f} us/level), kernels {kern / 1e6:.2f} ms, gaps {gaps / 1e6:.2f} ms "
      f"({gaps / 1e3 / (b - a):.2f} us per kernel boundary)")
by = {}
for s, e, k in rows[a:b + 1]:
    by.setdefault(k, [0, 0])
    by[k][0] += 1
    by[k][1] += e - s
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} {n:5d} launches  {t / 1e6:7.2f} ms  avg {t / 1e3 / n:6.1f} us")
pairs = {}
for i in range(a, b):
    g = max(0, rows[i + 1][0] - rows[i][1])
    key = rows[i][2] + " -> " + rows[i + 1][2]
    pairs.setdefault(key, [0, 0, 0])
    pairs[key][0] += 1
    pairs[key][1] += g
    pairs[key][2] = max(pairs[key][2], g)
print("\ngaps by kernel pair (count, total ms, max us):")
for k, (n, t, mx) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:52s} {n:5d}  {t / 1e6:7.3f} ms  max {mx / 1e3:8.1f} us")
