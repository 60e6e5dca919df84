#!/usr/bin/env python3
"""Print the kernel timeline of one mid-build BFS level from a rocprofv3 kernel trace CSV."""
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
idx = [i for i, r in enumerate(rows) if r[2] == "k_bfs_classify"]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = rows[i0][0]
prev_end = None
for s, e, k in rows[i0:i1 + 1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {gap:6.1f}  {k}")
    prev_end = max(prev_end or 0, e)
