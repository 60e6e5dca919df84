#!/usr/bin/env python3
"""Condense a scripts/profile_gpu.sh output directory into a markdown summary (kernel stats from
--kernel-trace --stats, per-kernel FETCH_SIZE / WRITE_SIZE sums from the two --pmc passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    for tok in ("trg::(anonymous namespace)::", "void "):
        name = name.replace(tok, "")
    return name.split("(")[0][:60]


print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
st = find("trace", "*kernel_stats.csv")
if st:
    print("## kernel-trace --stats (bench.py --steps 2 --warmup 1)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    with open(st) as f:
        for r in csv.DictReader(f):
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                  f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | "
                  f"{float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()
traffic = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        print(f"## {ctr}: no counter file\n")
        continue
    tot = defaultdict(float)
    calls = defaultdict(int)
    with open(cc) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != ctr:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
    print(f"## --pmc {ctr} (bench.py --steps 1 --warmup 0); raw counter unit = KiB\n")
    print("| kernel | dispatches | sum (raw) | sum MB (raw*1024/1e6) | per dispatch KB |")
    print("|---|---|---|---|---|")
    for k in sorted(tot, key=lambda k: -tot[k]):
        print(f"| {k} | {calls[k]} | {tot[k]:.0f} | {tot[k] * 1024 / 1e6:.2f} | "
              f"{tot[k] * 1024 / 1e3 / max(1, calls[k]):.1f} |")
        traffic.setdefault(k, {})[ctr] = {"dispatches": calls[k], "raw_kib_total": tot[k]}
    print()
if traffic:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # exactly half of the bytes of a coalesced streaming read (calibrated here on k_bounds /
    # k_cell_count, whose reads are exactly 12 B/point), WRITE_SIZE reads bytes as they are.
    import json
    out_j = {}
    for k, v in traffic.items():
        f = v.get("FETCH_SIZE", {"dispatches": 0, "raw_kib_total": 0.0})
        w = v.get("WRITE_SIZE", {"dispatches": 0, "raw_kib_total": 0.0})
        n = max(1, f["dispatches"] or w["dispatches"])
        out_j[k] = {
            "dispatches": n,
            "fetch_bytes_per_dispatch_corrected": 2.0 * f["raw_kib_total"] * 1024 / n,
            "write_bytes_per_dispatch": w["raw_kib_total"] * 1024 / n,
        }
        out_j[k]["hbm_bytes_per_dispatch"] = (out_j[k]["fetch_bytes_per_dispatch_corrected"] +
                                              out_j[k]["write_bytes_per_dispatch"])
    # the record names the kernel sources it was measured on: bench.py quotes it only for these
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trg-planner_amd", "csrc")
    h = hashlib.sha256()
    for fsrc in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc")) +
                       [os.path.join(csrc, "trg_kernels.h"), os.path.join(csrc, "build.sh")]):
        h.update(os.path.basename(fsrc).encode())
        h.update(open(fsrc, "rb").read())
    out_j["kernel_source_sha256"] = h.hexdigest()
    with open(os.path.join(out, "traffic.json"), "w") as fjs:
        json.dump(out_j, fjs, indent=1, sort_keys=True)
