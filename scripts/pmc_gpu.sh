#!/bin/bash
# Run on the GPU box (inside gpurun): instruction-mix / stall counters of the map kernels.
# Usage: bash scripts/pmc_gpu.sh <tag> [workload]
set -uo pipefail
TAG="${1:-pmc}"
WL="${2:-c3}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp TRG_BENCH_FAST=1
i=0
SETS_LIMIT="${3:-5}"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  if [ "$i" -gt "$SETS_LIMIT" ]; then break; fi
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
  echo "pass $i ($set) rc=$?"
done
cd "$R"
python3 - "$OUT" <<'PY' > "$OUT/summary.md"
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*", "", k).split("::")[-1].strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
print("| kernel | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    print("| " + k + " | " + " | ".join(f"{agg[k].get(c, 0):.4g}" for c in names) + " |")
PY
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
cat "$OUT/summary.md" | head -20
