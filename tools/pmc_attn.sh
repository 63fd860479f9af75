#!/bin/bash
# Matrix-core busy fraction of the attention kernels per operand precision
# (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), the figure DESIGN.md quotes per kernel).
# Counters are collected in their own runs (kernel-trace only besides --pmc).
# usage (on the GPU box, from the repo root):  bash tools/pmc_attn.sh gpurun_out/pmc_attn
set -u
OUT=${1:-gpurun_out/pmc_attn}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() {
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python "$ROOT/tools/bench_attn.py" --iters 20 > "$ROOT/$OUT/$name.log" 2>&1)
}
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS
run grbm GRBM_GUI_ACTIVE
python - "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_fwd_kernel" in n or "attn_bwd_kernel" in n:
            key = n[n.index("attn_"):][:40] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", ""))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in sorted(agg.items()):
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"    {c:32s} {sum(v) / len(v):.6g}\n")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            m = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
            g = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles = g / 8, 1024 SIMDs
            fo.write(f"    mfma_busy / (1024 SIMDs x cycles)   {m / (128 * g):.4f}\n")
print(open(out + "/summary.txt").read())
PY
