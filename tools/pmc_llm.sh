#!/bin/bash
# Matrix-pipe busy fraction of every kernel of a LoRA-Llama decoder layer forward + backward (bench.py --llm-layer --llm-fp8) (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
# kernel cycles)); counters in their own passes (kernel-trace only besides --pmc).
# usage (GPU box, repo root):  bash tools/pmc_llm.sh gpurun_out/pmc_llm
set -u
OUT=${1:-gpurun_out/pmc_llm}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() {
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python "$ROOT/bench.py" --llm-layer --llm-fp8 --steps 6 --warmup 3 --no-cpu-baseline > "$ROOT/$OUT/$name.log" 2>&1)
}
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_BUSY_CU_CYCLES
run grbm GRBM_GUI_ACTIVE
python - "$ROOT/$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:60]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    fo.write("# per-launch means over a short --llm-layer --llm-fp8 run; busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)\n")
    fo.write("# (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs)\n")
    for k, cs in sorted(agg.items()):
        if "GRBM_GUI_ACTIVE" not in cs or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs:
            continue
        m = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
        g = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        i = sum(cs["SQ_INSTS_MFMA"]) / len(cs["SQ_INSTS_MFMA"])
        fo.write(f"{k:62s} launches {len(cs['GRBM_GUI_ACTIVE']):4d}  cycles/XCD {g / 8:10.0f}  mfma insts {i:12.0f}  busy {m / (128 * g):.4f}\n")
print(open(out + "/summary.txt").read())
PY
