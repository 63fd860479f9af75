#!/bin/bash
# PMC passes over the token-GEMM micro-benchmark (tools/bench_gemm.py); counters are collected
# in their own runs (kernel-trace only besides --pmc).
# usage (on the GPU box, from the repo root):  bash tools/pmc_gemm.sh gpurun_out/pmc_gemm
set -u
OUT=${1:-gpurun_out/pmc_gemm}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() {
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python "$ROOT/tools/bench_gemm.py" > "$ROOT/$OUT/$name.log" 2>&1)
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run grbm GRBM_GUI_ACTIVE
python - "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32_kernel" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("gemm_f32_kernel")[1][:22], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in sorted(agg.items()):
        fo.write(" ".join(map(str, k)) + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"    {c:32s} {sum(v) / len(v):.6g}\n")
print(open(out + "/summary.txt").read())
PY
