#!/bin/bash
# PMC passes over the frozen-encoder micro-benchmark (tools/bench_sa.py); counters are
# collected in their own runs (no --stats / trace domains besides kernel-trace).
# usage (on the GPU box, from the repo root):  bash tools/pmc_sa.sh gpurun_out/pmc_sa
set -u
OUT=${1:-gpurun_out/pmc_sa}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() {  # name, counters...
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python "$ROOT/tools/bench_sa.py" --iters 5 > "$ROOT/$OUT/$name.log" 2>&1)
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq3 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
python - "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("sa1_kernel", "sa1_split_kernel", "sa1_rows_kernel", "sa1_plan_kernel", "sa2_kernel", "sa2_split_kernel", "sa2_rows_kernel",
                    "sa2_plan_kernel", "sa3_kernel", "sa3_split_kernel", "sa3_split4_kernel", "sa3_tiles_kernel", "fps_kernel", "fps_query_kernel", "fps_query_plan_kernel",
                    "ball_query_kernel"):
            if key in r["Kernel_Name"]:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    fo.write("# rocprofv3 --pmc passes over tools/bench_sa.py (960 objects per launch), per-launch means\n")
    fo.write("# FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE under-reports wide streams by 2x on gfx950)\n")
    for k, cs in sorted(agg.items()):
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"    {c:32s} {sum(v) / len(v):.6g}\n")
print(open(out + "/summary.txt").read())
PY
