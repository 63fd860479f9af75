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
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
run tcc TCC_HIT TCC_MISS TCC_REQ
python - "$ROOT/$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in agg.items():
        if not any(s in k for s in ("sa1_kernel", "sa2_kernel", "sa3_kernel", "fps_kernel")):
            continue
        line = k + ": " + ", ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items()))
        print(line); fo.write(line + "\n")
PY
