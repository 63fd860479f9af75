#!/bin/bash
# Where the scene blocks' time goes: instruction mix and pipe-busy counters of every kernel of one bench
# step, counters in their own passes (kernel-trace only besides --pmc).
# usage (GPU box, repo root):  bash tools/pmc_blocks.sh gpurun_out/pmc_blocks
set -u
OUT=${1:-gpurun_out/pmc_blocks}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() {
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$ROOT/$OUT/$name.log" 2>&1)
}
run p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run p2 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run p3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p4 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p5 SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run p6 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run p7 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
run p8 FETCH_SIZE
run p9 WRITE_SIZE
python - "$ROOT/$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:44]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
with open(out + "/summary.txt", "w") as fo:
    fo.write("# per-launch means; SQ_* summed over all waves / SEs; GRBM_GUI_ACTIVE summed over the 8 XCDs\n")
    for k, cs in sorted(agg.items()):
        if not any(t in k for t in ("scene_block", "scene_attn", "wgrad", "split_pack", "sa2", "rows_linear", "anchor_front", "pos_embed")):
            continue
        fo.write(k + "\n")
        for c in names:
            if c in cs:
                fo.write(f"    {c:40s} {sum(cs[c]) / len(cs[c]):16.0f}   (n={len(cs[c])})\n")
print(open(out + "/summary.txt").read())
PY
