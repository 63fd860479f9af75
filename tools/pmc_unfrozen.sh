#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, KiB) and matrix-pipe busy fraction of every kernel of the UNFROZEN-backbone
# step (`bench.py --unfrozen --batch 16`); counters in their own passes (kernel-trace only besides --pmc).
# usage (GPU box, repo root):  bash tools/pmc_unfrozen.sh gpurun_out/pmc_unf
set -u
OUT=${1:-gpurun_out/pmc_unf}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
run() {
  local name=$1; shift
  # (one derived memory counter per pass: FETCH_SIZE + WRITE_SIZE together exceed what one pass can collect, and the
  #  profiler then aborts without exiting -- hence also the timeout)
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- \
      python "$ROOT/bench.py" --unfrozen --batch 16 --steps 4 --warmup 2 --no-cpu-baseline > "$ROOT/$OUT/$name.log" 2>&1)
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run grbm GRBM_GUI_ACTIVE
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt" -- \
    python "$ROOT/bench.py" --unfrozen --batch 16 --steps 4 --warmup 2 --no-cpu-baseline > "$ROOT/$OUT/kt.log" 2>&1)
python - "$ROOT/$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n)[:52]
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
with open(out + "/summary.txt", "w") as fo:
    fo.write("# bench.py --unfrozen --batch 16: per-launch means.  FETCH / WRITE in MB (KiB counters x 1.024e-3; FETCH_SIZE\n")
    fo.write("# under-reports wide streams by up to 2x on gfx950: MI355X_MICROARCH.md), avg us from --kernel-trace --stats of the\n")
    fo.write("# same command, TB/s = (FETCH + WRITE) / avg us, busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)\n")
    rows = []
    for k, cs in agg.items():
        if "FETCH_SIZE" not in cs or k not in dur:
            continue
        mean = lambda v: sum(v) / len(v)
        f_, w_ = mean(cs["FETCH_SIZE"]) * 1.024e-3, mean(cs["WRITE_SIZE"]) * 1.024e-3
        us, calls = dur[k]
        busy = mean(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / (128 * mean(cs["GRBM_GUI_ACTIVE"])) if "GRBM_GUI_ACTIVE" in cs and "SQ_VALU_MFMA_BUSY_CYCLES" in cs else float("nan")
        rows.append((us * calls, k, calls, us, f_, w_, busy))
    for tot, k, calls, us, f_, w_, busy in sorted(rows, reverse=True)[:28]:
        fo.write(f"{k:54s} calls {calls:4d}  avg {us:8.1f} us  fetch {f_:8.1f} MB  write {w_:8.1f} MB  {(f_ + w_) / us:6.2f} TB/s  mfma busy {busy:.3f}\n")
print(open(out + "/summary.txt").read())
PY
