cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$(pwd); OUT=gpurun_out/tl; mkdir -p $OUT
for v in 1 0; do
(cd /tmp && MSR3D_PACK_EARLY=$v rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/kt$v" -- python "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/kt$v.log" 2>&1)
F=$(ls "$OUT"/kt$v/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tools/trace_step.py "$F" > "$OUT/timeline_early$v.txt"
rm -rf $OUT/kt$v
done
tail -3 $OUT/timeline_early1.txt
