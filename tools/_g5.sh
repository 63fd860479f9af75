cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROOT=$(pwd); OUT=gpurun_out/fs; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/kt" -- python "$ROOT/bench.py" --full-step --llm-fp8 --steps 3 --warmup 2 --no-cpu-baseline > "$ROOT/$OUT/kt.log" 2>&1)
F=$(ls "$OUT"/kt/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tools/trace_step_agg.py "$F" > "$OUT/full_step_fp8_agg11.txt"
rm -rf $OUT/kt
head -14 $OUT/full_step_fp8_agg11.txt | cut -c1-150
