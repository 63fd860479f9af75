cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_scene_blocks_gpu.py tests/test_golden_fullsize_gpu.py -q > $OUT/t.log 2>&1; echo "rc=$?" >> $OUT/t.log; tail -15 $OUT/t.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err; python -c "
import json;j=json.load(open('$OUT/bench.json'));print(j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['kernel_ms'])"
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt" -- python "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/kt.log" 2>&1)
F=$(ls "$OUT"/kt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" "$OUT/bench_kernel_stats.csv"
F=$(ls "$OUT"/kt/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tools/trace_step.py "$F" > "$OUT/step_timeline.txt"
rm -rf "$OUT/kt"
cat $OUT/step_timeline.txt | tail -60
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/ktf" -- python "$ROOT/bench.py" --full-step --steps 4 --warmup 2 > "$ROOT/$OUT/ktf.log" 2>&1)
F=$(ls "$OUT"/ktf/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" "$OUT/full_step_kernel_stats.csv"
rm -rf "$OUT/ktf"
head -40 $OUT/full_step_kernel_stats.csv
python bench.py --full-step --batch 20 --steps 4 --warmup 2 2>/dev/null | tail -1 > $OUT/full_step_window20.json; head -c 600 $OUT/full_step_window20.json
