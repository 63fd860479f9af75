cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lora_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/prof_lora_grad.py 2>&1 | grep pair
python bench.py --full-step --llm-fp8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
ROOT=$(pwd); OUT=gpurun_out/fs; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/kt" -- python "$ROOT/bench.py" --full-step --llm-fp8 --steps 3 --warmup 2 --no-cpu-baseline > "$ROOT/$OUT/kt.log" 2>&1)
F=$(ls "$OUT"/kt/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tools/trace_step_agg.py "$F" > "$OUT/full_step_fp8_agg3.txt"
F=$(ls "$OUT"/kt/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python - "$F" > "$OUT/layer_seq.txt" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
# print a window in the middle of forward and one in the middle of backward: 90 kernels each
n = len(seg)
for lo in (n // 6, 2 * n // 3):
    prev = None
    for r in seg[lo:lo + 90]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); name = re.sub(r'^void ', '', name)[:70]
        print(f"{(e-s)/1e3:8.1f} gap {((s-prev)/1e3 if prev else 0):6.1f}  {name}")
        prev = e
    print('-----')
PY
rm -rf $OUT/kt
head -24 $OUT/full_step_fp8_agg3.txt | cut -c1-150
