cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04u; mkdir -p $OUT; ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt" -- python "$ROOT/bench.py" --no-cpu-baseline --unfrozen --steps 10 --warmup 2 > "$ROOT/$OUT/kt.log" 2>&1)
F=$(ls "$OUT"/kt/*/*kernel_stats.csv 2>/dev/null | head -1); cp "$F" $OUT/unfrozen_kernel_stats.csv; rm -rf $OUT/kt
python - <<'P'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r04u/unfrozen_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:45]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'void ','',n)[:70]
    print(f"{n:70s} calls {r['Calls']:>5s} tot ms {float(r['TotalDurationNs'])/1e6:8.2f} avg us {float(r['AverageNs'])/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
P
