cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for m in 1 0; do MSR3D_MERGE_LAUNCHES=$m python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;j=json.loads(sys.stdin.read());print('merge $m',round(j['value']),j['ms_per_step'],j['ms_per_step_percentiles']['p50'])"; done; done
