cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sa_split_gpu.py tests/test_sa_fused_gpu.py tests/test_fullsize_gpu.py tests/test_golden_fullsize_gpu.py -q -x 2>&1 | tail -5
for t in 4 2; do MSR3D_SA3_TILE=$t python bench.py --no-cpu-baseline --time-all-kernels 2>/dev/null | tail -1 | python -c "
import json,sys;j=json.loads(sys.stdin.read());print('sa3 tile $t',round(j['value']),j['ms_per_step'],j['kernels_ms'])"; done
