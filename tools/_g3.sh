cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_scene_blocks_gpu.py tests/test_golden_fullsize_gpu.py tests/test_train_step_gpu.py -q -x 2>&1 | tail -4
for i in 1 2 3; do timeout 300 python -m pytest tests/test_scene_blocks_gpu.py -q -x -k wgrad 2>&1 | tail -1; done
for h in 1 0; do MSR3D_WGRAD_HALVES=$h python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('halves=$h',j['value'],j['ms_per_step'],j['ms_per_step_percentiles']['p50'])"; done
