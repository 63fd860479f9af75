cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -m gpu -q 2>&1 > gpurun_out/full_suite.log; tail -5 gpurun_out/full_suite.log; grep -n "^E " gpurun_out/full_suite.log | head -20
