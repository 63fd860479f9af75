cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -m gpu -q 2>&1 > gpurun_out/full_suite.log; tail -3 gpurun_out/full_suite.log; grep -n "^FAILED" gpurun_out/full_suite.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh gpurun_out/r04v5 r04_v5 2>&1 | tail -3
