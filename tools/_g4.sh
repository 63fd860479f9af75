cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh gpurun_out/r04v2 r04_v2 2>&1 | tail -5
