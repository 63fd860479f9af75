cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh gpurun_out/r04 r04_v1 2>&1 | tail -30
