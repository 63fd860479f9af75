cd $GRAFT_REPO_ROOT
for a in 0 1 2 3; do echo "abl $a"; MSR3D_FA_ABL=$a python tools/prof_attn.py 2>&1 | grep forward; done
