#!/bin/bash
# Which GPU test files make some library compile device code at RUN TIME (through comgr, cached under ~/.cache/comgr)?
# Round 4's driver run aborted inside a GPU backward whose only foreign code was a vendor library that does exactly that.
# Prints the cache's growth per test file (each file in its own pytest process, cache kept between files).
rm -rf ~/.cache/comgr ~/.cache/miopen ~/.config/miopen
prev=0
for f in tests/test_*_gpu.py; do
  MSR3D_GPU_INPROC=1 python -m pytest "$f" -x -q -m gpu -p no:cacheprovider > /tmp/cu.log 2>&1
  rc=$?
  now=$(du -sk ~/.cache/comgr 2>/dev/null | cut -f1); now=${now:-0}
  n=$(ls ~/.cache/comgr 2>/dev/null | wc -l)
  echo "$f rc=$rc comgr_kb=$now (+$((now - prev))) entries=$n miopen=$(du -sk ~/.cache/miopen 2>/dev/null | cut -f1)"
  prev=$now
done
