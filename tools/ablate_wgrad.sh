#!/bin/bash
# Ablation builds of the pipe weight-gradient tile (csrc/wgrad_split.hip, WGP_ABLATE) -> tools/_prof/libwgp_<n>.so; run
# HERE (hipcc), then on the GPU box:  python tools/ablate_wgrad.py
#   0 full | 1 no MFMAs | 2 no split arithmetic | 3 no global loads | 4 no fragment reads | 5 no stash stores
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_prof
for n in 0 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -I include \
      -DWGP_ABLATE=$n -shared msr3d_amd/csrc/wgrad_split.hip -o tools/_prof/libwgp_$n.so &
done
wait
ls -la tools/_prof/libwgp_*.so
