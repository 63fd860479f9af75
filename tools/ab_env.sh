#!/bin/bash
# Interleaved A/B of bench.py under environment settings:  bash tools/ab_env.sh ROUNDS "" "MSR3D_WGRAD_STREAM=1" ...
# (one line per run: setting, samples/s, ms per step, p50).  A setting may carry bench.py flags after "--":
#   "MSR3D_SA3_TILES=0 -- --no-pipeline"
R=$1; shift
for i in $(seq 1 $R); do
  for e in "$@"; do
    envs="${e%%--*}"; flags=""; [[ "$e" == *"--"* ]] && flags="${e#*-- }"
    env $envs python bench.py --no-cpu-baseline --no-extra --census-steps 0 --steps 40 $flags 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-72s %9.1f  %.4f  p50 %.4f' % ('$e' or 'default', d['value'], d['ms_per_step'], d['ms_per_step_percentiles']['p50']))"
  done
done
