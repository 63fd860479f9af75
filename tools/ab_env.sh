#!/bin/bash
# Interleaved A/B of bench.py under environment settings:  bash tools/ab_env.sh ROUNDS "" "MSR3D_WGRAD_STREAM=1" ...
# (one line per run: setting, samples/s, ms per step, p50)
R=$1; shift
for i in $(seq 1 $R); do
  for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --no-extra --census-steps 0 --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-40s %9.1f  %.4f  p50 %.4f' % ('$e' or 'default', d['value'], d['ms_per_step'], d['ms_per_step_percentiles']['p50']))"
  done
done
