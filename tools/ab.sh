#!/bin/bash
# Interleaved same-box A/B of the headline step:  bash tools/ab.sh "ENV_A=.." "ENV_B=.." [rounds]
# prints ms_per_step (mean) and the p50 of each run
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', round(j['ms_per_step'],4), round(j['ms_per_step_percentiles']['p50'],4))"
  done
done
