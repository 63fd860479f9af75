#!/bin/bash
# One-step kernel timeline of the default bench (rocprofv3 --kernel-trace):  bash tools/timeline.sh OUT.txt [ENV=..]
export TMPDIR=/tmp
ROOT=$(pwd); OUT=${1:-gpurun_out/timeline.txt}; shift
D=$(mktemp -d /tmp/tl.XXXX)
(cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extra > "$D/log" 2>&1)
F=$(ls "$D"/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$F" ] && python tools/trace_step.py "$F" > "$OUT"
rm -rf "$D"
