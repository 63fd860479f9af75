#!/bin/bash
# Everything profiles/ holds for a round, in one go on the GPU box:  bash tools/collect_profiles.sh gpurun_out/r03 r03_v1
# (bench line with CPU baseline, rocprofv3 kernel stats + one-step timeline of the same command, PMC passes,
#  the variant lines, the secondary LLM-layer line, the per-op CPU table)
set -u
OUT=${1:-gpurun_out/prof}
TAG=${2:-rXX}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p "$OUT"
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
tail -c 400 "$OUT/${TAG}_bench.err"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/kt" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extra > "$ROOT/$OUT/kt.log" 2>&1)
F=$(ls "$OUT"/kt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" "$OUT/${TAG}_bench_kernel_stats.csv"
rm -rf "$OUT/kt"
# one step's launches in order: the un-pipelined schedule (one stream), census off (no event pairs between the kernels)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/kt1" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extra --no-pipeline --census-steps 0 > "$ROOT/$OUT/kt1.log" 2>&1)
F=$(ls "$OUT"/kt1/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tools/trace_step.py "$F" > "$OUT/${TAG}_step_timeline.txt"
rm -rf "$OUT/kt1"
bash tools/pmc_step.sh "$OUT/pmc_step" > /dev/null 2>&1; cp "$OUT/pmc_step/summary.txt" "$OUT/${TAG}_pmc_step.txt"; rm -rf "$OUT/pmc_step"
bash tools/pmc_sa.sh "$OUT/pmc_sa" > /dev/null 2>&1; cp "$OUT/pmc_sa/summary.txt" "$OUT/${TAG}_pmc_sa.txt"; rm -rf "$OUT/pmc_sa"
bash tools/pmc_blocks.sh "$OUT/pmc_blocks" > /dev/null 2>&1; cp "$OUT/pmc_blocks/summary.txt" "$OUT/${TAG}_pmc_blocks.txt"; rm -rf "$OUT/pmc_blocks"
V="$OUT/${TAG}_variants.jsonl"; : > "$V"
v() { echo "# $*" >> "$V.cmds"; "$@" 2>/dev/null | tail -1 >> "$V"; }
v python bench.py --no-cpu-baseline --batch 4 --accum 5
v python bench.py --no-cpu-baseline --batch 4 --accum 5 --micro-steps
v python bench.py --no-cpu-baseline --batch 4 --accum 5 --micro-steps --no-window
v python bench.py --no-cpu-baseline --skip-padded
v python bench.py --no-cpu-baseline --no-pipeline
v python bench.py --no-cpu-baseline --situation-type as_object
v python bench.py --no-cpu-baseline --dense
v env MSR3D_TRAIN_MMA=bf16 python bench.py --no-cpu-baseline
v python bench.py --no-cpu-baseline --from-store
v python bench.py --no-cpu-baseline --host-inputs
v python bench.py --no-cpu-baseline --objects 120 --points 2048 --llm-hidden 5120 --batch 8 --situation-type as_object
v env MSR3D_ATTN_MMA=bf16 python bench.py --no-cpu-baseline --objects 120 --points 2048 --llm-hidden 5120 --batch 8 --situation-type as_object
v env MSR3D_ATTN_MMA=fp8_bf16 python bench.py --no-cpu-baseline --objects 120 --points 2048 --llm-hidden 5120 --batch 8 --situation-type as_object
v env MSR3D_TRAINABLE=strips python bench.py --no-cpu-baseline
v env MSR3D_SA_MMA=f32 python bench.py --no-cpu-baseline
v env MSR3D_SA_ROWS=0 python bench.py --no-cpu-baseline
v env MSR3D_SA_MMA=split2 python bench.py --no-cpu-baseline
v env MSR3D_WGRAD_MIXED=0 python bench.py --no-cpu-baseline
v env MSR3D_WGRAD_PIPE=0 python bench.py --no-cpu-baseline
v env MSR3D_WGRAD_STREAM=1 python bench.py --no-cpu-baseline
v env MSR3D_PACK_FORK=1 python bench.py --no-cpu-baseline
v env MSR3D_SA3_TILE=2 python bench.py --no-cpu-baseline
v env MSR3D_WGRAD_HALVES=1 python bench.py --no-cpu-baseline
v env MSR3D_ATTN_FWD_WAVES=4 MSR3D_ATTN_FWD_SPLIT=0 python bench.py --no-cpu-baseline
v env MSR3D_ATTN_FWD_SPLIT=0 python bench.py --no-cpu-baseline
v env MSR3D_ATTN_BWD=2 python bench.py --no-cpu-baseline
v env MSR3D_ATTN_BWD=0 python bench.py --no-cpu-baseline
v env MSR3D_FC_SPLIT=0 python bench.py --no-cpu-baseline
v env MSR3D_FFN_WAVES=4 python bench.py --no-cpu-baseline
v env MSR3D_SA_PLAN12=0 python bench.py --no-cpu-baseline
v env MSR3D_SA_PLAN_IN_SAMPLING=0 python bench.py --no-cpu-baseline
v env MSR3D_SA3_TILES=0 python bench.py --no-cpu-baseline
v env MSR3D_FPS_QUERY=0 python bench.py --no-cpu-baseline
v env MSR3D_BENCH_FORCE_DIST=1 MSR3D_DP_GRAPH_COMM=0 python bench.py --no-cpu-baseline
v env MSR3D_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline
v env MSR3D_BENCH_FORCE_DIST=1 MSR3D_DP_GRAPH_COMM=1 python bench.py --no-cpu-baseline
v env MSR3D_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --batch 4 --accum 5
v env MSR3D_BENCH_FORCE_DIST=1 MSR3D_DP_GRAPH_COMM=1 python bench.py --no-cpu-baseline --unfrozen --batch 16 --steps 10 --warmup 2
v python bench.py --no-cpu-baseline --unfrozen --batch 16 --steps 10 --warmup 2
python bench.py --llm-layer --steps 10 --warmup 3 2>/dev/null | tail -1 > "$OUT/${TAG}_llm_layer.json"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/ktl" -- python "$ROOT/bench.py" --llm-layer --steps 10 --warmup 3 > /dev/null 2>&1)
F=$(ls "$OUT"/ktl/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" "$OUT/${TAG}_llm_layer_kernel_stats.csv"
rm -rf "$OUT/ktl"
python bench.py --llm-stack 4 --steps 10 --warmup 3 2>/dev/null | tail -1 > "$OUT/${TAG}_llm_stack.json"
python bench.py --full-step --steps 8 --warmup 2 2>/dev/null | tail -1 > "$OUT/${TAG}_full_step.json"
python bench.py --full-step --batch 20 --steps 4 --warmup 2 2>/dev/null | tail -1 > "$OUT/${TAG}_full_step_window20.json"
env MSR3D_BENCH_FORCE_DIST=1 python bench.py --full-step --steps 8 --warmup 2 2>/dev/null | tail -1 > "$OUT/${TAG}_full_step_rccl1.json"
python bench.py --full-step --llm-fp8 --steps 8 --warmup 2 2>/dev/null | tail -1 > "$OUT/${TAG}_full_step_fp8.json"
python bench.py --full-step --llm-fp8 --batch 20 --steps 4 --warmup 2 2>/dev/null | tail -1 > "$OUT/${TAG}_full_step_fp8_window20.json"
python bench.py --llm-layer --llm-fp8 --steps 10 --warmup 3 2>/dev/null | tail -1 > "$OUT/${TAG}_llm_layer_fp8.json"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/ktf" -- python "$ROOT/bench.py" --full-step --llm-fp8 --steps 4 --warmup 2 > /dev/null 2>&1)
F=$(ls "$OUT"/ktf/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" "$OUT/${TAG}_full_step_fp8_kernel_stats.csv"
# (the stats above include the model's construction: init kernels, weight transposes; ONE step's kernels, aggregated:)
F=$(ls "$OUT"/ktf/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && python tools/trace_step_agg.py "$F" > "$OUT/${TAG}_full_step_fp8_one_step.txt"
rm -rf "$OUT/ktf"
bash tools/pmc_llm.sh "$OUT/pmc_llm" > /dev/null 2>&1; cp "$OUT/pmc_llm/summary.txt" "$OUT/${TAG}_pmc_llm_layer.txt"; rm -rf "$OUT/pmc_llm"
python tools/prof_fp8_gemm.py 2>/dev/null | grep "^M=" > "$OUT/${TAG}_fp8_gemm.txt"
python tools/prof_attn.py 2>/dev/null | grep "^B=" > "$OUT/${TAG}_attn.txt"
python tools/bench_bf16_gemm.py 2>/dev/null | grep -v amdgpu > "$OUT/${TAG}_bf16_gemm.txt"
python -m pytest tests/test_seq_ce_gpu.py -q -s -k roofline 2>/dev/null | grep seq_ce > "$OUT/${TAG}_seq_ce.txt"
python bench.py --cpu-ops --round-tag "$TAG" > "$OUT/${TAG}_cpu_ops.log" 2>&1; cp "profiles/${TAG}_cpu_ops.json" "$OUT/${TAG}_cpu_ops.json"
timeout 300 python tools/prof_blocks.py > "$OUT/${TAG}_block_stamps.txt" 2>&1
timeout 300 python tools/prof_sa_rows.py > "$OUT/${TAG}_sa_rows_stamps.txt" 2>&1
timeout 200 python tools/bench_wgrad.py > "$OUT/${TAG}_wgrad_forms.txt" 2>&1
[ -x tools/_prof/last_arriver ] && timeout 60 tools/_prof/last_arriver > "$OUT/${TAG}_last_arriver.txt" 2>&1
timeout 120 python tools/bench_sa.py > "$OUT/${TAG}_encoder_kernels.txt" 2>&1
timeout 120 python tools/bench_fps.py > "$OUT/${TAG}_sampling_launch.txt" 2>&1
ls -la "$OUT"
