#!/bin/bash
# Round-6 profiles (GPU box, through gpurun): rocprofv3 kernel stats of ONE bench step per GEMM arithmetic -- each run is a PURE step of that mode
# (--no-exact --no-other-gemm --no-parity16 --no-parity64: nothing of another arithmetic in the file) -- and optionally the VQ traffic PMC passes.
#   MODES="fp32 f16x2 exact" PMC=0 bash tools/profile_round6.sh
# PMC passes never share a run with trace domains other than --kernel-trace.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_r6
mkdir -p "$O"
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-token-check --no-kernel-roofs --no-other-gemm --no-exact --no-parity16 --no-parity64"
for mode in ${MODES:-fp32 f16x2 exact}; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$mode" -o bench -- python "$R/bench.py" $COMMON --gemm $mode > "$O/bench_$mode.log" 2>&1
  grep -o '{"metric.*' "$O/bench_$mode.log" > "$O/r6_bench_${mode}_under_rocprof.json"
  f=$(find "$O/$mode" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r6_bench_${mode}_kernel_stats.csv"
  find "$O/$mode" -name "*kernel_trace.csv" -delete
done
if [ "${PMC:-0}" = "1" ]; then
  bash "$R/tools/pmc_vq_traffic.sh" > "$O/pmc_vq_traffic.log" 2>&1
  cp "$R/gpurun_out/pmc_vq/vq_traffic.json" "$O/vq_traffic.json"
fi
ls -la "$O"
