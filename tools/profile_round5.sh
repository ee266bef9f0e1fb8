#!/bin/bash
# Round-5 profiles (GPU box, through gpurun): rocprofv3 kernel stats of the bench step in both GEMM arithmetics (defaults: exact-order VAE
# encoder AND decoder, exact-order Q-Former encoder), the stamped HBM-side traffic of the VQ kernels, the encoder in both modes.
# PMC passes never share a run with trace domains other than --kernel-trace.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_r5
mkdir -p "$O"
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-token-check --no-kernel-roofs --no-other-gemm --no-latency"
for mode in ${MODES:-fp32 f16x2}; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$mode" -o bench -- python "$R/bench.py" $COMMON --gemm $mode > "$O/bench_$mode.log" 2>&1
  grep -o '{"metric.*' "$O/bench_$mode.log" > "$O/r5_bench_${mode}_under_rocprof.json"
  f=$(find "$O/$mode" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r5_bench_${mode}_kernel_stats.csv"
  find "$O/$mode" -name "*kernel_trace.csv" -delete
done
if [ "${PMC:-1}" = "1" ]; then
  bash "$R/tools/pmc_vq_traffic.sh" > "$O/pmc_vq_traffic.log" 2>&1
  cp "$R/gpurun_out/pmc_vq/vq_traffic.json" "$O/vq_traffic.json"
fi
python "$R/tools/bench_encoder_modes.py" 64 > "$O/r5_encoder_modes.txt" 2>&1
python "$R/tools/bench_vae_exact.py" 64 > "$O/r5_vae_exact_bench.txt" 2>&1
ls -la "$O"
