#!/bin/bash
# Round-4 profiles (GPU box, through gpurun): rocprofv3 kernel stats of the bench step in both GEMM arithmetics (default VAE mode = exact-order
# encoder), the stamped HBM-side traffic of the VQ kernels (tools/pmc_vq_traffic.sh; the default VQ path is now the one-MFMA coarse pass),
# a 2-rank dry run of `bench.py --gpus 2` on the one GPU over gloo (the N > 1 flow after this round's dist.py changes), configs[2].
# PMC passes never share a run with trace domains other than --kernel-trace.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_r4
mkdir -p "$O"
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-token-check --no-kernel-roofs --no-other-gemm --no-latency"
for mode in fp32 f16x2; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$mode" -o bench -- python "$R/bench.py" $COMMON --gemm $mode > "$O/bench_$mode.log" 2>&1
  grep -o '{"metric.*' "$O/bench_$mode.log" > "$O/r4_bench_${mode}_under_rocprof.json"
  f=$(find "$O/$mode" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r4_bench_${mode}_kernel_stats.csv"
  find "$O/$mode" -name "*kernel_trace.csv" -delete
done
bash "$R/tools/pmc_vq_traffic.sh" > "$O/pmc_vq_traffic.log" 2>&1
cp "$R/gpurun_out/pmc_vq/vq_traffic.json" "$O/vq_traffic.json"
SELFTOK_ONE_GPU=1 SELFTOK_DIST_BACKEND=gloo timeout 600 python "$R/bench.py" --gpus 2 --batch 8 $COMMON 2> "$O/dryrun2.err" | grep -o '{"metric.*' > "$O/r4_bench_dryrun_2ranks_gloo_one_gpu.json"
timeout 900 python "$R/bench.py" --tokens 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-token-check --no-kernel-roofs --no-latency 2>/dev/null | grep -o '{"metric.*' > "$O/r4_bench_config2_k1024.json"
# where the waves of the two new hot kernels spend their cycles (SQ counters, one PMC pass each, kernel-trace only)
bash "$R/tools/pmc_sq.sh" xconv xconv_kernel -- python "$R/tools/bench_vae_exact.py" 16 > "$O/pmc_sq_xconv.log" 2>&1
cp "$R/gpurun_out/pmc_sq_xconv/summary.json" "$O/r4_pmc_sq_xconv.json"
bash "$R/tools/pmc_sq.sh" vq vq_f16_kernel -- python "$R/tools/pmc_vq.py" 32768 > "$O/pmc_sq_vq.log" 2>&1
cp "$R/gpurun_out/pmc_sq_vq/summary.json" "$O/r4_pmc_sq_vq.json"
ls -la "$O"
