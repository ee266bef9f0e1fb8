#!/bin/bash
# GPU (through gpurun): rocprofv3 kernel stats of one bench step in f16x2 mode (the MMDiT block Linears on the pre-split f16x2 kernel).
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_r2b
mkdir -p "$O"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/f16x2" -o bench -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-token-check --no-kernel-roofs --no-other-gemm --gemm f16x2 > "$O/bench_f16x2.log" 2>&1
grep -o '{"metric.*' "$O/bench_f16x2.log" > "$O/r2_bench_f16x2_presplit_under_rocprof.json"
f=$(find "$O/f16x2" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r2_bench_f16x2_presplit_kernel_stats.csv"
find "$O/f16x2" -name "*kernel_trace.csv" -delete
head -12 "$O/r2_bench_f16x2_presplit_kernel_stats.csv" | cut -c1-160
