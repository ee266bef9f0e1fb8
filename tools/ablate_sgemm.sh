#!/bin/bash
# build container: csrc/gemm_fp32.hip with -DSG_ABL=1..4 (timing-only ablations of its k-loop, see the source) linked against the product's other objects
# -> tools/microbench/libselftok_sgabl<n>.so, loaded by tools/ablate_sgemm.py through SELFTOK_HIP_LIB.  Run AFTER the product build (csrc/build/*.o).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/tools/microbench/tune_build
mkdir -p $O
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
      -DSG_ABL=$n -I $R/include -c $R/selftoktokenizer_amd/csrc/gemm_fp32.hip -o $O/gemm_fp32_abl$n.o &
done
wait
for n in 1 2 3 4; do
  objs=$(ls $R/selftoktokenizer_amd/csrc/build/*.o | grep -v gemm_fp32.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/microbench/libselftok_sgabl$n.so $objs $O/gemm_fp32_abl$n.o
done
echo built
