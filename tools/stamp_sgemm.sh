#!/bin/bash
# build container: csrc/gemm_fp32.hip with -DSG_STAMP (in-kernel s_memtime stamps, see the source) -> tools/microbench/libselftok_sgstamp.so (tools/stamp_sgemm.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/tools/microbench/tune_build
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
    -DSG_STAMP -I $R/include -c $R/selftoktokenizer_amd/csrc/gemm_fp32.hip -o $O/gemm_fp32_stamp.o
objs=$(ls $R/selftoktokenizer_amd/csrc/build/*.o | grep -v gemm_fp32.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/microbench/libselftok_sgstamp.so $objs $O/gemm_fp32_stamp.o
echo built
