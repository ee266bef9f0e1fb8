#!/bin/bash
# build container: all of csrc/*.hip with -DSELFTOK_TUNE (kernel-variant switches read from the environment; the product build has
# none) -> tools/microbench/libselftok_tune.so, loaded by the tools/bench_*.py A/B scripts through SELFTOK_HIP_LIB.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/tools/microbench/tune_build
mkdir -p $O
for f in $R/selftoktokenizer_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
      -DSELFTOK_TUNE "$@" -I $R/include -c $f -o $O/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/microbench/libselftok_tune.so $O/*.o
echo built $R/tools/microbench/libselftok_tune.so
