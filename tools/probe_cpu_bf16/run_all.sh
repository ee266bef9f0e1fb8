#!/bin/bash
# build container only: regenerates profiles/r4_cpu_bf16_orders.txt
set -e
cd "$(dirname "$0")"
OUT=../../profiles/r4_cpu_bf16_orders.txt
gcc -O2 -fopenmp -ffp-contract=off -march=native -shared -fPIC -o libamxconv.so amxconv_emul.c
gcc -O2 -fopenmp -ffp-contract=off -march=native -shared -fPIC -o libgn.so groupnorm_emul.c -lm
gcc -O2 -fopenmp -ffp-contract=off -march=native -shared -fPIC -o libattn.so attention_emul.c -lm
gcc -O2 -ffp-contract=off -march=native -o expf_emul expf_emul.c -lm
{
  echo "# torch $(python -c 'import torch; print(torch.__version__)'), $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), $(nproc) threads; $(date -u +%F)"
  python -c "import torch; print(torch.__config__.show().split(chr(10))[4])"
  echo; echo "== fprev_amx_tile.py 32  (rows/cols = channel index; value = unit summands outside the lowest common subtree of the +M / -M pair)"
  python fprev_amx_tile.py 32
  echo; echo "== readout_amx_tile_fp32.py"
  python readout_amx_tile_fp32.py
  echo; echo "== fprev_conv_in.py (leaf index = (kh*3+kw)*3 + ic)"
  python fprev_conv_in.py
  for cfg in "128 128 256 3" "256 256 128 3" "512 512 64 3" "512 512 32 3" "128 128 256 3 2" "256 256 128 3 2" "512 512 64 3 2" "128 256 128 1" "512 512 32 1"; do
    echo; echo "== fprev_conv_chunks.py $cfg (IC OC H ksize [stride])"; python fprev_conv_chunks.py $cfg 2>&1 | cut -c1-240 | head -12
  done
  echo; echo "== check_conv_emul.py 1 2  [(order, mismatches, outputs, seconds)]; order 0 = (kh,kw,block), 1 = (block,kh,kw), 3 = block-major with private sums, 2 = conv_in"
  python check_conv_emul.py 1 2
  echo; echo "== onehot_groupnorm_lanes.py"; python onehot_groupnorm_lanes.py
  echo; echo "== check_groupnorm_stats.py"; python check_groupnorm_stats.py
  echo; echo "== check_groupnorm_emul.py (variant bit0: fma in the apply, bit1: float sqrt, bit2: bias by fma)"; python check_groupnorm_emul.py
  echo; echo "== check_attention_emul.py 4"; python check_attention_emul.py 4
  echo; echo "== expf_emul"; ./expf_emul
} 2>&1 | tee $OUT
