#!/bin/bash
# CPU: compile gemm_split.hip to ISA and print the main loop's memory / matrix instruction order (schedule audit)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I /root/repo/include -S --cuda-device-only /root/repo/selftoktokenizer_amd/csrc/gemm_split.hip -o /tmp/gemm_split.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "kernelILi0ELi0" | grep -i "vgprs\|spill\|error"
awk '/^_ZN7selftok19linear_f16x2_kernelILi0ELi0/,/s_endpgm/' /tmp/gemm_split.s > /tmp/k0.s
awk '/Inner Loop Header/,/s_branch|s_cbranch_scc0/' /tmp/k0.s | grep -n "s_waitcnt\|s_barrier\|ds_read_b128\|mfma\|global_load\|ds_write" | awk '{print $1" "$2" "$3}' | head -${1:-64} | tr '\n' ';' | sed 's/v_mfma_f32_32x32x16_f16/MFMA/g; s/ds_read_b128/DSR/g; s/global_load_lds_dwordx4/DMA/g; s/global_load_dwordx4/GLD/g; s/ds_write2st64_b64/DSW/g'
echo
