#!/bin/bash
# Round profile: rocprofv3 kernel stats of the default bench command + separate PMC passes for the VQ kernel.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_final
mkdir -p "$O"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -o vq -- python "$R/tools/pmc_vq.py" 32768 > "$O/pmc_$c.log" 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -o vq -- python "$R/tools/pmc_vq.py" 32768 > "$O/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -o bench -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$O/bench.log" 2>&1
grep -o '{"metric.*' "$O/bench.log" > "$O/bench_under_rocprof.json"
mv "$O/bench/bench_kernel_trace.csv" /tmp/ 2>/dev/null
ls -R "$O" | head -30
