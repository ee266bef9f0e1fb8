#!/bin/bash
# rocprofv3 PMC passes (separate runs, kernel-trace only) over two sampler steps of the C2 workload; writes per-kernel summaries.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/pmc_dit
mkdir -p "$O"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/$n" -o run -- python "$R/tools/pmc_dit.py" > "$O/$n.log" 2>&1
  python "$R/tools/pmc_summarize.py" "$O/$n" run > "$O/$n.summary.jsonl"
  mv "$O/$n/run_counter_collection.csv" /tmp/ 2>/dev/null
  mv "$O/$n/run_kernel_trace.csv" /tmp/ 2>/dev/null
done
head -7 "$O"/*.summary.jsonl | cut -c1-330
