#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): rocprofv3 kernel stats of the bench step in both arithmetic modes, and
# separate PMC passes (never mixed with trace domains other than --kernel-trace) for the attention kernels.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/prof_r2
mkdir -p "$O"
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-token-check --no-kernel-roofs --no-other-gemm"
for mode in fp32 f16x2; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$mode" -o bench -- python "$R/bench.py" $COMMON --gemm $mode > "$O/bench_$mode.log" 2>&1
  grep -o '{"metric.*' "$O/bench_$mode.log" > "$O/bench_${mode}_under_rocprof.json"
  f=$(find "$O/$mode" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r2_bench_${mode}_kernel_stats.csv"
  find "$O/$mode" -name "*kernel_trace.csv" -delete
done
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/attn_$tag" -- python "$R/tools/bench_attn.py" > "$O/attn_$tag.log" 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_r2"
with open(out + "/r2_attn_pmc_summary.txt", "w") as fo:
    for f in sorted(glob.glob(out + "/attn_*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attn64" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            # bench_attn launches n_ctx = 512, 358, 20 x 13 launches per kernel: report the n_ctx=512 group (largest values)
            v = sorted(v)[-13:]
            print(f"{k[0]} {k[1]}: mean of the 13 largest (n_ctx=512) {sum(v)/len(v):.5g}", file=fo)
print(open(out + "/r2_attn_pmc_summary.txt").read())
PY
ls "$O"
