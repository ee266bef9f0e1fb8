#!/bin/bash
# GPU: PMC passes over linear_f16x2_kernel at M=22912 N=4608 K=1536 (one pass per counter group; gpurun refuses mixed trace modes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python $R/tools/bench_gemm.py 1 > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_gemm"
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "linear_f16x2" in r["Kernel_Name"]:
            kern = "pre_kernel" if "pre_kernel" in r["Kernel_Name"] else "kernel"
            acc[(kern, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, k), v in sorted(acc.items()):
        print(f"linear_f16x2_{kern} {k}: mean {sum(v)/len(v):.4g} over {len(v)} launches")
PY
