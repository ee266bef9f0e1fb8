#!/bin/bash
# GPU: where the fp32 GEMM kernels spend their cycles and what they pull through L2 (two rocprofv3 PMC passes, kernel-trace only).
#   bash tools/pmc_sgemm.sh <tag> [M K N]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
tag=${1:-sgemm}; shift
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/sq -- python $R/tools/pmc_sgemm.py "$@" > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum \
    --kernel-trace --output-format csv -d $O/tcc -- python $R/tools/pmc_sgemm.py "$@" > $O/tcc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d $O/inst -- python $R/tools/pmc_sgemm.py "$@" > $O/inst.log 2>&1
OUT="$O" python - <<'PY'
import csv, glob, os, collections, json
out = os.environ["OUT"]
res = collections.defaultdict(dict)
for sub in ("sq", "tcc", "inst"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "gemm" in k or "Cijk" in k:
                acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "gemm" in k or "Cijk" in k:
                dur[k[:48]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
        res[k][f"avg_us_{sub}_pass"] = round(sum(dur[k]) / max(1, len(dur[k])) / 1e3, 1)
for k, row in res.items():
    wc = row.get("SQ_WAVE_CYCLES") or 1
    row["frac_parked_waitcnt_barrier"] = round(row.get("SQ_WAIT_ANY", 0) / wc, 4)
    row["frac_issue_stall"] = round(row.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
    row["frac_active_issue"] = round(row.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4)
    if row.get("GRBM_GUI_ACTIVE"):
        row["mfma_busy_over_gui_active_per_simd"] = round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (row["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if row.get("TCC_HIT_sum") is not None:
        row["l2_hit_rate"] = round(row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row.get("TCC_MISS_sum", 0)), 4)
        row["fabric_read_MB_at_64B_per_req"] = round(row.get("TCC_EA0_RDREQ_sum", 0) * 64 / 1e6, 1)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
