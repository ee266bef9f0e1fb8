#!/bin/bash
# GPU: where the waves of a kernel spend their cycles (one rocprofv3 PMC pass, kernel-trace only):
#   usage: bash tools/pmc_sq.sh <tag> <kernel-name-substring> -- <command...>
# SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (MI355X_MICROARCH.md);
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs ...) = matrix-pipe utilisation.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
tag=$1; pat=$2; shift 3
O=$R/gpurun_out/pmc_sq_$tag
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/run -- "$@" > $O/run.log 2>&1
PAT="$pat" OUT="$O" python - <<'PY'
import csv, glob, os, collections, json
out, pat = os.environ["OUT"], os.environ["PAT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/run/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat in k:
            acc[k[-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/run/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat in k:
            dur[k[-60:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for k, cs in acc.items():
    row = {c: sum(v) / len(v) for c, v in cs.items()}
    row["launches"] = len(next(iter(cs.values())))
    if k in dur:
        row["avg_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
    wc = row.get("SQ_WAVE_CYCLES", 0) or 1
    row["frac_parked_waitcnt_barrier"] = round(row.get("SQ_WAIT_ANY", 0) / wc, 4)
    row["frac_issue_stall"] = round(row.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
    row["frac_active_issue"] = round(row.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4)
    if row.get("GRBM_GUI_ACTIVE"):
        row["mfma_busy_over_gui_active_per_simd"] = round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (row["GRBM_GUI_ACTIVE"] / 8.0), 4)
    res[k] = row
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
