#!/bin/bash
# GPU: HBM-side traffic of the VQ kernels at N = 32768 (separate --pmc passes, kernel-trace only) -> gpurun_out/pmc_vq_r2/summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/pmc_vq_r2
mkdir -p $O
for mode in f16 fp32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${mode}_$c -- python $R/tools/pmc_vq.py 32768 $mode > $O/${mode}_$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os, json, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_vq_r2"
res = {}
for mode in ("f16", "fp32"):
    acc = collections.defaultdict(list)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/{mode}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "vq_" in k and "pack" not in k:
                    acc[(k.split("(")[0].split("<")[0][-28:], c)].append(float(r["Counter_Value"]))
    d = {f"{k[0]}:{k[1]}_KB": round(sum(v) / len(v), 1) for k, v in sorted(acc.items())}
    fetch = sum(v for k, v in d.items() if "FETCH" in k) * 1024 * 2          # gfx950: FETCH_SIZE reports half of wide coalesced reads
    write = sum(v for k, v in d.items() if "WRITE" in k) * 1024
    res[mode] = {"per_kernel": d, "fetch_bytes_corrected": fetch, "write_bytes": write, "total_bytes": fetch + write}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
