#!/bin/bash
# GPU: HBM-side traffic of the VQ kernels at N = 32768, measured for THIS build.
#   * separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel-trace only) over tools/pmc_vq.py 32768 {f16|fp32};
#   * the same two passes over tools/microbench/fetch_calib (known byte counts in the VQ kernels' access patterns) give the
#     correction factor of each counter for each pattern on this box;
#   * result -> gpurun_out/pmc_vq/vq_traffic.json, stamped with the sha256 of the kernel sources (bench.py refuses a stale stamp).
# Copy the result to profiles/vq_traffic.json to have bench.py quote it.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/pmc_vq
mkdir -p $O
CAL=$R/tools/microbench/fetch_calib
[ -x $CAL ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $CAL $R/tools/microbench/fetch_calib.hip
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- $CAL > $O/calib_$c.log 2>&1
  for mode in f16 fp32; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${mode}_$c -- python $R/tools/pmc_vq.py 32768 $mode > $O/${mode}_$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os, json, collections, sys
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R)
out = R + "/gpurun_out/pmc_vq"

def counters(tag, counter, want):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
            if want(k):
                acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}          # mean per launch (KB)

known = json.loads([l for l in open(f"{out}/calib_FETCH_SIZE.log") if l.startswith("{")][-1])
cal_f = counters("calib", "FETCH_SIZE", lambda k: k.startswith("calib_read"))
cal_w = counters("calib", "WRITE_SIZE", lambda k: k.startswith("calib_write"))
factor = {k: known[k] / (v * 1024.0) for k, v in {**cal_f, **cal_w}.items() if v > 0}
res = {"calibration": {"known_bytes": known, "counter_KB": {**cal_f, **cal_w}, "true_bytes_per_counted_byte": {k: round(v, 4) for k, v in factor.items()}}}
f16 = factor.get("calib_read_stream16", 2.0)
fc8, ft4 = factor.get("calib_read_cand8", 2.0), factor.get("calib_read_tile4", 2.0)
w8, w16 = factor.get("calib_write_cand8", 1.0), factor.get("calib_write_stream16", 1.0)
for mode, key in (("f16", "N32768_f16"), ("fp32", "N32768")):
    fe = counters(mode, "FETCH_SIZE", lambda k: k.startswith("vq_") and "pack" not in k)
    wr = counters(mode, "WRITE_SIZE", lambda k: k.startswith("vq_") and "pack" not in k)
    det, total, lo, hi = {}, 0.0, 0.0, 0.0
    for k in sorted(set(fe) | set(wr)):
        fin = "finalize" in k
        # main kernels read with 16 B per lane (rows + LDS-DMA) and store 8-byte candidates; the finalize kernels mix 8-byte candidate
        # reads, 4-byte gathers of code tiles and 16-byte row loads: its factor is bracketed by the calibrated patterns and the
        # quoted figure uses their mean
        ffac = ((fc8 + ft4) / 2.0) if fin else f16
        fb, wb = fe.get(k, 0.0) * 1024 * ffac, wr.get(k, 0.0) * 1024 * (w16 if fin else w8)
        det[k] = {"FETCH_SIZE_KB": round(fe.get(k, 0.0), 1), "WRITE_SIZE_KB": round(wr.get(k, 0.0), 1), "fetch_factor": round(ffac, 4),
                  "fetch_bytes": round(fb), "write_bytes": round(wb)}
        total += fb + wb
        lo += fe.get(k, 0.0) * 1024 * (min(fc8, ft4, f16) if fin else f16) + wb
        hi += fe.get(k, 0.0) * 1024 * (max(fc8, ft4, f16) if fin else f16) + wb
    res[key] = round(total)
    res[key + "_detail"] = {"per_kernel": det, "bracket_bytes": [round(lo), round(hi)]}
import bench
res["source_stamp"] = bench.source_stamp()
res["method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (kernel-trace only) over tools/pmc_vq.py 32768, mean of 5 launches per kernel; "
                 "counter -> bytes by the per-pattern factors measured in the same passes with tools/microbench/fetch_calib (known byte counts)")
json.dump(res, open(out + "/vq_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
