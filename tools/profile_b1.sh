#!/bin/bash
# B = 1 latency: rocprofv3 kernel stats of encode + 50-step decode of ONE image in both GEMM arithmetics (GPU box, through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/b1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for mode in ${MODES:-fp32 f16x2}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o b1 -- python $R/tools/profile_b1.py 3 $mode > $O/prof_$mode.log 2>&1
  grep "pass" $O/prof_$mode.log
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1); cp "$f" $O/b1_${mode}_kernel_stats.csv; rm -rf $O/$mode; head -12 $O/b1_${mode}_kernel_stats.csv | cut -c1-120
done
