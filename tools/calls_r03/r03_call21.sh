#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/c21
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c21/prof -o b -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/c21/bench.log 2>&1
cd $R
f=$(find gpurun_out/c21/prof -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-160
