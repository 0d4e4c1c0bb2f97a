#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/c32
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c32/prof -o d -- python $R/tools/exp/topk_bench.py > $R/gpurun_out/c32/topk.log 2>&1
cd $R
tail -5 gpurun_out/c32/topk.log
f=$(find gpurun_out/c32/prof -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-90,120-220
