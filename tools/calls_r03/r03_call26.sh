#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/c26
cd /tmp && export TMPDIR=/tmp
DR_FORCE_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c26/prof -o s -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 3 --events off > $R/gpurun_out/c26/bench.log 2>&1
cd $R
f=$(find gpurun_out/c26/prof -name "*kernel_trace.csv" | head -1)
python tools/exp/timeline.py $f hash_bucket_i64_kernel -3 12
python tools/trace_gaps.py $f hash_bucket_i64_kernel -3 30
tail -1 gpurun_out/c26/bench.log | cut -c1-200
