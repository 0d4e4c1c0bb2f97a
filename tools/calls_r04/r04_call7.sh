#!/bin/bash
# timeline of the sharded world-1 step (rocprofv3 kernel trace) + kernel stats of the single-GPU step
cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r04
DR_FORCE_SHARDED=1 DR_BENCH_EVENTS=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04/trace_sh -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/gpurun_out/r04/trace_sh.log 2>&1
T=$(ls -t $(find $R/gpurun_out/r04/trace_sh -name "*kernel_trace.csv") | head -1)
python $R/tools/exp/timeline.py $T bf3_gemm_tn_rs_kernel -3 8 > $R/gpurun_out/r04/timeline_sharded_world1.txt
cat $R/gpurun_out/r04/timeline_sharded_world1.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/prof_single -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r04/prof_single.log 2>&1
TRACE=$(ls -t $(find $R/gpurun_out/r04/prof_single -name "*kernel_trace.csv") | head -1)
python $R/tools/trace_timed_stats.py "$TRACE" --last 20 > $R/gpurun_out/r04/kernel_stats_timed.csv
head -12 $R/gpurun_out/r04/kernel_stats_timed.csv
rm -rf $R/gpurun_out/r04/trace_sh/*/*.db $R/gpurun_out/r04/prof_single/*/*.db
