#!/bin/bash
# One steady-state step of an engine as a timeline + the idle gaps per stream (rocprofv3 kernel trace, per-kernel HIP events off so nothing
# but the step's own launches is on the streams).  usage: timeline_call.sh <name> <anchor kernel> [env ...] -- [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; anchor=$2; shift 2
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
export DR_BENCH_STRICT=0
rm -rf $R/gpurun_out/tl_$name
env "${envs[@]}" timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$name -- python $R/bench.py --steps 20 --warmup 20 --no-cpu-baseline --events off "$@" > $R/gpurun_out/tl_$name.log 2>&1
T=$(ls -t $(find $R/gpurun_out/tl_$name -name "*kernel_trace.csv") | head -1)
{ tail -1 $R/gpurun_out/tl_$name.log | cut -c1-200
  python $R/tools/trace_step.py $T $anchor -4
  python $R/tools/trace_gaps.py $T $anchor -4 2
  python $R/tools/trace_gaps.py $T $anchor -6 2 | head -3; } > $R/gpurun_out/timeline_$name.txt 2>&1
rm -rf $R/gpurun_out/tl_$name
