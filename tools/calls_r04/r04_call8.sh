#!/bin/bash
# sharded world-1: wgrad after the last owner-side K4 (DR_SH_K4_FIRST=1) vs beside it (0); timeline of the new order
cd /root/repo
mkdir -p gpurun_out/r04
run() { tag=$1; shift; env "$@" DR_FORCE_SHARDED=1 DR_BENCH_EVENTS=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json; python - $tag <<'PY'
import json,sys
d=json.loads(open("/tmp/l.json").read())
print("SH", sys.argv[1], d["ms_per_step"], d["config"]["final_loss"], flush=True)
PY
}
run warm DR_SH_K4_FIRST=1
for rep in 1 2 3; do
  run k4first DR_SH_K4_FIRST=1
  run beside DR_SH_K4_FIRST=0
done
cd /tmp && export TMPDIR=/tmp
R=/root/repo
DR_FORCE_SHARDED=1 DR_BENCH_EVENTS=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04/trace_sh2 -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/gpurun_out/r04/trace_sh2.log 2>&1
T=$(ls -t $(find $R/gpurun_out/r04/trace_sh2 -name "*kernel_trace.csv") | head -1)
python $R/tools/exp/timeline.py $T bf3_gemm_tn_rs_kernel -3 20 > $R/gpurun_out/r04/timeline_sharded_world1_k4first.txt
cat $R/gpurun_out/r04/timeline_sharded_world1_k4first.txt
rm -rf $R/gpurun_out/r04/trace_sh2/*/*.db
