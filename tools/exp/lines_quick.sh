#!/bin/bash
# quick look at the secondary bench lines (no CPU baseline, no strict-fp32 leg)
export DR_BENCH_STRICT=0
run() { echo "[$1] $(env $2 timeout -s KILL 300 python bench.py --no-cpu-baseline $3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("final_loss"), (d.get("metric_pass") or {}).get("topk_ms_per_batch"))')"; }
run default "A=1" ""
run adam "A=1" "--optimizer adam"
run zipf "A=1" "--ids zipf"
run c2 "A=1" "--preset c2"
run dssm "A=1" "--model dssm"
run sharded "DR_FORCE_SHARDED=1" ""
run dcn "A=1" "--model dcn"
