#!/bin/bash
# as ab_env.sh, with bench.py's default per-kernel events (every 4th step)
export DR_BENCH_STRICT=0
for i in 1 2; do
for v in "$@"; do
  echo "[$v] $(env $v timeout -s KILL 200 python bench.py --no-cpu-baseline $AB_ARGS 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("final_loss"), d["config"].get("settle_steps_untimed"), [(r["kernel"], r["avg_us"]) for r in d["roofline_all"]][:8])')"
done; done
