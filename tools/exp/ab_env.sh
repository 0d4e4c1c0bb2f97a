#!/bin/bash
# A/B of environment switches on the default bench command: tools/exp/ab_env.sh "VAR=a" "VAR=b OTHER=c" ...  (two alternating rounds)
export DR_BENCH_STRICT=0
for i in 1 2; do
for v in "$@"; do
  echo "[$v] $(env $v timeout -s KILL 200 python bench.py --no-cpu-baseline --events off $AB_ARGS 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("final_loss"))')"
done; done
