#!/bin/bash
# does a long run hold the short run's rate?  100 / 2000 / 100 steps in one call
export DR_BENCH_STRICT=0
for n in 100 2000 100 4000; do
  echo "steps $n: $(timeout -s KILL 300 python bench.py --steps $n --warmup 5 --no-cpu-baseline --events off 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("final_loss"))')"
done
