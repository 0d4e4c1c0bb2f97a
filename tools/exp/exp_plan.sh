#!/bin/bash
# EXPERIMENT (round 6): what the next batch's slot plan costs the step.  DR_EXP_SKIP_PLAN=1 (engine.py): after 8 steps the side chain is
# not launched any more (the buffers keep an older batch's ids + plan: a valid step on stale ids, timing only).
export DR_BENCH_STRICT=0
for i in 1 2; do
for v in base skipplan; do
  unset DR_EXP_SKIP_PLAN
  [ $v = skipplan ] && export DR_EXP_SKIP_PLAN=1
  echo "$v $(timeout -s KILL 200 python bench.py --no-cpu-baseline --events off $@ 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("final_loss"))')"
done; done
