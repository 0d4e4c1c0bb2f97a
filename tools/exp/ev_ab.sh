#!/bin/bash
# what bench.py's per-kernel events cost its own timed region: headline kernel only (default) / every kernel (rounds 1-5) / none
export DR_BENCH_STRICT=0
for i in 1 2; do
for v in "A=1" "DR_BENCH_EVENTS_ALL=1" "DR_BENCH_EVENTS=0"; do
  echo "[$v] $(env $v timeout -s KILL 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(d["ms_per_step"], r.get("kernel"), r.get("avg_us"), r.get("launches"), [(x["kernel"], x["avg_us"], x["launches"]) for x in d.get("roofline_all", [])][:4])')"
done; done
