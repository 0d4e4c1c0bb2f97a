#!/bin/bash
for v in "A=1" "DR_SIDE_R=0" "DR_PREFETCH_EARLY=1" "DR_SIDE_R=0 DR_PREFETCH_EARLY=1 DR_FUSE_PLAN_FRONT=0"; do
  echo "[$v] $(env $v timeout -s KILL 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], (d.get("strict_fp32") or {}).get("ms_per_step"))')"
done
echo "[standalone bf16x3] $(DR_GEMM_SPLIT=bf16x3 DR_BENCH_STRICT=0 timeout -s KILL 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
