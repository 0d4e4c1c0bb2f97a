#!/bin/bash
# soak: the run-to-run bit-identity tests (a race between the streams would show up as a difference), the schedule-switch / fused-K4
# bit-identity tests, the slot-plan tests and the two-rank sharded tests, several times over; then a long bench run whose loss must stay finite
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_benchcfg.py -m gpu -q -x -k "three_steps or reproducible" 2>&1 | grep -E "passed|failed" | sed "s/^/repro $i: /"
done
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_gpu_fused_k4.py tests/test_gpu_models.py -m gpu -q -x -k "fused or schedule or prefetch or tail" 2>&1 | grep -E "passed|failed" | sed "s/^/engine schedules $i: /"
  timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "plan or sorted or hash_sort" 2>&1 | grep -E "passed|failed" | sed "s/^/slot plan $i: /"
done
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_sharded_two_rank.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | sed "s/^/two-rank $i: /"
done
timeout 600 python bench.py --steps 2000 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('long run', d['steps'], d['ms_per_step'], d['config'].get('final_loss'))"
