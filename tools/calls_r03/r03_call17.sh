#!/bin/bash
set -x
mkdir -p gpurun_out/c17
cd /root/repo
timeout 300 python tools/exp/rs64_bench.py > gpurun_out/c17/rs32.log 2>&1
DR_BF3_RS64=1 timeout 300 python tools/exp/rs64_bench.py > gpurun_out/c17/rs64.log 2>&1
DR_BF3_RS64=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bf3" 2>&1 | tail -3 > gpurun_out/c17/pytest_rs64.log
cat gpurun_out/c17/rs32.log gpurun_out/c17/rs64.log gpurun_out/c17/pytest_rs64.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c17/bench_rs32.json 2> gpurun_out/c17/bench_rs32.err
DR_BF3_RS64=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c17/bench_rs64.json 2> gpurun_out/c17/bench_rs64.err
for f in gpurun_out/c17/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], sorted(((k,v) for k,v in d.get("kernels_us",{}).items()), key=lambda kv:-kv[1])[:5])
PY
done
