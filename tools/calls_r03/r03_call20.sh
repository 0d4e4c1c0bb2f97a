#!/bin/bash
mkdir -p gpurun_out/c20
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sorted" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_benchcfg.py -m gpu -x -q -k "zipf" 2>&1 | tail -3
for v in a b; do
for det in 1 0; do
DR_K4_DETERMINISTIC=$det timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c20/bench_det${det}_$v.json
DR_K4_DETERMINISTIC=$det timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>/dev/null > gpurun_out/c20/bench_zipf_det${det}_$v.json
done
done
for f in gpurun_out/c20/bench_*det*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
done
