#!/bin/bash
mkdir -p gpurun_out/c36
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py -m gpu -x -q -k "sorted or missing or empty" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -6
timeout 1200 python -m pytest tests/test_gpu_benchcfg.py tests/test_gpu_models.py -m gpu -x -q -k "deepfm or engine" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -6
for v in a b; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [(r['kernel'][:20], r['avg_us']) for r in d['roofline_all']])"
DR_K4_DETERMINISTIC=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('det0', d['ms_per_step'])"
done
timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('zipf', d['ms_per_step'])"
