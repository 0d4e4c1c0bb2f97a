#!/bin/bash
mkdir -p gpurun_out/c27
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "rs64 or bf3_linear" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -5
for rep in 1 2; do
for f in 0 2 1; do
DR_BF3_RS64=$f timeout 300 python tools/exp/rs64_bench.py 2>/dev/null | grep -E "forward|dgrad layer|dgrad with|square|accum" | sed "s/^/rs64=$f /" | cut -c1-110
done
done
for rep in 1 2; do
for f in 0 2; do
DR_BF3_RS64=$f timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c27/bench_$f_$rep.json
python - gpurun_out/c27/bench_$f_$rep.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("rs64=%s"%sys.argv[2], d["ms_per_step"], [(r["kernel"][:20], r["avg_us"]) for r in d["roofline_all"][:4]])
PY
done
done
