#!/bin/bash
mkdir -p gpurun_out/c23
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sorted" 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
timeout 900 python -m pytest tests/test_gpu_benchcfg.py -m gpu -x -q -k "lin_side or three_steps" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
for v in a b; do
for ls in 0 1; do
DR_LIN_SIDE=$ls timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c23/bench_ls${ls}_$v.json
done
done
DR_LIN_SIDE=1 timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>/dev/null > gpurun_out/c23/bench_zipf_ls1.json
DR_LIN_SIDE=0 timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>/dev/null > gpurun_out/c23/bench_zipf_ls0.json
for f in gpurun_out/c23/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], [(r["kernel"][:22], r["avg_us"]) for r in d["roofline_all"]], {k[:30]: v["event_us_while_overlapped"] for k, v in d.get("overlapped_side_stream", {}).items()})
PY
done
