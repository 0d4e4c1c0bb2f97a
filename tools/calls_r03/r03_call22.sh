#!/bin/bash
mkdir -p gpurun_out/c22
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
for det in 1 0; do
DR_K4_DETERMINISTIC=$det timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c22/bench_det${det}.json
DR_K4_DETERMINISTIC=$det timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>/dev/null > gpurun_out/c22/bench_zipf_det${det}.json
done
for f in gpurun_out/c22/bench_*det*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], [(r["kernel"], r["avg_us"]) for r in d["roofline_all"] if "hot" in r["kernel"] or "pool" in r["kernel"]])
PY
done
