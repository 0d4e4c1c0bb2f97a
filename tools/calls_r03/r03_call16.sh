#!/bin/bash
set -x
mkdir -p gpurun_out/c16
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "engine" 2>&1 | tail -3
for v in a b off; do
  E=auto; [ $v = off ] && E=off
  timeout 300 python bench.py --no-cpu-baseline --events $E > gpurun_out/c16/bench_$v.json 2> gpurun_out/c16/bench_$v.err
done
for f in gpurun_out/c16/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"] and (d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launches"]), [(r["kernel"][:16], r["avg_us"]) for r in d.get("roofline_all",[])[:4]])
PY
done
