#!/bin/bash
set -x
mkdir -p gpurun_out/c11
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/c11/pytest.log
tail -6 gpurun_out/c11/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c11/bench_single.json 2> gpurun_out/c11/bench_single.err
DR_NO_CONCAT=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c11/bench_single_concat.json 2> gpurun_out/c11/bench_single_concat.err
timeout 300 python bench.py --no-cpu-baseline --ids zipf > gpurun_out/c11/bench_single_zipf.json 2> gpurun_out/c11/bench_single_zipf.err
timeout 300 python bench.py --no-cpu-baseline --optimizer adam > gpurun_out/c11/bench_single_adam.json 2> gpurun_out/c11/bench_single_adam.err
for f in gpurun_out/c11/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], d["config"]["final_loss"])
    for r in d.get("roofline_all",[])[:7]: print("   ", r["kernel"], r["avg_us"], r["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
    print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
