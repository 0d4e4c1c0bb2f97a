#!/bin/bash
set -x
mkdir -p gpurun_out/c7
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_two_tower.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_two_rank.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/c7/pytest.log
tail -6 gpurun_out/c7/pytest.log
timeout 300 python bench.py --model dssm --no-cpu-baseline > gpurun_out/c7/bench_dssm.json 2> gpurun_out/c7/bench_dssm.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c7/bench_dssm.json').read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], d.get("metric_pass"))
    for r in d.get("roofline_all",[])[:10]: print("   ", r["kernel"], r["avg_us"], r["frac"])
except Exception as e:
    print("FAILED", e); print(open('gpurun_out/c7/bench_dssm.err').read()[-2000:])
PY
