#!/bin/bash
set -x
mkdir -p gpurun_out/c10
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/c10/pytest.log
tail -3 gpurun_out/c10/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c10/bench_single.json 2> gpurun_out/c10/bench_single.err
timeout 300 python bench.py --no-cpu-baseline --optimizer adam > gpurun_out/c10/bench_single_adam.json 2> gpurun_out/c10/bench_single_adam.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c10/bench_sharded_mb2.json 2> gpurun_out/c10/bench_sharded_mb2.err
timeout 400 python bench.py --model dcn --no-cpu-baseline > gpurun_out/c10/bench_dcn.json 2> gpurun_out/c10/bench_dcn.err
timeout 300 python bench.py --preset c2 --no-cpu-baseline > gpurun_out/c10/bench_c2.json 2> gpurun_out/c10/bench_c2.err
for f in gpurun_out/c10/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], {k:v["avg_us"] for k,v in d.get("exchange_phases",{}).items()})
    if d.get("roofline"): print("   headline:", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_with_plan_charged"), {k:v["event_us_while_overlapped"] for k,v in d.get("overlapped_side_stream",{}).items()})
    for r in d.get("roofline_all",[])[:9]: print("   ", r["kernel"], r["avg_us"], r["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
    print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
