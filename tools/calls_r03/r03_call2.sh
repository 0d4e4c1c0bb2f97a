#!/bin/bash
# round 3, GPU call 2: new slot plan (no rocPRIM), sharded side-stream forwards, bench-config parity tests, bench lines
set -x
mkdir -p gpurun_out/c2
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/c2/pytest.log
tail -5 gpurun_out/c2/pytest.log
timeout 200 python tools/exp/plan_bench.py > gpurun_out/c2/plan_bench.log 2>&1
cat gpurun_out/c2/plan_bench.log
timeout 300 python bench.py > gpurun_out/c2/bench_single.json 2> gpurun_out/c2/bench_single.err
timeout 300 python bench.py --ids zipf --no-cpu-baseline > gpurun_out/c2/bench_single_zipf.json 2> gpurun_out/c2/bench_single_zipf.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c2/bench_sharded_mb2.json 2> gpurun_out/c2/bench_sharded_mb2.err
DR_FORCE_SHARDED=1 DR_FWD_STREAMS=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c2/bench_sharded_mb2_nofwdstreams.json 2> gpurun_out/c2/bench_sharded_mb2_nofwdstreams.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --micro-batches 1 > gpurun_out/c2/bench_sharded_mb1.json 2> gpurun_out/c2/bench_sharded_mb1.err
for f in gpurun_out/c2/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], {k:v["avg_us"] for k,v in d.get("exchange_phases",{}).items()})
    print("   headline:", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_with_plan_charged"), d.get("overlapped_side_stream"))
    for r in d.get("roofline_all",[])[:8]: print("   ", r["kernel"], r["avg_us"], r["frac"])
    if "cpu_baseline" in d: print("   cpu:", d["cpu_baseline"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
    print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
