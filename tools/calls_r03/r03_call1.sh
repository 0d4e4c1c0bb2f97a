#!/bin/bash
# round 3, GPU call 1: full GPU test suite (incl. the new two-ranks-on-one-GPU tests), bench lines single / sharded world-1
set -x
mkdir -p gpurun_out/c1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/c1/pytest.log
timeout 300 python bench.py > gpurun_out/c1/bench_single.json 2> gpurun_out/c1/bench_single.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c1/bench_sharded_mb2.json 2> gpurun_out/c1/bench_sharded_mb2.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline --micro-batches 1 > gpurun_out/c1/bench_sharded_mb1.json 2> gpurun_out/c1/bench_sharded_mb1.err
DR_FORCE_SHARDED=1 DR_FUSE_K3=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c1/bench_sharded_mb2_unfused.json 2> gpurun_out/c1/bench_sharded_mb2_unfused.err
cd /tmp && export TMPDIR=/tmp
DR_FORCE_SHARDED=1 timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1/prof_sharded -o sh -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/c1/prof_sharded.log 2>&1
cd $GRAFT_REPO_ROOT
# keep only the summaries
find gpurun_out/c1/prof_sharded -name "*kernel_trace.csv" -size +20M -delete
tail -3 gpurun_out/c1/pytest.log
for f in gpurun_out/c1/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], {k:v["avg_us"] for k,v in d.get("exchange_phases",{}).items()})
    for r in d.get("roofline_all",[])[:8]: print("   ", r["kernel"], r["avg_us"], r["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
