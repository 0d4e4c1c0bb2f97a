#!/bin/bash
set -x
mkdir -p gpurun_out/c14
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_benchcfg.py -m gpu -x -q -k "deepfm or dcn" 2>&1 | tail -4
for v in pf1 concat pf1b concatb pf1c concatc; do
  case $v in
    pf1*) export DR_WGRAD_PF=1 DR_NO_CONCAT=1;;
    pf0*) export DR_WGRAD_PF=0 DR_NO_CONCAT=1;;
    concat*) export DR_WGRAD_PF=1 DR_NO_CONCAT=0;;
  esac
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c14/bench_$v.json 2> gpurun_out/c14/bench_$v.err
done
for f in gpurun_out/c14/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["config"]["final_loss"], [(r["kernel"][:18], r["avg_us"]) for r in d.get("roofline_all",[])[:4]])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
    print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
