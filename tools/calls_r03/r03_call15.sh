#!/bin/bash
set -x
mkdir -p gpurun_out/c15
cd $GRAFT_REPO_ROOT
for v in on off on2 off2; do
  case $v in on*) E=on;; off*) E=off;; esac
  timeout 300 python bench.py --no-cpu-baseline --events $E > gpurun_out/c15/bench_$v.json 2> gpurun_out/c15/bench_$v.err
done
for f in gpurun_out/c15/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"])
PY
done
