#!/bin/bash
# A/B of the shipped libdr_hotpath.so against an alternative build (tools/exp/_alt/<name>.so), alternating, inside ONE gpurun call.
# usage: ab_lib.sh <alt .so> <reps> [bench args...]      prints ms/step and the main kernels' event times per run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ALT=$1; REPS=$2; shift 2
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/ab_new.so
for rep in $(seq 1 $REPS); do
  for n in new alt; do
    if [ $n = new ]; then cp /tmp/ab_new.so $L; else cp $ALT $L; fi
    timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - $n <<'PY'
import json,sys
try:
    d=json.loads(open("/tmp/ab_line.json").read())
    print("AB", sys.argv[1], d["ms_per_step"], [(r["kernel"], r["avg_us"]) for r in d["roofline_all"][:6]], flush=True)
except Exception as e:
    print("AB", sys.argv[1], "failed", e)
PY
  done
done
cp /tmp/ab_new.so $L
