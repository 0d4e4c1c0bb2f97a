#!/bin/bash
# A/B/C... of the shipped libdr_hotpath.so ("base") against several alternative builds, round-robin, inside ONE gpurun call.
# usage: ab_multi.sh <reps> <alt1.so> [<alt2.so> ...] -- [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
REPS=$1; shift
ALTS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ALTS+=("$1"); shift; done
[ "$1" = "--" ] && shift
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/ab_base.so
for rep in $(seq 1 $REPS); do
  for n in base "${ALTS[@]}"; do
    if [ $n = base ]; then cp /tmp/ab_base.so $L; else cp $n $L; fi
    timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - $(basename $n .so | sed 's/libdr_hotpath_//') <<'PY'
import json,sys
try:
    d=json.loads(open("/tmp/ab_line.json").read())
    print("AB %-12s %.4f ms  " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%s %.0f" % (r["kernel"].replace("linear_","").replace("emb_",""), r["avg_us"]) for r in d["roofline_all"][:7]), flush=True)
except Exception as e:
    print("AB", sys.argv[1], "failed", e)
PY
  done
done
cp /tmp/ab_base.so $L
