#!/bin/bash
# A/B of an engine switch on the bench box: tools/exp/ab_bench.sh VAR "extra bench args"  -> ms/step with VAR=1 and VAR=0, twice each
var=$1; shift
for rep in 1 2; do
  for f in 1 0; do
    env $var=$f timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$var=$f" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print(sys.argv[1], "ms/step", d["ms_per_step"], "value", d["value"], "loss", d["config"].get("final_loss"))
for k in d.get("kernels", [])[:12]:
    print("     ", k)
PY
  done
done
