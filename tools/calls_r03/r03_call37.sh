#!/bin/bash
# the default bench line five times in a row on one box (final tree): run-to-run spread of the step and of K4's event time
mkdir -p gpurun_out/c37
cd /root/repo
for i in 1 2 3 4 5; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c37/line_$i.json
python - gpurun_out/c37/line_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
print(d["ms_per_step"], r["kernel"], r["avg_us"], r["frac"], r.get("frac_with_plan_alone_charged"))
PY
done
