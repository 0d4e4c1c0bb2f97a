#!/bin/bash
mkdir -p gpurun_out/c18
cd /root/repo
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/new.so
for rep in 1 2; do
  for v in head new; do
    if [ $v = head ]; then cp tools/exp/_alt/libdr_hotpath_head.so $L; else cp /tmp/new.so $L; fi
    timeout 300 python tools/exp/rs64_bench.py 2>/dev/null | grep -E "forward|dgrad layer|square" | sed "s/^/$v $rep /"
    timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c18/bench_${v}_$rep.json
    python - gpurun_out/c18/bench_${v}_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"])
PY
  done
done
cp /tmp/new.so $L
