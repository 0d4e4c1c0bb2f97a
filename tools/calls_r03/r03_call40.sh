#!/bin/bash
cd /root/repo
for args in "--preset c2" "--optimizer adam" "--ids zipf" "--model dcn" "--gemm native"; do
  echo -n "$args: "; timeout 300 python bench.py $args --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
    print(d['ms_per_step'], r.get('kernel'), r.get('frac'), r.get('kernel_alone_us'), r.get('plan_alone_us'))
except Exception as e:
    print('FAILED', e)"
done
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
