#!/bin/bash
cd /root/repo
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], {k: r.get(k) for k in ('frac','avg_us','kernel_alone_us','frac_kernel_alone','plan_alone_us','frac_with_plan_alone_charged')})"
done
timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('zipf', d['ms_per_step'], {k: r.get(k) for k in ('frac','avg_us','kernel_alone_us','frac_kernel_alone','plan_alone_us')})"
