#!/bin/bash
cd /root/repo
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k:r[k] for k in ('kernel','frac','avg_us','plan_event_us_while_overlapped','frac_with_plan_charged','plan_alone_us','frac_with_plan_alone_charged') if k in r})"
timeout 300 python bench.py --no-cpu-baseline --ids zipf 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k:r[k] for k in ('kernel','frac','avg_us','plan_event_us_while_overlapped','frac_with_plan_charged','plan_alone_us','frac_with_plan_alone_charged') if k in r})"
