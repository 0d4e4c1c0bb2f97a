#!/bin/bash
# sharded world-1 step with the clock-stamp stall measurement; A/B of the nontemporal shard kernels (2 rounds)
cd /root/repo
mkdir -p gpurun_out/r04
run() { tag=$1; shift; env "$@" DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json; python - $tag <<'PY'
import json,sys
d=json.loads(open("/tmp/l.json").read()); ex=d.get("exchange") or {}
print("SH", sys.argv[1], d["ms_per_step"], [(r["kernel"], r["avg_us"]) for r in d["roofline_all"][:5]], {k:v["avg_us"] for k,v in (d.get("exchange_phases") or {}).items()}, ex.get("exposed_parts_us_per_step"), flush=True)
PY
}
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/base.so
for rep in 1 2; do
  run base_ev A=1
  run base_noev DR_BENCH_EVENTS=0
  cp tools/exp/_alt/libdr_hotpath_shardnt0.so $L
  run shardnt0_ev A=1
  run shardnt0_noev DR_BENCH_EVENTS=0
  cp /tmp/base.so $L
done
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_sharded_2.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_2.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r04/bench_2.json').read().strip().splitlines()[-1]); print('single', d['ms_per_step'], d['roofline'])"
