#!/bin/bash
# sharded world-1 step: why 4.07 ms in call 4 (2.12 in call 1)?  repeat in isolation, events off, then A/B the nontemporal shard kernels
cd /root/repo
mkdir -p gpurun_out/r04
run() { tag=$1; shift; env "$@" DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json; python - $tag <<'PY'
import json,sys
d=json.loads(open("/tmp/l.json").read()); ex=d.get("exchange") or {}
print("SH", sys.argv[1], d["ms_per_step"], [(r["kernel"], r["avg_us"]) for r in d["roofline_all"][:5]], {k:v["avg_us"] for k,v in (d.get("exchange_phases") or {}).items()}, ex.get("exposed_parts_us_per_step"), flush=True)
PY
}
run first A=1
run second A=1
run noevents DR_BENCH_EVENTS=0
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/base.so
cp tools/exp/_alt/libdr_hotpath_shardnt0.so $L
run shardnt0 A=1
run shardnt0_noev DR_BENCH_EVENTS=0
cp /tmp/base.so $L
run base_again A=1
run base_noev DR_BENCH_EVENTS=0
