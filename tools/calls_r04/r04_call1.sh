#!/bin/bash
# K4's ablation ladder + this box's baseline bench lines (default, sharded world 1)
cd /root/repo
mkdir -p gpurun_out/r04
timeout 900 python tools/exp/k4_ladder.py > gpurun_out/r04/k4_ladder.log 2>&1
tail -5 gpurun_out/r04/k4_ladder.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_base.json 2>gpurun_out/r04/bench_base.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_sharded_base.json 2>gpurun_out/r04/bench_sharded_base.err
python - <<'PY'
import json
for f in ("bench_base", "bench_sharded_base"):
    try:
        d = json.loads(open("gpurun_out/r04/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], [(r["kernel"], r["avg_us"]) for r in d["roofline_all"]][:8])
    except Exception as e:
        print(f, "failed", e)
PY
