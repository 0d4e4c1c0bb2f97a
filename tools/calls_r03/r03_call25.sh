#!/bin/bash
mkdir -p gpurun_out/c25
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pack or bf3_linear" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
timeout 1200 python -m pytest tests/test_gpu_sharded_two_rank.py tests/test_gpu_models.py -m gpu -x -q -k "shard" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
for v in a b; do
DR_FUSE_PACK=1 DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c25/bench_sh_fp1_$v.json
DR_FUSE_PACK=0 DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/c25/bench_sh_fp0_$v.json
done
for f in gpurun_out/c25/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], [(r["kernel"][:22], r["avg_us"]) for r in d["roofline_all"]], {k[:24]: v["avg_us"] for k, v in d.get("exchange_phases", {}).items()})
PY
done
