#!/bin/bash
# sharded world-1: per-micro-batch wgrads (DR_SH_WGRAD_SPLIT=1) vs one wgrad at the end; two-rank parity with the split on
cd /root/repo
mkdir -p gpurun_out/r04
run() { tag=$1; shift; env "$@" DR_FORCE_SHARDED=1 DR_BENCH_EVENTS=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json; python - $tag <<'PY'
import json,sys
d=json.loads(open("/tmp/l.json").read())
print("SH", sys.argv[1], d["ms_per_step"], d["config"]["final_loss"], flush=True)
PY
}
run warm A=1
for rep in 1 2 3; do
  run whole DR_SH_WGRAD_SPLIT=0
  run split DR_SH_WGRAD_SPLIT=1
  run split_k4first DR_SH_WGRAD_SPLIT=1 DR_SH_K4_FIRST=1
done
DR_SH_WGRAD_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_sharded_two_rank.py -m gpu -q -x -k "deepfm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "emb_pool_fwd or topk or pack or rows_gather or shard" 2>&1 | tail -3
