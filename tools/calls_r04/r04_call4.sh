#!/bin/bash
# nontemporal hints in the three wide GEMMs (fused forward's row gather, wgrad's row gather, dgrad's output stores): A/B each
cd /root/repo
mkdir -p gpurun_out/r04
A=tools/exp/_alt
bash tools/exp/ab_multi.sh 2 $A/libdr_hotpath_nt_fwd.so $A/libdr_hotpath_nt_wgrad.so $A/libdr_hotpath_nt_dgrad.so $A/libdr_hotpath_nt_all.so -- 2>&1 | tee gpurun_out/r04/ab_nt_gemm.log
timeout 600 python -m pytest tests/test_bench_launch.py -m gpu -q -x 2>&1 | tail -15
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_sharded_1.json 2>gpurun_out/r04/bench_sharded_1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_sharded_1.json").read().strip().splitlines()[-1])
print("sharded world1", d["ms_per_step"], json.dumps(d.get("exchange")))
PY
