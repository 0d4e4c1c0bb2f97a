#!/bin/bash
mkdir -p gpurun_out/c35
cd /root/repo
timeout 300 python bench.py --model dssm --no-cpu-baseline > gpurun_out/c35/line_dssm.log 2>&1
tail -1 gpurun_out/c35/line_dssm.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_sharded_two_rank.py -m gpu -q -x -k "retrieval or tower or topk" 2>&1 | grep -E "passed|failed" | head -3
