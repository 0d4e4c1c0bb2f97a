#!/bin/bash
# nontemporal hints in the shipped K4: ladder (S now carries them), correctness tests, A/B of the bench step against a DR_K4_NT=0 build
cd /root/repo
mkdir -p gpurun_out/r04
timeout 600 python tools/exp/k4_ladder.py > gpurun_out/r04/k4_ladder3.log 2>&1
grep K4LADDER gpurun_out/r04/k4_ladder3.log | tail -40
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "emb_bwd_sorted or emb_pool_bwd" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_benchcfg.py -m gpu -q -x -k "sgd_uniform or reproducible or adam or zipf" 2>&1 | tail -3
bash tools/exp/ab_lib.sh tools/exp/_alt/libdr_hotpath_k4nt0.so 2 2>&1 | tee gpurun_out/r04/ab_k4nt.log
bash tools/exp/ab_lib.sh tools/exp/_alt/libdr_hotpath_k4nt0.so 1 --ids zipf 2>&1 | tee gpurun_out/r04/ab_k4nt_zipf.log
bash tools/exp/ab_lib.sh tools/exp/_alt/libdr_hotpath_k4nt0.so 1 --optimizer adam 2>&1 | tee gpurun_out/r04/ab_k4nt_adam.log
