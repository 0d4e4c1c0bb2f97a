#!/bin/bash
# K4 ladder, second pass: nontemporal row loads, first-order weight write-only (old value streamed), Infinity-Cache freshness; random RMW sweep
cd /root/repo
mkdir -p gpurun_out/r04
timeout 900 python tools/exp/k4_ladder.py > gpurun_out/r04/k4_ladder2.log 2>&1
tail -3 gpurun_out/r04/k4_ladder2.log
timeout 300 python tools/exp/exp_rmw.py > gpurun_out/r04/exp_rmw.log 2>&1
cat gpurun_out/r04/exp_rmw.log | grep EXPRMW
