#!/bin/bash
cd /root/repo
for cfg in "256 0" "256 32768" "256 65536" "64 32768" "32 32768" "16 32768" "64 65536" "32 65536"; do
  set -- $cfg
  echo -n "first_mb=$1 scan_cols=$2: "
  DR_TOPK_FIRST_MB=$1 DR_TOPK_SCAN_COLS=$2 timeout 300 python tools/exp/topk_bench.py 2>&1 | tail -1
done
