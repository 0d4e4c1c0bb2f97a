#!/bin/bash
# Ablations of the register-split GEMM (forward layer 0 and dgrad shapes), both wave shapes.  Needs the library built with
# -DDR_BF3_ABLATE (tools/exp/_alt/libdr_hotpath_ablate.so); run on the GPU box from the repo root.
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/new.so
cp tools/exp/_alt/libdr_hotpath_ablate.so $L
for ms in ${RS_MS:-0 1}; do
  for dbg in ${RS_DBG:-0 32 64 96 2 1}; do
    DR_BF3_RS64=$ms DR_BF3_RS_DBG=$dbg timeout 200 python tools/exp/rs64_bench.py 2>/dev/null | grep -E "forward|dgrad layer|square" | awk -v m=$ms -v d=$dbg '{for (i = 1; i <= NF; ++i) if ($i == "us") printf "rs64=%s dbg=%-3s %-24s %s us\n", m, d, $1" "$2" "$3, $(i-1)}'
  done
done
cp /tmp/new.so $L
