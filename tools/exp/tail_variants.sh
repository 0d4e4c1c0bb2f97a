#!/bin/bash
# EXPERIMENT: variants / ablations of the one-pass tower tail (csrc/tower_tail.hip, -DDR_TAIL_DBG=<bits>), one library each.
#   in the container:  bash tools/exp/tail_variants.sh build "0 1 2 3 16 32 64 128"
#   on the GPU box:    bash tools/exp/tail_variants.sh "0 1 2 3 16 32 64 128"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  mkdir -p tools/exp/_alt
  for v in $2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=131072 -DDR_TAIL_DBG=$v \
      -Rpass-analysis=kernel-resource-usage -Iinclude -Ideep_recommenders_amd/csrc -c deep_recommenders_amd/csrc/tower_tail.hip -o /tmp/tt_$v.o 2>&1 \
      | grep -A8 "tower_tail_fused_kernelILi8" | grep -E "VGPRs:|VGPRs Spill|Scratch" | tr '\n' ' '; echo " <- DBG=$v"
    objs=$(ls deep_recommenders_amd/lib/*.o | grep -v tower_tail.o | tr '\n' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_alt/libdr_hotpath_tail$v.so $objs /tmp/tt_$v.o || exit 1
  done
  exit 0
fi
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/tail_new.so
for v in $1; do
  cp tools/exp/_alt/libdr_hotpath_tail$v.so $L
  echo -n "DBG=$v  "; timeout -s KILL 120 python tools/exp/tail_bench.py 2>/dev/null | grep TAIL
done
cp /tmp/tail_new.so $L
