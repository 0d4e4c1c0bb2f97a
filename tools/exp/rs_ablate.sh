#!/bin/bash
# Ablations of the register-split GEMM (forward layer 0 and dgrad shapes), both wave shapes.  Needs the library built with
# -DDR_BF3_ABLATE (`bash tools/exp/rs_ablate.sh build` in the container writes tools/exp/_alt/libdr_hotpath_ablate.so, which travels with
# the snapshot); then `gpurun -- bash tools/exp/rs_ablate.sh` on the GPU box.  RS_MS / RS_DBG select wave shapes / ablations.
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then      # in the container, before the gpurun call: the library with the ablation launches compiled in
  mkdir -p tools/exp/_alt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=131072 -DDR_BF3_ABLATE \
    -Iinclude -Ideep_recommenders_amd/csrc -c deep_recommenders_amd/csrc/bf3_gemm.hip -o /tmp/bf3_abl.o || exit 1
  objs=$(ls deep_recommenders_amd/lib/*.o | grep -v bf3_gemm.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_alt/libdr_hotpath_ablate.so $objs /tmp/bf3_abl.o
  exit $?
fi
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/new.so
cp tools/exp/_alt/libdr_hotpath_ablate.so $L
for ms in ${RS_MS:-0 1}; do
  for dbg in ${RS_DBG:-0 32 64 96 2 1}; do
    DR_BF3_RS64=$ms DR_BF3_RS_DBG=$dbg timeout 200 python tools/exp/rs64_bench.py 2>/dev/null | grep -E "forward|dgrad layer|square" | awk -v m=$ms -v d=$dbg '{for (i = 1; i <= NF; ++i) if ($i == "us") printf "rs64=%s dbg=%-3s %-24s %s us\n", m, d, $1" "$2" "$3, $(i-1)}'
  done
done
cp /tmp/new.so $L
