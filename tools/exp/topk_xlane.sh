#!/bin/bash
# EXPERIMENT (round 5): the sorted top-k list's cross-lane steps (csrc/topk_list.h, DR_TOPK_XLANE) and the selection kernel's loads.
#   old : the list as written in round 1 (every cross-lane step a ds_bpermute; one 256-byte load per 64 entries, waited for on the spot)
#   x0  : the new selection loop (4 x 64 entries per round, next round's loads in flight) with the ds_bpermute list
#   new : that loop + readlane / ballot-popcount / DPP wave_shr list (what the tree ships if it wins)
#   in the container:   bash tools/exp/topk_xlane.sh build
#   on the GPU box:     bash tools/exp/topk_xlane.sh
# The GPU part runs the retrieval / IVF parity tests on the new library FIRST, then times the three libraries alternately and prints a
# fingerprint of the top-100 result of each (must be equal).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=131072"
if [ "$1" = build ]; then
  mkdir -p tools/exp/_alt /tmp/topk_old
  for f in topk_list.h retrieval.hip ivf.hip; do git show ${OLD_REV:-2186acb}:deep_recommenders_amd/csrc/$f > /tmp/topk_old/$f; done
  others=$(ls deep_recommenders_amd/lib/*.o | grep -v -e retrieval.o -e ivf.o | tr '\n' ' ')
  for f in retrieval ivf; do
    $CC -Iinclude -I/tmp/topk_old -Ideep_recommenders_amd/csrc -c /tmp/topk_old/$f.hip -o /tmp/topk_old_$f.o || exit 1
    $CC -DDR_TOPK_XLANE=0 -Iinclude -Ideep_recommenders_amd/csrc -c deep_recommenders_amd/csrc/$f.hip -o /tmp/topk_x0_$f.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_alt/libdr_hotpath_topk_old.so $others /tmp/topk_old_retrieval.o /tmp/topk_old_ivf.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_alt/libdr_hotpath_topk_x0.so $others /tmp/topk_x0_retrieval.o /tmp/topk_x0_ivf.o || exit 1
  ls -la tools/exp/_alt/libdr_hotpath_topk_*.so
  exit 0
fi
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/xl_new.so
timeout -s KILL 500 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_kernels.py tests/test_gpu_benchcfg.py tests/test_gpu_fullsize.py -q -m gpu \
  -k "topk or top_k or retrieval or brute or streaming or ivf or faiss or select" 2>&1 | tail -4
for rep in 1 2; do
  for v in old x0 new; do
    if [ $v = new ]; then cp /tmp/xl_new.so $L; else cp tools/exp/_alt/libdr_hotpath_topk_$v.so $L; fi
    echo -n "$v   "; timeout -s KILL 200 python tools/exp/topk_prof.py 2>/dev/null | grep topk
  done
done
cp /tmp/xl_new.so $L
