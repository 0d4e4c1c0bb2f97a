#!/bin/bash
# EXPERIMENT (prepared at the end of round 4, never run: the GPU budget was gone): the top-K filter epilogue with a per-register early
# reject (-DDR_TOPK_FAST_REJECT=1, csrc/bf3_gemm.hip EPI == 4).  Expected from profiles/r04_topk_kernel_stats.csv: the 31 steady-state
# scan launches at 312-320 us (19.5 us per tile) move towards the ~12 us per tile of matrix + ingest time, the exact top-100 of config 5
# from 13.8 to ~11 ms.
#   in the container:   bash tools/exp/topk_fast_reject.sh build
#   on the GPU box:     gpurun --timeout 600 -- bash tools/exp/topk_fast_reject.sh
# The GPU part runs the retrieval parity tests against the alternative library FIRST (ties, top-k order and the oracle comparisons must
# not change: the reject only skips registers in which no column passes), then times both libraries alternately.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "$1" = build ]; then
  mkdir -p tools/exp/_alt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=131072 -DDR_TOPK_FAST_REJECT=1 \
    -Iinclude -Ideep_recommenders_amd/csrc -c deep_recommenders_amd/csrc/bf3_gemm.hip -o /tmp/bf3_fr.o || exit 1
  objs=$(ls deep_recommenders_amd/lib/*.o | grep -v bf3_gemm.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_alt/libdr_hotpath_fastreject.so $objs /tmp/bf3_fr.o
  exit $?
fi
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/fr_new.so
cp tools/exp/_alt/libdr_hotpath_fastreject.so $L
timeout -s KILL 500 python -m pytest tests/test_gpu_retrieval.py tests/test_gpu_benchcfg.py -q -m gpu -k "topk or top_k or retrieval or brute or streaming" 2>&1 | tail -4
for rep in 1 2; do
  cp /tmp/fr_new.so $L; echo -n "shipped      "; timeout -s KILL 200 python tools/exp/topk_prof.py 2>/dev/null | grep topk
  cp tools/exp/_alt/libdr_hotpath_fastreject.so $L; echo -n "fast reject  "; timeout -s KILL 200 python tools/exp/topk_prof.py 2>/dev/null | grep topk
done
cp /tmp/fr_new.so $L
