#!/bin/bash
# K4 (emb_bwd_sorted_kernel) at other occupancies / unrolls: builds variant libraries (run with "build" in the container), then
# (on the GPU box) swaps each in and runs the bench.  Variants: minwaves 5, 6, 8 (register caps 96 / 80 / 64), U = 2, 8.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -pragma-unroll-threshold=131072 -Iinclude -Ideep_recommenders_amd/csrc"
VARS="mw5:-DDR_K4_MINWAVES=5 mw6:-DDR_K4_MINWAVES=6 mw8:-DDR_K4_MINWAVES=8 u2:-DDR_K4_U=2 u8:-DDR_K4_U=8 u2mw8:-DDR_K4_U=2,-DDR_K4_MINWAVES=8"
if [ "$1" = build ]; then
  mkdir -p tools/exp/_alt
  objs=$(ls deep_recommenders_amd/lib/*.o | grep -v emb_sorted.o | tr '\n' ' ')
  for v in $VARS; do
    n=${v%%:*}; d=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc $FLAGS $d -c deep_recommenders_amd/csrc/emb_sorted.hip -o /tmp/es_$n.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "emb_bwd_sorted_kernelILi16ELi[0-9]ELb0E" | grep -E "VGPRs:|Occupancy|VGPRs Spill" | sed "s/.*remark: */$n /" | tr '\n' ' '; echo
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_alt/libdr_hotpath_k4_$n.so $objs /tmp/es_$n.o
  done
  exit 0
fi
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/new.so
mkdir -p gpurun_out/k4occ
for rep in 1 2; do
for n in base mw5 mw6 mw8 u2 u8 u2mw8; do
  if [ $n = base ]; then cp /tmp/new.so $L; else cp tools/exp/_alt/libdr_hotpath_k4_$n.so $L; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 30 2>/dev/null > gpurun_out/k4occ/bench_${n}_$rep.json
  python - gpurun_out/k4occ/bench_${n}_$rep.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], [(r["kernel"], r["avg_us"]) for r in d["roofline_all"] if r["kernel"]=="emb_pool_bwd"])
PY
done
done
cp /tmp/new.so $L
