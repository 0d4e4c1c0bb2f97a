#!/bin/bash
# Ablations of the f16x2 register-split GEMM (tools/exp/h2_ablate.py), one process per setting.  Build the ablation library in the
# container first (`bash tools/exp/rs_ablate.sh build`), then on the GPU box:  gpurun --timeout 900 -- bash tools/exp/h2_ablate.sh
# H2_DBG selects the settings (default: the single ablations and the no-MFMA combinations added at the end of round 4, which have not
# been run yet: 34 no MFMA + no fragment reads, 3 no MFMA + no weight DMA, 10 no MFMA + A from cache, 98 no MFMA + no reads + no split,
# 35 no MFMA + no reads + no DMA, 43 that + A from cache).
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=deep_recommenders_amd/lib/libdr_hotpath.so
cp $L /tmp/h2abl_new.so
cp tools/exp/_alt/libdr_hotpath_ablate.so $L
for dbg in ${H2_DBG:-0 32 64 96 2 1 34 3 10 98 35 43}; do
  DR_BF3_RS64=0 DR_BF3_RS_DBG=$dbg timeout -s KILL 200 python tools/exp/h2_ablate.py 2>/dev/null | grep H2ABL
done
cp /tmp/h2abl_new.so $L
