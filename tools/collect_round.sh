#!/bin/bash
# Everything the round's profiles/ needs, in one gpurun call: full GPU tests, rocprofv3 stats + PMC traffic of the default
# bench command, the bench lines of every configuration.  Results land in gpurun_out/; copy them into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
# the driver's own command first (-x), then nothing is hidden: the full -rA report is kept
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu -rA 2>&1 | tail -420 > gpurun_out/pytest_gpu_full.log
grep -E "passed|failed" gpurun_out/pytest_gpu_full.log | tail -2 | tee gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu_full.log | head
bash tools/collect_profiles.sh
cd $R
# SQ counters (wave-cycle split, MFMA busy cycles, LDS activity) of the default kernels
bash tools/collect_sq_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_summary.txt gpurun_out/pmc_sq_summary_default.txt
cd $R
timeout -s KILL 400 python bench.py > gpurun_out/line_default.log 2>&1
timeout -s KILL 200 python bench.py --optimizer adam --no-cpu-baseline > gpurun_out/line_adam.log 2>&1
timeout -s KILL 200 python bench.py --ids zipf --no-cpu-baseline > gpurun_out/line_zipf.log 2>&1
timeout -s KILL 200 python bench.py --preset c2 --no-cpu-baseline > gpurun_out/line_c2.log 2>&1
timeout -s KILL 400 python bench.py --model dcn > gpurun_out/line_dcn.log 2>&1
timeout -s KILL 400 python bench.py --model dssm > gpurun_out/line_dssm.log 2>&1
DR_FUSE_K3=0 DR_PREFETCH_PLAN=0 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_unfused.log 2>&1
DR_NO_CONCAT=0 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_concat.log 2>&1
DR_FORCE_SHARDED=1 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_sharded_world1.log 2>&1
DR_FORCE_SHARDED=1 timeout -s KILL 300 python bench.py --model dcn --no-cpu-baseline > gpurun_out/line_dcn_sharded_world1.log 2>&1
DR_PREFETCH_EARLY=0 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_plan_beside_k4.log 2>&1
# round 6 (second half): the prefetch chain as rounds 3 - 5 scheduled it (start of the step, K1 / transpose / plan as three calls, one side stream)
DR_PREFETCH_EARLY=1 DR_FUSE_PLAN_FRONT=0 DR_SIDE_R=0 DR_BENCH_STRICT=0 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_chain_r5_schedule.log 2>&1
# what the chain costs the step (tools/exp/exp_plan.sh: the chain not launched at all, timing only), and one step as a timeline
bash tools/exp/exp_plan.sh > gpurun_out/plan_cost.log 2>&1
bash tools/exp/timeline_call.sh default bf3_emb_linear_kernel -- > /dev/null 2>&1
bash tools/exp/timeline_call.sh sharded bf3_gemm_tn_rs_kernel DR_FORCE_SHARDED=1 -- > /dev/null 2>&1
cd $R
# round 6: the three-kernel backward of round 5 (dgrad, wgrad, K4) beside the fused default; the 8-wave GEMM kernel everywhere
DR_FUSE_K4=0 DR_BENCH_STRICT=0 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_k4_unfused.log 2>&1
DR_H2_OCC=0 DR_BENCH_STRICT=0 timeout -s KILL 300 python bench.py --model dcn --no-cpu-baseline > gpurun_out/line_dcn_8wave.log 2>&1
# round 6 experiment drivers: the 16-wave kernel against the 8-wave one (bit-identity + times), the fused dgrad + K4 against dgrad + K4
timeout -s KILL 300 python tools/exp/occ_bench.py 2>&1 | grep "OCC=" > gpurun_out/occ_bench.log
timeout -s KILL 300 python tools/exp/fused_k4_bench.py 2>&1 | grep FUSEDK4 > gpurun_out/fused_k4_bench.log
timeout -s KILL 300 python tools/exp/fused_k4_bench.py 2000000 zipf 2>&1 | grep FUSEDK4 >> gpurun_out/fused_k4_bench.log
# the six-product mode of rounds 2-3 beside the default (f16x2) lines: default, DCN, sharded
DR_GEMM_SPLIT=bf16x3 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_bf16x3.log 2>&1
DR_GEMM_SPLIT=bf16x3 timeout -s KILL 300 python bench.py --model dcn --no-cpu-baseline > gpurun_out/line_dcn_bf16x3.log 2>&1
DR_GEMM_SPLIT=bf16x3 DR_FORCE_SHARDED=1 timeout -s KILL 200 python bench.py --no-cpu-baseline > gpurun_out/line_sharded_world1_bf16x3.log 2>&1
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for f in default adam zipf c2 dcn dssm unfused concat sharded_world1 dcn_sharded_world1 plan_beside_k4 chain_r5_schedule k4_unfused dcn_8wave bf16x3 dcn_bf16x3 sharded_world1_bf16x3; do
  python - $f <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/line_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1], d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
# the raw per-dispatch traces are tens of MB (gpurun merges at most 64 MiB back): the summaries above are what profiles/ keeps
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_sq gpurun_out/prof_stats gpurun_out/tl_default gpurun_out/tl_sharded
du -sh gpurun_out
