#!/bin/bash
# One rocprofv3 --pmc pass with the SQ occupancy / stall counters over the default bench command (own run: no stats, no traces
# besides the kernel trace).  Output: gpurun_out/pmc_sq/*counter_collection.csv -> tools/sq_pmc_summary.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export DR_BENCH_STRICT=0
rm -rf $R/gpurun_out/pmc_sq
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
tail -2 $R/gpurun_out/pmc_sq.log | cut -c1-200
python $R/tools/sq_pmc_summary.py $(find $R/gpurun_out/pmc_sq -name "*counter_collection.csv") > $R/gpurun_out/pmc_sq_summary.txt 2>&1
cat $R/gpurun_out/pmc_sq_summary.txt
