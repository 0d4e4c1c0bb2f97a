#!/bin/bash
# round 3: the judged profiles of the default bench command (kernel stats, timed stats, PMC traffic) + the step's gap picture
set -x
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh
cat gpurun_out/pmc_traffic.txt | head -30
TRACE=$(ls -t $(find gpurun_out/prof_stats -name "*kernel_trace.csv") | head -1)
python tools/trace_gaps.py "$TRACE" bf3_emb_linear_kernel -3 8 > gpurun_out/step_gaps.txt 2>&1
head -80 gpurun_out/step_gaps.txt
rm -rf gpurun_out/prof_stats gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
