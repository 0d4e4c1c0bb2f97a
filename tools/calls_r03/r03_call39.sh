#!/bin/bash
mkdir -p gpurun_out/c39
cd /root/repo
timeout 400 python bench.py > gpurun_out/c39/line_default.log 2>&1
tail -1 gpurun_out/c39/line_default.log | cut -c1-300
