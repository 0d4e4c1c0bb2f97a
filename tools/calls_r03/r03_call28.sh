#!/bin/bash
mkdir -p gpurun_out/c28
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -8
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/c28/line_default.log 2>&1
tail -1 gpurun_out/c28/line_default.log | cut -c1-400
