#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -k "topk or top_k or retrieval or two_tower or brute or streaming or dssm" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head -8
timeout 300 python tools/exp/topk_bench.py 2>&1 | tail -1
timeout 300 python bench.py --model dssm --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['metric_pass'])"
