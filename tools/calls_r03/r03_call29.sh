#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q 2>&1 | tail -30
