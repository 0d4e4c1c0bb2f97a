#!/bin/bash
# build (so the in-tree .so files match the sources), check the C-ABI symbols, then hand the command to gpurun
set -e
cd /root/repo
python __graft_entry__.py > /dev/null
python -m pytest tests/test_cabi_symbols.py -q -x 2>&1 | tail -1
set +e
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-1500} -- "$@"
