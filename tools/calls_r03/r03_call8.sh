#!/bin/bash
set -x
mkdir -p gpurun_out/c8
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "adam" 2>&1 | tail -15; cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c8/prof -o tk -- python $GRAFT_REPO_ROOT/tools/exp/topk_bench.py > $GRAFT_REPO_ROOT/gpurun_out/c8/topk.log 2>&1
cd $GRAFT_REPO_ROOT
grep topk_mips gpurun_out/c8/topk.log
python - <<'PY'
import sqlite3, glob, re
f=glob.glob('gpurun_out/c8/prof/*.db')[0]
db=sqlite3.connect(f); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=cur.execute(f"select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start), max(k.end-k.start), sum(k.end-k.start) from {kd} k join {ks} s on k.kernel_id=s.id group by s.kernel_name order by 6 desc").fetchall()
for r in rows[:10]:
    n=re.sub(r'\(.*','',r[0])[:60]
    print("%-60s n=%5d avg=%8.1f min=%8.1f max=%8.1f total_ms=%8.2f"%(n,r[1],r[2]/1e3,r[3]/1e3,r[4]/1e3,r[5]/1e6))
rows=cur.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {ks} s on k.kernel_id=s.id order by k.start").fetchall()
# last 12 kernels: gaps
for a,b in zip(rows[-14:-1], rows[-13:]):
    print("%-40s dur %7.1f  gap to next %6.1f"%(re.sub(r'\(.*','',a[2])[-40:], (a[1]-a[0])/1e3, (b[0]-a[1])/1e3))
PY
rm -rf gpurun_out/c8/prof
