#!/bin/bash
set -x
mkdir -p gpurun_out/c3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c3/prof_plan -o plan -- python $GRAFT_REPO_ROOT/tools/exp/plan_bench.py > $GRAFT_REPO_ROOT/gpurun_out/c3/plan_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob, re
f=glob.glob('gpurun_out/c3/prof_plan/*.db')[0]
db=sqlite3.connect(f); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=cur.execute(f"select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start), max(k.end-k.start) from {kd} k join {ks} s on k.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
for r in rows:
    n=re.sub(r'\(.*','',r[0])[:70]
    print("%-70s n=%5d avg=%9.1f us min=%9.1f max=%9.1f"%(n,r[1],r[2]/1e3,r[3]/1e3,r[4]/1e3))
PY
rm -rf gpurun_out/c3/prof_plan
