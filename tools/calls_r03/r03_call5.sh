#!/bin/bash
set -x
mkdir -p gpurun_out/c5
cd $GRAFT_REPO_ROOT
timeout 200 python tools/exp/plan_bench.py > gpurun_out/c5/plan_bench.log 2>&1
cat gpurun_out/c5/plan_bench.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sorted or plan or adam" 2>&1 | tail -15 > gpurun_out/c5/pytest_plan.log
tail -5 gpurun_out/c5/pytest_plan.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c5/bench_single.json 2> gpurun_out/c5/bench_single.err
timeout 300 python bench.py --ids zipf --no-cpu-baseline > gpurun_out/c5/bench_single_zipf.json 2> gpurun_out/c5/bench_single_zipf.err
DR_FORCE_SHARDED=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c5/bench_sharded_mb2.json 2> gpurun_out/c5/bench_sharded_mb2.err
for f in gpurun_out/c5/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], {k:v["avg_us"] for k,v in d.get("exchange_phases",{}).items()})
    print("   headline:", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_with_plan_charged"), {k:v["event_us_while_overlapped"] for k,v in d.get("overlapped_side_stream",{}).items()})
    for r in d.get("roofline_all",[])[:8]: print("   ", r["kernel"], r["avg_us"], r["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
    print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
bash tools/exp/ldsdma_fetch_calib.sh
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c5/prof_plan -o plan -- python $GRAFT_REPO_ROOT/tools/exp/plan_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob, re
f=glob.glob('gpurun_out/c5/prof_plan/*.db')[0]
db=sqlite3.connect(f); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=cur.execute(f"select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start), max(k.end-k.start) from {kd} k join {ks} s on k.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
for r in rows[:14]:
    n=re.sub(r'\(.*','',r[0])[:60]
    print("%-60s n=%5d avg=%8.1f min=%8.1f max=%8.1f"%(n,r[1],r[2]/1e3,r[3]/1e3,r[4]/1e3))
PY
rm -rf gpurun_out/c5/prof_plan
