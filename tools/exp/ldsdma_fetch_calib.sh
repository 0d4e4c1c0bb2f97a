#!/bin/bash
# builds and runs tools/exp/ldsdma_fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE; prints counter / known bytes per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_calib
mkdir -p $OUT
[ -x $R/tools/exp/ldsdma_fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/exp/ldsdma_fetch_calib $R/tools/exp/ldsdma_fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -- $R/tools/exp/ldsdma_fetch_calib > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
nrows = 1 << 22
known = {"stream_dwordx4": nrows * 256, "gather_ldsdma<0>": nrows * 128 + nrows * 4, "gather_ldsdma<1>": nrows * 256 + nrows * 4}
res = {}
for k, v in sorted(agg.items()):
    kb = sum(v) / len(v)
    name = [n for n in known if n.split("<")[0] in k and (("<" not in n) or n[n.index("<"):] in k)]
    kn = known[name[0]] if name else None
    res[k] = {"FETCH_SIZE_KB_per_launch": kb, "known_bytes": kn, "counter_over_known": (kb * 1024 / kn) if kn else None, "launches": len(v)}
    print("%-70s FETCH_SIZE %.1f MB  known %.1f MB  ratio %.3f" % (k[:70], kb * 1024 / 1e6, (kn or 0) / 1e6, (kb * 1024 / kn) if kn else float("nan")))
json.dump(res, open(out + "/fetch_calib.json", "w"), indent=1)
PY
rm -rf $OUT/pmc
