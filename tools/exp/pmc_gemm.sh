cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_$tag -- python $R/tools/exp/gemm_only.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmcg_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:120]
        if "gemm_f32" not in k: continue
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print("PMCG", k)
    for c,vals in sorted(v.items()):
        print("PMCG    %-32s %.4g  (n=%d)"%(c, sum(vals)/len(vals), len(vals)))
PY
