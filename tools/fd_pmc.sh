cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -o out -- python $R/tools/time_float.py 1 > $R/gpurun_out/pmc_$n.log 2>&1)
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ.get("GRAFT_REPO_ROOT",".")
for f in sorted(glob.glob(R+"/gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:60]
        if "decode_fused" not in k and "syn_fused" not in k and "upsample_step" not in k: continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[k+row["Counter_Name"]]+=1
    for k,v in agg.items():
        print(os.path.basename(os.path.dirname(os.path.dirname(f))), k)
        for c,val in v.items(): print("   ",c,val/cnt[k+c], "per launch over", cnt[k+c])
PY
