#!/bin/bash
# Instruction mix and wait fractions of the entropy kernel on kodak24 (rocprofv3 --pmc, own runs) -> gpurun_out/sq_pmc.txt
set -u
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$i -- python $REPO/tools/prof_workload.py kodak24 2 keep_float > $OUT/$i.log 2>&1
done
cd $REPO
python - <<'PY' | tee gpurun_out/sq_pmc.txt
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(float))
for f in glob.glob("gpurun_out/sq_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "entropy_pipe" in r["Kernel_Name"]:
            vals[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(vals.items()):
    per = sorted(d.values())
    print(f"{c:28s} per launch, summed over the chip: {per[len(per)//2]:.4g}")
PY
rm -rf $OUT
