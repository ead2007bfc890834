#!/bin/bash
# LDS pipe utilisation of the entropy kernel on kodak24 (own rocprofv3 --pmc run, no other traces): SQ_LDS_IDX_ACTIVE,
# SQ_LDS_BANK_CONFLICT, SQ_INSTS_LDS, SQ_BUSY_CYCLES per kernel -> gpurun_out/lds_pmc.txt
set -u
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/lds_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/a -- python $REPO/tools/prof_workload.py kodak24 2 keep_float > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python $REPO/tools/prof_workload.py kodak24 2 keep_float > $OUT/b.log 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/lds_pmc.txt
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/lds_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "entropy_pipe" in r["Kernel_Name"]:
            vals[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in sorted(vals.items()):
    per = [sum(v) for v in d.values()]
    print(c, "per launch (sum over the chip):", [round(x) for x in per])
PY
rm -rf $OUT
