#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 64 128 192 224 256; do timeout 300 python tools/diag_wide.py $n 3 2>&1 | grep streams | tee -a gpurun_out/diag_wide.txt; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_wide -o wide --output-format csv -- python $GRAFT_REPO_ROOT/tools/diag_wide.py 256 1 1 > /dev/null 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_wide/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows[-14:]:
        print("%-70s grid %8s  start %10.3f ms  end %10.3f ms  queue %s" % (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Queue_Id")))
PY
