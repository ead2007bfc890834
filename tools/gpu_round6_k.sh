#!/bin/bash
# r06: the warp kernel with one polynomial per sin / cos where the quadrant is wave-uniform: parity (video tests), kernel time, GOP phases
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "video or python_surface or workloads_match_the_oracle" 2>&1 | tail -4 | tee gpurun_out/gpu_tests_k.log
CCD_VIDEO_TIMING=1 timeout 600 python tools/prof_gop.py 4 2>&1 | grep -v amdgpu.ids | grep "decoded\|reconstructed\|ccd_decode_video " | tee gpurun_out/gop_timing_k.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gop_k -o gop --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_gop.py 3 > /dev/null 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_gop_k/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-80s calls %5s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
