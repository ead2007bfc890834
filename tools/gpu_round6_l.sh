#!/bin/bash
# r06: warp coefficients ahead of the references: parity, then the GOP's phases with and without
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "video or python_surface or workloads_match_the_oracle or overlapped" 2>&1 | tail -4 | tee gpurun_out/gpu_tests_l.log
for mb in default 0; do
  echo "CCD_VIDEO_COEF_MB=$mb" | tee -a gpurun_out/gop_timing_coef.txt
  if [ $mb = default ]; then unset CCD_VIDEO_COEF_MB; else export CCD_VIDEO_COEF_MB=$mb; fi
  CCD_VIDEO_TIMING=1 timeout 600 python tools/prof_gop.py 5 2>&1 | grep -v amdgpu.ids | grep "decoded\|reconstructed\|ccd_decode_video " | tee -a gpurun_out/gop_timing_coef.txt
done
unset CCD_VIDEO_COEF_MB
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gop_l -o gop --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_gop.py 3 > /dev/null 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_gop_l/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("%-80s calls %5s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
