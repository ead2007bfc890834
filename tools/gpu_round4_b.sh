#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_entropy.py "$@" 2>&1 | tee gpurun_out/ab_entropy_thresholds.log
CCD_LIB=cool_chic_amd/libccd_fdprof.so timeout 300 python tools/time_float.py 1 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/time_float_prof.log
