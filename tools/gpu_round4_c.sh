#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_parity and (fused_ or production)" 2>&1 | tail -8 | tee gpurun_out/gpu_tests_fused.log
timeout 300 python tools/time_float.py 1 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/time_float.log
CCD_LIB=cool_chic_amd/libccd_fdprof.so timeout 300 python tools/time_float.py 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/time_float_prof.log
