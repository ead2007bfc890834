#!/bin/bash
# One gpurun call: the GPU parity suite, then A/B of the entropy stage (libraries given as label=path ...).
#   gpurun --timeout 900 -- 'bash tools/gpu_check.sh r02=cool_chic_amd/libccd_r02.so new=cool_chic_amd/libccd.so'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/gpu_tests.log
if [ $# -gt 0 ]; then
  timeout 400 python tools/ab_entropy.py "$@" 2>&1 | tee gpurun_out/ab_entropy.log
fi
