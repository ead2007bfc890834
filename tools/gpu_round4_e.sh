#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rate_model" 2>&1 | tail -3 | tee gpurun_out/gpu_tests_rate.log
timeout 1500 bash tools/collect_profiles.sh 2>&1 | tail -12
for i in 1 2 3; do timeout 120 python tools/prof_rate.py 2>&1 | grep '^{'; done | tee gpurun_out/rate_runs.log
