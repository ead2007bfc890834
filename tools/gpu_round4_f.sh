#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_entropy.py "$@" 2>&1 | tee gpurun_out/ab_entropy_features.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_parity or arm_sweep or dynamic_operand or mfma or workloads_match or ragged or fuzzed or repeated or video_ipb or many_streams" 2>&1 | tail -5 | tee gpurun_out/gpu_tests_subset.log
