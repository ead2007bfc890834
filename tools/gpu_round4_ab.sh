#!/bin/bash
# One gpurun call: A/B of the entropy stage (base = HEAD's kernel, new = working tree, with / without the fixed-shape instantiation),
# parity of the new kernel, light and per-task profiles of a landscape and a portrait stream.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_entropy.py "$@" 2>&1 | tee gpurun_out/ab_entropy.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_parity or arm_sweep or dynamic_operand or workloads_match or ragged or fuzzed or repeated" 2>&1 | tail -5 | tee gpurun_out/gpu_tests_subset.log
for i in 0 3; do
  CCD_LIB=cool_chic_amd/libccd_prof1.so timeout 200 python tools/prof_grids.py $i 2>&1 | tail -7 | tee -a gpurun_out/prof_grids.log
  CCD_LIB=cool_chic_amd/libccd_prof2.so timeout 200 python tools/prof_tasks_stream.py $i 2>&1 | tail -2 | tee -a gpurun_out/prof_tasks.log
done
for hw in "512 768" "256 384" "128 192"; do
  CCD_LIB=cool_chic_amd/libccd_prof2.so timeout 200 python tools/prof_tasks.py $hw 2>&1 | tail -1 | tee -a gpurun_out/prof_tasks.log
done
