#!/bin/bash
# r05: the fused float kernel with ONE 3x3 tile (30 + 9 KB of LDS per workgroup instead of 58 + 9): at 2 waves per SIMD (main) and
# with the register allocation capped for 3 (fdw3)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/time_float.py 1 8 2>&1 | grep -v amdgpu.ids | grep "fused" | sed 's/^/1tile-w2  /' | tee gpurun_out/time_float_1tile.log
CCD_LIB=cool_chic_amd/libccd_fdw3.so timeout 300 python tools/time_float.py 1 8 2>&1 | grep -v amdgpu.ids | grep "fused" | sed 's/^/1tile-w3  /' | tee -a gpurun_out/time_float_1tile.log
timeout 900 python -m pytest tests -m gpu -q -x -k "stream_parity or workloads or float_stage or fuzzed or full_size or video or many_streams" 2>&1 | tail -4 | tee gpurun_out/gpu_tests_e.log
CCD_LIB=cool_chic_amd/libccd_fdw3.so timeout 900 python -m pytest tests -m gpu -q -x -k "stream_parity or workloads or float_stage" 2>&1 | tail -4 | tee -a gpurun_out/gpu_tests_e.log
