#!/bin/bash
# r05 second GPU call: float kernel with the next tile's inputs requested during the last 3x3 layer (main) against the same
# build without (nopf); entropy A/B of the window-parameter prefetch (winpf); parity of the float paths.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/time_float.py 1 8 2>&1 | grep -v amdgpu.ids | grep "pre+fused" | sed 's/^/main  /' | tee gpurun_out/time_float_ab.log
CCD_LIB=cool_chic_amd/libccd_nopf.so timeout 300 python tools/time_float.py 1 8 2>&1 | grep -v amdgpu.ids | grep "pre+fused" | sed 's/^/nopf  /' | tee -a gpurun_out/time_float_ab.log
timeout 400 python tools/ab_entropy.py base=cool_chic_amd/libccd.so winpf=cool_chic_amd/libccd_winpf.so base2=cool_chic_amd/libccd.so winpf2=cool_chic_amd/libccd_winpf.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_entropy_winpf.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "stream_parity or workloads or float_stage or fuzzed or full_size or video" 2>&1 | tail -5 | tee gpurun_out/gpu_tests_b.log
