#!/bin/bash
# r06: chained 4-symbol parts (ccd_dec_parts4.inc) against r05's (-DCCD_NO_CHAIN4), then 4-pixel tasks on the half-size grids with them
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/ab_entropy.py nc4=cool_chic_amd/libccd_nc4.so c4=cool_chic_amd/libccd.so c4_t40=cool_chic_amd/libccd_c4_t40.so c4_t40_s=cool_chic_amd/libccd_c4_t40_s.so c4_t40_se=cool_chic_amd/libccd_c4_t40_se.so c4_t30_s=cool_chic_amd/libccd_c4_t30_s.so nc4_again=cool_chic_amd/libccd_nc4.so c4_again=cool_chic_amd/libccd.so 2>&1 | tee gpurun_out/ab_chain4.txt
timeout 1500 python -m pytest tests -m gpu -q -x -k "arm_sweep or stream_parity or ragged or streamed_body or fuzzed or repeated or workloads_match or full_size or video" 2>&1 | tail -4 | tee gpurun_out/gpu_tests_j.log
