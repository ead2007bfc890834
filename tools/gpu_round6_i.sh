#!/bin/bash
# r06: 4-pixel tasks on the half-size grids (CCD_T8=40) as the tree stands - calibration of the chain model before chained 4-symbol parts
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_entropy.py base=cool_chic_amd/libccd.so t8_40=cool_chic_amd/libccd_t8_40.so 2>&1 | tee gpurun_out/ab_t8_40.txt
for lib in prof1 t8_40_prof1; do for s in 0 3; do echo "== $lib stream $s" | tee -a gpurun_out/prof_grids_t8_40.txt; CCD_LIB=cool_chic_amd/libccd_$lib.so timeout 300 python tools/prof_grids.py $s 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/prof_grids_t8_40.txt; done; done
