#!/bin/bash
# r06 first GPU call: the fat-MFMA exactness probe (verdict item 3a), baseline entropy timing + per-grid profile of the tree as r05 left it.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/ubench/mfma_probe_k 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mfma_probe_k.log
timeout 600 python tools/ab_entropy.py base=cool_chic_amd/libccd.so base2=cool_chic_amd/libccd.so 2>&1 | tee gpurun_out/ab_entropy_base.txt
for s in 0 3; do CCD_LIB=cool_chic_amd/libccd_prof1.so timeout 300 python tools/prof_grids.py $s 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/prof_grids_base.txt; done
