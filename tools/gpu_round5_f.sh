#!/bin/bash
# r05: determinism stress of the final tree (the pipelined kernel's digests must equal r04's: its code did not change; the generic
# kernel changed: 1024 threads, v_readlane, shared Laplace code)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/stress.py 200 1; timeout 600 python tools/stress.py 20 11; CCD_FORCE_GENERIC=1 timeout 900 python tools/stress.py 30 1; CCD_FORCE_GENERIC=1 timeout 900 python tools/stress.py 5 11 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stress.txt
