#!/bin/bash
# r06: stream / queue concurrency facts (tools/ubench/queues.hip) and where ccd_decode_video's time beyond the cool-chics goes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/ubench/queues 2>&1 | grep -v amdgpu.ids | tee gpurun_out/queues.txt
CCD_VIDEO_TIMING=1 timeout 600 python tools/prof_gop.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gop_timing.txt
