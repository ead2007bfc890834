#!/bin/bash
# r06 robustness: the GPU suite with other stream / queue settings than the measured default (8 hardware queues; 8 unmeasured side
# streams = r05's assumption; one stream; chain groups off)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
K="overlapped or repeated or two_host or arm_sweep or video or workloads_match or prepare or python_surface"
( export GPU_MAX_HW_QUEUES=8; echo "GPU_MAX_HW_QUEUES=8"; python -c "import torch; from cool_chic_amd._lib import lib; print('concurrent streams', lib().ccd_concurrent_streams(0))" 2>&1 | grep concurrent; timeout 1200 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -2 ) | tee gpurun_out/robust.txt
( export CCD_SIDE_STREAMS=8; echo "CCD_SIDE_STREAMS=8"; timeout 1200 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -2 ) | tee -a gpurun_out/robust.txt
( export CCD_SIDE_STREAMS=1; echo "CCD_SIDE_STREAMS=1"; timeout 1200 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -2 ) | tee -a gpurun_out/robust.txt
( export CCD_OVERLAP=0; echo "CCD_OVERLAP=0"; timeout 1200 python -m pytest tests -m gpu -q -x -k "$K and not overlapped" 2>&1 | tail -2 ) | tee -a gpurun_out/robust.txt
( export GPU_MAX_HW_QUEUES=8; timeout 600 python bench.py --steps 10 --warmup 2 --legs clic41,wide --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=8 bench', d['value'], d['ms_per_step'], d['float_ms_exposed'], d['entropy_launches'], d['concurrent_streams'], d['baseline_configs']['clic41']['ms_per_step'], d['baseline_configs']['clic41']['float_ms_exposed'], d['more_frames_in_flight']['ms_per_step'])" ) | tee -a gpurun_out/robust.txt
