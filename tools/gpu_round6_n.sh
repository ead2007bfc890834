#!/bin/bash
set -u
export TMPDIR=/tmp
K="overlapped or repeated or two_host or arm_sweep or video or workloads_match or prepare or python_surface"
( export CCD_SIDE_STREAMS=8; timeout 1200 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | grep -v amdgpu | tail -30 )
( export CCD_OVERLAP=0; timeout 1200 python -m pytest tests -m gpu -q -x -k "$K and not overlapped" 2>&1 | grep -v amdgpu | tail -30 )
