#!/bin/bash
# One gpurun call: A/B of the entropy stage only (libraries given as label=path ...), log -> gpurun_out/ab_entropy.log
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/ab_entropy.py "$@" 2>&1 | tee gpurun_out/ab_entropy.log
