#!/bin/bash
# r06: chain groups at a full chip with their boundaries on multiples of 8 streams (63 + 193 -> 64 + 192)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/diag_wide3.txt
for n in 240 248 256 250; do timeout 300 python tools/diag_wide.py $n 4 2>&1 | grep streams | tee -a gpurun_out/diag_wide3.txt; done
timeout 900 python bench.py --steps 10 --warmup 2 --legs wide,clic41 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['verified']['ok'], 'clic41', d['baseline_configs']['clic41']['ms_per_step'], 'wide', d['more_frames_in_flight']['ms_per_step'], d['more_frames_in_flight']['value'], d['more_frames_in_flight']['verified']['ok'])" | tee -a gpurun_out/diag_wide3.txt
timeout 900 python tools/stress.py 10 8 2>&1 | grep -v amdgpu | tee -a gpurun_out/diag_wide3.txt
