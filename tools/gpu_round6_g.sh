#!/bin/bash
# r06: compile-time ARM shape for lop.cfg (the GOP's I frames): parity, then A/B of the GOP and of kodak24_hq
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "fixed_shape or stream_parity or arm_sweep or video or dynamic_operand" 2>&1 | tail -5 | tee gpurun_out/gpu_tests_g.log
for fs in 1 0; do
  echo "CCD_FIXED_SHAPE=$fs" | tee -a gpurun_out/gop_timing_lop.txt
  CCD_FIXED_SHAPE=$fs CCD_VIDEO_TIMING=1 timeout 600 python tools/prof_gop.py 3 2>&1 | grep -v amdgpu.ids | grep "decoded\|ccd_decode_video " | tee -a gpurun_out/gop_timing_lop.txt
done
timeout 900 python bench.py --steps 10 --warmup 2 --legs gop1080p33,kodak24_hq --no-cpu-baseline --no-live-traffic > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_g.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms_per_step", "float_ms_exposed", "entropy_launches")}, d["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print("  ", k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("float_ms_exposed"), v.get("entropy_launches"), v.get("resident_coolchics_ms"))
PY
