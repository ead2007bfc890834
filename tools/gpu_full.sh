#!/bin/bash
# One gpurun call: GPU parity suite, exhaustive CDF sweep, bench with all legs.  Logs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 | tee gpurun_out/gpu_tests.log
if [ "${1:-}" = "sweep" ]; then
  timeout 900 python tools/cdf_sweep.py --which both 2>&1 | tee gpurun_out/cdf_sweep.log | tail -12
fi
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 600 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "verified", "stage_ms_per_step")})
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"))
for k in ("more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing"):
    if k in d: print(k, d[k]["value"], d[k].get("verified", {}).get("ok"))
PY
