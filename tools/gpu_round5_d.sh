#!/bin/bash
# r05: the generic entropy kernel after its r05 changes (shared Laplace math, rows one symbol ahead, v_readlane): parity + the cliff leg
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "generic or laplace or fuzzed or arm_sweep or python_surface" 2>&1 | tail -3 | tee gpurun_out/gpu_tests_d.log
timeout 600 python bench.py --steps 10 --warmup 2 --legs cliffs --no-cpu-baseline --no-live-traffic > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_d.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step")}, d["verified"]["ok"])
print("from_bytes", d["from_bytes"]["value"], d["from_bytes"]["ms_per_step"], d["from_bytes"]["ratio_to_value"], d["from_bytes"]["verified"]["ok"])
print({a: b for a, b in d["fallback_cliffs"].items() if a not in ("verified", "workload")}, d["fallback_cliffs"]["verified"]["ok"])
PY
