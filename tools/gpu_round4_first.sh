#!/bin/bash
# First GPU call of round 4: the whole GPU parity suite (no -x: every failure is wanted), then the default bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -60 | tee gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 1500 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "verified", "stage_ms_per_step", "scaling")})
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("cpu_baseline", {}).get("value"), v.get("cpu_baseline", {}).get("cores"))
for k in ("more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "rate_model", "cpu_baseline"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample")})
print(d["roofline_float_stages"][0]["frac"], d["serial_chain_bound"]["frac"])
PY
