#!/bin/bash
# Full GPU suite + bench after the float-path rework
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -25 | tee gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 1200 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "scaling")}, d["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("float_ms"))
for k in ("more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "rate_model"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample")})
print(json.dumps(d["roofline_float_stages"][0])[:900]); print(d["serial_chain_bound"]["frac"], d["roofline"])
for r in d.get("float_stages_sweep", []): print(r)
PY
