#!/bin/bash
# Round-end GPU call of r05: the whole GPU suite, the default bench (what the driver runs), rocprofv3 summaries (kernel stats + PMC).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -12 | tee gpurun_out/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 600 gpurun_out/bench_full.err
timeout 1500 bash tools/collect_profiles.sh 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "scaling")}, d["verified"]["ok"])
print("from_bytes", {k: v for k, v in d["from_bytes"].items() if k not in ("what", "verified")}, d["from_bytes"]["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("bpp"), v.get("rare_path_symbols"), v.get("full_searches"), v.get("entropy_kernel_widths_nv"), v.get("cpu_baseline", {}).get("value"), v.get("cpu_baseline", {}).get("cores"))
for k in ("fallback_cliffs", "more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "rate_model", "cpu_baseline", "wide_envelope_network", "entropy_ms_by_orientation"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample", "verified_png_readback", "workload")})
print(json.dumps(d["roofline_float_stages"][0])[:700]); print(d["serial_chain_bound"]); print(d["roofline"])
PY
