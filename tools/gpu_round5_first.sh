#!/bin/bash
# r05 first GPU call: MFMA operand-class probe, the whole GPU suite, the default bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
MFMA_PROBE_EXACT_ONLY=1 timeout 120 tools/ubench/mfma_probe 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mfma_probe_classes.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 | tee gpurun_out/gpu_tests.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 1500 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "scaling")}, d["verified"]["ok"])
print("from_bytes", {k: v for k, v in d["from_bytes"].items() if k not in ("what",)})
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("bpp"), v.get("rare_path_symbols"), v.get("full_searches"), v.get("entropy_kernel_widths_nv"))
for k in ("fallback_cliffs", "more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "rate_model", "cpu_baseline", "wide_envelope_network", "entropy_ms_by_orientation"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample", "verified_png_readback", "workload")})
print(json.dumps(d["roofline_float_stages"][0])[:700]); print(d["serial_chain_bound"]); print(d["roofline"])
PY
