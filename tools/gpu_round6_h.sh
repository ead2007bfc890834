#!/bin/bash
# r06: the default bench (what the driver runs) on the tree with chain groups / measured streams / XCD order, the determinism
# stress (kodak24 x 1 and x 8: launches forked over side streams), rocprofv3 summaries (kernel stats + PMC)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 900 gpurun_out/bench_full.err
timeout 600 python tools/stress.py 40 1 2>&1 | grep -v amdgpu | tee gpurun_out/stress.txt
timeout 900 python tools/stress.py 12 8 2>&1 | grep -v amdgpu | tee -a gpurun_out/stress.txt
timeout 2400 bash tools/collect_profiles.sh 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms_per_step", "scaling", "float_ms_exposed", "entropy_launches", "concurrent_streams")}, d["verified"]["ok"])
print("from_bytes", {k: v for k, v in d["from_bytes"].items() if k not in ("what", "verified")}, d["from_bytes"]["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("float_ms_exposed"), v.get("entropy_launches"), v.get("resident_coolchics_ms"), v.get("bpp"), v.get("cpu_baseline", {}).get("value"), v.get("cpu_baseline", {}).get("cores"))
for k in ("fallback_cliffs", "more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "rate_model", "cpu_baseline", "wide_envelope_network", "entropy_ms_by_orientation"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample", "verified_png_readback", "workload")})
print(json.dumps(d["roofline_float_stages"][0])[:1200]); print(d["serial_chain_bound"]); print(d["roofline"])
PY
