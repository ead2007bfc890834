#!/bin/bash
# Round-end GPU call of r06 (final tree): the whole GPU suite, the driver's bench command, rocprofv3 summaries, stress
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=4 2>&1 | tail -8 | tee gpurun_out/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 300 gpurun_out/bench_full.err
timeout 600 python tools/stress.py 40 1 2>&1 | grep -v amdgpu | tee gpurun_out/stress.txt
timeout 900 python tools/stress.py 12 8 2>&1 | grep -v amdgpu | tee -a gpurun_out/stress.txt
timeout 2400 bash tools/collect_profiles.sh 2>&1 | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms_per_step", "float_ms_exposed", "entropy_launches", "concurrent_streams")}, d["verified"]["ok"])
print("from_bytes", d["from_bytes"]["value"], d["from_bytes"]["ms_per_step"], d["from_bytes"]["ratio_to_value"], d["from_bytes"]["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, round(v["value"], 1), round(v["ms_per_step"], 2), v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("float_ms_exposed"), v.get("entropy_launches"), v.get("resident_coolchics_ms"))
for k in ("more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "wide_envelope_network", "entropy_ms_by_orientation"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample", "verified_png_readback", "workload")})
c = d["fallback_cliffs"]; print("cliffs", c["generic_entropy_ms"], c["ratio_to_pipelined_entropy"], c["generic_float_ms"], c["ratio_to_fused_float"], c["picture_wider_than_the_symbol_ring"]["ratio_ns_per_symbol"])
print([(round(r["frac"], 4), round(r["ms_per_launch"], 4), r.get("traffic")) for r in d["roofline_float_stages"]])
print(d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d["roofline"]["launches_per_step"], d["roofline"]["traffic"], d["serial_chain_bound"]["frac"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_core"]["value"], "rate", d["rate_model"]["frac"])
PY
