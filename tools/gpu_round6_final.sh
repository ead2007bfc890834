#!/bin/bash
# Round-end GPU call of r06: build() as the driver runs it, the whole GPU suite, smoke(), the default bench (what the driver runs)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -v "hipcc\|amdgpu.ids" | tee gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -12 | tee gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 400 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "stage_ms_per_step", "float_ms_exposed", "entropy_launches", "concurrent_streams", "gpus_active")}, d["verified"]["ok"])
print("from_bytes", d["from_bytes"]["value"], d["from_bytes"]["ratio_to_value"], d["from_bytes"]["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, round(v["value"], 1), round(v["ms_per_step"], 2), v.get("verified", {}).get("ok"), v.get("float_ms_exposed"), v.get("entropy_launches"))
print([ (round(r["frac"], 4), round(r["ms_per_launch"], 4)) for r in d["roofline_float_stages"]])
print(d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d["roofline"]["launches_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
