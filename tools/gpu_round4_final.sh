#!/bin/bash
# Round-end GPU call: the whole GPU suite, the default bench, per-grid / per-task profiles, rocprofv3 summaries (kernel stats + PMC).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14 | tee gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 600 gpurun_out/bench_full.err
rm -f gpurun_out/prof_grids.log gpurun_out/prof_tasks.log
for i in 0 3; do
  CCD_LIB=cool_chic_amd/libccd_prof1.so timeout 200 python tools/prof_grids.py $i 2>&1 | grep -v amdgpu.ids | tail -7 | tee -a gpurun_out/prof_grids.log
  CCD_LIB=cool_chic_amd/libccd_prof2.so timeout 200 python tools/prof_tasks_stream.py $i 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/prof_tasks.log
done
for hw in "512 768" "256 384" "128 192"; do
  CCD_LIB=cool_chic_amd/libccd_prof2.so timeout 200 python tools/prof_tasks.py $hw 2>&1 | tail -1 | tee -a gpurun_out/prof_tasks.log
done
timeout 1500 bash tools/collect_profiles.sh 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "scaling")}, d["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("float_ms"), v.get("cpu_baseline", {}).get("value"), v.get("cpu_baseline", {}).get("cores"))
for k in ("more_frames_in_flight", "end_to_end_from_bytes", "with_png_packing", "cc_decode_file_to_png", "rate_model", "cpu_baseline", "wide_envelope_network", "entropy_ms_by_orientation"):
    if k in d: print(k, {a: b for a, b in d[k].items() if a not in ("what", "note", "verified", "sample", "verified_png_readback", "workload")})
print(json.dumps(d["roofline_float_stages"][0])[:700]); print(d["serial_chain_bound"]); print(d["roofline"])
PY
