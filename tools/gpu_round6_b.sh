#!/bin/bash
# r06: float path overlapped with the entropy tail (chain groups): parity of the new run, then the bench legs that show the exposed float time
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "overlapped or repeated_runs or two_host_threads or workloads_match or python_surface or video_ipb or prepare" 2>&1 | tail -5 | tee gpurun_out/gpu_tests_b.log
timeout 900 python bench.py --steps 10 --warmup 2 --legs clic41,wide,gop1080p33,clic41_alt --no-cpu-baseline --no-live-traffic > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
tail -c 800 gpurun_out/bench_b.err
CCD_OVERLAP=0 timeout 900 python bench.py --steps 10 --warmup 2 --legs clic41,wide,gop1080p33,clic41_alt --no-cpu-baseline --no-live-traffic > gpurun_out/bench_b_nooverlap.json 2> gpurun_out/bench_b_nooverlap.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_b.json", "gpurun_out/bench_b_nooverlap.json"):
    d = json.load(open(f))
    print(f, {k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "float_ms_exposed", "entropy_launches")}, d["verified"]["ok"])
    print("  from_bytes", d["from_bytes"]["value"], d["from_bytes"]["ms_per_step"], d["from_bytes"]["ratio_to_value"], d["from_bytes"]["verified"]["ok"])
    for k, v in d.get("baseline_configs", {}).items():
        print("  ", k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("float_ms_exposed"), v.get("entropy_launches"), v.get("resident_coolchics_ms"))
    if "more_frames_in_flight" in d: print("  wide", d["more_frames_in_flight"]["value"], d["more_frames_in_flight"]["ms_per_step"], d["more_frames_in_flight"]["verified"]["ok"])
PY
