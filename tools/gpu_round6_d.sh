#!/bin/bash
# r06: launches on measured-concurrent side streams; the GOP's reconstruction kernels under rocprofv3; overlap A/B again
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "overlapped or repeated_runs or two_host_threads or video or arm_sweep or prepare" 2>&1 | tail -5 | tee gpurun_out/gpu_tests_d.log
CCD_VIDEO_TIMING=1 timeout 600 python tools/prof_gop.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gop_timing.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gop -o gop --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_gop.py 3 > /dev/null 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_gop/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-90s calls %5s total %10.3f ms avg %9.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
timeout 900 python bench.py --steps 10 --warmup 2 --legs clic41,wide,gop1080p33,clic41_alt --no-cpu-baseline --no-live-traffic > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err
tail -c 600 gpurun_out/bench_d.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_d.json",):
    d = json.load(open(f))
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "stage_ms_per_step", "float_ms_exposed", "entropy_launches", "concurrent_streams")}, d["verified"]["ok"])
    print("  from_bytes", d["from_bytes"]["value"], d["from_bytes"]["ms_per_step"], d["from_bytes"]["ratio_to_value"], d["from_bytes"]["verified"]["ok"])
    for k, v in d.get("baseline_configs", {}).items():
        print("  ", k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"), v.get("float_ms"), v.get("float_ms_exposed"), v.get("entropy_launches"), v.get("resident_coolchics_ms"))
    if "more_frames_in_flight" in d: print("  wide", d["more_frames_in_flight"]["value"], d["more_frames_in_flight"]["ms_per_step"], d["more_frames_in_flight"]["verified"]["ok"])
PY
