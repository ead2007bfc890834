#!/bin/bash
# r06: whole GPU suite on the tree (chain groups, measured side streams, per-frame plane copies, XCD-ordered work lists), then the
# float path's traffic / time with and without the XCD order, the GOP's phases, the streams-in-flight sweep
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -14 | tee gpurun_out/gpu_tests_f.log
for v in xcd natural; do
  if [ $v = natural ]; then export CCD_NO_XCD_ORDER=1; else unset CCD_NO_XCD_ORDER; fi
  timeout 900 python bench.py --steps 10 --warmup 2 --legs float --no-cpu-baseline > gpurun_out/bench_f_$v.json 2> gpurun_out/bench_f_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_f_$v.json"))
f = d["roofline_float_stages"][0]
print("$v", {k: d.get(k) for k in ("value", "ms_per_step", "float_ms_exposed")}, d["stage_ms_per_step"], "traffic", f["traffic"], "alg", f["algorithmic_bytes"], "frac", round(f["frac"], 4), d["roofline"]["traffic"], d["traffic_from"][:40])
for r in d.get("float_stages_sweep", []): print("   ", r["frames"], r["path"][:60], round(r["ms"], 4), round(r["frac_of_fp32_peak"], 4))
PY
done
unset CCD_NO_XCD_ORDER
CCD_VIDEO_TIMING=1 timeout 600 python tools/prof_gop.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gop_timing.txt
for n in 24 224 256; do timeout 300 python tools/diag_wide.py $n 3 2>&1 | grep streams | tee -a gpurun_out/diag_wide2.txt; done
