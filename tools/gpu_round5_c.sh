#!/bin/bash
# r05 third GPU call: 8-pixel tasks published as two 4-pixel parts (main) against whole tasks (nosplit): A/B, per-grid profile,
# the whole GPU suite (a structural change of the hand-over), a short bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python tools/ab_entropy.py nosplit=cool_chic_amd/libccd_nosplit.so split=cool_chic_amd/libccd.so nosplit2=cool_chic_amd/libccd_nosplit.so split2=cool_chic_amd/libccd.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_entropy_split8.txt
rm -f gpurun_out/prof_grids_split8.txt
for lib in prof1ns prof1; do
  for i in 0 3; do
    echo "== $lib" | tee -a gpurun_out/prof_grids_split8.txt
    CCD_LIB=cool_chic_amd/libccd_$lib.so timeout 200 python tools/prof_grids.py $i 2>&1 | grep -v amdgpu.ids | tail -7 | tee -a gpurun_out/prof_grids_split8.txt
  done
done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/gpu_tests_c.log
timeout 600 python bench.py --steps 10 --warmup 2 --legs kodak24_hq,clic41,uhd4k --no-cpu-baseline --no-live-traffic > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_c.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stage_ms_per_step")}, d["verified"]["ok"])
print("from_bytes", d["from_bytes"]["value"], d["from_bytes"]["ms_per_step"], d["from_bytes"]["ratio_to_value"], d["from_bytes"]["verified"]["ok"])
for k, v in d.get("baseline_configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v.get("verified", {}).get("ok"), v.get("entropy_ms"))
PY
