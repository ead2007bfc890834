#!/bin/bash
# Runs on the GPU box (gpurun): FETCH_SIZE and WRITE_SIZE of tools/ubench/pmc_calib (1 GiB streamed per kernel, one access width
# each), one rocprofv3 --pmc pass per counter -> gpurun_out/pmc_calibration.json (factor = true bytes / counter bytes).
set -u
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_calib
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $REPO/tools/ubench/pmc_calib > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $REPO/tools/ubench/pmc_calib > "$OUT/write.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
N = float(1 << 30)
res = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    for k, v in sorted(vals.items()):
        v = v[1:] or v  # first launch: cold
        mean = sum(v) / len(v)
        res.setdefault(k, {})[counter + "_bytes"] = mean
        res[k][counter + "_over_true"] = mean / N
doc = {"true_bytes_per_kernel": int(N), "unit": "counter value x 1024 (rocprofv3 reports KiB)", "kernels": res}
json.dump(doc, open(os.path.join(os.path.dirname(out), "pmc_calibration.json"), "w"), indent=1)
for k, v in res.items():
    print(k[:60], {a: round(b, 4) for a, b in v.items() if a.endswith("over_true")})
PY
rm -rf "$OUT/fetch" "$OUT/write"
