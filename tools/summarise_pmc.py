"""gpurun_out/prof/{fetch,write} (tools/collect_profiles.sh: one rocprofv3 --pmc run per counter) -> profiles/<round>/<tag>_pmc_traffic.json

    python tools/summarise_pmc.py [gpurun_out/prof] [profiles/r03] [kodak24]

Bytes per launch = mean of the counter over the launches of a kernel (first launch of every kernel dropped: warm-up
with cold caches), counter unit KiB, CORRECTED with the factors measured on this GPU generation by tools/pmc_calibrate.sh
(profiles/r03/pmc_calibration.json: 1 GiB streamed per kernel with 1 / 2 / 4 / 16 bytes per lane): FETCH_SIZE reports exactly
half of the bytes read whatever the access width -> x 2; WRITE_SIZE is exact for 1 / 4 / 8 / 16 bytes per lane -> x 1.
Kernels launched several times per step (the six pyramid levels) also get a per-step figure."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {  # short name -> (substring of the kernel name, launches per bench step or None)
    "entropy_pipe_kernel": ("entropy_pipe_kernel", None),
    "upsample_step_kernel": ("upsample_step_kernel", 6),
    "syn_fused_kernel": ("syn_fused_kernel", None),
    "decode_fused_kernel": ("decode_fused_kernel", None),   # every instantiation of a step (pyramid launch + tiles): summed
    "rate_kernel_v4": ("rate_kernel_v4", None),
    "png_filter_huff_kernel": ("png_filter_huff_kernel", None),
    "png_emit_kernel": ("png_emit_kernel", None),
    "png_crc_kernel": ("png_crc_kernel", None),
}


RUNS = {"kodak24": 3, "clic41": 4, "uhd4k": 4, "kodak24_hq": 4}  # runs of the batch in the profiled command (tools/prof_workload.py: 1 + steps)


def read(dir_, counter):
    vals = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(dir_, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    vals[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0))
    return vals


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
    dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04"
    tag = sys.argv[3] if len(sys.argv) > 3 else "kodak24"
    fetch, write = read(os.path.join(src, "fetch"), "FETCH_SIZE"), read(os.path.join(src, "write"), "WRITE_SIZE")
    out = {}
    for short, (sub, per_step) in KERNELS.items():
        names = [n for n in fetch if sub in n]
        if not names:
            continue
        # one entry per short name: the mean over the launches of every kernel whose name holds `sub` (first launch dropped),
        # summed over those kernels (the fused float path is two launches per step: the pyramid launch and the tiles)
        e = {"full_names": names, "launches": 0}
        tot_f = tot_w = 0.0
        for name in names:
            # r06: a kernel may be launched several times per step (chain groups: one entropy / pyramid / fused launch per group).
            # Where the number of runs of the profiled command is known (RUNS), the figure is bytes per STEP: the launches of the
            # first run dropped, the others summed and divided by the runs left
            runs = RUNS.get(tag)
            k = len(fetch[name]) // runs if runs and len(fetch[name]) % runs == 0 else 1
            f, w = fetch[name][k:] or fetch[name], write.get(name, [0.0])[k:] or write.get(name, [0.0])
            e["launches"] += len(f)
            e.setdefault("launches_per_step_by_kernel", {})[name[:120]] = k
            tot_f += sum(f) / len(f) * k
            tot_w += sum(w) / len(w) * k
            if len(names) > 1:
                e.setdefault("per_kernel", {})[name[:120]] = {"fetch_bytes": sum(f) / len(f) * k, "write_bytes": sum(w) / len(w) * k}
        if per_step:
            e["launches_per_step"] = per_step
            e["fetch_bytes_per_step"] = tot_f * per_step
            e["write_bytes_per_step"] = tot_w * per_step
        else:
            e["fetch_bytes"] = tot_f
            e["write_bytes"] = tot_w
        out[short] = e
    cmd = {"kodak24": "python tools/prof_workload.py kodak24 2 keep_float [= bench.py's live pass, measure_traffic_live: the metric's batch alone]",
           "kodak256": "python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --legs none --scaling throughput",
           "rate": "python tools/prof_rate.py"}.get(tag, f"python tools/prof_workload.py {tag} 3")
    doc = {
        "command": cmd + " (tools/collect_profiles.sh; one rocprofv3 run per counter)",
        "unit": "bytes per step of the batch (= per launch where a kernel is launched once per step; launches_per_step_by_kernel says): rocprofv3 FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024 x 1",
        "calibration": "profiles/r03/pmc_calibration.json (tools/pmc_calibrate.sh, tools/ubench/pmc_calib.hip): on gfx950 FETCH_SIZE = 0.500 of "
                       "the bytes streamed with 1, 2, 4 and 16 bytes per lane; WRITE_SIZE = 1.000 with 1, 4, 8 and 16 bytes per lane",
        "kernels": out,
    }
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, tag + "_pmc_traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
