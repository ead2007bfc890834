#!/usr/bin/env python3
"""One image workload of cool_chic_amd/synth.py decoded a few times in one batch - the command tools/collect_profiles.sh puts
under rocprofv3 for the BASELINE configurations that are legs of bench.py (clic41, uhd4k), so that their kernels do not mix
with kodak24's in the statistics.     python tools/prof_workload.py clic41 [steps] [keep_float]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from cool_chic_amd import DecodeBatch, synth  # noqa: E402


def main():
    name = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    keep_float = len(sys.argv) > 3 and sys.argv[3] == "keep_float"  # like bench.py's metric batch (the library default)
    wl = synth.workload(name)
    b = DecodeBatch(0, keep_float=keep_float)
    for s in wl["streams"]:
        b.add(*synth.split_image_stream(s), 8, 0)
    b.run(); b.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    b.wait()
    dt = (time.perf_counter() - t0) / steps
    px = sum(h * w for h, w in wl["sizes"])
    print(json.dumps({"workload": name, "frames": len(wl["streams"]), "mpixels": px / 1e6, "ms_per_step": dt * 1e3, "mpx_per_s": px / dt / 1e6,
                      "symbols": int(sum(b.header(i).n_symbols for i in range(len(wl["streams"]))))}))
    b.close()


if __name__ == "__main__":
    main()
