#!/usr/bin/env python3
"""Times the float stages (upsampling + synthesis + integer samples) of kodak24 x COPIES on one MI355X:
fused kernel (ccd_fused.hip) with and without the f32 output, and the unfused path (6 upsampling launches + synthesis
kernel).  HIP events on the launch stream, inputs resident (the entropy stage ran once before)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cool_chic_amd import DecodeBatch  # noqa: E402


def main():
    copies_list = [int(a) for a in sys.argv[1:]] or [1, 8]
    items, _ = bench.build_kodak24(0)
    stream = torch.cuda.current_stream(0)
    sh = stream.cuda_stream
    for copies in copies_list:
        px = copies * sum(h * w for *_, (h, w) in items)
        for label, opts in (("fused, planes only", dict(fused_dec=1, keep_float=False)),
                            ("fused, planes + f32", dict(fused_dec=1, keep_float=True)),
                            ("pre+fused, planes only", dict(fused_dec=2, keep_float=False)),
                            ("pre+fused, planes+f32", dict(fused_dec=2, keep_float=True)),
                            ("unfused (r01 path)", dict(fused_dec=False))):
            b = DecodeBatch(0, **opts)
            for _ in range(copies):
                for hdr, nn, lat, _ in items:
                    b.add(hdr, nn, lat, 8, 0)
            b.run(sh)
            b.wait(sh)
            reps = 20
            res = {}
            for stages in ((1, 2), (1,), (2,)):
                for _ in range(3):
                    for st in stages:
                        b.run(sh, stage=st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(0)
                e0.record(stream)
                for _ in range(reps):
                    for st in stages:
                        b.run(sh, stage=st)
                e1.record(stream)
                torch.cuda.synchronize(0)
                res[stages] = e0.elapsed_time(e1) / reps
            ms = res[(1, 2)]
            split = f"   [stage 1 alone {res[(1,)]:.4f} ms, stage 2 alone {res[(2,)]:.4f} ms]"
            flop = 1724.0 * px  # algorithmic: synthesis 2 x 672 + upsampling ~380 flop / px
            print(f"{copies * len(items):4d} frames  {label:22s} {ms:8.4f} ms  {px / ms / 1e3:9.1f} Mpx/s  {flop / ms / 1e9:7.1f} TFLOP/s algorithmic"
                  f"  ({flop / ms / 1e9 / 157.3 * 100:5.1f} % of fp32 peak)" + split, flush=True)
            if opts.get("fused_dec"):
                import ctypes as C
                import numpy as np
                from cool_chic_amd._lib import lib
                prof = np.zeros(16, np.uint64)
                if lib().ccd_debug_fd_profile(prof.ctypes.data, 1) == 1 and prof[9]:
                    lib().ccd_debug_fd_profile(prof.ctypes.data, 1)
                    b.run(sh, stage=2)
                    torch.cuda.synchronize(0)
                    lib().ccd_debug_fd_profile(prof.ctypes.data, 1)
                    n = float(prof[9])
                    names = ["params/WG", "top sync+geom+S1", "S2 coarse levels", "S3 ups pass0", "S3 ups pass1", "S3 mfma pass0", "S3 mfma pass1", "S4 conv+epilogue", "tile total", "tiles", "S1 issue", "S1 barrier1", "S1 wait+stores", "S1 barrier2", "S2 phase A"]
                    print("      cycles per tile (wave 0): " + ", ".join(f"{nm} {float(prof[i]) / n:.0f}" for i, nm in enumerate(names) if i != 9), flush=True)
            b.close()


if __name__ == "__main__":
    main()
