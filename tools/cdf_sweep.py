#!/usr/bin/env python3
"""Exhaustive check of the f64 Laplace CDF on the device (hard part H2).

A symbol's boundaries are floor(16777088 * cdf(s -+ 0.5)) with a float64 `exp` of the ENCODER's libm (constriction's
QuantizedLaplace); one boundary that floors differently desynchronises a stream silently.  The reachable
set is finite: 32768 mu indices x 2561 scale indices x 127 boundaries (s = -63 .. 63; the left bound of -64 is 0 and the
right bound of s is the left bound of s + 1) = 1.0658e10.  This tool evaluates ALL of them on the GPU, with the production
kernel's table builder (window_left: home-made exp + Markstein quotient) and with the generic kernel's (laplace_left: the
device library's exp), and compares every one with libm on the host (oracle/cc_oracle.c::ora_laplace_lefts_check, one
thread per scale index).

    python tools/cdf_sweep.py [--which pipe|generic|both] [--stride N] [--chunk 8] [--threads T]
"""
import argparse
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def sweep(which: int, scales, chunk: int, threads: int, device: int = 0, log=print):
    """Returns (boundaries checked, mismatches, first offenders as (scale_idx, mu_idx, s, device, libm))."""
    import torch  # noqa: F401  (HIP runtime first, like every user of libccd)
    from cool_chic_amd._lib import check, lib
    from oracle import oracle_py

    oracle_py.build()
    n_checked = n_bad = 0
    offenders = []
    t0 = time.perf_counter()
    runs = []  # maximal runs of consecutive scale indices, at most `chunk` long
    for c in scales:
        if runs and runs[-1][-1] + 1 == c and len(runs[-1]) < chunk:
            runs[-1].append(c)
        else:
            runs.append([c])
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pending = []
        for k, run in enumerate(runs):
            buf = np.empty((len(run), 32768, 127), dtype=np.uint32)
            check(lib().ccd_debug_laplace_sweep(device, which, run[0], len(run), buf.ctypes.data), "ccd_debug_laplace_sweep")
            pending.append([(c, pool.submit(oracle_py.laplace_lefts_check, c, buf[i])) for i, c in enumerate(run)])
            while len(pending) > 4 or (k == len(runs) - 1 and pending):  # bound the buffers in flight
                for c, fut in pending.pop(0):
                    n, bad = fut.result()
                    n_checked += 32768 * 127
                    n_bad += n
                    offenders += [(c, *map(int, row)) for row in bad][:max(0, 64 - len(offenders))]
            if k % 32 == 0:
                log(f"  scale index {run[0]:4d}: {n_checked / 1e9:7.3f}e9 checked, {n_bad} differ, {time.perf_counter() - t0:6.1f} s")
    return n_checked, n_bad, offenders


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", choices=["pipe", "generic", "both"], default="both")
    ap.add_argument("--stride", type=int, default=1, help="every N-th scale index (1 = exhaustive)")
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 8))
    args = ap.parse_args()
    scales = list(range(0, 2561, args.stride))
    rc = 0
    for name, which in (("pipe", 0), ("generic", 1)):
        if args.which not in (name, "both"):
            continue
        print(f"{name}: {'window_left (ccd_entropy_pipe.hip)' if which == 0 else 'laplace_left (ccd_entropy.hip)'}, "
              f"{len(scales)} scale indices x 32768 mu indices x 127 boundaries, {args.threads} host threads", flush=True)
        n, bad, off = sweep(which, scales, args.chunk, args.threads, log=lambda m: print(m, flush=True))
        print(f"{name}: {bad} of {n} ({n / 1e9:.4f}e9) boundaries differ from libm", flush=True)
        for o in off:
            print("   scale_idx %d mu_idx %d s %d: device %d libm %d" % o, flush=True)
        rc |= bad != 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
