#!/usr/bin/env python3
"""would "mode-run speculation" pay in the decoder wave?  (CPU only, via the oracle.)

For every symbol in DECODE order (wavefront order x + 10 y, latent.py:240-265) the most probable symbol of its quantised
Laplace window is round(mu).  A scalar-only recurrence over PREDICTED symbols can only replace the per-symbol search while
the decoded symbols keep equalling the prediction, so what matters is the run length of consecutive hits *within a
16-symbol batch* (the decoder's unit: all (L, P) rows of a batch are known up front).  Prints per grid: hit rate, mean run
of hits, the distribution of the leading hit run of a batch, and the expected number of symbols a batch decodes before the
first miss.

    python tools/experiments/mode_runs.py            (kodim14 + a portrait variant + a 2K synthetic stream)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import oracle_py  # noqa: E402


def decode_order(h, w):
    """(y, x) of every symbol in decode order + the step boundaries."""
    ys, xs, steps = [], [], []
    if w <= 9:
        for c in range(h * w):
            ys.append(c // w); xs.append(c % w); steps.append(1)
        return np.array(ys), np.array(xs), steps
    for c in range(w + 10 * (h - 1)):
        if c < w:
            y0, x0 = 0, c
        else:
            y0, x0 = (c - w) // 10 + 1, w - 10 + (c - w) % 10
        n = min(h - y0, x0 // 10 + 1)
        for i in range(n):
            ys.append(y0 + i); xs.append(x0 - 10 * i)
        steps.append(n)
    return np.array(ys), np.array(xs), steps


def analyse(name, stream):
    fh, ccs = oracle_py.split_stream(stream)[1][0]
    r = oracle_py.decode_coolchic(*ccs[0], stop_after_entropy=True)
    print(f"== {name}: {r['n_symbols']} symbols, {8 * len(ccs[0][2]) / r['n_symbols']:.3f} bit/symbol")
    tot_hit = tot = 0
    for g in range(min(4, r["n_grids"])):
        lat = r["latent"][g]
        h, w = lat.shape
        ys, xs, steps = decode_order(h, w)
        sym = lat[ys, xs].astype(np.int64)
        ms = r["mu_scale_idx"][g]
        mode = np.clip(np.rint(ms[:, 0] / 256.0 - 64.0), -64, 63).astype(np.int64)
        hit = sym == mode
        tot_hit += int(hit.sum()); tot += hit.size
        # batches of 16 inside a step (the decoder's unit)
        lead, pos = [], 0
        for n in steps:
            for b0 in range(0, n, 16):
                hb = hit[pos + b0: pos + min(n, b0 + 16)]
                miss = np.flatnonzero(~hb)
                lead.append(int(miss[0]) if miss.size else len(hb))
            pos += n
        lead = np.array(lead)
        # run lengths of consecutive hits over the whole grid
        d = np.diff(np.concatenate([[0], hit.astype(np.int8), [0]]))
        runs = np.flatnonzero(d == -1) - np.flatnonzero(d == 1)
        hist = np.bincount(np.minimum(lead, 16), minlength=17)
        print(f"  grid {g} ({h}x{w}): hit rate {hit.mean():.3f}, mean hit run {runs.mean() if runs.size else 0:.2f}, "
              f"leading hits per 16-batch: mean {lead.mean():.2f}, P(>=6) {np.mean(lead >= 6):.3f}, P(all) {np.mean(lead >= 16):.3f}")
        print("     leading-run histogram 0..16:", " ".join(str(int(v)) for v in hist))
    print(f"  first 4 grids: hit rate {tot_hit / tot:.3f}")


def main():
    from cool_chic_amd import synth

    streams, sizes = synth.kodak24()
    analyse("kodim14 (real stream)", streams[0])
    analyse("kodak24[3] (portrait variant)", streams[3])
    analyse("2048x1365 synthetic (clic41[4])", synth.image_stream(1365, 2048, 2004))


if __name__ == "__main__":
    main()
