#!/usr/bin/env python3
"""Golden vectors for the rate model (build container only): imports the REFERENCE's compute_rate
(coolchic/component/core/arm.py:448-485) and evaluates it, in float32 on one CPU thread, on 2^16 symbols:

    python tests/golden/gen/dump_rate.py            # writes tests/golden/rate.npz

Inputs are stored as small integers the test expands exactly: x = int8 symbol, mu = mu_idx / 256 - 64 (the ARM's mu grid),
scale = the .cool format's Laplace scale table at scale_idx (include/ccd_scale_table.inc) - what the decoder's ARM can produce -
plus the reference's output `rate` (float32).  The first 6000 symbols are steered: symbol at the mode, half-way between
two grid points, far tails that hit the 2^-16 clamp (16 bits exactly), the smallest and the largest scale."""
import os
import re
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(1, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)
from coolchic.component.core.arm import compute_rate  # noqa: E402


def scale_table() -> np.ndarray:
    text = open(os.path.join(ROOT, "include", "ccd_scale_table.inc")).read()
    bits = np.array([int(t, 16) for t in re.findall(r"0x([0-9a-f]{8})u", text)], dtype=np.uint32)
    assert bits.size == 2561
    return bits.view(np.float32)


def main():
    n = 1 << 16
    rng = np.random.default_rng(20260927)
    x = rng.integers(-64, 64, n).astype(np.int8)
    mu_idx = np.clip(np.round((x.astype(np.float64) + rng.normal(0, 2.0, n) + 64) * 256), 0, 32767).astype(np.int16)
    scale_idx = rng.integers(700, 2561, n).astype(np.int16)
    mu_of = lambda idx: idx.astype(np.float64) / 256 - 64
    # steered cases
    mu_idx[:1000] = ((x[:1000].astype(np.int32) + 64) * 256).astype(np.int16)              # symbol at the mode
    mu_idx[1000:2000] = np.clip((x[1000:2000].astype(np.int32) + 64) * 256 + 128, 0, 32767)  # half-way
    x[2000:3000] = np.clip(np.round(mu_of(mu_idx[2000:3000])) + rng.choice([-60, 60], 1000), -64, 63).astype(np.int8)
    scale_idx[2000:3000] = rng.integers(0, 900, 1000)                                       # far tail, narrow law: clamp
    scale_idx[3000:4000] = 0
    scale_idx[4000:5000] = 2560
    x[5000:6000] = rng.choice([-64, 63], 1000).astype(np.int8)
    tab = scale_table()
    xf = torch.from_numpy(x.astype(np.float32))
    mu = torch.from_numpy(mu_of(mu_idx).astype(np.float32))
    sc = torch.from_numpy(tab[scale_idx.astype(np.int64)])
    with torch.no_grad():
        rate = compute_rate(xf, mu, sc).numpy().astype(np.float32)
    assert np.isfinite(rate).all() and rate.max() <= 16.0 and (rate == 16.0).sum() > 100
    out = os.path.join(ROOT, "tests", "golden", "rate.npz")
    np.savez_compressed(out, x=x, mu_idx=mu_idx, scale_idx=scale_idx, rate=rate)
    print("wrote", out, os.path.getsize(out), "bytes; clamped:", int((rate == 16.0).sum()), "mean bits", float(rate.mean()))


if __name__ == "__main__":
    main()
