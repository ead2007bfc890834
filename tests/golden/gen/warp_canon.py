#!/usr/bin/env python3
"""How the float32 operation sequence of F.grid_sample (bilinear / bicubic, border, align_corners=True) on the
Warper's grid was pinned (build container only: needs the CPU PyTorch the reference runs on).

A numpy restatement with every rounding explicit is compared BIT FOR BIT with torch on random pictures and flows; the
variant that matches 100 % is the canon used by oracle/cc_oracle.c (warp_pixel_native) and csrc/ccd_inter.hip:

  linspace(-1, 1, n)[i] = fma(step, i, -1) for i < n / 2, fma(-step, n - 1 - i, 1) above       (step = 2 / (n - 1))
  grid = linspace + flow / ((size - 1) / 2);  source index = (grid + 1) * ((size - 1) / 2)
  bilinear: clip the index; w = x - floor x, e = 1 - w, ...; nw * (s e), then fma(ne, s w, .), fma(sw, n e, .), fma(se, n w, .)
  bicubic : outer coefficients ((A x - 5A) x + 8A) x - 4A with every step rounded; inner fma(fma(A+2, x, -(A+3)) * x, x, 1);
            row = fma(p0, c0, p1 * c1) + p2 * c2 + p3 * c3;  column = fma(r3, d3, fma(r2, d2, fma(r1, d1, r0 * d0)))

    python tests/golden/gen/warp_canon.py         # prints the matching fractions (1.0 expected)
"""
import numpy as np
import torch
import torch.nn.functional as F

f32, f64 = np.float32, np.float64


def fma(a, b, c):  # exact product of two float32 in float64, one rounding
    return (np.asarray(a, f64) * np.asarray(b, f64) + np.asarray(c, f64)).astype(f32)


def mul(a, b):
    return (a * b).astype(f32)


def linspace(n):
    step = f32(f32(2.0) / f32(n - 1))
    i = np.arange(n)
    lo = (f64(step) * i - 1.0).astype(f32)
    hi = (1.0 - f64(step) * (n - 1 - i)).astype(f32)
    return np.where(i < n // 2, lo, hi).astype(f32)


def main():
    torch.manual_seed(0)
    H, W = 128, 224
    x = torch.rand(1, 3, H, W)
    flow = torch.randn(1, 2, H, W) * 3
    th = torch.linspace(-1.0, 1.0, W, dtype=torch.float32).view(1, 1, 1, W).expand(1, -1, H, -1)
    tv = torch.linspace(-1.0, 1.0, H, dtype=torch.float32).view(1, 1, H, 1).expand(1, -1, -1, W)
    grid = torch.cat([th, tv], 1) + torch.cat([flow[:, 0:1] / ((W - 1.0) / 2.0), flow[:, 1:2] / ((H - 1.0) / 2.0)], 1)
    ref = {m: F.grid_sample(x, grid.permute(0, 2, 3, 1), mode=m, padding_mode="border", align_corners=True).numpy()[0]
           for m in ("bilinear", "bicubic")}
    fl, xn = flow.numpy()[0], x.numpy()[0]
    gx = (linspace(W)[None, :] + (fl[0] / f32((W - 1.0) / 2.0)).astype(f32)).astype(f32)
    gy = (linspace(H)[:, None] + (fl[1] / f32((H - 1.0) / 2.0)).astype(f32)).astype(f32)
    print("grid       ", float((gx == grid.numpy()[0, 0]).mean()), float((gy == grid.numpy()[0, 1]).mean()))
    ixu = ((gx + f32(1)) * f32((W - 1) / 2)).astype(f32)
    iyu = ((gy + f32(1)) * f32((H - 1) / 2)).astype(f32)
    # ---- bilinear
    ix = np.minimum(f32(W - 1), np.maximum(ixu, f32(0)))
    iy = np.minimum(f32(H - 1), np.maximum(iyu, f32(0)))
    x0, y0 = np.floor(ix), np.floor(iy)
    w = (ix - x0).astype(f32); e = (f32(1) - w).astype(f32); n = (iy - y0).astype(f32); s = (f32(1) - n).astype(f32)
    x0i, y0i = x0.astype(int), y0.astype(int)
    x1i, y1i = np.minimum(x0i + 1, W - 1), np.minimum(y0i + 1, H - 1)
    out = np.empty((3, H, W), f32)
    for c in range(3):
        acc = mul(xn[c][y0i, x0i], mul(s, e))
        acc = fma(xn[c][y0i, x1i], mul(s, w), acc)
        acc = fma(xn[c][y1i, x0i], mul(n, e), acc)
        out[c] = fma(xn[c][y1i, x1i], mul(n, w), acc)
    print("bilinear   ", float((out == ref["bilinear"]).mean()))
    # ---- bicubic
    A = f32(-0.75)
    xf, yf = np.floor(ixu), np.floor(iyu)
    tx, ty = (ixu - xf).astype(f32), (iyu - yf).astype(f32)

    def inner(t):
        u = mul(fma(np.full_like(t, A + f32(2)), t, np.full_like(t, -(A + f32(3)))), t)
        return fma(u, t, np.full_like(t, f32(1)))

    def outer(t):
        u = mul(np.full_like(t, A), t)
        u = (u - f32(5) * A).astype(f32)
        u = mul(u, t)
        u = (u + f32(8) * A).astype(f32)
        u = mul(u, t)
        return (u - f32(4) * A).astype(f32)

    def coeffs(t):
        return [outer((t + f32(1)).astype(f32)), inner(t), inner((f32(1) - t).astype(f32)), outer((f32(2) - t).astype(f32))]

    cx, cy = coeffs(tx), coeffs(ty)
    x0i, y0i = xf.astype(int), yf.astype(int)
    for c in range(3):
        rows = []
        for i in range(4):
            yy = np.clip(y0i - 1 + i, 0, H - 1)
            p = [xn[c][yy, np.clip(x0i - 1 + j, 0, W - 1)] for j in range(4)]
            acc = fma(p[0], cx[0], mul(p[1], cx[1]))
            acc = (acc + mul(p[2], cx[2])).astype(f32)
            rows.append((acc + mul(p[3], cx[3])).astype(f32))
        acc = fma(rows[1], cy[1], mul(rows[0], cy[0]))
        acc = fma(rows[2], cy[2], acc)
        out[c] = fma(rows[3], cy[3], acc)
    print("bicubic    ", float((out == ref["bicubic"]).mean()))


if __name__ == "__main__":
    main()
