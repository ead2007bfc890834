#!/usr/bin/env python3
"""The rate model (ccd_compute_rate, 2^26 symbols, 12 launches) - the command tools/collect_profiles.sh puts under rocprofv3 for the
one HBM-bound kernel of the build (16 B per symbol).     python tools/prof_rate.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from cool_chic_amd._lib import check, lib  # noqa: E402

n = 1 << 26
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randint(-64, 64, (n,), generator=g, device=dev).float()
mu = x + torch.randn(n, generator=g, device=dev) * 1.5
sc = torch.exp(torch.rand(n, generator=g, device=dev) * 4 - 2)
out = torch.empty_like(x)
st = torch.cuda.current_stream(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def once():
    check(lib().ccd_compute_rate(0, C.c_void_p(st.cuda_stream or None), C.c_void_p(x.data_ptr()), C.c_void_p(mu.data_ptr()),
                                 C.c_void_p(sc.data_ptr()), n, C.c_void_p(out.data_ptr()), None), "ccd_compute_rate")


once(); once()
torch.cuda.synchronize()
e0.record(st)
for _ in range(10):
    once()
e1.record(st)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"workload": "rate_model", "symbols": n, "ms_per_launch": ms, "algorithmic_gbs": 16.0 * n / ms / 1e6}))
