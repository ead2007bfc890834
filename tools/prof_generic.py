#!/usr/bin/env python3
"""Per-phase ticks of the GENERIC entropy kernel (ccd_entropy.hip) for a landscape and a portrait kodak24 stream, thread 0's view
(each phase with the barrier behind it): gather, stabiliser + hidden layers, output layer, 128-entry tables, symbol loop.
    python -c "from cool_chic_amd import _build; _build.build_variant('genprof', '-DCCD_GEN_PROFILE')"
    CCD_FORCE_GENERIC=1 CCD_LIB=cool_chic_amd/libccd_genprof.so python tools/prof_generic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cool_chic_amd import DecodeBatch
items, _ = bench.build_kodak24(0)
for idx in (0, 3):
    b = DecodeBatch(0); b.add(*items[idx][:3], 8, 0); b.run(stage=0); b.wait()
    st = b.slot_stats(0)
    names = ["gather", "layers", "output", "tables", "symbols"]
    tot = sum(int(st[40 + k]) for k in range(5))
    print("stream", idx, {n: int(st[40 + k]) * 1024 for k, n in enumerate(names)}, "Mticks total", tot * 1024 / 1e6)
    b.close()
