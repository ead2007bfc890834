#!/usr/bin/env python3
"""Per-grid decoder counters of one kodak24 stream (light profile build):
    CCD_LIB=cool_chic_amd/libccd_prof1.so python tools/prof_grids.py [stream index = 0 (landscape); 3 = first portrait]
Needs a library built with -DCCD_PIPE_PROFILE=1 (python -c "from cool_chic_amd import _build; _build.build_variant('prof1', '-DCCD_PIPE_PROFILE=1')")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
items, _ = bench.build_kodak24(0)
b = DecodeBatch(0)
b.add(*items[idx][:3], 8, 0)
for _ in range(2):
    b.run(stage=0); b.wait()
hdr = b.header(0)
st = np.zeros(64, np.int32)
lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
u = st[4:24].view(np.uint64)
print(f"stream {idx}: rare-path symbols {st[62]}, full searches {st[63]}")
tot = 0
for g in range(min(4, hdr.n_grids)):
    h, w = b.latent(0, g).shape[-2:]
    ticks, polls = int(st[50 + 3 * g]) * 1024, int(st[51 + 3 * g])
    steps = w + 10 * (h - 1) if w > 9 else h * w
    tot += ticks
    print(f"  grid {g} {h}x{w}: {ticks / 1e6:7.2f} Mticks  {ticks / (h * w):6.1f} / symbol  {ticks / steps:7.0f} / step ({steps} steps, {h * w / steps:.1f} px)  "
          f"polls {polls} ({polls / steps:.2f} / step)")
print(f"  four finest grids: {tot / 1e6:.1f} Mticks")
