"""Entropy-stage time of every stream of the kodak24 workload decoded alone (one workgroup), to see the spread
that the max-over-streams step time hides.  Run on the GPU box: python tools/per_stream.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib

items, streams = bench.build_kodak24(0)
rows = []
for i, (hdr, nn, lat, (h, w)) in enumerate(items):
    b = DecodeBatch(0)
    b.add(hdr, nn, lat, 8, 0)
    for _ in range(2):
        torch.cuda.synchronize(); t = time.time(); b.run(stage=0); b.wait(); dt = time.time() - t
    st = np.zeros(64, np.int32); lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
    rows.append((i, h, w, len(lat), dt * 1e3))
    b.close()
for r in rows:
    print("stream %2d %dx%d payload %6d B (%.3f bpp)  entropy %.1f ms" % (r[0], r[1], r[2], r[3], r[3] * 8 / (r[1] * r[2]), r[4]))
print("max %.1f ms, mean %.1f ms" % (max(r[4] for r in rows), sum(r[4] for r in rows) / len(rows)))
