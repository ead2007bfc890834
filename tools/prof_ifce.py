import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib
items, _ = bench.build_kodak24(0)
for idx in (0, 3):
    b = DecodeBatch(0); b.add(*items[idx][:3], 8, 0)
    for _ in range(2): b.run(stage=0); b.wait()
    st = np.zeros(64, np.int32); lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
    u = st[4:24].view(np.uint64)
    print(idx, "total %.2f M, decoder wait %.2f work %.2f, ifce passes %.2f M, barriers %.2f M" % tuple(float(x) / 1e6 for x in u[:5]), " ifce per grid (Kticks, grids 0-3):", [int(st[24 + 2 * g]) for g in range(4)])
    b.close()
