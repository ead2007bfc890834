"""Light-profile counters (CCD_PIPE_PROFILE=1 build; =2 adds the producer-0 task timeline) of one kodak24 stream decoded alone: python tools/prof_stream.py IDX"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib

items, _ = bench.build_kodak24(0)
for idx in [int(a) for a in sys.argv[1:]] or [0]:
    hdr, nn, lat, (h, w) = items[idx]
    b = DecodeBatch(0)
    b.add(hdr, nn, lat, 8, 0)
    for _ in range(2):
        torch.cuda.synchronize(); t = time.time(); b.run(stage=0); b.wait(); dt = time.time() - t
    st = np.zeros(64, np.int32); lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
    hh = b.header(0)
    print("stream %d %dx%d: entropy %.1f ms" % (idx, h, w, dt * 1e3))
    for g in range(4):
        n = hh.grid_h[g] * hh.grid_w[g]
        tot, polls, ev = int(st[50 + 3 * g]) * 1024, int(st[51 + 3 * g]), int(st[52 + 3 * g])
        print("  grid %d (%dx%d): total %.1fM ticks = %.0f / symbol; %d polls of a ready counter (~%.0f%% of the grid's time at 200 ticks each), %d long waits" % (g, hh.grid_h[g], hh.grid_w[g], tot / 1e6, tot / n, polls, 100.0 * polls * 200 / max(tot, 1), ev))
    u = st[4:24].view(np.uint64); e = st[40:50].view(np.uint64); nt = max(int(e[4]), 1)
    g03 = sum(int(st[50 + 3 * g]) * 1024 for g in range(4))
    print("  grids 0-3 together %.1fM ticks of about %.0fM (wall time x 2.4 GHz): the rest is the coarser grids, the feature passes and the set-up" % (g03 / 1e6, dt * 2.4e3))
    print("  producer 0: %d tasks; per task: idle before early wait %.0f, early work %.0f, late wait %.0f, late work %.0f; between tasks (loop control, IFCE prefetch issue) %.0f (CCD_PIPE_PROFILE=2 builds only)" % (nt, u[6] / nt, u[7] / nt, u[8] / nt, u[9] / nt, e[0] / nt))
    print("  polls of a ready counter inside the decoder's asm region (each ~250 ticks of stall not in the per-grid figures): %d" % int(st[38]))
    print("  symbol-loop exits (renormalisation / miss / sentinel): %d, of which full 128-way searches: %d" % (int(st[62]), int(st[63])))
    print("  grid 0 stall Mticks by batch position in the step:", [round(int(x) * 1024 / 1e6, 2) for x in st[32:38]])
    b.close()
