"""Level-2 profile (-DCCD_PIPE_PROFILE=2 build) of producer 0's tasks on a kodim14-architecture stream cropped to H x W
(512 768: 8-pixel tasks on the top grid; 256 384: 4-pixel; 128 192: 2-pixel everywhere): ticks per task spent idle before
the early wait, in the early work, in the late wait, in the late work (left neighbour -> table rows counted in).
    CCD_LIB=cool_chic_amd/libccd_prof2.so python tools/prof_tasks.py 128 192"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cool_chic_amd import DecodeBatch, writer
from cool_chic_amd._lib import lib
from oracle import oracle_py as O

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 192)
bs = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "kodim14.cool"), "rb").read()
hdr, nn, lat = O.split_stream(bs)[1][0][1][0]
b = DecodeBatch(0); b.add(hdr, nn, lat, 8, 0); b.run(stage=0); b.wait()
donor = writer.parse_cc_header(hdr)
latents = [b.latent(0, g) for g in range(donor.n_grids)]
b.close()
arch = writer.derive_arch(donor, img_size=(H, W))
stream = writer.encode_stream(writer.cc_header_bytes(arch), nn, [np.ascontiguousarray(a[: arch.grid_h[g], : arch.grid_w[g]]) for g, a in enumerate(latents)])
h2, n2, l2 = O.split_stream(stream)[1][0][1][0]
b = DecodeBatch(0); b.add(h2, n2, l2, 8, 0)
for _ in range(2):
    b.run(stage=0); b.wait()
st = np.zeros(64, np.int32); lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
u = st[4:24].view(np.uint64); e = st[40:50].view(np.uint64)
nsym = int(arch.n_symbols); nt = max(int(e[4]), 1)
print("%dx%d: %d symbols, decoder %.0f ticks / symbol; producer 0: %d tasks, per task: idle %.0f  early work %.0f  late wait %.0f  late work %.0f  between %.0f" %
      (H, W, nsym, float(u[0]) / nsym, nt, u[6] / nt, u[7] / nt, u[8] / nt, u[9] / nt, e[0] / nt))
