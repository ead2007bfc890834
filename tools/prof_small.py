"""Heavy-profile (CCD_PIPE_PROFILE=3 build) phase times of the producers on a kodim14-architecture stream cropped to
H x W (default 128 x 192: every grid runs 2-pixel tasks; 256 x 384: the top grid runs 4-pixel tasks)."""
import os, sys, time
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
    torch.cuda.synchronize(); t = time.time(); b.run(stage=0); b.wait(); dt = time.time() - t
st = np.zeros(64, np.int32); lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
u = st[4:24].view(np.uint64); e = st[40:50].view(np.uint64)
nsym = int(arch.n_symbols); nt = max(int(e[4]), 1)
print("%dx%d: %d symbols, entropy %.2f ms = %.0f ticks/symbol" % (H, W, nsym, dt * 1e3, float(u[0]) / nsym))
print(" decoder: wait %.2fM work %.2fM (%.0f / symbol)" % (u[1] / 1e6, u[2] / 1e6, float(u[2]) / nsym))
print(" producer1: %d tasks; per task wait %.0f gather %.0f mlp %.0f (reload %.0f stab %.0f hidden %.0f out+meta %.0f) table %.0f" % (nt, u[6] / nt, u[7] / nt, u[8] / nt, e[0] / nt, e[1] / nt, e[2] / nt, e[3] / nt, u[9] / nt))
