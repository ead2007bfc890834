"""Determinism stress of the pipelined entropy kernel: the kodak24 batch decoded N times, every latent grid and every
integer plane hashed after each run and compared with the first run (which the parity tests tie to the oracle)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from cool_chic_amd import DecodeBatch

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 50
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # kodak24 x copies streams in the batch (8: most CUs busy)
items, _ = bench.build_kodak24(0)
items = items * copies
b = DecodeBatch(0)
for hdr, nn, lat, _ in items:
    b.add(hdr, nn, lat, 8, 0)

def digest():
    h = hashlib.sha256()
    for s in range(len(items)):
        for g in range(b.header(s).n_grids):
            h.update(np.ascontiguousarray(b.latent(s, g)).tobytes())
        for p in b.planes(s):
            h.update(np.ascontiguousarray(p).tobytes())
    return h.hexdigest()

ref = None
for it in range(n_iter):
    b.run(); b.wait()
    d = digest()
    if ref is None:
        ref = d
    elif d != ref:
        print("MISMATCH at iteration", it); sys.exit(1)
print("%d iterations, %d streams each: identical (%s)" % (n_iter, len(items), ref[:16]))
