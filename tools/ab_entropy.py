#!/usr/bin/env python3
"""A/B of entropy-kernel builds on ONE box: for each library variant given as label=path, the entropy stage of kodak24
(24 streams in one launch, HIP events) and of the first landscape / portrait stream alone, latents checked against the
first variant.   python tools/ab_entropy.py base=cool_chic_amd/libccd.so t16=cool_chic_amd/libccd_t16.so ...
A variant may carry environment settings: valu=cool_chic_amd/libccd.so:CCD_MFMA_ARM=0,CCD_X=1"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from cool_chic_amd import DecodeBatch
items, streams = bench.build_kodak24(0)
st = torch.cuda.current_stream(0); sh = st.cuda_stream
def ms(b, reps=8):
    for _ in range(2): b.run(sh, stage=0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(st)
    for _ in range(reps): b.run(sh, stage=0)
    e1.record(st); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
b = DecodeBatch(0)
for hdr, nn, lat, _ in items: b.add(hdr, nn, lat, 8, 0)
b.run(sh); b.wait(sh)
h = hashlib.sha256()
for s in range(len(items)):
    for g in range(b.header(s).n_grids): h.update(b.latent(s, g).tobytes())
t_all = ms(b); b.close()
out = []
for idx in (0, 3):
    b1 = DecodeBatch(0); b1.add(*items[idx][:3], 8, 0); b1.run(sh); b1.wait(sh); out.append(ms(b1)); b1.close()
print("RESULT %%.3f %%.3f %%.3f %%s" %% (t_all, out[0], out[1], h.hexdigest()[:16]))
''' % ROOT


def main():
    ref = None
    for arg in sys.argv[1:]:
        label, path = arg.split("=", 1)
        extra = {}
        if ":" in path:
            path, settings = path.split(":", 1)
            extra = dict(kv.split("=", 1) for kv in settings.split(","))
        env = dict(os.environ, CCD_LIB=os.path.abspath(path), **extra)
        try:
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=90)
        except subprocess.TimeoutExpired:
            print(label, "TIMEOUT (hang)", flush=True)
            continue
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(label, "FAILED", r.stderr[-400:])
            continue
        t_all, t_l, t_p, sha = line[0].split()[1:]
        ref = ref or sha
        print(f"{label:12s} kodak24 {float(t_all):7.2f} ms   landscape alone {float(t_l):7.2f} ms   portrait alone {float(t_p):7.2f} ms   "
              f"latents {'== first' if sha == ref else '!= first  <-- WRONG'}", flush=True)


if __name__ == "__main__":
    main()
