#!/usr/bin/env python3
"""CPU baseline (ii) of SURVEY.md section 8d / BASELINE.md section 3 (build container only): the REFERENCE's own PyTorch
decoder on samples/bitstreams/kodim14.cool, imported from /root/reference through the shims of tests/golden/gen/shims, with
the pure-Python stand-in for constriction's range decoder replaced by the C oracle's coder (ctypes), so that the figure is
not dominated by a Python range decoder.  torch threads 1 and 8.  Writes profiles/r02/reference_pytorch_container.json,
which bench.py quotes as a static field (the reference cannot travel to the GPU box).

    python tools/ref_baseline.py
"""
import ctypes as C
import json
import os
import sys
import time

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "gen", "shims"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle_py  # noqa: E402

oracle_py.build()
L = C.CDLL(os.path.join(ROOT, "oracle", "libcc_oracle.so"))
L.ora_rc_decoder_new.restype = C.c_void_p
L.ora_rc_decoder_new.argtypes = [C.c_char_p, C.c_size_t]
L.ora_rc_decode_many.restype = C.c_int
L.ora_rc_decode_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
L.ora_rc_decoder_free.argtypes = [C.c_void_p]

import constriction.stream.queue as q  # noqa: E402  (the shim)

SCALES = np.load("/root/reference/coolchic/bitstream/component/mu_scale.npy")  # [mu table | scale table], rangecoder.py:36-38
SCALE_TAB = np.asarray(SCALES[-2561:], dtype=np.float32)


class CRangeDecoder:
    """constriction.stream.queue.RangeDecoder over the C oracle's coder."""

    def __init__(self, words):
        w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
        self.h = L.ora_rc_decoder_new(w.tobytes(), w.size * 4)

    def decode(self, model, mus, scales):
        mus = np.asarray(mus, dtype=np.float32)
        scales = np.asarray(scales, dtype=np.float32)
        mu_idx = np.rint((mus.astype(np.float64) + 64.0) * 256.0).astype(np.int32)
        sc_idx = np.searchsorted(SCALE_TAB, scales).astype(np.int32)
        assert np.array_equal(SCALE_TAB[sc_idx], scales)
        out = np.empty(mus.shape[0], dtype=np.int32)
        rc = L.ora_rc_decode_many(self.h, mu_idx.ctypes.data, sc_idx.ctypes.data, int(mus.shape[0]), out.ctypes.data)
        if rc != 0:
            raise ValueError("invalid compressed data")
        return out

    def __del__(self):
        L.ora_rc_decoder_free(self.h)


q.RangeDecoder = CRangeDecoder
import coolchic.bitstream.component.rangecoder as rcmod  # noqa: E402

if hasattr(rcmod, "constriction"):
    rcmod.constriction.stream.queue.RangeDecoder = CRangeDecoder
import coolchic.bitstream.decode as bdec  # noqa: E402

PATH = "/root/reference/samples/bitstreams/kodim14.cool"
res = {"what": "reference decode_video(kodim14.cool) in the build container, torch CPU, constriction replaced by the C oracle's range "
               "decoder behind the import shim (tools/ref_baseline.py)",
       "host": f"{os.cpu_count()} cores, torch {torch.__version__}", "pixels": 512 * 768, "runs": {}}
want = None
for threads in (1, 8):
    torch.set_num_threads(threads)
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        frames = bdec.decode_video(PATH, decoded_path=None, verbosity=0)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    img = np.round(frames["0"].data.numpy()[0] * 255).astype(np.uint8)
    if want is None:
        z = np.load(os.path.join(ROOT, "tests", "golden", "kodim14.npz"))
        want = z["frame0.data"]
    res["runs"][f"threads_{threads}"] = {"seconds_per_frame": best, "mpixel_per_s": 512 * 768 / best / 1e6,
                                         "samples_differing_from_fixture": int((img != want).sum())}
os.makedirs(os.path.join(ROOT, "profiles", "r02"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "r02", "reference_pytorch_container.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
