import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

IMAGE_STREAMS = ["kodim14", "rgb192", "yuv420_8b", "yuv420_10b", "yuv444_10b", "cr192", "bicubic190", "bilinear190", "mop192", "vhop192", "odd191x127", "odd100x37", "odd18x65", "hq192", "yuv444_8b"]
SMALL_STREAMS = ["rgb192", "yuv420_8b", "yuv420_10b", "yuv444_10b", "cr192", "bicubic190", "bilinear190", "mop192", "vhop192", "odd191x127", "odd100x37", "odd18x65", "hq192", "yuv444_8b"]
# I / P / B videos encoded by the reference encoder: vid5 = lop presets (+ its two warp-filter variants); vid3_* = the other
# decoder presets of cfg/dec (tests/golden/gen/encode_presets.sh): intra vhop + residue hop + motion mop, intra mop + residue
# mop + motion lop, intra lop + residue vlop + motion mop; vid3_ldp = low-delay I P P (a P frame predicted from a P frame)
VIDEO_STREAMS = ["vid5", "vid5_w2", "vid5_w4", "vid3_hop", "vid3_mop", "vid3_vlop", "vid3_ldp"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".cool"), "rb") as f:
        bs = f.read()
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        j = json.load(f)
    return bs, z, j


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.build()
    return oracle_py


def reference_planes(z, j, frame="0"):
    """Integer planes the REFERENCE decoder produced (fixture), as a list of 3 arrays."""
    fr = j["frames"][frame]
    if fr["frame_data_type"] == "yuv420":
        return [z[f"frame{frame}.{k}"] for k in "yuv"]
    d = z[f"frame{frame}.data"]
    return [d[0], d[1], d[2]]


def load_arm_sweep():
    """tests/golden/arm_sweep.npz (tests/golden/gen/make_arm_sweep.py): {name: (stream bytes, [sha256 of every latent grid the
    REFERENCE decoder decoded, finest first])} for ARM shapes d<inputs>_h<hidden layers>_i<IFCE features>."""
    z = np.load(os.path.join(GOLDEN, "arm_sweep.npz"))
    return {str(n): (z[f"{n}.stream"].tobytes(), [str(x) for x in z[f"{n}.latent_sha256"]]) for n in z["names"]}
