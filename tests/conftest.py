import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

IMAGE_STREAMS = ["kodim14", "rgb192", "yuv420_8b", "yuv420_10b", "yuv444_10b", "cr192", "bicubic190", "bilinear190"]
SMALL_STREAMS = ["rgb192", "yuv420_8b", "yuv420_10b", "yuv444_10b", "cr192", "bicubic190", "bilinear190"]


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
