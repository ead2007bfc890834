#!/usr/bin/env python3
"""PPM golden files (build container only): the bytes the REFERENCE's save_frame_data_to_file / write_ppm
(io/io.py:53-105, io/format/ppm.py:161-203) writes for an 8-bit, a 10-bit and a 16-bit RGB FrameData whose samples cover the
whole range.  -> tests/golden/ppm{8,10,16}.ppm + ppm_planes.npz (the integer planes they were written from)."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(1, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
from coolchic.io.framedata import FrameData  # noqa: E402
from coolchic.io.io import save_frame_data_to_file  # noqa: E402


def main():
    out_dir = os.path.abspath(os.path.join(HERE, ".."))
    rng = np.random.default_rng(7)
    arrays = {}
    for bd in (8, 10, 16):
        maxv = 2 ** bd - 1
        planes = rng.integers(0, maxv + 1, (3, 5, 7), dtype=np.int64)
        planes[0, 0, :4] = [0, maxv, 255 % (maxv + 1), 256 % (maxv + 1)]   # range ends and the byte boundary
        planes[1, 0, :2] = [257 % (maxv + 1), maxv - 1]
        arrays[f"ppm{bd}"] = planes.astype(np.uint16)
        data = torch.from_numpy(planes.astype(np.float32) / np.float32(maxv))[None]   # [1, 3, H, W] in [0, 1] on the bit-depth grid
        path = os.path.join(out_dir, f"ppm{bd}.ppm")
        save_frame_data_to_file(FrameData(bd, "rgb", data), path)
        print(path, os.path.getsize(path))
    np.savez_compressed(os.path.join(out_dir, "ppm_planes.npz"), **arrays)


if __name__ == "__main__":
    main()
