"""FrameData + PNG / PPM / YUV writers (reference: coolchic/io/{framedata,io}.py, io/format/*.py)."""
import os
from dataclasses import dataclass, field
from typing import Any, Tuple

import numpy as np


@dataclass
class FrameData:
    """Same fields as the reference's FrameData (io/framedata.py:16-44): `data` is a [1, 3, H, W]
    float32 tensor in [0, 1] already on the bit-depth grid, or a {"y","u","v"} dict for yuv420."""

    bitdepth: int
    frame_data_type: str
    data: Any
    img_size: Tuple[int, int] = field(init=False)
    n_pixels: int = field(init=False)

    def __post_init__(self):
        ref = self.data.get("y") if self.frame_data_type == "yuv420" else self.data
        self.img_size = tuple(ref.shape[-2:])
        self.n_pixels = self.img_size[0] * self.img_size[1]

    def integer_planes(self):
        """[plane0, plane1, plane2] as integer numpy arrays (value = round(x * (2^bitdepth - 1)))."""
        maxv = 2 ** self.bitdepth - 1
        dt = np.uint8 if self.bitdepth == 8 else np.uint16

        def q(t):
            return np.round(t.detach().cpu().numpy().astype(np.float32) * maxv).astype(dt)

        if self.frame_data_type == "yuv420":
            return [q(self.data[k])[0, 0] for k in ("y", "u", "v")]
        return [q(self.data)[0, c] for c in range(3)]


def save_frame_data_to_file(frame_data: FrameData, file_path: str, append: bool = False) -> None:
    """io/io.py:53-105."""
    ext = os.path.splitext(file_path)[1]
    assert ext in (".yuv", ".png", ".ppm"), f"expected a .yuv, .png or .ppm path, found {file_path}"
    if ext == ".png":
        # io/format/png.py:44-62.  The file is packed on the GPU (csrc/ccd_png.hip): only compressed bytes leave HBM.
        assert frame_data.frame_data_type == "rgb" and frame_data.bitdepth == 8, "PNG output needs 8-bit RGB"
        import torch

        from .png import device_png_bytes

        data = frame_data.data
        if not data.is_cuda:
            data = data.cuda()
        planes = torch.round(data[0].to(torch.float32) * 255.0).to(torch.uint8)
        with open(file_path, "wb") as f:
            f.write(device_png_bytes(planes))
        return
    planes = frame_data.integer_planes()
    if ext == ".ppm":
        assert frame_data.frame_data_type == "rgb"
        h, w = planes[0].shape
        maxv = 2 ** frame_data.bitdepth - 1
        img = np.stack(planes, axis=-1)
        with open(file_path, "wb") as f:  # io/format/ppm.py:161-203 (binary P6, big-endian above 8 bits)
            f.write(f"P6\n{w} {h}\n{maxv}\n".encode())
            f.write(img.astype(">u2" if maxv > 255 else np.uint8).tobytes())
    else:
        assert frame_data.frame_data_type in ("yuv420", "yuv444")
        # io/format/yuv.py:152-160: uint16 only when bitdepth == 10, uint8 otherwise (reference quirk)
        dt = np.uint16 if frame_data.bitdepth == 10 else np.uint8
        with open(file_path, "ab" if append else "wb") as f:
            for p in planes:
                p.astype(dt).tofile(f)
