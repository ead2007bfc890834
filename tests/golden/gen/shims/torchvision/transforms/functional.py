import numpy as np
import torch


def to_tensor(pic):
    """HWC uint8 PIL image -> CHW float32 in [0, 1]."""
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    if t.dtype == torch.uint8:
        return t.to(torch.float32).div(255)
    return t.to(torch.float32)
