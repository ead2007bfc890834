"""P / B frame reconstruction on the MI355X (reference: coolchic/bitstream/decode.py:156-206,
component/intercoding/{warp,globalmotion}.py) through ccd_inter_reconstruct."""
import ctypes as C
from typing import List

import torch

from .._lib import check, lib
from ..io import FrameData

_FDT_INDEX = {"rgb": 0, "yuv420": 1, "yuv444": 2, "flow": 3}


def _integer_planes(fd: FrameData, device: torch.device) -> List[torch.Tensor]:
    """FrameData -> three integer planes on the device (u8 for 8-bit, u16 above)."""
    maxv = 2 ** fd.bitdepth - 1
    dt = torch.uint8 if fd.bitdepth == 8 else torch.uint16
    if fd.frame_data_type == "yuv420":
        src = [fd.data[k][0, 0] for k in ("y", "u", "v")]
    else:
        src = [fd.data[0, c] for c in range(3)]
    return [torch.round(p.to(device=device, dtype=torch.float32) * maxv).to(torch.int32).to(dt).contiguous() for p in src]


def reconstruct_inter_frame(frame_header, residue: torch.Tensor, motion: torch.Tensor,
                            reference_frames: List[FrameData]) -> FrameData:
    """residue: [1, 4|5, H, W], motion: [1, 2|4, H, W] CUDA float tensors (outputs of the two cool-chics)."""
    frame_type = frame_header.get_value("frame_type")
    bitdepth = frame_header.get_value("bitdepth")
    fdt = frame_header.get_value("frame_data_type")
    dev = residue.device
    h, w = residue.shape[-2:]
    n_refs = 2 if frame_type == "B" else 1
    if len(reference_frames) < n_refs:
        raise ValueError(f"a {frame_type} frame needs {n_refs} reference frame(s)")
    for fd in reference_frames[:n_refs]:
        # the references' planes are read with THIS frame's sample layout (a 4:2:0 reference has quarter-size chroma planes)
        if fd.frame_data_type != fdt or fd.bitdepth != bitdepth:
            raise ValueError(f"reference frame is {fd.frame_data_type} {fd.bitdepth}-bit, the frame is {fdt} {bitdepth}-bit")
    if fdt == "yuv420" and (h % 2 or w % 2):
        raise ValueError("4:2:0 frames need even sizes")
    refs = [_integer_planes(fd, dev) for fd in reference_frames[:n_refs]]
    dt = torch.uint8 if bitdepth == 8 else torch.uint16
    ch, cw = (h // 2, w // 2) if fdt == "yuv420" else (h, w)
    out = [torch.empty((h, w), dtype=dt, device=dev), torch.empty((ch, cw), dtype=dt, device=dev),
           torch.empty((ch, cw), dtype=dt, device=dev)]

    def ptrs(planes):
        return (C.c_void_p * 3)(*[p.data_ptr() for p in planes])

    gflow = (C.c_int32 * 4)(*(list(frame_header.get_value("global_flow")) + [0, 0, 0, 0])[:4])
    res = residue.contiguous()
    mot = motion.contiguous()
    stream = torch.cuda.current_stream(dev).cuda_stream
    check(lib().ccd_inter_reconstruct(dev.index or 0, C.c_void_p(stream or None), 1 if frame_type == "P" else 2, h, w, bitdepth,
                                      _FDT_INDEX[fdt], C.c_void_p(res.data_ptr()), C.c_void_p(mot.data_ptr()), ptrs(refs[0]),
                                      ptrs(refs[1]) if n_refs == 2 else None, gflow, frame_header.get_value("warp_filter_size"),
                                      ptrs(out)), "ccd_inter_reconstruct")
    maxv = float(2 ** bitdepth - 1)
    f = [p.to(torch.float32).div(maxv)[None, None] for p in out]
    if fdt == "yuv420":
        return FrameData(bitdepth, fdt, {"y": f[0], "u": f[1], "v": f[2]})
    return FrameData(bitdepth, fdt, torch.cat(f, dim=1))
