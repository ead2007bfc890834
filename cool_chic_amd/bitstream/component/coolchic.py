"""encode_decode_coolchic(mode="decode") on the MI355X.

Mirror of coolchic/bitstream/component/coolchic.py:29-207 of the reference: same arguments, same
return shape ([1, C, H, W] float32 tensor, None), same ValueError for missing arguments. The tensor
lives on the GPU (the reference decoder is CPU-only and has no device argument; pass
device="cpu" to get a host copy)."""
import ctypes as C
from typing import List, Literal, Optional, Tuple

import torch

from ..._lib import check, lib
from ..header import CoolChicHeader


def encode_decode_coolchic(
    header: CoolChicHeader,
    bytes_nn: bytes,
    mode: Literal["encode", "decode"],
    dec_bytes_latent: Optional[bytes] = None,
    enc_quantized_latent: Optional[List[torch.Tensor]] = None,
    verbosity: int = 0,
    device: str = "cuda:0",
) -> Tuple[torch.Tensor, Optional[bytes]]:
    if mode == "encode":
        if enc_quantized_latent is None:
            raise ValueError(
                "Trying to encode cool_chic latent without indicating the quantized latent value. "
                "Found enc_quantized_latent=None. It should be a list of integer Tensor."
            )
        raise NotImplementedError("the MI355X build accelerates the decode path only (SURVEY.md section 8)")
    if mode == "decode" and dec_bytes_latent is None:
        raise ValueError(
            "Trying to encode cool_chic latent with dec_bytes_latent=None. "
            "The argument dec_bytes_latent should represent the bytes of the bitstream."
        )
    if not torch.cuda.is_available():
        raise RuntimeError("cool_chic_amd has no CPU fallback: an MI355X (gfx950) device is required")
    want = torch.device(device)
    gpu = want if want.type == "cuda" else torch.device("cuda:0")
    c = header.c
    out = torch.empty((1, c.out_channels, c.img_size[0], c.img_size[1]), dtype=torch.float32, device=gpu)
    stream = torch.cuda.current_stream(gpu).cuda_stream
    check(lib().ccd_decode_coolchic(header.raw, len(header.raw), bytes_nn, len(bytes_nn), dec_bytes_latent,
                                    len(dec_bytes_latent), gpu.index or 0, C.c_void_p(stream or None),
                                    C.c_void_p(out.data_ptr()), 1), "ccd_decode_coolchic")
    if verbosity:
        print(header.pretty_string())
    return (out if want.type == "cuda" else out.cpu()), None
