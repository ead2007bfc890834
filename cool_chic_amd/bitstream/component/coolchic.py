"""encode_decode_coolchic on the MI355X.

Mirror of coolchic/bitstream/component/coolchic.py:29-207 of the reference: same arguments, same
return shape ([1, C, H, W] float32 tensor, None | bytes), same ValueError for missing arguments. The tensor
lives on the GPU (the reference is CPU-only and has no device argument; pass device="cpu" to get a host copy).

mode="encode" (the boundary's second caller, bitstream/encode.py:83-89): the quantised latents are range-coded by the
bitstream writer, which walks the decoder's integer ARM / IFCE path on the host like the reference (latent.py:168-173);
the header's n_bytes_latent is filled in, the returned bytes are header + NN payload + latent payload, and the
synthesis output is what the decoder makes of exactly those bytes."""
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
        import numpy as np

        from ... import writer

        latents = [np.ascontiguousarray(t.detach().cpu().numpy()).reshape(t.shape[-2], t.shape[-1]) for t in enc_quantized_latent]
        c = header.c
        if len(latents) != c.n_grids or any(a.shape != (c.grid_h[g], c.grid_w[g]) for g, a in enumerate(latents)):
            raise ValueError("enc_quantized_latent does not match the header's latent grids")
        if any(a.min() < -64 or a.max() > 63 or not np.array_equal(a, np.round(a)) for a in latents if a.size):
            raise ValueError("quantised latents must be integers in [-64, 63]")  # AC_MAX_VAL, constants.py:11
        coolchic_bytes = writer.encode_coolchic(c, bytes_nn, [a.astype(np.int8) for a in latents])
        rest = header.read_header(coolchic_bytes)  # the written header carries n_bytes_latent (coolchic.py:169)
        dec_bytes_latent = rest[len(bytes_nn):]
        out, _ = encode_decode_coolchic(header, bytes_nn, "decode", dec_bytes_latent=dec_bytes_latent, verbosity=verbosity, device=device)
        return out, coolchic_bytes
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
    stream = torch.cuda.current_stream(gpu).cuda_stream
    if verbosity >= 2:
        # coolchic.py:38,69,92,150,171,199-205: the reference prints the seconds of its four sections (network decode, IFCE
        # set-up, latents, upsampling + synthesis).  Same line here, from the batch API run stage by stage with a wait behind
        # each: parse + fixed-point conversion + upload | 0 (the feature pass is part of the entropy kernel) | entropy stage |
        # float stages + the copy of the output.
        import time

        from ...batch import DecodeBatch

        t0 = time.time()
        batch = DecodeBatch(gpu.index or 0)
        try:
            batch.add(header.raw, bytes_nn, dec_bytes_latent, 0, 0)
            torch.cuda.synchronize(gpu)
            time_neural_net = time.time() - t0
            t0 = time.time()
            batch.run(stream, stage=0)
            batch.wait(stream)
            time_latent = time.time() - t0
            t0 = time.time()
            batch.run(stream, stage=1)
            batch.run(stream, stage=2)
            batch.wait(stream)
            out = torch.as_tensor(batch.output_device(0), device=gpu).clone()
            torch.cuda.synchronize(gpu)
            time_syn = time.time() - t0
        finally:
            batch.close()
        print(header.pretty_string())
        print(f"{time_neural_net:6.2f} {0.0:6.2f} {time_latent:6.2f} {time_syn:6.2f} ")
        return (out if want.type == "cuda" else out.cpu()), None
    out = torch.empty((1, c.out_channels, c.img_size[0], c.img_size[1]), dtype=torch.float32, device=gpu)
    check(lib().ccd_decode_coolchic(header.raw, len(header.raw), bytes_nn, len(bytes_nn), dec_bytes_latent,
                                    len(dec_bytes_latent), gpu.index or 0, C.c_void_p(stream or None),
                                    C.c_void_p(out.data_ptr()), 1), "ccd_decode_coolchic")
    if verbosity:
        print(header.pretty_string())
    return (out if want.type == "cuda" else out.cpu()), None
