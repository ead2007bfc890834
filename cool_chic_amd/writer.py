"""Bitstream writer helpers (SURVEY.md section 8f "next-2"): range-encode latents into a .cool file and
derive synthetic Kodak-shaped streams from a real one for the benchmark configurations.

Reference: coolchic/bitstream/encode.py:24-95 (framing), rangecoder.py:46-76 (entropy coder)."""
import ctypes as C
from typing import List, Sequence

import numpy as np

from ._lib import CCHeader, check, lib


def range_encode(symbols: np.ndarray, mu_idx: np.ndarray, scale_idx: np.ndarray) -> bytes:
    """constriction RangeEncoder + QuantizedLaplace(-64, 63) over table indices (one call = one payload)."""
    s = np.ascontiguousarray(symbols, dtype=np.int8)
    m = np.ascontiguousarray(mu_idx, dtype=np.int32)
    c = np.ascontiguousarray(scale_idx, dtype=np.int32)
    out = C.POINTER(C.c_uint8)()
    n = check(lib().ccd_range_encode(s.ctypes.data, m.ctypes.data, c.ctypes.data, s.size, C.byref(out)), "ccd_range_encode")
    try:
        return bytes(np.ctypeslib.as_array(out, shape=(n,))) if n else b""
    finally:
        lib().ccd_free(out)


def encode_stream(cc_header: bytes, bytes_nn: bytes, latents: Sequence[np.ndarray], bitdepth: int = 8,
                  frame_data_type: int = 0, img_size=None) -> bytes:
    """One-intra-frame .cool file with the architecture of `cc_header`, NN payload `bytes_nn` and the given
    quantised latents (index 0 = finest grid). `img_size` overrides the header's (H, W)."""
    h = CCHeader()
    check(lib().ccd_read_cc_header(cc_header, len(cc_header), C.byref(h)), "ccd_read_cc_header")
    if img_size is not None:
        h.img_size[0], h.img_size[1] = int(img_size[0]), int(img_size[1])
    arrs = [np.ascontiguousarray(a, dtype=np.int8) for a in latents]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    out = C.POINTER(C.c_uint8)()
    n = check(lib().ccd_encode_stream(C.byref(h), bytes_nn, len(bytes_nn), ptrs, int(bitdepth), int(frame_data_type),
                                      C.byref(out)), "ccd_encode_stream")
    try:
        return bytes(np.ctypeslib.as_array(out, shape=(n,)))
    finally:
        lib().ccd_free(out)


def grid_sizes(img_size, cc_header: bytes):
    """(h, w) of every grid of a stream with this architecture at another image size."""
    h = CCHeader()
    check(lib().ccd_read_cc_header(cc_header, len(cc_header), C.byref(h)), "ccd_read_cc_header")
    levels = []
    lo, hi = h.latent_resolution[0], h.latent_resolution[1]
    first, last = lo, hi
    if h.flag_hyperlatent:
        first = min(lo, hi, h.hyperlatent_resolution[0], h.hyperlatent_resolution[1])
        last = max(lo, hi, h.hyperlatent_resolution[0], h.hyperlatent_resolution[1])
    for lv in range(first, last + 1):
        if lo <= lv <= hi:
            levels.append(lv)
        if h.flag_hyperlatent and h.hyperlatent_resolution[0] <= lv <= h.hyperlatent_resolution[1]:
            levels.append(lv)
    return [(-(-int(img_size[0]) // (1 << lv)), -(-int(img_size[1]) // (1 << lv))) for lv in levels], levels


def variant_latents(latents: List[np.ndarray], levels: List[int], seed: int, transpose: bool = False) -> List[np.ndarray]:
    """A deterministic, statistically realistic variation of real latents: one spatial roll of the whole
    pyramid (consistent across levels) plus optional transposition (portrait <-> landscape)."""
    rng = np.random.default_rng(seed)
    top = max(levels)
    # roll by a multiple of the coarsest cell so every level shifts by an integer amount
    ry = int(rng.integers(0, max(latents[0].shape[0] >> top, 1))) << top
    rx = int(rng.integers(0, max(latents[0].shape[1] >> top, 1))) << top
    out = []
    for a, lv in zip(latents, levels):
        b = np.roll(a, (ry >> lv, rx >> lv), axis=(0, 1))
        out.append(np.ascontiguousarray(b.T if transpose else b))
    return out
