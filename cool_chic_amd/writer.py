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


def network_layout(arch: CCHeader) -> List[int]:
    """Transmitted integers per group (arm.w arm.b ifce.w ifce.b ups.w ups.b syn.w syn.b) for this architecture."""
    n = (C.c_int64 * 8)()
    check(lib().ccd_network_layout(C.byref(arch), n), "ccd_network_layout")
    return list(n)


def encode_network(arch: CCHeader, values: np.ndarray) -> bytes:
    """Exp-Golomb NN payload (neuralnet.py:27-90) of the quantised parameters `values` (stream order) with the
    orders in arch.nn_expgol_cnt.  Sets arch.nn_n_bytes / arch.nn_n_bit_pad like encode_frame does."""
    v = np.ascontiguousarray(values, dtype=np.int32)
    out = C.POINTER(C.c_uint8)()
    pad = C.c_int32(0)
    n = check(lib().ccd_encode_network(C.byref(arch), v.ctypes.data, v.size, C.byref(pad), C.byref(out)), "ccd_encode_network")
    try:
        payload = bytes(np.ctypeslib.as_array(out, shape=(n,))) if n else b""
    finally:
        lib().ccd_free(out)
    arch.nn_n_bytes, arch.nn_n_bit_pad = len(payload), pad.value
    return payload


def encode_coolchic(arch: CCHeader, bytes_nn: bytes, latents: Sequence[np.ndarray]) -> bytes:
    """Cool-chic header + NN payload + range-coded latents of one cool-chic (bitstream/encode.py:83-92)."""
    arrs = [np.ascontiguousarray(a, dtype=np.int8) for a in latents]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    out = C.POINTER(C.c_uint8)()
    n = check(lib().ccd_encode_coolchic(C.byref(arch), bytes_nn, len(bytes_nn), ptrs, C.byref(out)), "ccd_encode_coolchic")
    try:
        return bytes(np.ctypeslib.as_array(out, shape=(n,)))
    finally:
        lib().ccd_free(out)


def frame_header_bytes(display_index: int, frame_type: str, frame_data_type: int, bitdepth: int, index_references=(),
                       global_flow=(), warp_filter_size: int = 8) -> bytes:
    from ._lib import FrameHeader

    f = FrameHeader()
    f.display_index, f.frame_type, f.frame_data_type, f.bitdepth = display_index, "IPB".index(frame_type), frame_data_type, bitdepth
    f.n_refs = f.frame_type
    for i, r in enumerate(index_references):
        f.index_references[i] = int(r)
    for i, g in enumerate(global_flow):
        f.global_flow[i] = int(g)
    f.warp_filter_size = warp_filter_size
    buf = (C.c_uint8 * 64)()
    n = check(lib().ccd_write_frame_header(C.byref(f), buf, 64), "ccd_write_frame_header")
    return bytes(buf[:n])


def video_header_bytes(n_frames: int, intra_pos: Sequence[int], p_pos: Sequence[int]) -> bytes:
    from ._lib import VideoHeader

    v = VideoHeader()
    v.n_frames, v.n_intras, v.n_p_frames = n_frames, len(intra_pos), len(p_pos)
    for i, x in enumerate(intra_pos):
        v.intra_pos[i] = int(x)
    for i, x in enumerate(p_pos):
        v.p_pos[i] = int(x)
    cap = 16 + 2 * (len(intra_pos) + len(p_pos))
    buf = (C.c_uint8 * cap)()
    n = check(lib().ccd_write_video_header(C.byref(v), buf, cap), "ccd_write_video_header")
    return bytes(buf[:n])


def cc_header_bytes(arch: CCHeader) -> bytes:
    buf = (C.c_uint8 * 128)()
    n = check(lib().ccd_write_cc_header(C.byref(arch), buf, 128), "ccd_write_cc_header")
    return bytes(buf[:n])


def derive_arch(donor: CCHeader, **changes) -> CCHeader:
    """Copy of `donor` with some transmitted fields replaced (tuples for array fields, `syn_layers` = list of
    (out_ft, k, mode, non_linearity)); the derived geometry is recomputed by a serialise / parse round trip."""
    a = CCHeader.from_buffer_copy(bytes(donor))
    for k, v in changes.items():
        if k == "syn_layers":
            a.n_layer_synthesis = len(v)
            for i, (o, ks, m, nl) in enumerate(v):
                a.syn_layer[i].out_ft, a.syn_layer[i].k_size, a.syn_layer[i].mode, a.syn_layer[i].non_linearity = o, ks, m, nl
        elif isinstance(v, (tuple, list)):
            for i, x in enumerate(v):
                getattr(a, k)[i] = int(x)
        else:
            setattr(a, k, int(v))
    return parse_cc_header(cc_header_bytes(a))


def _cycle_cols(w: np.ndarray, n_cols: int) -> np.ndarray:
    return w[:, np.arange(n_cols) % w.shape[1]]


def fits_fast_path(cc_header: bytes, bytes_nn: bytes) -> bool:
    """Whether this cool-chic's integer ARM provably fits the pipelined entropy kernel's 32-bit operands (include/ccd.h)."""
    return check(lib().ccd_network_fits_fast_path(cc_header, len(cc_header), bytes_nn, len(bytes_nn)), "ccd_network_fits_fast_path") == 1


def adapt_network(donor: CCHeader, donor_ints: np.ndarray, arch: CCHeader, noise_gain: float = 0.25, ifce_gain: float = 0.25) -> np.ndarray:
    """Quantised parameters for `arch` grown from a trained donor network: groups of equal size are copied,
    the others are extended by cycling the donor's rows / columns (IFCE inputs the donor never saw scaled by ifce_gain,
    upsampling filters, the first synthesis layer's input channels; common-randomness channels get the donor columns
    scaled by noise_gain)."""
    dl, al = network_layout(donor), network_layout(arch)
    d = np.split(np.asarray(donor_ints, dtype=np.int64), np.cumsum(dl)[:-1])
    out = []
    for k in range(8):
        if dl[k] == al[k]:
            out.append(d[k]); continue
        if al[k] == 0:
            out.append(np.zeros(0, np.int64)); continue
        if dl[k] == 0:
            raise ValueError("the donor has no parameters in group %d" % k)
        if k == 2:  # ifce.w: per grid [n_out][n_in(g)]
            n_out = arch.output_feature_ifce
            fin_d = [f for f in donor.input_features_ifce[:donor.n_grids] if f > 0]
            w0 = d[k][:n_out * fin_d[0]].reshape(donor.output_feature_ifce, fin_d[0])
            w0 = w0[np.arange(n_out) % w0.shape[0]]
            def grown(f):  # inputs the donor never saw (more, coarser grids) get a quarter of the weight
                w = _cycle_cols(w0, f).astype(np.float64)
                w[:, w0.shape[1]:] *= ifce_gain
                return np.round(w).astype(np.int64).ravel()
            out.append(np.concatenate([grown(f) for f in arch.input_features_ifce[:arch.n_grids] if f > 0]))
        elif k == 4:  # ups.w: n_ups x (k/2) transposed-conv halves, then n_ups x ceil(k_pre/2)
            nd, na = donor.latent_resolution[1], arch.latent_resolution[1]
            if (donor.ups_k_size, donor.ups_preconcat_k_size) != (arch.ups_k_size, arch.ups_preconcat_k_size):
                raise ValueError("upsampling kernel sizes must match the donor's")
            ku, kp = (arch.ups_k_size + 1) // 2, (arch.ups_preconcat_k_size + 1) // 2
            t = d[k][:nd * ku].reshape(nd, ku); p = d[k][nd * ku:].reshape(nd, kp)
            out.append(np.concatenate([t[np.arange(na) % nd].ravel(), p[np.arange(na) % nd].ravel()]))
        elif k == 6:  # syn.w: output_transform, stabiliser, main layers
            c, ci_d, ci_a = donor.out_channels, donor.input_feature_synthesis, arch.input_feature_synthesis
            if arch.out_channels != c or arch.n_layer_synthesis != donor.n_layer_synthesis:
                raise ValueError("synthesis depth / output channels must match the donor's")
            pos = c * c
            parts = [d[k][:pos]]
            st_d = ci_d // 2 if donor.flag_common_randomness else ci_d
            st_a = ci_a // 2 if arch.flag_common_randomness else ci_a
            if donor.linear_stabiliser_synth:
                parts.append(_cycle_cols(d[k][pos:pos + c * st_d].reshape(c, st_d), st_a).ravel()); pos += c * st_d
            l0 = donor.syn_layer[0]
            w0 = d[k][pos:pos + l0.out_ft * ci_d * l0.k_size ** 2].reshape(l0.out_ft, ci_d, -1); pos += w0.size
            w0a = w0[:, np.arange(ci_a) % ci_d, :].astype(np.float64)
            if arch.flag_common_randomness:
                w0a[:, ci_a // 2:, :] *= noise_gain
            parts.append(np.round(w0a).astype(np.int64).ravel())
            parts.append(d[k][pos:])
            out.append(np.concatenate(parts))
        else:
            out.append(d[k][np.arange(al[k]) % dl[k]])
        if out[-1].size != al[k]:
            raise ValueError("group %d: built %d values, the layout wants %d" % (k, out[-1].size, al[k]))
    return np.concatenate(out).astype(np.int32)


def tile_latents(donor_latents: Sequence[np.ndarray], donor: CCHeader, arch: CCHeader) -> List[np.ndarray]:
    """Latent grids for `arch` made of tiled copies of a real pyramid: grid (level, kind) of the donor is
    repeated to cover the new size; levels the donor does not have are zero."""
    def keyed(h):
        keys, lv_prev = [], None
        sizes, levels = grid_sizes(tuple(h.img_size), cc_header_bytes(h))
        for i, lv in enumerate(levels):
            keys.append((lv, bool(h.is_hyperlatent[i])))
        return keys, sizes
    dk, _ = keyed(donor)
    ak, asz = keyed(arch)
    src = dict(zip(dk, donor_latents))
    out = []
    for key, (gh, gw) in zip(ak, asz):
        a = src.get(key)
        if a is None:
            out.append(np.zeros((gh, gw), np.int8)); continue
        reps = (-(-gh // a.shape[0]), -(-gw // a.shape[1]))
        out.append(np.ascontiguousarray(np.tile(a, reps)[:gh, :gw]).astype(np.int8))
    return out


def parse_cc_header(raw: bytes) -> CCHeader:
    h = CCHeader()
    check(lib().ccd_read_cc_header(raw, len(raw), C.byref(h)), "ccd_read_cc_header")
    return h


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
