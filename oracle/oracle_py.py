"""ctypes binding of the CPU oracle (oracle/libcc_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module. The product package (cool_chic_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcc_oracle.so")

MAX_GRIDS = 40
MAX_SYN = 8
MAX_ARM = 9


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("cc_oracle.c", "cc_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class SynLayer(C.Structure):
    _fields_ = [("out_ft", C.c_int), ("k_size", C.c_int), ("mode", C.c_int), ("nl", C.c_int)]


class CCHeader(C.Structure):
    _fields_ = [
        ("linear_stabiliser_synth", C.c_int), ("n_layer_synthesis", C.c_int), ("ups_k_size", C.c_int),
        ("ups_preconcat_k_size", C.c_int), ("output_feature_ifce", C.c_int), ("spatial_context_arm", C.c_int),
        ("linear_stabiliser_arm", C.c_int), ("n_hidden_layers_arm", C.c_int), ("img_size", C.c_int * 2),
        ("latent_resolution", C.c_int * 2), ("n_latent_grids", C.c_int), ("flag_hyperlatent", C.c_int),
        ("flag_common_randomness", C.c_int), ("final_upsampling_type", C.c_int), ("nn_q_step_log2", C.c_int * 8),
        ("nn_expgol_cnt", C.c_int * 8), ("nn_n_bytes", C.c_int), ("nn_n_bit_pad", C.c_int),
        ("n_bytes_latent", C.c_int), ("n_bytes_header", C.c_int), ("has_ifce_resolution", C.c_int),
        ("ifce_resolution", C.c_int * 2), ("hyperlatent_resolution", C.c_int * 2), ("syn_layer", SynLayer * MAX_SYN),
    ]


class Geometry(C.Structure):
    _fields_ = [
        ("n_grids", C.c_int), ("grid_h", C.c_int * MAX_GRIDS), ("grid_w", C.c_int * MAX_GRIDS),
        ("is_hyper", C.c_int * MAX_GRIDS), ("input_features_ifce", C.c_int * MAX_GRIDS), ("flag_ifce", C.c_int),
        ("input_feature_synthesis", C.c_int), ("total_context_arm", C.c_int), ("n_ups", C.c_int),
    ]


class CCResult(C.Structure):
    _fields_ = [
        ("hdr", CCHeader), ("geo", Geometry), ("n_nn_ints", C.c_int), ("nn_ints", C.POINTER(C.c_int64)),
        ("arm_n_layers", C.c_int), ("arm_dim", C.c_int), ("arm_w", C.POINTER(C.c_int64) * MAX_ARM),
        ("arm_b", C.POINTER(C.c_int64) * MAX_ARM), ("arm_ws", C.POINTER(C.c_int64)), ("arm_bs", C.c_int64 * 2),
        ("latent", C.POINTER(C.c_int8) * MAX_GRIDS), ("mu_scale_idx", C.POINTER(C.c_int32) * MAX_GRIDS),
        ("ctx_ifce", C.POINTER(C.c_int32) * MAX_GRIDS), ("n_symbols", C.c_uint64), ("words_consumed", C.c_uint64),
        ("dense_c", C.c_int), ("dense_h", C.c_int), ("dense_w", C.c_int), ("dense", C.POINTER(C.c_float)),
        ("out_c", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int), ("syn_out", C.POINTER(C.c_float)),
        ("out", C.POINTER(C.c_float)),
    ]


class VideoHeader(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("n_intras", C.c_int), ("n_p_frames", C.c_int), ("n_bytes_header", C.c_int),
                ("intra_pos", C.c_int * 4096), ("p_pos", C.c_int * 4096)]


class FrameHeader(C.Structure):
    _fields_ = [("display_index", C.c_int), ("frame_type", C.c_int), ("frame_data_type", C.c_int),
                ("bitdepth", C.c_int), ("n_bytes_header", C.c_int), ("n_refs", C.c_int),
                ("index_references", C.c_int * 2), ("global_flow", C.c_int * 4), ("warp_filter_size", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("display_index", C.c_int), ("frame_type", C.c_int), ("frame_data_type", C.c_int),
                ("bitdepth", C.c_int), ("h", C.c_int), ("w", C.c_int), ("ch", C.c_int), ("cw", C.c_int),
                ("plane", C.POINTER(C.c_uint16) * 3)]


class Video(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("frames", C.POINTER(Frame))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ora_read_video_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(VideoHeader)]
        L.ora_read_frame_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FrameHeader)]
        L.ora_read_cc_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(CCHeader)]
        L.ora_geometry_from_header.argtypes = [C.POINTER(CCHeader), C.POINTER(Geometry)]
        L.ora_decode_exp_golomb.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.c_int,
                                            C.POINTER(C.c_int64)]
        L.ora_laplace_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ora_laplace_bounds.restype = None
        L.ora_laplace_lefts_check.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_laplace_lefts_check.restype = C.c_int64
        L.ora_scale_table.argtypes = [C.c_int]
        L.ora_scale_table.restype = C.c_float
        L.ora_decode_coolchic.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                          C.c_int, C.POINTER(CCResult)]
        L.ora_cc_result_free.argtypes = [C.POINTER(CCResult)]
        L.ora_cc_result_free.restype = None
        L.ora_decode_video.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Video)]
        L.ora_video_free.argtypes = [C.POINTER(Video)]
        L.ora_video_free.restype = None
        L.ora_rc_encoder_new.restype = C.c_void_p
        L.ora_rc_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ora_rc_encode.restype = None
        L.ora_rc_get_compressed.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32))]
        L.ora_rc_get_compressed.restype = C.c_size_t
        L.ora_rc_encoder_free.argtypes = [C.c_void_p]
        L.ora_rc_encoder_free.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).reshape(shape).copy()


def read_cc_header(raw: bytes):
    """(CCHeader, Geometry) of one cool-chic header: sections 2 and 3 of cc_oracle.c."""
    L = lib()
    h, g = CCHeader(), Geometry()
    used = L.ora_read_cc_header(raw, len(raw), C.byref(h))
    if used < 0:
        raise OracleError(f"cc header: {used}")
    rc = L.ora_geometry_from_header(C.byref(h), C.byref(g))
    if rc < 0:
        raise OracleError(f"geometry: {rc}")
    return h, g


def split_stream(bitstream: bytes):
    """Walk a .cool file: returns (video_header, [(frame_header, [(hdr_bytes, nn_bytes, lat_bytes), ...])])."""
    L = lib()
    vh = VideoHeader()
    used = L.ora_read_video_header(bitstream, len(bitstream), C.byref(vh))
    if used < 0:
        raise OracleError(f"video header: {used}")
    pos = used
    frames = []
    for _ in range(vh.n_frames):
        fh = FrameHeader()
        rest = bitstream[pos:]
        used = L.ora_read_frame_header(rest, len(rest), C.byref(fh))
        if used < 0:
            raise OracleError(f"frame header: {used}")
        pos += used
        ccs = []
        for _cc in range(2 if fh.frame_type in (1, 2) else 1):
            ch = CCHeader()
            rest = bitstream[pos:]
            used = L.ora_read_cc_header(rest, len(rest), C.byref(ch))
            if used < 0:
                raise OracleError(f"cc header: {used}")
            hdr = bitstream[pos:pos + used]
            pos += used
            nn = bitstream[pos:pos + ch.nn_n_bytes]
            pos += ch.nn_n_bytes
            lat = bitstream[pos:pos + ch.n_bytes_latent]
            pos += ch.n_bytes_latent
            ccs.append((hdr, nn, lat))
        frames.append((fh, ccs))
    return vh, frames


def decode_coolchic(hdr: bytes, nn: bytes, lat: bytes, stop_after_entropy: bool = False) -> dict:
    L = lib()
    r = CCResult()
    rc = L.ora_decode_coolchic(hdr, len(hdr), nn, len(nn), lat, len(lat), int(stop_after_entropy), C.byref(r))
    try:
        if rc < 0:
            raise OracleError(f"ora_decode_coolchic failed: {rc}")
        g = r.geo
        n = g.n_grids
        out = {
            "n_grids": n,
            "grid_hw": [(g.grid_h[i], g.grid_w[i]) for i in range(n)],
            "is_hyper": [bool(g.is_hyper[i]) for i in range(n)],
            "input_features_ifce": [g.input_features_ifce[i] for i in range(n)],
            "nn_ints": _np(r.nn_ints, (r.n_nn_ints,), np.int64),
            "n_symbols": int(r.n_symbols),
            "words_consumed": int(r.words_consumed),
            "latent": [], "mu_scale_idx": [], "ctx_ifce": [],
            "arm_w": [], "arm_b": [],
        }
        dim = r.arm_dim
        for l in range(r.arm_n_layers):
            o = 2 if l == r.arm_n_layers - 1 else dim
            out["arm_w"].append(_np(r.arm_w[l], (dim, o), np.int64))
            out["arm_b"].append(_np(r.arm_b[l], (o,), np.int64))
        out["arm_ws"] = _np(r.arm_ws, (dim, 2), np.int64)
        out["arm_bs"] = np.array([r.arm_bs[0], r.arm_bs[1]], dtype=np.int64)
        n_ifce = r.hdr.output_feature_ifce
        for i in range(n):
            hw = out["grid_hw"][i]
            out["latent"].append(_np(r.latent[i], hw, np.int8))
            out["mu_scale_idx"].append(_np(r.mu_scale_idx[i], (hw[0] * hw[1], 2), np.int32))
            out["ctx_ifce"].append(_np(r.ctx_ifce[i], (n_ifce,) + hw, np.int32) if r.ctx_ifce[i] else None)
        if not stop_after_entropy:
            out["dense"] = _np(r.dense, (r.dense_c, r.dense_h, r.dense_w), np.float32)
            out["syn_out"] = _np(r.syn_out, (r.out_c, r.dense_h, r.dense_w), np.float32)
            out["out"] = _np(r.out, (r.out_c, r.out_h, r.out_w), np.float32)
        return out
    finally:
        L.ora_cc_result_free(C.byref(r))


def decode_video(bitstream: bytes) -> list:
    """Returns a list (display order) of dicts with integer planes."""
    L = lib()
    v = Video()
    rc = L.ora_decode_video(bitstream, len(bitstream), C.byref(v))
    try:
        if rc < 0:
            raise OracleError(f"ora_decode_video failed: {rc}")
        frames = []
        for i in range(v.n_frames):
            f = v.frames[i]
            planes = [_np(f.plane[0], (f.h, f.w), np.uint16)] + [_np(f.plane[p], (f.ch, f.cw), np.uint16) for p in (1, 2)]
            frames.append({"display_index": f.display_index, "frame_type": "IPB"[f.frame_type],
                           "frame_data_type": ["rgb", "yuv420", "yuv444", "flow"][f.frame_data_type],
                           "bitdepth": f.bitdepth, "planes": planes})
        return frames
    finally:
        L.ora_video_free(C.byref(v))


def laplace_bounds(mu_idx: int, scale_idx: int, s: int):
    L = lib()
    a, b = C.c_uint32(), C.c_uint32()
    L.ora_laplace_bounds(mu_idx, scale_idx, s, C.byref(a), C.byref(b))
    return a.value, b.value


def rc_encode(symbols, mu_idx, scale_idx) -> bytes:
    L = lib()
    e = L.ora_rc_encoder_new()
    try:
        for s, m, c in zip(symbols, mu_idx, scale_idx):
            L.ora_rc_encode(e, int(s), int(m), int(c))
        p = C.POINTER(C.c_uint32)()
        n = L.ora_rc_get_compressed(e, C.byref(p))
        arr = np.ctypeslib.as_array(p, shape=(n,)).astype("<u4").copy() if n else np.zeros(0, "<u4")
        C.CDLL(None).free(p)
        return arr.tobytes()
    finally:
        L.ora_rc_encoder_free(e)


def laplace_lefts_check(scale_idx: int, dev: np.ndarray, cap: int = 16):
    """dev: uint32 [32768, 127] = a device's left cumulatives for every (mu_idx, s = -63 .. 63) at scale_idx.  Returns
    (number of mismatches against libm, first mismatches as rows (mu_idx, s, device, libm)).  Releases the GIL."""
    L = lib()
    assert dev.dtype == np.uint32 and dev.shape == (32768, 127) and dev.flags.c_contiguous
    bad = np.zeros((cap, 4), dtype=np.int64)
    n = int(L.ora_laplace_lefts_check(int(scale_idx), dev.ctypes.data, bad.ctypes.data, cap))
    return n, bad[:min(n, cap)]
