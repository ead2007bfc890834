"""ctypes view of the C ABI declared in include/ccd.h (libccd.so).

There is no CPU fallback: if the library is missing it is built with hipcc; if that fails, or no
gfx950 device is usable when a decode is requested, the call raises.
"""
import ctypes as C
import os

from . import _build

MAX_GRIDS = 40
MAX_SYN_LAYERS = 8
MAX_REFS = 2

OK = 0
ERR_NAMES = {-1: "TRUNCATED", -2: "VALUE", -3: "INVALID_DATA", -4: "UNSUPPORTED", -5: "NOMEM", -6: "HIP", -7: "ARG"}


class CcdError(RuntimeError):
    def __init__(self, code: int, where: str = ""):
        self.code = code
        msg = lib().ccd_strerror(code).decode()
        super().__init__(f"{where}: {msg} (CCD_ERR_{ERR_NAMES.get(code, code)})" if where else msg)


class VideoHeader(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_intras", C.c_int32), ("n_p_frames", C.c_int32),
                ("n_bytes_header", C.c_int32), ("intra_pos", C.c_int32 * 4096), ("p_pos", C.c_int32 * 4096)]


class FrameHeader(C.Structure):
    _fields_ = [("display_index", C.c_int32), ("frame_type", C.c_int32), ("frame_data_type", C.c_int32),
                ("bitdepth", C.c_int32), ("n_bytes_header", C.c_int32), ("n_refs", C.c_int32),
                ("index_references", C.c_int32 * MAX_REFS), ("global_flow", C.c_int32 * (2 * MAX_REFS)),
                ("warp_filter_size", C.c_int32)]


class SynLayer(C.Structure):
    _fields_ = [("out_ft", C.c_int32), ("k_size", C.c_int32), ("mode", C.c_int32), ("non_linearity", C.c_int32)]


class CCHeader(C.Structure):
    _fields_ = [
        ("linear_stabiliser_synth", C.c_int32), ("n_layer_synthesis", C.c_int32), ("ups_k_size", C.c_int32),
        ("ups_preconcat_k_size", C.c_int32), ("output_feature_ifce", C.c_int32), ("spatial_context_arm", C.c_int32),
        ("linear_stabiliser_arm", C.c_int32), ("n_hidden_layers_arm", C.c_int32), ("img_size", C.c_int32 * 2),
        ("latent_resolution", C.c_int32 * 2), ("n_latent_grids", C.c_int32), ("flag_hyperlatent", C.c_int32),
        ("flag_common_randomness", C.c_int32), ("final_upsampling_type", C.c_int32),
        ("nn_q_step_log2", C.c_int32 * 8), ("nn_expgol_cnt", C.c_int32 * 8), ("nn_n_bytes", C.c_int32),
        ("nn_n_bit_pad", C.c_int32), ("n_bytes_latent", C.c_int32), ("n_bytes_header", C.c_int32),
        ("has_ifce_resolution", C.c_int32), ("ifce_resolution", C.c_int32 * 2),
        ("hyperlatent_resolution", C.c_int32 * 2), ("syn_layer", SynLayer * MAX_SYN_LAYERS),
        ("n_grids", C.c_int32), ("grid_h", C.c_int32 * MAX_GRIDS), ("grid_w", C.c_int32 * MAX_GRIDS),
        ("is_hyperlatent", C.c_int32 * MAX_GRIDS), ("input_features_ifce", C.c_int32 * MAX_GRIDS),
        ("input_feature_synthesis", C.c_int32), ("total_context_arm", C.c_int32), ("out_channels", C.c_int32),
        ("n_symbols", C.c_int64),
    ]


class Frame(C.Structure):
    _fields_ = [("display_index", C.c_int32), ("frame_type", C.c_int32), ("frame_data_type", C.c_int32),
                ("bitdepth", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("ch", C.c_int32), ("cw", C.c_int32),
                ("plane", C.POINTER(C.c_uint16) * 3)]


class PngItem(C.Structure):
    _fields_ = [("r", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32),
                ("out", C.c_void_p), ("cap", C.c_size_t)]


class Video(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("frames", C.POINTER(Frame))]


# name -> (restype, argtypes): every symbol include/ccd.h declares
_u8p = C.POINTER(C.c_uint8)
SIGNATURES = {
    "ccd_strerror": (C.c_char_p, [C.c_int]),
    "ccd_version": (C.c_char_p, []),
    "ccd_read_video_header": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(VideoHeader)]),
    "ccd_read_frame_header": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(FrameHeader)]),
    "ccd_read_cc_header": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(CCHeader)]),
    "ccd_get_coding_structure": (C.c_int, [C.POINTER(VideoHeader), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ccd_decode_coolchic": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int]),
    "ccd_batch_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ccd_batch_destroy": (None, [C.c_void_p]),
    "ccd_batch_add": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                C.c_int, C.c_int]),
    "ccd_batch_size": (C.c_int, [C.c_void_p]),
    "ccd_batch_header": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(CCHeader)]),
    "ccd_batch_prepare": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ccd_batch_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ccd_batch_entropy_launches": (C.c_int, [C.c_void_p]),
    "ccd_batch_launch_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ccd_concurrent_streams": (C.c_int, [C.c_int]),
    "ccd_debug_chain_groups": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccd_batch_run_stage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ccd_batch_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ccd_batch_slot_status": (C.c_int, [C.c_void_p, C.c_int]),
    "ccd_batch_slot_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ccd_batch_slot_kernels": (C.c_int, [C.c_void_p, C.c_int]),
    "ccd_batch_planes_layout": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ccd_batch_copy_planes_async": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "ccd_pool_trim": (None, [C.c_int]),
    "ccd_network_fits_fast_path": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "ccd_network_kernel_class": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "ccd_debug_fd_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "ccd_batch_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ccd_batch_output": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ccd_batch_dense": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ccd_batch_latent": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "ccd_batch_plane": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ccd_batch_copy_latent": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccd_batch_copy_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ccd_batch_copy_output": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ccd_batch_copy_dense": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ccd_decode_video": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Video)]),
    "ccd_inter_reconstruct": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "ccd_video_free": (None, [C.POINTER(Video)]),
    "ccd_range_encode": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_u8p)]),
    "ccd_encode_stream": (C.c_int64, [C.POINTER(CCHeader), C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_int,
                                      C.c_int, C.POINTER(_u8p)]),
    "ccd_encode_coolchic": (C.c_int64, [C.POINTER(CCHeader), C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                                        C.POINTER(_u8p)]),
    "ccd_network_layout": (C.c_int, [C.POINTER(CCHeader), C.POINTER(C.c_int64)]),
    "ccd_encode_network": (C.c_int64, [C.POINTER(CCHeader), C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(_u8p)]),
    "ccd_write_cc_header": (C.c_int, [C.POINTER(CCHeader), C.c_void_p, C.c_size_t]),
    "ccd_write_frame_header": (C.c_int, [C.POINTER(FrameHeader), C.c_void_p, C.c_size_t]),
    "ccd_write_video_header": (C.c_int, [C.POINTER(VideoHeader), C.c_void_p, C.c_size_t]),
    "ccd_free": (None, [C.c_void_p]),
    "ccd_compute_rate": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ccd_png_bound": (C.c_size_t, [C.c_int, C.c_int]),
    "ccd_png_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ccd_png_destroy": (None, [C.c_void_p]),
    "ccd_png_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                               C.c_void_p]),
    "ccd_png_finish": (C.c_int64, [C.c_void_p, C.c_void_p]),
    "ccd_png_pack_batch": (C.c_int, [C.c_void_p, C.POINTER(PngItem), C.c_int, C.c_void_p]),
    "ccd_png_finish_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "ccd_debug_laplace_sweep": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ccd_debug_laplace_bounds": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p]),
}

_lib = None


def lib():
    """Loads (building first if needed) libccd.so. Raises if the HIP extension cannot be produced."""
    global _lib
    if _lib is None:
        # CCD_LIB: an alternative build of the same library (profiling variants made by _build.build_variant)
        path = os.environ.get("CCD_LIB") or _build.build_lib()
        # PyTorch-ROCm bundles its own libamdhip64.so.7; load it FIRST so that libccd.so (same SONAME)
        # binds to that copy. Two HIP runtimes in one process cannot both own the device.
        import torch  # noqa: F401

        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int, where: str = "") -> int:
    if code < 0:
        raise CcdError(code, where)
    return code
