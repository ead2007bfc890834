"""Video / frame / cool-chic headers: the reference's AbstractHeader API (read_header, get_value)
over the C parser in libccd.so (reference: coolchic/bitstream/header/header.py:72-88, 130-377)."""
import ctypes as C
from typing import Any, Optional

from .. import _lib
from .._lib import check, lib

FRAME_TYPES = ["I", "P", "B"]
FRAME_DATA_TYPES = ["rgb", "yuv420", "yuv444", "flow"]
FINAL_UPSAMPLING = ["nearest", "bilinear", "bicubic"]
_NN_SLOTS = ["arm.weight", "arm.bias", "ifce.weight", "ifce.bias", "upsampling.weight", "upsampling.bias",
             "synthesis.weight", "synthesis.bias"]


class _Header:
    _ctype = None
    _reader = ""

    def __init__(self):
        self.c = self._ctype()
        self.raw = b""

    def read_header(self, raw_data: bytes) -> bytes:
        """Parses the header at the start of raw_data and returns the remaining bytes."""
        used = check(getattr(lib(), self._reader)(raw_data, len(raw_data), C.byref(self.c)), self._reader)
        self.raw = bytes(raw_data[:used])
        return raw_data[used:]

    def get_value(self, key: str) -> Optional[Any]:
        return self._values().get(key)

    def pretty_string(self) -> str:
        return "".join(f"{k:<30}{str(v):<40}\n" for k, v in self._values().items())


class VideoHeader(_Header):
    _ctype = _lib.VideoHeader
    _reader = "ccd_read_video_header"

    def _values(self):
        c = self.c
        return {"n_frames": c.n_frames, "n_intras": c.n_intras, "n_p_frames": c.n_p_frames,
                "n_bytes_header": c.n_bytes_header, "intra_pos": list(c.intra_pos[: c.n_intras]),
                "p_pos": list(c.p_pos[: c.n_p_frames])}


    def get_coding_structure(self):
        """header.py:161 -> CodingStructure(n_frames, intra_pos, p_pos) (utils/codingstructure.py:226-436) through the C ABI:
        the frames in CODING order as dicts {display_order, frame_type, index_references, depth}.  Raises ValueError where
        the reference asserts (first frame not intra, last frame neither I nor P, a frame both I and P)."""
        import numpy as np

        n = self.c.n_frames
        disp, typ, dep = (np.zeros(max(n, 1), dtype=np.int32) for _ in range(3))
        refs = np.zeros((max(n, 1), 2), dtype=np.int32)
        got = lib().ccd_get_coding_structure(C.byref(self.c), disp.ctypes.data, typ.ctypes.data, refs.ctypes.data, dep.ctypes.data)
        if got < 0:
            raise ValueError("the video header does not describe a coding structure the reference accepts "
                             f"(n_frames={n}, intra_pos={self.get_value('intra_pos')}, p_pos={self.get_value('p_pos')})")
        return [{"display_order": int(disp[k]), "frame_type": FRAME_TYPES[typ[k]], "index_references": [int(r) for r in refs[k] if r >= 0],
                 "depth": int(dep[k])} for k in range(got)]


class FrameHeader(_Header):
    _ctype = _lib.FrameHeader
    _reader = "ccd_read_frame_header"

    def _values(self):
        c = self.c
        v = {"display_index": c.display_index, "frame_type": FRAME_TYPES[c.frame_type],
             "frame_data_type": FRAME_DATA_TYPES[c.frame_data_type], "bitdepth": c.bitdepth,
             "n_bytes_header": c.n_bytes_header, "index_references": list(c.index_references[: c.n_refs]),
             "global_flow": list(c.global_flow[: 2 * c.n_refs])}
        if c.n_refs:
            v["warp_filter_size"] = c.warp_filter_size
        return v


class CoolChicHeader(_Header):
    _ctype = _lib.CCHeader
    _reader = "ccd_read_cc_header"

    def _values(self):
        c = self.c
        v = {k: getattr(c, k) for k in ("linear_stabiliser_synth", "n_layer_synthesis", "ups_k_size",
                                        "ups_preconcat_k_size", "output_feature_ifce", "spatial_context_arm",
                                        "linear_stabiliser_arm", "n_hidden_layers_arm", "n_latent_grids",
                                        "flag_hyperlatent", "flag_common_randomness", "nn_n_bytes", "nn_n_bit_pad",
                                        "n_bytes_latent", "n_bytes_header")}
        v["img_size"] = list(c.img_size)
        v["latent_resolution"] = list(c.latent_resolution)
        v["final_upsampling_type"] = FINAL_UPSAMPLING[c.final_upsampling_type]
        v["nn_q_step"] = {k: 2.0 ** c.nn_q_step_log2[i] for i, k in enumerate(_NN_SLOTS)}
        v["nn_expgol_cnt"] = {k: c.nn_expgol_cnt[i] for i, k in enumerate(_NN_SLOTS)}
        v["ifce_resolution"] = list(c.ifce_resolution) if c.has_ifce_resolution else None
        v["hyperlatent_resolution"] = list(c.hyperlatent_resolution) if c.flag_hyperlatent else None
        for i in range(c.n_layer_synthesis):
            s = c.syn_layer[i]
            v[f"syn_layer_{i}"] = f"{s.out_ft}-{s.k_size}-{['linear', 'residual'][s.mode]}-{['none', 'relu'][s.non_linearity]}"
        return v

    # derived geometry (reference: CoolChicEncoderParameter, component/core/coolchic.py:149-225)
    def size_per_latent(self):
        return [(self.c.grid_h[g], self.c.grid_w[g]) for g in range(self.c.n_grids)]
