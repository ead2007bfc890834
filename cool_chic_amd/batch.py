"""DecodeBatch: many cool-chics in flight on one MI355X (wraps the ccd_batch_* C ABI)."""
import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import CCHeader, check, lib

FRAME_DATA_TYPES = ["rgb", "yuv420", "yuv444", "flow"]


class _DevArray:
    """Zero-copy view of library-owned device memory for torch.as_tensor(..., device='cuda')."""

    def __init__(self, ptr: int, shape: Tuple[int, ...], typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner  # keeps the batch (and its arena) alive


class DecodeBatch:
    """One slot per cool-chic; all slots decode concurrently in run().

    Inputs (latent payload words, network parameters) are uploaded to HBM by add(); run() only
    enqueues kernels on `stream` (a hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream).
    """

    OPT_FUSED_DEC, OPT_KEEP_FLOAT, OPT_MFMA_ARM, OPT_RANGE_BITS, OPT_OVERLAP, OPT_TIME_LAUNCHES = 1, 2, 3, 4, 5, 6  # include/ccd.h

    def __init__(self, device: int = 0, fused_dec: Optional[bool] = None, keep_float: Optional[bool] = None,
                 mfma_arm: Optional[int] = None, range_bits: Optional[int] = None, overlap: Optional[bool] = None):
        """fused_dec=False: unfused float path (materialises dense()); keep_float=False: rgb / yuv444 intra slots
        write integer planes only (output() is then unavailable for them); mfma_arm=1: the integer ARM on the matrix
        cores where the stream allows (2..22: test hook, see ccd.h); range_bits=8..14: test hook that lowers the feature
        limit of the pipelined entropy kernel's dynamic operand check.
        overlap=False: run() puts every float-path launch behind the join of the entropy launches (A/B, tests); by default the
        frames whose streams finish early are synthesised while the longest chains still decode (ccd.h, ccd_batch_run).
        None = library default: fused_dec on, keep_float on, mfma_arm OFF (the vector-ALU ARM is the faster one),
        production limit (15 bits), overlap on."""
        self._h = C.c_void_p()
        check(lib().ccd_batch_create(int(device), C.byref(self._h)), "ccd_batch_create")
        self.device = int(device)
        if fused_dec is not None:
            # False / True, or 2: the fused kernel behind the batch's pyramid steps (level-1 stack pre-computed once per frame)
            # True = "the fused float path" = the library's default form of it (2); 0 / 1 / 2 select a form explicitly (ccd.h)
            check(lib().ccd_batch_set_option(self._h, self.OPT_FUSED_DEC, 2 if fused_dec is True else int(fused_dec)), "ccd_batch_set_option")
        if keep_float is not None:
            check(lib().ccd_batch_set_option(self._h, self.OPT_KEEP_FLOAT, int(bool(keep_float))), "ccd_batch_set_option")
        if mfma_arm is not None:
            check(lib().ccd_batch_set_option(self._h, self.OPT_MFMA_ARM, int(mfma_arm)), "ccd_batch_set_option")
        if range_bits is not None:
            check(lib().ccd_batch_set_option(self._h, self.OPT_RANGE_BITS, int(range_bits)), "ccd_batch_set_option")
        if overlap is not None:
            check(lib().ccd_batch_set_option(self._h, self.OPT_OVERLAP, int(bool(overlap))), "ccd_batch_set_option")
        self._meta: List[Tuple[int, int]] = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().ccd_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self._meta)

    def add(self, cc_header: bytes, bytes_nn: bytes, bytes_latent: bytes, bitdepth: int = 0,
            frame_data_type: int = 0) -> int:
        slot = check(lib().ccd_batch_add(self._h, cc_header, len(cc_header), bytes_nn, len(bytes_nn), bytes_latent,
                                         len(bytes_latent), int(bitdepth), int(frame_data_type)), "ccd_batch_add")
        self._meta.append((int(bitdepth), int(frame_data_type)))
        return slot

    def header(self, slot: int) -> CCHeader:
        h = CCHeader()
        check(lib().ccd_batch_header(self._h, slot, C.byref(h)), "ccd_batch_header")
        return h

    def prepare(self, stream: int = 0):
        """Uploads the launch tables of the slots added so far without launching (ccd_batch_prepare): for callers that then
        make `stream` wait for other work before run()."""
        check(lib().ccd_batch_prepare(self._h, C.c_void_p(stream or None)), "ccd_batch_prepare")

    def run(self, stream: int = 0, stage: Optional[int] = None):
        st = C.c_void_p(stream or None)
        if stage is None:
            check(lib().ccd_batch_run(self._h, st), "ccd_batch_run")
        else:
            check(lib().ccd_batch_run_stage(self._h, st, int(stage)), "ccd_batch_run_stage")

    def entropy_launches(self) -> int:
        """Entropy launches per run (kernel instantiations x chain groups) as the launch tables were last built."""
        return check(lib().ccd_batch_entropy_launches(self._h), "ccd_batch_entropy_launches")

    def time_launches(self, on: bool = True):
        """Timing events around every entropy launch (on the stream it runs on); launch_ms() reads the last run's."""
        check(lib().ccd_batch_set_option(self._h, self.OPT_TIME_LAUNCHES, int(bool(on))), "ccd_batch_set_option")

    def launch_ms(self):
        """[(ms, streams)] of every entropy launch of the last run, longest expected chains first (needs time_launches())."""
        ms = (C.c_float * 32)()
        ns = (C.c_int * 32)()
        n = check(lib().ccd_batch_launch_ms(self._h, ms, ns, 32), "ccd_batch_launch_ms")
        return [(float(ms[i]), int(ns[i])) for i in range(n)]

    def wait(self, stream: int = 0):
        check(lib().ccd_batch_wait(self._h, C.c_void_p(stream or None)), "ccd_batch_wait")

    def slot_status(self, slot: int) -> int:
        return lib().ccd_batch_slot_status(self._h, slot)

    def slot_stats(self, slot: int) -> np.ndarray:
        """Raw counters of the entropy kernel after wait() (ccd.h): [0] status, [1] words read, [2..3] symbols, [39] pixels
        the pipelined kernel redid in int64."""
        out = np.zeros(64, dtype=np.int32)
        check(lib().ccd_batch_slot_stats(self._h, slot, out.ctypes.data), "ccd_batch_slot_stats")
        return out

    # ---- results -------------------------------------------------------------------------------
    def slot_kernels(self, slot: int) -> int:
        """bit 0: pipelined entropy kernel, bit 1: fused synthesis kernel, bit 2: fused upsampling + synthesis kernel,
        bit 3: the ARM on the matrix cores, bit 4: the pipelined kernel's instantiation with the device check of IFCE features,
        bit 5: its instantiation with a compile-time ARM shape (HOP), bit 6: the fused float kernel behind the pyramid launch,
        bit 7: the network is outside the finite envelope of the float stages (vector-ALU float kernels only, see ccd.h)."""
        return check(lib().ccd_batch_slot_kernels(self._h, slot), "ccd_batch_slot_kernels")

    def latent(self, slot: int, grid: int) -> np.ndarray:
        h = self.header(slot)
        out = np.empty((h.grid_h[grid], h.grid_w[grid]), dtype=np.int8)
        check(lib().ccd_batch_copy_latent(self._h, slot, grid, out.ctypes.data, None), "copy_latent")
        return out

    def output(self, slot: int) -> np.ndarray:
        h = self.header(slot)
        out = np.empty((h.out_channels, h.img_size[0], h.img_size[1]), dtype=np.float32)
        check(lib().ccd_batch_copy_output(self._h, slot, out.ctypes.data, None), "copy_output")
        return out

    def dense(self, slot: int) -> np.ndarray:
        h = self.header(slot)
        g0 = next(g for g in range(h.n_grids) if not h.is_hyperlatent[g])
        out = np.empty((h.input_feature_synthesis, h.grid_h[g0], h.grid_w[g0]), dtype=np.float32)
        check(lib().ccd_batch_copy_dense(self._h, slot, out.ctypes.data, None), "copy_dense")
        return out

    def plane_shape(self, slot: int, plane: int) -> Tuple[int, int]:
        ph, pw = C.c_int(), C.c_int()
        if not lib().ccd_batch_plane(self._h, slot, plane, C.byref(ph), C.byref(pw)):
            raise ValueError("slot has no integer planes (added with bitdepth=0)")
        return ph.value, pw.value

    def planes(self, slot: int) -> List[np.ndarray]:
        bd, _ = self._meta[slot]
        res = []
        for p in range(3):
            shape = self.plane_shape(slot, p)
            out = np.empty(shape, dtype=np.uint8 if bd == 8 else np.uint16)
            check(lib().ccd_batch_copy_plane(self._h, slot, p, out.ctypes.data, None), "copy_plane")
            res.append(out)
        return res

    def all_planes(self, stream: int = 0) -> List[List[np.ndarray]]:
        """Integer planes of EVERY slot on the host: one device -> host copy per slot (its three planes are one block) into
        one pinned buffer, one wait - the writer's path for an image set.  Returns [slot][plane] numpy views of that buffer."""
        import torch

        n = len(self._meta)
        total, off = C.c_size_t(), (C.c_size_t * 3)()
        layout, base = [], 0
        for s in range(n):
            check(lib().ccd_batch_planes_layout(self._h, s, C.byref(total), off), "ccd_batch_planes_layout")
            layout.append((base, [off[0], off[1], off[2]]))
            base += total.value
        host = torch.empty(base, dtype=torch.uint8, pin_memory=True)  # torch caches pinned blocks: no page-locking per call
        ptrs = (C.c_void_p * n)(*[host.data_ptr() + b0 for b0, _ in layout])
        st = C.c_void_p(stream or None)
        check(lib().ccd_batch_copy_planes_async(self._h, 0, n, ptrs, st), "ccd_batch_copy_planes_async")
        check(lib().ccd_batch_wait(self._h, st), "ccd_batch_wait")
        arr = host.numpy()
        out = []
        for s, (b0, offs) in enumerate(layout):
            bd, _ = self._meta[s]
            dt = np.uint8 if bd == 8 else np.uint16
            planes = []
            for p in range(3):
                h, w = self.plane_shape(s, p)
                planes.append(arr[b0 + offs[p]: b0 + offs[p] + h * w * dt().itemsize].view(dt).reshape(h, w))
            out.append(planes)
        return out

    def planes_layout(self) -> Tuple[int, List[Tuple[int, List[int]]]]:
        """(total bytes, [(block offset, [plane offsets inside the block]) per slot]) of every slot's plane block laid end to
        end: the host buffer copy_planes_async() fills."""
        total, off = C.c_size_t(), (C.c_size_t * 3)()
        layout, base = [], 0
        for s in range(len(self._meta)):
            check(lib().ccd_batch_planes_layout(self._h, s, C.byref(total), off), "ccd_batch_planes_layout")
            layout.append((base, [off[0], off[1], off[2]]))
            base += total.value
        return base, layout

    def copy_planes_async(self, host_ptr: int, layout, stream: int = 0) -> None:
        """One device -> host copy per slot into the (pinned) buffer at `host_ptr`, laid out as planes_layout() says; returns
        without waiting - the caller waits on `stream` (wait())."""
        n = len(self._meta)
        ptrs = (C.c_void_p * n)(*[host_ptr + b0 for b0, _ in layout])
        check(lib().ccd_batch_copy_planes_async(self._h, 0, n, ptrs, C.c_void_p(stream or None)), "ccd_batch_copy_planes_async")

    def plane_views(self, host: np.ndarray, layout) -> List[List[np.ndarray]]:
        """[slot][plane] numpy views of a host buffer filled by copy_planes_async()."""
        out = []
        for s, (b0, offs) in enumerate(layout):
            bd, _ = self._meta[s]
            dt = np.uint8 if bd == 8 else np.uint16
            planes = []
            for p in range(3):
                h, w = self.plane_shape(s, p)
                planes.append(host[b0 + offs[p]: b0 + offs[p] + h * w * dt().itemsize].view(dt).reshape(h, w))
            out.append(planes)
        return out

    def planes_block_device(self, slot: int) -> Tuple[_DevArray, List[Tuple[int, Tuple[int, int]]]]:
        """The slot's three integer planes as the ONE device block they occupy (uint8 bytes) + [(byte offset, (h, w))] per
        plane: a consumer that must outlive the batch copies the block once and views the planes in its copy."""
        total, off = C.c_size_t(), (C.c_size_t * 3)()
        check(lib().ccd_batch_planes_layout(self._h, slot, C.byref(total), off), "ccd_batch_planes_layout")
        ptr = lib().ccd_batch_plane(self._h, slot, 0, None, None)
        return _DevArray(ptr, (total.value,), "|u1", self), [(off[p], self.plane_shape(slot, p)) for p in range(3)]

    def output_device(self, slot: int) -> _DevArray:
        h = self.header(slot)
        ptr = lib().ccd_batch_output(self._h, slot)
        if not ptr:
            raise ValueError("slot has no float output (keep_float=False and integer planes written directly)")
        return _DevArray(ptr, (1, h.out_channels, h.img_size[0], h.img_size[1]), "<f4", self)

    def plane_device(self, slot: int, plane: int) -> _DevArray:
        bd, _ = self._meta[slot]
        ph, pw = C.c_int(), C.c_int()
        ptr = lib().ccd_batch_plane(self._h, slot, plane, C.byref(ph), C.byref(pw))
        return _DevArray(ptr, (ph.value, pw.value), "|u1" if bd == 8 else "<u2", self)
