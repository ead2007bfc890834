"""PNG files packed on the MI355X (C ABI: ccd_png_* in include/ccd.h, kernels in csrc/ccd_png.hip).

Reference: coolchic/io/format/png.py:44-62 write_png - same input (an 8-bit RGB picture), same result for every PNG
reader; the bytes differ from PIL's because the deflate stream is built by the device packer (literal-only
dynamic-Huffman blocks over adaptively filtered scanlines), not by zlib.  Only the finished file crosses PCIe."""
import ctypes as C

import torch

from .._lib import PngItem, check, lib


class PngPacker:
    """One packer = one workspace in HBM; packs are enqueued on the caller's stream."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().ccd_png_create(int(device), C.byref(self._h)), "ccd_png_create")
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().ccd_png_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def bound(h: int, w: int) -> int:
        n = lib().ccd_png_bound(int(h), int(w))
        if n == 0:
            raise ValueError(f"PNG sides must lie in 1..16383, got {h}x{w}")
        return n

    def pack_async(self, r: int, g: int, b: int, h: int, w: int, out: torch.Tensor, stream: int = 0) -> None:
        """r, g, b: device addresses of [h][w] uint8 planes; out: uint8 CUDA tensor of >= bound(h, w) bytes."""
        check(lib().ccd_png_pack(self._h, C.c_void_p(r), C.c_void_p(g), C.c_void_p(b), int(h), int(w),
                                 C.c_void_p(out.data_ptr()), out.numel(), C.c_void_p(stream or None)), "ccd_png_pack")

    def finish(self, stream: int = 0) -> int:
        return check(lib().ccd_png_finish(self._h, C.c_void_p(stream or None)), "ccd_png_finish")

    def pack_batch_async(self, items, stream: int = 0) -> None:
        """items: sequence of (r, g, b, h, w, out) - device addresses of the planes and a uint8 CUDA tensor per picture.
        Every deflate block of every picture becomes a workgroup of the same five launches."""
        arr = (PngItem * len(items))()
        for a, (r, g, b, h, w, out) in zip(arr, items):
            a.r, a.g, a.b, a.h, a.w, a.out, a.cap = r, g, b, int(h), int(w), out.data_ptr(), out.numel()
        self._n = len(items)
        check(lib().ccd_png_pack_batch(self._h, arr, len(items), C.c_void_p(stream or None)), "ccd_png_pack_batch")

    def finish_batch(self, stream: int = 0):
        sizes = (C.c_int64 * self._n)()
        check(lib().ccd_png_finish_batch(self._h, C.c_void_p(stream or None), sizes, self._n), "ccd_png_finish_batch")
        return list(sizes)

    def pack_many(self, pictures) -> list:
        """pictures: list of [3, H, W] uint8 CUDA tensors -> list of PNG byte strings (one set of launches)."""
        pictures = [p.contiguous() for p in pictures]
        for p in pictures:
            if p.dtype != torch.uint8 or p.dim() != 3 or p.shape[0] != 3 or not p.is_cuda:
                raise ValueError("PNG output needs [3, H, W] uint8 tensors on the GPU")
        outs = [torch.empty(self.bound(p.shape[1], p.shape[2]) + 4, dtype=torch.uint8, device=p.device) for p in pictures]
        stream = torch.cuda.current_stream(pictures[0].device).cuda_stream
        items = []
        for p, o in zip(pictures, outs):
            _, h, w = p.shape
            items.append((p.data_ptr(), p.data_ptr() + h * w, p.data_ptr() + 2 * h * w, h, w, o))
        self.pack_batch_async(items, stream)
        sizes = self.finish_batch(stream)
        return [o[:n].cpu().numpy().tobytes() for o, n in zip(outs, sizes)]

    def pack(self, planes: torch.Tensor) -> bytes:
        """planes: [3, H, W] uint8 CUDA tensor (r, g, b) -> the bytes of a .png."""
        if planes.dtype != torch.uint8 or planes.dim() != 3 or planes.shape[0] != 3 or not planes.is_cuda:
            raise ValueError("PNG output needs a [3, H, W] uint8 tensor on the GPU")
        planes = planes.contiguous()
        _, h, w = planes.shape
        out = torch.empty(self.bound(h, w) + 4, dtype=torch.uint8, device=planes.device)
        stream = torch.cuda.current_stream(planes.device).cuda_stream
        base, step = planes.data_ptr(), h * w
        self.pack_async(base, base + step, base + 2 * step, h, w, out, stream)
        n = self.finish(stream)
        return out[:n].cpu().numpy().tobytes()


_packers = {}


def device_png_bytes(planes: torch.Tensor) -> bytes:
    """[3, H, W] uint8 CUDA tensor -> PNG bytes, with one cached packer per device."""
    dev = planes.device.index or 0
    if dev not in _packers:
        _packers[dev] = PngPacker(dev)
    return _packers[dev].pack(planes)
