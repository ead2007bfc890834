"""Times the device PNG packer against PIL (zlib level 6, the reference's writer) on the same pictures."""
import io
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cool_chic_amd.io.png import PngPacker  # noqa: E402


def photo(h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(yy * 3 + xx) % 256, (yy + xx * 2) % 256, (yy * xx) % 256]).astype(np.int32)
    return np.clip(base // 2 + rng.normal(0, 6, (3, h, w)).round().astype(np.int32) + 40, 0, 255).astype(np.uint8)


def main():
    from PIL import Image

    p = PngPacker(0)
    for h, w in ((512, 768), (1080, 1920), (2160, 3840)):
        planes = photo(h, w)
        d = torch.from_numpy(planes).cuda()
        out = torch.empty(p.bound(h, w) + 4, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        step = h * w
        for _ in range(3):
            p.pack_async(d.data_ptr(), d.data_ptr() + step, d.data_ptr() + 2 * step, h, w, out, st)
            n = p.finish(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            p.pack_async(d.data_ptr(), d.data_ptr() + step, d.data_ptr() + 2 * step, h, w, out, st)
        e1.record()
        n = p.finish(st)
        gpu_ms = e0.elapsed_time(e1) / reps
        t = time.time()
        host = out[:n].cpu()
        d2h_ms = (time.time() - t) * 1e3
        b = io.BytesIO()
        t = time.time()
        Image.fromarray(np.ascontiguousarray(planes.transpose(1, 2, 0)), mode="RGB").save(b, format="PNG")
        pil_ms = (time.time() - t) * 1e3
        raw = 3 * h * w
        print(f"{h}x{w}: device pack {gpu_ms:.3f} ms ({raw / gpu_ms / 1e6:.1f} GB/s of pixels), file {n} B "
              f"(PIL {len(b.getvalue())} B), D2H of the file {d2h_ms:.2f} ms, PIL save on the host {pil_ms:.1f} ms")
        assert bytes(host.numpy().tobytes()[:8]) == b"\x89PNG\r\n\x1a\n"
    # 24 Kodak-sized pictures in one set of launches
    pics = [torch.from_numpy(photo(512, 768, seed=i)).cuda() for i in range(24)]
    outs = [torch.empty(p.bound(512, 768) + 4, dtype=torch.uint8, device="cuda") for _ in pics]
    items = [(q.data_ptr(), q.data_ptr() + 512 * 768, q.data_ptr() + 2 * 512 * 768, 512, 768, o) for q, o in zip(pics, outs)]
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        p.pack_batch_async(items, st)
        p.finish_batch(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        p.pack_batch_async(items, st)
    e1.record()
    sizes = p.finish_batch(st)
    ms = e0.elapsed_time(e1) / 20
    print(f"24 x 512x768 in one batch: {ms:.3f} ms ({24 * 3 * 512 * 768 / ms / 1e6:.1f} GB/s of pixels), {sum(sizes)} B")


if __name__ == "__main__":
    main()
