"""Random pictures through the device PNG packer, each compared byte for byte with the CPU restatement and read back with PIL."""
import io
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from cool_chic_amd.io.png import PngPacker  # noqa: E402
from oracle import png_pack  # noqa: E402


def main(n_iter=150, seed=0):
    from PIL import Image

    rng = np.random.default_rng(seed)
    p = PngPacker(0)
    n_bytes = 0
    for it in range(n_iter):
        h, w = int(rng.integers(1, 260)), int(rng.integers(1, 400))
        kind = it % 4
        if kind == 0:
            pl = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
        elif kind == 1:
            pl = np.minimum(rng.geometric(rng.uniform(0.05, 0.9), (3, h, w)), 255).astype(np.uint8)
        elif kind == 2:
            yy, xx = np.mgrid[0:h, 0:w]
            pl = np.stack([(yy * rng.integers(1, 5) + xx) % 256, (yy + xx * rng.integers(1, 5)) % 256, (yy * xx) % 256]).astype(np.uint8)
        else:
            pl = np.full((3, h, w), rng.integers(0, 256), np.uint8)
            pl[:, rng.integers(0, h), rng.integers(0, w)] ^= 0xFF
        if it % 10 == 9:  # a batch of the last few pictures as well
            pics = [pl] + [rng.integers(0, 256, (3, int(rng.integers(1, 40)), int(rng.integers(1, 40))), dtype=np.uint8) for _ in range(5)]
            got = p.pack_many([torch.from_numpy(q).cuda() for q in pics])
            for q, g in zip(pics, got):
                assert g == png_pack.pack_rgb8(q), ("batch", it, q.shape)
            continue
        png = p.pack(torch.from_numpy(pl).cuda())
        assert png == png_pack.pack_rgb8(pl), (it, h, w, kind)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(png))).transpose(2, 0, 1), pl)
        n_bytes += len(png)
    print(f"{n_iter} random pictures: device PNG == oracle PNG, PIL reads them back ({n_bytes} bytes)")


if __name__ == "__main__":
    main()
