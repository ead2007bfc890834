"""PNG packing (reference: coolchic/io/format/png.py:44-62).

PNG bytes are not normative, so the bar has two parts: (1) the picture an independent reader (PIL / zlib - the
reference's own writer library) decodes is pixel-exact, at every size; (2) the device's bytes equal those of the CPU
restatement oracle/png_pack.py (filter choice, code lengths, block layout and checksums are integer and deterministic)."""
import io
import zlib

import numpy as np
import pytest


def _pictures(h, w, seed=0):
    rng = np.random.default_rng(seed + 7919 * h + w)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(yy * 3 + xx) % 256, (yy + xx * 2) % 256, (yy * xx) % 256]).astype(np.uint8)
    photo = np.clip(smooth.astype(np.int32) // 2 + rng.normal(0, 6, (3, h, w)).round().astype(np.int32) + 40, 0, 255).astype(np.uint8)
    return {
        "random": rng.integers(0, 256, (3, h, w), dtype=np.uint8),
        "smooth": smooth,
        "photo": photo,
        "zeros": np.zeros((3, h, w), np.uint8),
        # counts falling off geometrically: optimal codes longer than 15 bits, the length limiter runs
        "skewed": np.minimum(rng.geometric(0.55, (3, h, w)) * 3, 255).astype(np.uint8),
    }


def _read_png(png: bytes) -> np.ndarray:
    from PIL import Image

    im = Image.open(io.BytesIO(png))
    assert im.mode == "RGB"
    return np.asarray(im).transpose(2, 0, 1)


SIZES = [(1, 1), (1, 7), (7, 1), (2, 3), (17, 33), (64, 64), (100, 300), (33, 1111)]


# ------------------------------------------------------------------------------------------------ CPU: the restatement itself
def test_oracle_png_is_read_back_exactly():
    from oracle import png_pack

    for h, w in SIZES:
        for kind, planes in _pictures(h, w).items():
            png = png_pack.pack_rgb8(planes)
            assert len(png) <= png_pack.bound(h, w)
            assert np.array_equal(_read_png(png), planes), (h, w, kind)


def test_oracle_checksums_and_code_lengths():
    from oracle import png_pack

    rng = np.random.default_rng(1)
    for n in (0, 1, 511, 512, 513, 4096, 70001):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert png_pack.crc32_chunked(d) == zlib.crc32(d)
    scan = rng.integers(0, 256, (37, 301), dtype=np.uint8)
    assert png_pack.adler32_rows(scan) == zlib.adler32(scan.tobytes())
    # Fibonacci counts: the optimal code is 29 bits deep; the limited one is complete and <= 15 bits
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    hist = np.zeros(257, np.int64)
    hist[:30] = fib
    hist[256] = 1
    lens = png_pack.code_lengths(hist)
    assert lens.max() == 15 and sum(2.0 ** -int(v) for v in lens if v) == 1.0
    assert all(lens[i] >= lens[i + 1] for i in range(29))
    # two symbols only (a constant block + end-of-block)
    hist = np.zeros(257, np.int64)
    hist[0], hist[256] = 1000, 1
    assert list(png_pack.code_lengths(hist)[[0, 256]]) == [1, 1]


def test_bound_matches_the_library():
    """Host-side arithmetic of the C ABI (no device work)."""
    from cool_chic_amd import _lib
    from oracle import png_pack

    L = _lib.lib()
    for h, w in SIZES + [(512, 768), (2160, 3840), (16383, 16383)]:
        assert L.ccd_png_bound(h, w) == png_pack.bound(h, w)
    assert L.ccd_png_bound(0, 5) == 0 and L.ccd_png_bound(5, 16384) == 0


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def packer():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cool_chic_amd.io.png import PngPacker

    p = PngPacker(0)
    yield p
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_device_png_equals_oracle_and_reads_back(packer, h, w):
    import torch

    from oracle import png_pack

    for kind, planes in _pictures(h, w).items():
        png = packer.pack(torch.from_numpy(planes).cuda())
        assert np.array_equal(_read_png(png), planes), (h, w, kind)
        assert png == png_pack.pack_rgb8(planes), (h, w, kind)


@pytest.mark.gpu
def test_device_png_of_the_kodak_fixture(packer):
    import torch

    from conftest import load_golden, reference_planes
    from oracle import png_pack

    _, z, j = load_golden("kodim14")
    planes = np.stack(reference_planes(z, j)).astype(np.uint8)
    png = packer.pack(torch.from_numpy(planes).cuda())
    assert np.array_equal(_read_png(png), planes)
    assert png == png_pack.pack_rgb8(planes)
    # same workspace, smaller then larger picture (workspace growth), then the first one again
    for h, w in ((40, 50), (700, 900), (512, 768)):
        pl = planes[:, :h, :w] if (h <= 512 and w <= 768) else _pictures(h, w)["photo"]
        pl = np.ascontiguousarray(pl)
        assert np.array_equal(_read_png(packer.pack(torch.from_numpy(pl).cuda())), pl)
    assert packer.pack(torch.from_numpy(planes).cuda()) == png


@pytest.mark.gpu
def test_device_png_batch_of_mixed_sizes(packer):
    """One set of launches for many pictures: same bytes as one by one (= the oracle's)."""
    import torch

    from oracle import png_pack

    pics = [_pictures(h, w)[kind] for (h, w), kind in zip(SIZES + [(300, 200), (64, 512)],
                                                         ["random", "smooth", "photo", "zeros", "skewed"] * 2)]
    got = packer.pack_many([torch.from_numpy(p).cuda() for p in pics])
    for p, png in zip(pics, got):
        assert np.array_equal(_read_png(png), p)
        assert png == png_pack.pack_rgb8(p)
    # more pictures than one launch set takes (64)
    many = [_pictures(9 + i % 5, 11 + i % 7, seed=i)["photo"] for i in range(70)]
    for p, png in zip(many, packer.pack_many([torch.from_numpy(p).cuda() for p in many])):
        assert np.array_equal(_read_png(png), p)


@pytest.mark.gpu
def test_device_png_full_size_round_trips(packer):
    """BASELINE sizes (4K, 2K portrait) through the size-independent property: an independent reader gets the pixels."""
    import torch

    for h, w in ((2160, 3840), (2048, 1365), (2, 16383), (16383, 2), (3000, 1)):  # + the extreme aspect ratios of 14-bit sizes
        planes = _pictures(h, w)["photo"]
        png = packer.pack(torch.from_numpy(planes).cuda())
        assert len(png) <= packer.bound(h, w)
        assert np.array_equal(_read_png(png), planes)


@pytest.mark.gpu
def test_png_argument_errors(packer):
    import torch

    from cool_chic_amd import CcdError

    with pytest.raises(ValueError):
        packer.pack(torch.zeros(3, 4, 4, dtype=torch.float32, device="cuda"))
    small = torch.empty(16, dtype=torch.uint8, device="cuda")
    pl = torch.zeros(3, 8, 8, dtype=torch.uint8, device="cuda")
    with pytest.raises(CcdError):
        packer.pack_async(pl.data_ptr(), pl.data_ptr() + 64, pl.data_ptr() + 128, 8, 8, small)
    with pytest.raises(CcdError):
        packer.finish()  # nothing in flight
