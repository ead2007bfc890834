"""Synthetic .cool bitstreams shaped like BASELINE.json's configurations (hard part H6).

Only `samples/bitstreams/kodim14.cool` is a real Cool-chic stream and there is no network for datasets, so every
benchmark / full-size test input is MANUFACTURED here with the build's own bitstream writer: trained networks of a
reference-encoded donor stream (grown to more pyramid levels where the picture size asks for it), real latent pyramids
tiled / rolled / transposed to the new geometry (real symbol statistics, 0.6-0.9 bpp), framed exactly like the reference
frames them.  Used by bench.py and tests/; nothing here runs while decoding.

Sizes: Kodak = 18 landscape + 6 portrait 512x768; CLIC20-pro-valid = the 41 pixel counts of
results/v5.0/image-clic20-pro-valid.tsv (n_pixels column, 91.45 Mpx) factorised with an aspect ratio near 3:2 (the TSV
does not carry widths and heights); 1080p GOP = intra period 32, hierarchical B (docs/source/results/video.rst:28,
samples/encode.py:23-70); 4K = 3840x2160.  Latent / hyperlatent ranges follow the reference's "auto" rule
(utils/parsecli.py:82-117)."""
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import writer
from ._lib import CCHeader

_HERE = os.path.dirname(os.path.abspath(__file__))
# donor streams (the reference's kodim14.cool, the reference-encoded 5-frame video), their latent grids and network
# integers: package data written by tools/make_package_data.py
DONORS = os.path.join(_HERE, "data", "donors.npz")
_donors = None

# (H, W) per picture of CLIC20-pro-valid: H * W = the TSV's n_pixels, rows in the TSV's order
CLIC41_SIZES = [(1363, 2048), (1339, 2048), (1188, 2048), (1361, 2048), (1365, 2048), (1020, 1464), (1361, 2048), (1365, 2048),
                (1360, 2048), (1365, 2048), (1000, 1266), (1292, 1945), (1360, 2040), (1166, 1750), (1370, 2048), (1365, 2048),
                (1365, 2048), (1220, 1936), (1145, 1725), (1365, 2048), (1365, 2048), (640, 960), (1365, 2048), (1365, 2048),
                (772, 1158), (1366, 2048), (640, 906), (960, 1350), (1281, 1922), (1365, 2048), (384, 512), (1365, 2048),
                (1028, 1542), (1367, 2048), (790, 1264), (439, 720), (1033, 2048), (1366, 2048), (1232, 1836), (1325, 1988),
                (1365, 2048)]
assert len(CLIC41_SIZES) == 41 and sum(h * w for h, w in CLIC41_SIZES) == 91451931

Triple = Tuple[bytes, bytes, bytes]  # cool-chic header, NN payload, latent payload


def auto_resolution(n_pixels: int) -> int:
    """Coarsest latent level of the reference's "auto" rule (utils/parsecli.py:86-93)."""
    return 6 if n_pixels < 1_000_000 else (7 if n_pixels < 3_000_000 else 8)


def _pool(n: int) -> ThreadPoolExecutor:
    return ThreadPoolExecutor(max_workers=max(1, min(n, 32, os.cpu_count() or 4)))  # the writer's ARM walk releases the GIL


class _Donor:
    """Arrays of one donor stream: d["cc0.latent3"], d["cc0.nn_ints"]."""

    def __init__(self, z, name):
        self._z, self._p = z, name + "."

    def __getitem__(self, key):
        return self._z[self._p + key]


def _golden(name: str):
    global _donors
    if _donors is None:
        _donors = np.load(DONORS)
    return _donors[name + ".cool"].tobytes(), _Donor(_donors, name)


def split_image_stream(bs: bytes) -> Triple:
    """One-frame intra stream -> (cool-chic header, NN bytes, latent bytes)."""
    from .bitstream.header import CoolChicHeader, FrameHeader, VideoHeader

    rest = VideoHeader().read_header(bs)
    rest = FrameHeader().read_header(rest)
    ch = CoolChicHeader()
    rest = ch.read_header(rest)
    n_nn = ch.get_value("nn_n_bytes")
    return ch.raw, rest[:n_nn], rest[n_nn:n_nn + ch.get_value("n_bytes_latent")]


def _image_donor(name: str):
    """(stream, cc header bytes, NN bytes, parsed header, network integers, latent grids) of a one-frame donor stream."""
    bs, z = _golden(name)
    hdr, nn, lat = split_image_stream(bs)
    donor = writer.parse_cc_header(hdr)
    latents = [z[f"cc0.latent{g}"] for g in range(donor.n_grids)]
    return bs, hdr, nn, donor, z["cc0.nn_ints"], latents


def _kodim14():
    return _image_donor("kodim14")


def kodak24() -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """BASELINE configs[1]: 24 RGB 8-bit 512x768 streams (18 landscape, 6 portrait as in Kodak), HOP decoder.
    Stream 0 is the reference's kodim14.cool; the others carry kodim14's network and rolled / transposed copies of
    its latent pyramid.  Returns (streams, sizes)."""
    real, hdr, nn, donor, _, latents = _kodim14()
    _, levels = writer.grid_sizes((512, 768), hdr)
    jobs = [(1000 + i, i in (3, 8, 9, 16, 17, 18)) for i in range(1, 24)]

    def make(job):
        seed, portrait = job
        v = writer.variant_latents(latents, levels, seed, portrait)
        return writer.encode_stream(hdr, nn, v, img_size=(768, 512) if portrait else (512, 768))

    with _pool(len(jobs)) as ex:
        streams = [real] + list(ex.map(make, jobs))
    sizes = [(512, 768)] + [((768, 512) if p else (512, 768)) for _, p in jobs]
    return streams, sizes


def kodak24_wide_envelope() -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """kodak24 with a network OUTSIDE the r02 static envelope of the pipelined entropy kernel: the IFCE rows (weights and
    bias, every grid) that produce kodim14's LAST feature are doubled - that feature doubles, its worst case becomes 2^15.5
    >= 2^15 like the reference-encoded rgb192 / yuv444_10b / vid5-I networks - and the ARM's first-layer and stabiliser
    column that reads it is halved (one feature only: halving small integers is lossy).  Every stream is re-encoded against
    that network.  bench.py times it next to kodak24 (`wide_envelope_network`)."""
    _, hdr, _, donor, ints, latents = _kodim14()
    lay = writer.network_layout(donor)
    g = np.split(np.asarray(ints, dtype=np.int64).copy(), np.cumsum(lay)[:-1])
    dim, n_if = donor.total_context_arm, donor.output_feature_ifce
    pos = 0
    for k, f in enumerate(x for x in donor.input_features_ifce[:donor.n_grids] if x > 0):  # ifce.w per grid: [n_if][f]
        g[2][pos + (n_if - 1) * f: pos + n_if * f] *= 2
        g[3][k * n_if + n_if - 1] *= 2
        pos += n_if * f
    first = g[0][:dim * dim].reshape(dim, dim)            # arm.mlp.0 weight [out][in]: the IFCE columns are the last n_if
    first[:, dim - 1] = np.round(first[:, dim - 1] / 2.0)
    if donor.linear_stabiliser_arm:
        stab = g[0][-2 * dim:].reshape(2, dim)            # stabiliser_branch weight [2][in] is the last weight tensor of the ARM
        stab[:, dim - 1] = np.round(stab[:, dim - 1] / 2.0)
    nn = writer.encode_network(donor, np.concatenate(g).astype(np.int32))  # also sets the payload size / padding in `donor`
    hdr = writer.cc_header_bytes(donor)
    _, levels = writer.grid_sizes((512, 768), hdr)
    jobs = [(1000 + i, i in (3, 8, 9, 16, 17, 18)) for i in range(24)]

    def make(job):
        seed, portrait = job
        v = writer.variant_latents(latents, levels, seed, portrait) if seed != 1000 else list(latents)
        return writer.encode_stream(hdr, nn, v, img_size=(768, 512) if portrait else (512, 768))

    with _pool(len(jobs)) as ex:
        streams = list(ex.map(make, jobs))
    return streams, [((768, 512) if p else (512, 768)) for _, p in jobs]


def _image_arch(donor: CCHeader, h: int, w: int) -> CCHeader:
    v = auto_resolution(h * w)
    return writer.derive_arch(donor, img_size=(h, w), latent_resolution=(0, v), hyperlatent_resolution=(4, v),
                              n_latent_grids=(v + 1) + (v - 3))


def _rolled(latents: Sequence[np.ndarray], arch: CCHeader, seed: int) -> List[np.ndarray]:
    """Per-picture variation: the tiled pyramid rolled consistently across levels."""
    _, levels = writer.grid_sizes(tuple(arch.img_size), writer.cc_header_bytes(arch))
    return writer.variant_latents(list(latents), levels, seed, False)


def image_stream(h: int, w: int, seed: int = 0, donor_name: str = "kodim14") -> bytes:
    """One RGB 8-bit picture of any size: the donor's networks (kodim14: HOP) grown to the "auto" number of levels, its latents tiled."""
    _, _, _, donor, ints, latents = _image_donor(donor_name)
    arch = _image_arch(donor, h, w)
    nn = writer.encode_network(arch, writer.adapt_network(donor, ints, arch))
    lat = writer.tile_latents(latents, donor, arch)
    if seed:
        lat = _rolled(lat, arch, seed)
    return writer.encode_stream(writer.cc_header_bytes(arch), nn, lat)


def clic41() -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """BASELINE configs[2]: 41 RGB 8-bit pictures with CLIC20-pro-valid's pixel counts (91.45 Mpx), HOP decoder."""
    with _pool(len(CLIC41_SIZES)) as ex:
        streams = list(ex.map(lambda a: image_stream(a[1][0], a[1][1], 2000 + a[0]), enumerate(CLIC41_SIZES)))
    return streams, list(CLIC41_SIZES)


def kodak24_hq() -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """kodak24's geometry (18 landscape + 6 portrait 512x768) with the statistics of a HIGH-RATE stream: the network and the
    latent pyramid of `hq192` (the reference's 192x128 test picture encoded by the reference encoder at lambda = 1e-5: 2.5 bpp,
    12-15 % of its symbols behind the wide 62-symbol windows, LOP decoder) tiled to Kodak size, rolled / transposed per
    picture like kodak24.  What real content at the top of results/v5.0/image-kodak.tsv's rate range does to the entropy
    stage; bench.py leg `kodak24_hq`."""
    _, hdr, nn, donor, _, latents = _image_donor("hq192")
    arch = writer.derive_arch(donor, img_size=(512, 768))  # same levels ("auto": < 1 Mpx), so the donor's payload fits as it is
    hdr = writer.cc_header_bytes(arch)
    tiled = writer.tile_latents(latents, donor, arch)
    _, levels = writer.grid_sizes((512, 768), hdr)
    jobs = [(1500 + i, i in (3, 8, 9, 16, 17, 18)) for i in range(24)]

    def make(job):
        seed, portrait = job
        v = writer.variant_latents(tiled, levels, seed, portrait)
        return writer.encode_stream(hdr, nn, v, img_size=(768, 512) if portrait else (512, 768))

    with _pool(len(jobs)) as ex:
        streams = list(ex.map(make, jobs))
    return streams, [((768, 512) if p else (512, 768)) for _, p in jobs]


def clic41_alt() -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """clic41's 41 sizes with OTHER decoders than kodim14's HOP: picture i carries the network and the tiled latents of
    `vhop192` (i even: cfg/dec/intra/vhop.cfg, 26 ARM inputs -> entropy_pipe_kernel<7>, 64 hidden synthesis units) or `mop192`
    (i odd: intra/mop.cfg, 14 inputs -> <4>, 16 hidden units), both encoded by the reference encoder, grown to the "auto"
    levels of the size.  Two kernel instantiations in one batch at 2 K size; bench.py leg `clic41_alt`."""
    with _pool(len(CLIC41_SIZES)) as ex:
        streams = list(ex.map(lambda a: image_stream(a[1][0], a[1][1], 2500 + a[0], "mop192" if a[0] & 1 else "vhop192"),
                              enumerate(CLIC41_SIZES)))
    return streams, list(CLIC41_SIZES)


def clic41_subset(indices: Sequence[int]) -> List[bytes]:
    """Streams `indices` of clic41() (each is built independently of the others): what one rank of a sharded run manufactures."""
    idx = list(indices)
    if not idx:
        return []
    with _pool(len(idx)) as ex:
        return list(ex.map(lambda i: image_stream(CLIC41_SIZES[i][0], CLIC41_SIZES[i][1], 2000 + i), idx))


def uhd4k() -> Tuple[List[bytes], List[Tuple[int, int]]]:
    """BASELINE configs[4]: one 3840x2160 RGB 8-bit picture, latent 0-8 + hyperlatent 4-8 (14 grids, 11.1 M symbols)."""
    return [image_stream(2160, 3840, 0)], [(2160, 3840)]


def hierarchical_gop(intra_period: int) -> List[Tuple[int, str, List[int], int]]:
    """Coding order of one closed GOP I0 .. I<period> with dyadic hierarchical B frames (the structure
    utils/codingstructure.py:267-436 produces for p_pos = []): [(display index, type, references, depth)]."""
    out = [(0, "I", [], 0), (intra_period, "I", [], 0)]

    def split(a, b, depth):
        if b - a < 2:
            return
        m = (a + b) // 2
        out.append((m, "B", [a, b], depth))
        split(a, m, depth + 1)
        split(m, b, depth + 1)

    split(0, intra_period, 1)
    return out


def gop1080p(intra_period: int = 32, size: Tuple[int, int] = (1080, 1920), i_frames: str = "vid5") -> Tuple[bytes, dict]:
    """BASELINE configs[3]: a 1920x1080 YUV 4:2:0 8-bit GOP of intra_period + 1 frames (I0, I<period>, hierarchical B in
    between): the networks and headers of the reference-encoded `vid5` donor, frame by role (its I frame; its depth-1 B
    frame = residue + motion; its deeper B frames), residue cool-chics grown to the "auto" levels of the picture size,
    latents tiled.  i_frames="hop" swaps the I frames' cool-chic for kodim14's HOP network (samples/encode.py:23-70 codes
    the intra frames of real sequences with intra/hop; vid5 was encoded with the --debug preset).  Returns (stream, info)."""
    from .bitstream.decode import _split_frame
    from .bitstream.header import VideoHeader

    bs, z = _golden("vid5")
    vh = VideoHeader()
    rest = vh.read_header(bs)
    # donors by role: vid5 is I0 P4 B2 B1 B3 (coding order); cc indices count cool-chics in coding order
    role, cc_idx = {}, 0
    for _ in range(vh.get_value("n_frames")):
        fh, ccs, rest = _split_frame(rest)
        ftype, di = fh.get_value("frame_type"), fh.get_value("display_index")
        key = ftype if ftype != "B" else ("B1" if di == 2 else "B2")
        if key not in role:
            role[key] = (fh.c, [(ch.raw, nn, [z[f"cc{cc_idx + j}.latent{g}"] for g in range(ch.c.n_grids)], z[f"cc{cc_idx + j}.nn_ints"])
                                for j, (ch, nn, _) in enumerate(ccs)])
        cc_idx += len(ccs)
    if i_frames == "hop":
        # I frames as samples/encode.py:30-36 configures them for real sequences: intra/hop = kodim14's architecture and
        # trained network.  The default keeps vid5's own I-frame cool-chic (the --debug preset's LOP network).
        _, k_hdr, k_nn, _, k_ints, k_lat = _kodim14()
        role["I"] = (role["I"][0], [(k_hdr, k_nn, k_lat, k_ints)])
    H, W = size
    order = hierarchical_gop(intra_period)

    def coolchic(donor_cc, seed):
        hdr, nn, lat, ints = donor_cc
        donor = writer.parse_cc_header(hdr)
        if donor.latent_resolution[0] == 0:  # residue / intra: "auto" levels for this picture size
            v = auto_resolution(H * W)
            arch = writer.derive_arch(donor, img_size=(H, W), latent_resolution=(0, v),
                                      hyperlatent_resolution=(4, v) if donor.flag_hyperlatent else tuple(donor.hyperlatent_resolution),
                                      n_latent_grids=(v + 1) + ((v - 3) if donor.flag_hyperlatent else 0))
            nn = writer.encode_network(arch, writer.adapt_network(donor, ints, arch))
        else:                                # motion: latent 2-6 whatever the size (cfg/dec/motion/*.cfg)
            arch = writer.derive_arch(donor, img_size=(H, W))
        tiled = writer.tile_latents(lat, donor, arch)
        return writer.encode_coolchic(arch, nn, _rolled(tiled, arch, seed) if seed else tiled)

    def frame_bytes(item):
        k, (di, ftype, refs, depth) = item
        fh, ccs = role["I" if ftype == "I" else ("B1" if depth == 1 else "B2")]
        head = writer.frame_header_bytes(di, ftype, fh.frame_data_type, fh.bitdepth, refs, [0] * (2 * len(refs)), fh.warp_filter_size)
        return head + b"".join(coolchic(cc, 3000 + 7 * k + j) for j, cc in enumerate(ccs))

    with _pool(len(order)) as ex:
        body = list(ex.map(frame_bytes, enumerate(order)))
    stream = writer.video_header_bytes(intra_period + 1, [0, intra_period], []) + b"".join(body)
    info = {"frames": intra_period + 1, "size": size, "coding_order": [o[0] for o in order],
            "cool_chics": sum(1 if o[1] == "I" else 2 for o in order)}
    return stream, info


def planes_sha256(planes) -> str:
    """sha256 over the integer planes of one frame in order (y u v / r g b), each as little-endian uint16 rows (8-bit planes
    widened): the one definition shared by tests/golden/gen/hash_workloads.py (CPU oracle), bench.py and the GPU tests."""
    import hashlib

    h = hashlib.sha256()
    for p in planes:
        h.update(np.ascontiguousarray(p, dtype="<u2").tobytes())
    return h.hexdigest()


def workload(name: str) -> Dict:
    """{"streams": [...], "sizes": [(H, W) per frame], "video": bool} for "kodak24" | "clic41" | "uhd4k" | "gop1080p33" (BASELINE's
    configurations) | "kodak24_wide_envelope" | "kodak24_hq" | "clic41_alt" (variants of them)."""
    if name == "kodak24":
        s, z = kodak24()
        return {"streams": s, "sizes": z, "video": False}
    if name == "kodak24_wide_envelope":
        s, z = kodak24_wide_envelope()
        return {"streams": s, "sizes": z, "video": False}
    if name == "clic41":
        s, z = clic41()
        return {"streams": s, "sizes": z, "video": False}
    if name == "kodak24_hq":
        s, z = kodak24_hq()
        return {"streams": s, "sizes": z, "video": False}
    if name == "clic41_alt":
        s, z = clic41_alt()
        return {"streams": s, "sizes": z, "video": False}
    if name == "uhd4k":
        s, z = uhd4k()
        return {"streams": s, "sizes": z, "video": False}
    if name == "gop1080p33":
        s, info = gop1080p(32)
        return {"streams": [s], "sizes": [info["size"]] * info["frames"], "video": True, "info": info}
    raise ValueError(name)
