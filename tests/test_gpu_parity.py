"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference fixtures.

Bars: integer stages bit-exact; float stages bit-exact to the oracle's
canonical accumulation order; integer planes identical to the oracle and within 1 LSB /
<= 2e-5 mismatch fraction of what the reference decoder produced."""
import ctypes as C

import numpy as np
import pytest

from conftest import IMAGE_STREAMS, SMALL_STREAMS, VIDEO_STREAMS, load_golden, reference_planes
from float_classes import nan_aware_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cool_chic_amd import DecodeBatch, _lib

    _lib.lib()
    return DecodeBatch


def _decode(gpu, triples, bitdepth, fdt, **opts):
    b = gpu(0, **opts)
    for hdr, nn, lat in triples:
        b.add(hdr, nn, lat, bitdepth, fdt)
    b.run()
    b.wait()
    return b


@pytest.mark.parametrize("path", ["production", "mfma", "unfused", "fused_tile_pyramid", "fused_behind_pyramid_steps"])
@pytest.mark.parametrize("name", IMAGE_STREAMS)
def test_stream_parity(gpu, oracle, name, path):
    """path = "production": the library defaults - the pipelined entropy kernel with the ARM on the vector ALU, upsampling +
    synthesis + integer samples in the fused kernel (ccd_fused*.hip, every preset architecture); "mfma": the same with the ARM's
    layers on the matrix cores where the stream allows (limb-split int8; the measured, slower alternative); "unfused":
    per-level upsampling launches + synthesis kernel, which also exposes the dense stack; the two fused variants by name:
    "fused_tile_pyramid" (CCD_OPT_FUSED_DEC = 1: the whole pyramid per tile, one launch) and "fused_behind_pyramid_steps"
    (= 2: levels >= 1 once per frame by the batch's pyramid steps, level 0 + synthesis + samples in the fused kernel)."""
    bs, z, j = load_golden(name)
    _, frames = oracle.split_stream(bs)
    fh, ccs = frames[0]
    ref = oracle.decode_coolchic(*ccs[0])
    opts = {"production": {}, "mfma": {"mfma_arm": 1}, "unfused": {"fused_dec": False, "mfma_arm": 0},
            "fused_tile_pyramid": {"fused_dec": 1}, "fused_behind_pyramid_steps": {"fused_dec": 2}}[path]
    b = _decode(gpu, ccs[:1], fh.bitdepth, fh.frame_data_type, **opts)
    try:
        # every network the reference encoder produced runs the pipelined entropy kernel (the generic int64 kernel is the
        # safety net for weights beyond int32, reached in tests through CCD_FORCE_GENERIC)
        assert b.slot_kernels(0) & 1, "the pipelined entropy kernel must serve this stream"
        assert b.slot_stats(0)[39] == 0, "no pixel of a reference-encoded stream needs the int64 redo"
        if path == "production" and name == "kodim14":
            # the decoder's part-by-part mode (a batch whose later parts are still being built) is reached on a real stream:
            # the bit-exact latents below cover it
            assert b.slot_stats(0)[37] > 0, "no batch was decoded part by part"
        if path != "mfma":  # networks whose WORST-CASE feature leaves 16 bits run the instantiation that checks features
            assert bool(b.slot_kernels(0) & 16) == (name in ("rgb192", "cr192", "yuv444_10b", "yuv444_8b")), b.slot_kernels(0)
        # common randomness: the noise planes are extra input channels of the fused kernel's instantiations behind the pyramid
        # launch (CCD_OPT_FUSED_DEC = 2, the default); the one-launch variant (= 1) sends such a stream to the unfused path
        if path != "unfused" and not (name == "cr192" and path == "fused_tile_pyramid"):
            assert b.slot_kernels(0) & 4, "the fused float kernel must serve this stream"
        if name == "cr192" and path == "fused_tile_pyramid":
            assert not b.slot_kernels(0) & 4
        if path == "mfma" and name == "kodim14":
            assert b.slot_kernels(0) & 8, "the ARM of the reference's sample stream must run on the matrix cores"
        if path == "production":
            assert not b.slot_kernels(0) & 8
        if path == "unfused":
            assert not b.slot_kernels(0) & 12
        if path.startswith("fused_"):
            assert bool(b.slot_kernels(0) & 64) == (path == "fused_behind_pyramid_steps"), b.slot_kernels(0)
        if path == "production":
            assert b.slot_kernels(0) & 64, "default: the fused kernel behind the pyramid launch"
        # integer stage: every latent grid bit-exact with the oracle AND the reference fixture
        for g in range(ref["n_grids"]):
            got = b.latent(0, g)
            assert np.array_equal(got, ref["latent"][g]), f"grid {g} vs oracle"
            assert np.array_equal(got, z[f"cc0.latent{g}"]), f"grid {g} vs reference fixture"
        # float stages: bit-exact with the oracle
        if not b.slot_kernels(0) & 4:
            dense = b.dense(0)
            assert np.array_equal(dense.view(np.uint32), ref["dense"].view(np.uint32)), "Upsampling.forward"
        out = b.output(0)
        assert np.array_equal(out.view(np.uint32), ref["out"].view(np.uint32)), "synthesis output"
        # integer planes: identical to the oracle, within the reference's noise floor of the fixture
        want = oracle.decode_video(bs)[0]["planes"]
        planes = b.planes(0)
        n_diff = n_tot = 0
        for p, w, r in zip(planes, want, reference_planes(z, j)):
            assert np.array_equal(p.astype(np.uint16), w)
            d = np.abs(p.astype(np.int64) - r.astype(np.int64))
            assert d.max() <= 1
            n_diff += int((d != 0).sum())
            n_tot += d.size
        assert n_diff <= max(3, 2e-5 * n_tot)
    finally:
        b.close()


@pytest.mark.parametrize("bits", [4, 10, 16])
@pytest.mark.parametrize("name", ["kodim14", "yuv420_8b"])
def test_mfma_arm_exact_redo(gpu, oracle, name, bits):
    """Hidden activations travel through the matrix cores as three signed bytes; a task that meets a wider one is redone in
    plain int64.  Real streams never get there, so the width limit is lowered until (nearly) every / many / a few tasks do."""
    bs, z, j = load_golden(name)
    fh, ccs = oracle.split_stream(bs)[1][0]
    b = _decode(gpu, ccs[:1], fh.bitdepth, fh.frame_data_type, mfma_arm=bits)
    try:
        assert b.slot_kernels(0) & 8
        assert b.slot_status(0) == 0
        for g in range(len([k for k in z.files if k.startswith("cc0.latent")])):
            assert np.array_equal(b.latent(0, g), z[f"cc0.latent{g}"]), f"grid {g} vs reference fixture"
    finally:
        b.close()


@pytest.mark.parametrize("bits", [8, 9, 11])
@pytest.mark.parametrize("name", ["kodim14", "rgb192", "yuv444_10b"])
def test_dynamic_operand_redo(gpu, oracle, name, bits):
    """The pipelined entropy kernel multiplies 32-bit operands.  Weights and hidden activations are covered by the host's
    analysis of the weights; IFCE features are checked on the device: one that does not fit int16 is a sentinel in the
    feature plane and sends the pixels that read it through an int64 redo (exact_pixel).  No real stream gets there
    (features reach ~2^10 of 2^15), so the limit is lowered until many / some / a few pixels do: the latents must not change.
    rgb192 and yuv444_10b are networks whose WORST-CASE feature exceeds 2^15 (the r02 static envelope sent them to the
    generic kernel)."""
    bs, z, j = load_golden(name)
    fh, ccs = oracle.split_stream(bs)[1][0]
    b = _decode(gpu, ccs[:1], fh.bitdepth, fh.frame_data_type, range_bits=bits)
    try:
        assert b.slot_kernels(0) & 17 == 17, "the pipelined kernel's instantiation with the device check of the features"
        assert b.slot_status(0) == 0
        n_redo = int(b.slot_stats(0)[39])
        if bits == 8:
            assert n_redo > 0, "the lowered limit must drive pixels through the redo"
        for g in range(len([k for k in z.files if k.startswith("cc0.latent")])):
            assert np.array_equal(b.latent(0, g), z[f"cc0.latent{g}"]), f"grid {g} vs reference fixture ({n_redo} pixels redone)"
    finally:
        b.close()


@pytest.mark.parametrize("variant", ["static", "dyn14", "dyn9"])
def test_arm_sweep_every_instantiation(gpu, oracle, variant):
    """EVERY instantiation of the pipelined entropy kernel the library ships: 91 streams with 3 .. 32 ARM inputs (NV = ceil(inputs
    / 4) = 1 .. 8), 0 .. 3 (and 6 / 7) hidden layers, with and without IFCE, 32 x 320 pictures whose three finest grids run the
    producers' 8-, 4- and 2-pixel tasks - all in ONE batch, so that its up to 16 kernel instantiations fork over the side streams
    and join.  "static": the instantiation the network's worst-case feature selects; "dyn14": the one with the device check of the
    features forced for every stream (limit 2^14: nothing is redone); "dyn9": limit 2^9, pixels go through the int64 redo whose
    lane broadcast depends on NV.  Bars: latents == what the REFERENCE decoder decoded (sha256 per grid in the fixture), output
    and integer planes == the oracle's, bit for bit."""
    import hashlib

    from conftest import load_arm_sweep
    from cool_chic_amd._lib import lib

    sweep = load_arm_sweep()
    names = list(sweep)
    b = gpu(0, **({} if variant == "static" else {"range_bits": int(variant[3:])}))
    try:
        triples = []
        for n in names:
            hdr, nn, lat = oracle.split_stream(sweep[n][0])[1][0][1][0]
            triples.append((hdr, nn, lat))
            b.add(hdr, nn, lat, 8, 0)
        b.run()
        b.wait()
        seen, n_redo = set(), 0
        for i, n in enumerate(names):
            hdr, nn, lat = triples[i]
            cls = lib().ccd_network_kernel_class(hdr, len(hdr), nn, len(nn))
            k = b.slot_kernels(i)
            assert b.slot_status(i) == 0, n
            assert k & 1 and cls & 1, f"{n}: the pipelined entropy kernel must serve every shape of the sweep"
            assert bool(k & 16) == (variant != "static" or bool(cls & 16)), n
            assert k & 4, f"{n}: fused float kernel"
            seen.add(((cls >> 8) & 15, bool(k & 16)))
            n_redo += int(b.slot_stats(i)[39])
            h = b.header(i)
            got = [hashlib.sha256(np.ascontiguousarray(b.latent(i, g)).tobytes()).hexdigest() for g in range(h.n_grids)]
            assert got == sweep[n][1], f"{n}: latents vs the reference decoder"
        # every width ran; both variants of every width over the three runs of this test
        assert {nv for nv, _ in seen} == set(range(1, 9))
        if variant == "static":
            assert all((nv, False) in seen for nv in range(1, 9))
            assert n_redo == 0
        else:
            assert all((nv, True) in seen for nv in range(1, 9))
            assert (n_redo > 0) == (variant == "dyn9"), n_redo
        # float path and integer planes against the oracle for one stream per width and depth (the float path does not depend on the ARM)
        for i, n in enumerate(names):
            if not (n.endswith("_h2_i2") or n.endswith("_h0_i0") or "_h7_" in n or "_h6_" in n):
                continue
            ref = oracle.decode_coolchic(*triples[i])
            assert np.array_equal(b.output(i).view(np.uint32), ref["out"].view(np.uint32)), n
            want = oracle.decode_video(sweep[n][0])[0]["planes"]
            for p, w in zip(b.planes(i), want):
                assert np.array_equal(p.astype(np.uint16), w), n
    finally:
        b.close()


@pytest.mark.parametrize("name", ["rgb192", "yuv420_10b"])
def test_generic_fallback_kernels(gpu, oracle, name, monkeypatch):
    """The barrier-phased int64 entropy kernel and the per-layer synthesis kernels (used for networks
    outside the fast paths' envelope) give the same bits as the production kernels."""
    monkeypatch.setenv("CCD_FORCE_GENERIC", "1")
    bs, z, j = load_golden(name)
    fh, ccs = oracle.split_stream(bs)[1][0]
    ref = oracle.decode_coolchic(*ccs[0])
    b = _decode(gpu, ccs[:1], fh.bitdepth, fh.frame_data_type)
    try:
        for g in range(ref["n_grids"]):
            assert np.array_equal(b.latent(0, g), ref["latent"][g])
        assert np.array_equal(b.output(0).view(np.uint32), ref["out"].view(np.uint32))
        for p, w in zip(b.planes(0), oracle.decode_video(bs)[0]["planes"]):
            assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()


def test_many_streams_in_one_batch(gpu, oracle):
    """Several different architectures / sizes in flight together decode like they do alone."""
    triples, refs, meta = [], [], []
    for name in SMALL_STREAMS + ["kodim14"] + SMALL_STREAMS:
        bs, z, j = load_golden(name)
        fh, ccs = oracle.split_stream(bs)[1][0]
        triples.append(ccs[0])
        refs.append(oracle.decode_video(bs)[0]["planes"])
        meta.append((fh.bitdepth, fh.frame_data_type))
    b = gpu(0)
    try:
        for t, m in zip(triples, meta):
            b.add(*t, *m)
        for _ in range(2):  # a batch can be re-run
            b.run()
            b.wait()
            for i, want in enumerate(refs):
                for p, w in zip(b.planes(i), want):
                    assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()


@pytest.mark.parametrize("which", ["pipe", "generic"])
def test_laplace_boundaries_sweep(gpu, oracle, which):
    """Every left cumulative (32768 mu indices x 127 boundaries) of every 64th scale index - 1.7e8 of the 1.0658e10 reachable
    ones - computed by the production kernel's table builder ("pipe": window_left with its own exp and quotient) and by the
    generic kernel's, against libm.  The exhaustive sweep is tools/cdf_sweep.py; its log is profiles/r03/cdf_sweep.log."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cdf_sweep

    scales = list(range(0, 2561, 64)) + [1279, 1280, 1281, 2560]
    n, bad, off = cdf_sweep.sweep(0 if which == "pipe" else 1, sorted(set(scales)), 4, min(32, os.cpu_count() or 4), log=lambda m: None)
    assert n == len(set(scales)) * 32768 * 127
    assert bad == 0, f"{bad} boundaries differ from libm, first (scale_idx, mu_idx, s, device, libm): {off[:4]}"


def test_laplace_boundaries_on_device(gpu, oracle):
    """The f64 CDF boundaries the entropy kernel computes vs libm on the host (hard part H2)."""
    from cool_chic_amd._lib import check, lib

    rng = np.random.default_rng(1)
    n = 400_000
    mu = rng.integers(0, 32768, n).astype(np.int32)
    sc = rng.integers(0, 2561, n).astype(np.int32)
    s = rng.integers(-64, 64, n).astype(np.int32)
    # add structured cases: every scale at a few mus, all symbols
    mm, cc, ss = np.meshgrid(np.array([0, 1, 255, 16384, 16385, 20000, 32767]), np.arange(0, 2561, 7), np.arange(-64, 64))
    mu = np.concatenate([mu, mm.ravel().astype(np.int32)])
    sc = np.concatenate([sc, cc.ravel().astype(np.int32)])
    s = np.concatenate([s, ss.ravel().astype(np.int32)])
    left = np.empty(mu.size, np.uint32)
    right = np.empty(mu.size, np.uint32)
    check(lib().ccd_debug_laplace_bounds(0, mu.ctypes.data, sc.ctypes.data, s.ctypes.data, mu.size, left.ctypes.data,
                                         right.ctypes.data))
    L = oracle.lib()
    a, b = C.c_uint32(), C.c_uint32()
    bad = 0
    for i in range(0, mu.size, 1):
        L.ora_laplace_bounds(int(mu[i]), int(sc[i]), int(s[i]), C.byref(a), C.byref(b))
        if a.value != left[i] or b.value != right[i]:
            bad += 1
    assert bad == 0, f"{bad} of {mu.size} boundaries differ from the host"


def test_synthetic_variants_round_trip(gpu, oracle):
    """writer -> GPU decoder round trip at the Kodak size, both orientations (config 1 inputs)."""
    from cool_chic_amd import writer

    bs, z, j = load_golden("kodim14")
    hdr, nn, lat = oracle.split_stream(bs)[1][0][1][0]
    latents = [z[f"cc0.latent{g}"] for g in range(10)]
    _, levels = writer.grid_sizes((512, 768), hdr)
    streams, variants = [], []
    for seed, tr in ((11, False), (12, True), (13, False)):
        v = writer.variant_latents(latents, levels, seed, tr)
        streams.append(writer.encode_stream(hdr, nn, v, img_size=(768, 512) if tr else (512, 768)))
        variants.append(v)
    triples = [oracle.split_stream(s)[1][0][1][0] for s in streams]
    b = _decode(gpu, triples, 8, 0)
    try:
        for i, v in enumerate(variants):
            for g, a in enumerate(v):
                assert np.array_equal(b.latent(i, g), a), f"stream {i} grid {g}"
        # one full check against the oracle for the portrait stream
        want = oracle.decode_video(streams[1])[0]["planes"]
        for p, w in zip(b.planes(1), want):
            assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()


def test_corrupt_payload_is_reported_or_contained(gpu, oracle):
    """A damaged payload must not hang or crash: it either decodes to (wrong) symbols or reports
    CCD_ERR_INVALID_DATA, exactly like the oracle does on the same bytes."""
    from cool_chic_amd import CcdError

    bs, z, j = load_golden("rgb192")
    hdr, nn, lat = oracle.split_stream(bs)[1][0][1][0]
    bad = bytearray(lat)
    for i in range(40, 80):
        bad[i] ^= 0xA5
    bad = bytes(bad)
    try:
        ref = oracle.decode_coolchic(hdr, nn, bad, stop_after_entropy=True)
    except oracle.OracleError:
        ref = None
    b = gpu(0)
    try:
        b.add(hdr, nn, bad, 8, 0)
        b.run()
        if ref is None:
            with pytest.raises(CcdError):
                b.wait()
        else:
            b.wait()
            for g in range(ref["n_grids"]):
                assert np.array_equal(b.latent(0, g), ref["latent"][g])
    finally:
        b.close()


def test_python_surface(gpu, oracle, tmp_path):
    """decode_video / encode_decode_coolchic mirrors + the PNG writer."""
    import torch
    from PIL import Image

    from cool_chic_amd.bitstream.component.coolchic import encode_decode_coolchic
    from cool_chic_amd.bitstream.decode import decode_video
    from cool_chic_amd.bitstream.header import CoolChicHeader, FrameHeader, VideoHeader
    from conftest import GOLDEN
    import os

    out_png = str(tmp_path / "k.png")
    frames = decode_video(os.path.join(GOLDEN, "kodim14.cool"), decoded_path=out_png)
    assert list(frames) == ["0"]
    fd = frames["0"]
    assert fd.bitdepth == 8 and fd.frame_data_type == "rgb" and tuple(fd.data.shape) == (1, 3, 512, 768)
    bs, z, j = load_golden("kodim14")
    want = np.stack(oracle.decode_video(bs)[0]["planes"]).astype(np.uint8)
    png = np.asarray(Image.open(out_png)).transpose(2, 0, 1)
    assert np.array_equal(png, want)
    # boundary function
    rest = VideoHeader().read_header(bs)
    rest = FrameHeader().read_header(rest)
    ch = CoolChicHeader()
    rest = ch.read_header(rest)
    nn, lat = rest[: ch.get_value("nn_n_bytes")], rest[ch.get_value("nn_n_bytes"):]
    t, none = encode_decode_coolchic(ch, nn, "decode", dec_bytes_latent=lat)
    assert none is None and t.dtype == torch.float32 and tuple(t.shape) == (1, 3, 512, 768)
    ref = oracle.decode_coolchic(ch.raw, nn, lat)["out"]
    assert np.array_equal(t[0].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    with pytest.raises(ValueError):
        encode_decode_coolchic(ch, nn, "decode", dec_bytes_latent=None)
    # verbosity >= 2 (coolchic.py:199-205): the header, then one line of four section times; same tensor
    import contextlib
    import io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        t_v, _ = encode_decode_coolchic(ch, nn, "decode", dec_bytes_latent=lat, verbosity=2)
    assert torch.equal(t_v, t)
    last = buf.getvalue().rstrip("\n").split("\n")[-1].split()
    assert len(last) == 4 and all(float(x) >= 0.0 for x in last) and float(last[2]) > 0.0
    # the boundary's second caller (bitstream/encode.py:83-89): mode="encode" with the decoded latents as the encoder's
    # quantised latents gives back the reference encoder's own bytes (header + NN + latent payload) and the same output
    with pytest.raises(ValueError):
        encode_decode_coolchic(ch, nn, "encode")
    q = [torch.from_numpy(z[f"cc0.latent{g}"].astype(np.float32))[None, None] for g in range(ch.c.n_grids)]
    ch2 = CoolChicHeader()
    ch2.read_header(ch.raw)
    ch2.c.n_bytes_latent = 0  # "we don't know the number of bytes in the latent grids yet" (bitstream/encode.py:76-79)
    t2, enc = encode_decode_coolchic(ch2, nn, "encode", enc_quantized_latent=q)
    assert enc == ch.raw + nn + lat
    assert ch2.get_value("n_bytes_latent") == len(lat)
    assert torch.equal(t2, t)


@pytest.mark.parametrize("stream", VIDEO_STREAMS)
def test_video_ipb_parity(gpu, oracle, tmp_path, stream):
    """I/P/B YUV420 videos encoded by the reference encoder (vid5: 5 frames, sinc-8 warp; _w2 / _w4: the same stream with the
    2-tap bilinear / 4-tap bicubic grid_sample warp; vid3_*: 3 frames with the decoder presets vid5 does not use - intra
    vhop / mop, residue hop / mop / vlop, motion mop): C ABI (ccd_decode_video) and the Python mirror (decode_video /
    decode_frame) against the oracle (bit-exact) and the reference fixture (<= 1 LSB)."""
    import os

    from cool_chic_amd._lib import Video, check, lib
    from cool_chic_amd.bitstream.decode import decode_video
    from conftest import GOLDEN

    bs, z, j = load_golden(stream)
    want = oracle.decode_video(bs)
    v = Video()
    check(lib().ccd_decode_video(bs, len(bs), 0, C.byref(v)), "ccd_decode_video")
    try:
        nf = len(want)
        assert v.n_frames == nf == (5 if stream.startswith("vid5") else 3)
        if stream == "vid3_ldp":  # low delay: I P P, the second P frame predicted from the first
            assert [want[i]["frame_type"] for i in range(3)] == ["I", "P", "P"]
        n_diff = n_tot = 0
        for i in range(nf):
            f = v.frames[i]
            assert "IPB"[f.frame_type] == want[i]["frame_type"] and f.bitdepth == 8 and f.frame_data_type == 1
            shapes = [(f.h, f.w), (f.ch, f.cw), (f.ch, f.cw)]
            for p, (name, shape) in enumerate(zip("yuv", shapes)):
                got = np.ctypeslib.as_array(f.plane[p], shape=shape)
                assert np.array_equal(got, want[i]["planes"][p]), f"frame {i} plane {name} vs oracle"
                d = np.abs(got.astype(np.int64) - z[f"frame{i}.{name}"].astype(np.int64))
                assert d.max() <= 1
                n_diff += int((d != 0).sum())
                n_tot += d.size
        assert n_diff / n_tot <= 1e-4
    finally:
        lib().ccd_video_free(C.byref(v))
    # Python surface: decode_video writes a planar YUV file, frames in display order
    out_yuv = str(tmp_path / "v.yuv")
    frames = decode_video(os.path.join(GOLDEN, stream + ".cool"), decoded_path=out_yuv)
    assert list(frames) == [str(i) for i in range(nf)]
    raw = np.fromfile(out_yuv, dtype=np.uint8)
    expect = np.concatenate([np.concatenate([p.astype(np.uint8).ravel() for p in want[i]["planes"]]) for i in range(nf)])
    assert np.array_equal(raw, expect)


FULL_SIZE = {
    # CLIC-2K picture, latent 0-7 + hyperlatent 4-7 (12 grids)
    "clic2k": dict(img_size=(1365, 2048), latent_resolution=(0, 7), hyperlatent_resolution=(4, 7), n_latent_grids=12),
    # 3840x2160, latent 0-8 + hyperlatent 4-8 (14 grids, 11.1 M symbols)
    "uhd4k": dict(img_size=(2160, 3840), latent_resolution=(0, 8), hyperlatent_resolution=(4, 8), n_latent_grids=14),
}


@pytest.mark.parametrize("name", list(FULL_SIZE))
def test_full_size_configs(gpu, oracle, name):
    """BASELINE.json configs 2 and 4 at their full sizes: kodim14's trained networks grown to 12 / 14 grids,
    tiled real latents, written with the bitstream writer.  Round trip (every decoded grid equals the encoded
    one) plus integer planes bit-exact against the oracle."""
    import time

    from cool_chic_amd import writer

    bs, z, _ = load_golden("kodim14")
    hdr, _, _ = oracle.split_stream(bs)[1][0][1][0]
    donor = writer.parse_cc_header(hdr)
    arch = writer.derive_arch(donor, **FULL_SIZE[name])
    nn = writer.encode_network(arch, writer.adapt_network(donor, z["cc0.nn_ints"], arch))
    latents = writer.tile_latents([z[f"cc0.latent{g}"] for g in range(donor.n_grids)], donor, arch)
    stream = writer.encode_stream(writer.cc_header_bytes(arch), nn, latents)
    triple = oracle.split_stream(stream)[1][0][1][0]
    b = _decode(gpu, [triple], 8, 0)
    try:
        assert b.slot_status(0) == 0
        assert b.slot_kernels(0) & 15 == 5, "the reference configurations must run on the pipelined / fused kernels"
        t0 = time.time()
        b.run(); b.wait()
        dt = time.time() - t0
        print(f"\n{name}: {len(stream)} bytes, {arch.n_symbols} symbols, GPU decode {dt * 1e3:.1f} ms "
              f"({arch.img_size[0] * arch.img_size[1] / dt / 1e6:.1f} Mpx/s single stream)")
        for g, a in enumerate(latents):
            assert np.array_equal(b.latent(0, g), a), f"grid {g}"
        want = oracle.decode_video(stream)[0]["planes"]
        for p, w in zip(b.planes(0), want):
            assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()


def test_picture_wider_than_the_symbol_ring(gpu, oracle):
    """A 7 680 x 256 picture (the format allows 16 383 columns, header.py:244-307): wider than the 5 060 columns the pipelined
    entropy kernel's symbol ring holds, so the GENERIC entropy kernel serves it (slot_kernels bit 0 clear) while the float path
    stays on the fused kernels.  Round trip of every grid and integer planes against the oracle."""
    from cool_chic_amd import writer

    bs, z, _ = load_golden("kodim14")
    hdr, _, _ = oracle.split_stream(bs)[1][0][1][0]
    donor = writer.parse_cc_header(hdr)
    arch = writer.derive_arch(donor, img_size=(256, 7680))
    nn = writer.encode_network(arch, writer.adapt_network(donor, z["cc0.nn_ints"], arch))
    latents = writer.tile_latents([z[f"cc0.latent{g}"] for g in range(donor.n_grids)], donor, arch)
    stream = writer.encode_stream(writer.cc_header_bytes(arch), nn, latents)
    triple = oracle.split_stream(stream)[1][0][1][0]
    b = _decode(gpu, [triple], 8, 0)
    try:
        assert b.slot_status(0) == 0
        assert b.slot_kernels(0) & 1 == 0, "wider than the ring: the generic entropy kernel"
        assert b.slot_kernels(0) & 4, "the float path does not depend on the width"
        for g, a in enumerate(latents):
            assert np.array_equal(b.latent(0, g), a), f"grid {g}"
        want = oracle.decode_video(stream)[0]["planes"]
        for p, w in zip(b.planes(0), want):
            assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()


def test_video_warp_coefficients_ahead_of_the_references(oracle, monkeypatch):
    """With a scratch budget (CCD_VIDEO_COEF_MB; off by default: measured -0.6 % for 8 GB on the 1080p GOP) ccd_decode_video computes
    the sinc-8 warp's coefficients (f64 sin / cos of the flows) as soon as a frame's motion cool-chic is decoded and only gathers
    behind the references (ccd_inter.hip: inter_coef8_kernel + inter_apply8_kernel); without - or for frames beyond the budget, other
    filter sizes - the one-kernel form runs.  Same planes, both the oracle's, for the 5-frame I/P/B fixture and its 3-frame relatives."""
    from cool_chic_amd._lib import Video, check, lib

    for stream in ("vid5", "vid3_hop", "vid3_ldp"):
        bs, _, _ = load_golden(stream)
        want = oracle.decode_video(bs)
        got = {}
        for label, mb in (("in_kernel", None), ("ahead", "4096"), ("in_kernel_0", "0"), ("one_frame", "1")):
            if mb is None:
                monkeypatch.delenv("CCD_VIDEO_COEF_MB", raising=False)
            else:
                monkeypatch.setenv("CCD_VIDEO_COEF_MB", mb)
            v = Video()
            check(lib().ccd_decode_video(bs, len(bs), 0, C.byref(v)), "ccd_decode_video")
            try:
                planes = []
                for i in range(v.n_frames):
                    f = v.frames[i]
                    shapes = [(f.h, f.w), (f.ch, f.cw), (f.ch, f.cw)]
                    planes.append([np.ctypeslib.as_array(f.plane[p], shape=shapes[p]).copy() for p in range(3)])
                got[label] = planes
            finally:
                lib().ccd_video_free(C.byref(v))
        for label, planes in got.items():
            assert len(planes) == len(want)
            for i, fr in enumerate(planes):
                for p in range(3):
                    assert np.array_equal(fr[p], want[i]["planes"][p]), f"{stream} {label} frame {i} plane {p}"


def test_video_1080p_gop(gpu, oracle):
    """SURVEY 8d config 3 at full picture size (reduced GOP): the I/P/B/B/B structure, networks and headers of
    the reference-encoded `vid5` stream with its latents tiled to 1920x1080 4:2:0, re-written frame by frame.
    ccd_decode_video against the oracle, bit-exact, all five frames."""
    import time

    from cool_chic_amd import writer
    from cool_chic_amd._lib import Video, check, lib

    bs, z, _ = load_golden("vid5")
    vh, frames = oracle.split_stream(bs)
    H, W = 1080, 1920
    out = [writer.video_header_bytes(vh.n_frames, list(vh.intra_pos[:vh.n_intras]), list(vh.p_pos[:vh.n_p_frames]))]
    cc_idx = 0
    for fh, ccs in frames:
        out.append(writer.frame_header_bytes(fh.display_index, "IPB"[fh.frame_type], fh.frame_data_type, fh.bitdepth,
                                             list(fh.index_references[:fh.n_refs]), list(fh.global_flow[:2 * fh.n_refs]),
                                             fh.warp_filter_size))
        for hdr, nn, _lat in ccs:
            donor = writer.parse_cc_header(hdr)
            arch = writer.derive_arch(donor, img_size=(H, W))
            lat = [z[f"cc{cc_idx}.latent{g}"] for g in range(donor.n_grids)]
            out.append(writer.encode_coolchic(arch, nn, writer.tile_latents(lat, donor, arch)))
            cc_idx += 1
    stream = b"".join(out)
    t0 = time.time()
    want = oracle.decode_video(stream)
    t_cpu = time.time() - t0
    v = Video()
    t0 = time.time()
    check(lib().ccd_decode_video(stream, len(stream), 0, C.byref(v)), "ccd_decode_video")
    t_gpu = time.time() - t0
    print(f"\n1080p x {vh.n_frames} frames: {len(stream)} bytes; oracle {t_cpu:.1f} s, ccd_decode_video {t_gpu * 1e3:.0f} ms (incl. parse/upload)")
    try:
        assert v.n_frames == vh.n_frames
        for i in range(v.n_frames):
            f = v.frames[i]
            assert (f.h, f.w, f.ch, f.cw) == (H, W, H // 2, W // 2)
            for p, shape in enumerate([(f.h, f.w), (f.ch, f.cw), (f.ch, f.cw)]):
                got = np.ctypeslib.as_array(f.plane[p], shape=shape)
                assert np.array_equal(got, want[i]["planes"][p]), f"frame {i} plane {p}"
    finally:
        lib().ccd_video_free(C.byref(v))


@pytest.mark.parametrize("donor_name,sizes", [
    ("rgb192", [(1, 1), (1, 13), (13, 1), (7, 9), (9, 10), (10, 9), (17, 33), (64, 65), (100, 37), (129, 257)]),
    ("yuv420_8b", [(2, 2), (10, 18), (34, 66), (130, 94)]),
    ("cr192", [(5, 11), (63, 130)]),
])
def test_ragged_picture_sizes(gpu, oracle, donor_name, sizes):
    """Edge geometry: 1-pixel pictures, grids narrower than ten columns (raster coding order instead of the wavefront,
    latent.py:113-122), odd sizes (ceil at every pyramid level, crop after every x2 upsampling), tile borders of the
    float kernels, 4:2:0 chroma of small pictures.  Streams: the donor's trained networks, seeded random latents,
    written with the bitstream writer; latents and integer planes bit-exact against the oracle."""
    from cool_chic_amd import writer

    bs, z, j = load_golden(donor_name)
    (fh, ccs), = oracle.split_stream(bs)[1]
    hdr, nn, _ = ccs[0]
    donor = writer.parse_cc_header(hdr)
    rng = np.random.default_rng(20240926)
    streams, lat_in = [], []
    for (h, w) in sizes:
        arch = writer.derive_arch(donor, img_size=(h, w))
        nn_i = nn
        if writer.network_layout(arch) != writer.network_layout(donor):
            # the reference derives a grid's level from the HEIGHT ratio (component/core/coolchic.py:209-211): on 1-pixel-high
            # pictures every grid counts as level 0 and takes IFCE inputs - the network grows
            nn_i = writer.encode_network(arch, writer.adapt_network(donor, z["cc0.nn_ints"], arch))
        lat = [np.clip(np.rint(rng.laplace(0.0, 1.2, size=(arch.grid_h[g], arch.grid_w[g]))), -20, 20).astype(np.int8)
               for g in range(arch.n_grids)]
        streams.append(writer.encode_stream(writer.cc_header_bytes(arch), nn_i, lat, bitdepth=fh.bitdepth,
                                            frame_data_type=fh.frame_data_type))
        lat_in.append(lat)
    triples = [oracle.split_stream(s)[1][0][1][0] for s in streams]
    b = _decode(gpu, triples, fh.bitdepth, fh.frame_data_type)
    try:
        for i, (s, lat) in enumerate(zip(streams, lat_in)):
            assert b.slot_status(i) == 0
            for g, a in enumerate(lat):
                assert np.array_equal(b.latent(i, g), a), f"size {sizes[i]} grid {g}"
            want = oracle.decode_video(s)[0]["planes"]
            for p, (got, w_) in enumerate(zip(b.planes(i), want)):
                assert got.shape == w_.shape and np.array_equal(got.astype(np.uint16), w_), f"size {sizes[i]} plane {p}"
    finally:
        b.close()


@pytest.mark.parametrize("donor_name,sizes", [
    # grid 0 = the picture: streamed when its longest step ((W - 1) // 10 + 1 pixels) has >= 48 pixels or ends in a task of 1..4
    # pixels, W > 230, H >= 25 (8-pixel tasks).  The body then is steps [230, W + 10 (H - 24)):
    ("kodim14", [(25, 241), (25, 250), (40, 271), (33, 480), (64, 521), (300, 241), (57, 332)]),
    # ... and grid 1 (half size) as well: 40 x 250 / 45 x 261
    ("kodim14", [(80, 500), (90, 522)]),
    ("yuv420_8b", [(48, 482)]),
])
def test_streamed_body_geometry(gpu, oracle, donor_name, sizes):
    """The body of a wide grid runs as ONE stream of pixels (ccd_entropy_pipe.hip: StreamBody - batches and tasks cut without
    regard to wavefront steps, a task may hold pixels of two steps): sizes around every bound of that rule - the smallest width
    and height, bodies of a few steps, step lengths whose tail is 1, 2, 4 pixels, both the picture's and the half-size grid.
    Streams: the donor's trained networks, seeded random latents, written with the bitstream writer; latents and integer planes
    bit-exact against the oracle; stats word 36 says how many grids were streamed."""
    from cool_chic_amd import writer

    bs, z, j = load_golden(donor_name)
    (fh, ccs), = oracle.split_stream(bs)[1]
    hdr, nn, _ = ccs[0]
    donor = writer.parse_cc_header(hdr)
    rng = np.random.default_rng(20240927)
    streams, lat_in = [], []
    for (h, w) in sizes:
        arch = writer.derive_arch(donor, img_size=(h, w))
        assert writer.network_layout(arch) == writer.network_layout(donor)
        lat = [np.clip(np.rint(rng.laplace(0.0, 1.0 + 0.3 * g, size=(arch.grid_h[g], arch.grid_w[g]))), -30, 30).astype(np.int8)
               for g in range(arch.n_grids)]
        streams.append(writer.encode_stream(writer.cc_header_bytes(arch), nn, lat, bitdepth=fh.bitdepth, frame_data_type=fh.frame_data_type))
        lat_in.append(lat)
    triples = [oracle.split_stream(s)[1][0][1][0] for s in streams]
    b = _decode(gpu, triples, fh.bitdepth, fh.frame_data_type)
    try:
        for i, (s, lat) in enumerate(zip(streams, lat_in)):
            assert b.slot_status(i) == 0
            assert b.slot_kernels(i) & 1
            assert b.slot_stats(i)[36] >= 1, f"size {sizes[i]}: no grid was streamed"
            for g, a in enumerate(lat):
                assert np.array_equal(b.latent(i, g), a), f"size {sizes[i]} grid {g}"
            want = oracle.decode_video(s)[0]["planes"]
            for p_, (got, w_) in enumerate(zip(b.planes(i), want)):
                assert got.shape == w_.shape and np.array_equal(got.astype(np.uint16), w_), f"size {sizes[i]} plane {p_}"
    finally:
        b.close()


def test_repeated_runs_are_identical(gpu, oracle):
    """The pipelined entropy kernel hands work between waves through LDS flags; a lost or early hand-over would show as a
    run-to-run difference.  One batch (kodim14 twice + the small fixtures), twenty runs, every latent and plane equal."""
    import hashlib

    triples = []
    for name in ["kodim14", "rgb192", "yuv444_10b", "kodim14", "cr192", "bicubic190"]:
        bs, _, _ = load_golden(name)
        triples.append(oracle.split_stream(bs)[1][0][1][0])
    b = gpu(0)
    for t in triples:
        b.add(*t, 8, 0)
    try:
        ref = None
        for it in range(20):
            b.run(); b.wait()
            h = hashlib.sha256()
            for s in range(len(triples)):
                for g in range(b.header(s).n_grids):
                    h.update(np.ascontiguousarray(b.latent(s, g)).tobytes())
                for p in b.planes(s):
                    h.update(np.ascontiguousarray(p).tobytes())
            ref = ref or h.hexdigest()
            assert h.hexdigest() == ref, f"run {it} differs from run 0"
    finally:
        b.close()


def test_fixed_shape_instantiations_match_the_runtime_shape(gpu, oracle, monkeypatch):
    """The pipelined entropy kernel is instantiated with a compile-time ARM shape for the intra preset the benchmark sets use -
    hop.cfg (14 + 6 inputs: kodim14, kodak24, clic41, uhd4k), slot_kernels bit 5 (lop.cfg got one in r06 and lost it again: no
    gain).  CCD_FIXED_SHAPE=0 sends the same stream through the run-time-shape instantiation: identical latents and planes, and
    both are the reference's (the fixtures' expected values); the other networks never take a fixed shape."""
    names = ["kodim14", "rgb192", "cr192", "yuv444_10b", "hq192", "mop192"]
    triples = []
    for n in names:
        bs, _, _ = load_golden(n)
        fh, ccs = oracle.split_stream(bs)[1][0]
        triples.append((ccs[0], fh.bitdepth, fh.frame_data_type))

    def decode():
        b = gpu(0)
        for (hdr, nn, lat), bd, fdt in triples:
            b.add(hdr, nn, lat, bd, fdt)
        b.run(); b.wait()
        out = [([b.latent(s, g) for g in range(b.header(s).n_grids)], b.planes(s), b.slot_kernels(s)) for s in range(len(triples))]
        b.close()
        return out

    fixed = decode()
    monkeypatch.setenv("CCD_FIXED_SHAPE", "0")
    dyn = decode()
    for n, f, d in zip(names, fixed, dyn):
        want_fixed = n == "kodim14"   # only hop (14 + 6 inputs) has a compile-time instantiation
        assert bool(f[2] & 32) == want_fixed, (n, f[2])
        assert not d[2] & 32, (n, d[2])
        for a, b_ in zip(f[0], d[0]):
            assert np.array_equal(a, b_), n
        for a, b_ in zip(f[1], d[1]):
            assert np.array_equal(a, b_), n
        z = load_golden(n)[1]
        for g, a in enumerate(f[0]):
            assert np.array_equal(a, z[f"cc0.latent{g}"]), f"{n} grid {g} differs from the reference decoder's"


def test_overlapped_run_equals_staged_run(gpu, oracle):
    """ccd_batch_run overlaps the float path of the streams that finish early with the longest entropy chains (r06: chain groups -
    the slots of a kernel instantiation split by expected chain length, each group's pyramid + fused launches on the group's own
    stream, one join).  Same bits as the float stages behind the join (overlap=False), as three ccd_batch_run_stage calls, and
    after switching the option on a live batch - for a batch that mixes picture sizes (three chain groups), kernel
    instantiations, common randomness (its fused launch stays behind the join) and a final resize; prepare() on one stream and
    run() on another (the run orders itself behind the tables' copy)."""
    import hashlib

    import torch

    from cool_chic_amd import writer

    mixed = []  # several kernel instantiations (each its own launch), common randomness, a final resize
    for name in ["kodim14", "rgb192", "cr192", "bicubic190", "kodim14", "mop192", "odd191x127", "rgb192"]:
        bs, _, _ = load_golden(name)
        fh, ccs = oracle.split_stream(bs)[1][0]
        mixed.append((ccs[0], fh.bitdepth, fh.frame_data_type))
    # ONE instantiation, three chain groups: kodim14 twice, a slightly smaller and a half-size picture of the same network
    # (expected chains within 3 % / within 20 % / below 80 % of the longest)
    bs, _, _ = load_golden("kodim14")
    (fh, ccs), = oracle.split_stream(bs)[1]
    donor = writer.parse_cc_header(ccs[0][0])
    b0 = _decode(gpu, ccs[:1], fh.bitdepth, fh.frame_data_type)
    lats = [b0.latent(0, g) for g in range(donor.n_grids)]
    b0.close()
    sized = [(ccs[0], fh.bitdepth, fh.frame_data_type)] * 2
    for (h, w) in [(480, 704), (256, 384)]:
        arch = writer.derive_arch(donor, img_size=(h, w))
        st = writer.encode_stream(writer.cc_header_bytes(arch), ccs[0][1], [np.ascontiguousarray(a[: arch.grid_h[g], : arch.grid_w[g]]) for g, a in enumerate(lats)],
                                  bitdepth=fh.bitdepth, frame_data_type=fh.frame_data_type)
        sized.append((oracle.split_stream(st)[1][0][1][0], fh.bitdepth, fh.frame_data_type))
    from cool_chic_amd._lib import lib
    n_conc = lib().ccd_concurrent_streams(0)
    assert 1 <= n_conc <= 8  # (4 is the most the measurement looks for; CCD_SIDE_STREAMS=k skips it and takes k)
    _overlap_checks(gpu, sized, min(3, n_conc))   # one instantiation in three chain groups (as many as streams really run at once)
    _overlap_checks(gpu, mixed, None)


def _overlap_checks(gpu, triples, want_launches):
    import hashlib

    import torch

    def digest(b):
        h = hashlib.sha256()
        for s in range(len(triples)):
            for g in range(b.header(s).n_grids):
                h.update(np.ascontiguousarray(b.latent(s, g)).tobytes())
            for p in b.planes(s):
                h.update(np.ascontiguousarray(p).tobytes())
            h.update(np.ascontiguousarray(b.output(s)).tobytes())
        return h.hexdigest()

    def fresh(**opts):
        b = gpu(0, **opts)
        for (hdr, nn, lat), bd, fdt in triples:
            b.add(hdr, nn, lat, bd, fdt)
        return b

    ref = None
    b = fresh(overlap=False)
    try:
        b.run(); b.wait()
        ref = digest(b)
        n_inst = b.entropy_launches()
        assert want_launches is None or n_inst == 1
        for stage in range(3):
            b.run(stage=stage)
        b.wait()
        assert digest(b) == ref, "staged run differs"
    finally:
        b.close()
    b = fresh(overlap=True)  # (explicit: CCD_OVERLAP=0 in the environment would switch the default off)
    try:
        for it in range(6):
            b.run(); b.wait()
            assert digest(b) == ref, f"overlapped run {it} differs from the float stages behind the join"
        assert b.entropy_launches() == (want_launches or n_inst), (b.entropy_launches(), n_inst)
        for stage in range(3):  # the staged entry points on the tables of an overlapping batch (more, smaller launches)
            b.run(stage=stage)
        b.wait()
        assert digest(b) == ref
        from cool_chic_amd._lib import check, lib
        check(lib().ccd_batch_set_option(b._h, b.OPT_OVERLAP, 0), "set_option")  # live switch: the tables are rebuilt
        b.run(); b.wait()
        assert digest(b) == ref
        check(lib().ccd_batch_set_option(b._h, b.OPT_OVERLAP, 1), "set_option")
        b.run(); b.wait()
        assert digest(b) == ref
    finally:
        b.close()
    # prepare on one stream, run on another: no explicit ordering by the caller
    s1, s2 = torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)
    b = fresh(overlap=True)
    try:
        b.prepare(s1.cuda_stream)
        b.run(s2.cuda_stream); b.wait(s2.cuda_stream)
        assert digest(b) == ref
    finally:
        b.close()


def test_fuzzed_streams_never_hang_and_match_the_oracle(gpu, oracle):
    """Seeded mutations of real streams - bit flips in the latent payload, in the network payload and in the cool-chic
    header, truncated payloads - decoded by the device path and by the oracle: same verdict (both reject, or both
    decode to the same latents) and never a hang (bounded spins in the kernel, test timeout in CI)."""
    from cool_chic_amd import CcdError

    rng = np.random.default_rng(99)
    cases = []
    for name in ["rgb192", "yuv420_8b", "kodim14"]:
        bs, _, _ = load_golden(name)
        hdr, nn, lat = oracle.split_stream(bs)[1][0][1][0]
        n_mut = 6 if name != "kodim14" else 2
        for _ in range(n_mut):  # payload bit flips
            bad = bytearray(lat)
            for pos in rng.integers(8, len(bad), size=int(rng.integers(1, 6))):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            cases.append((hdr, nn, bytes(bad), False))
        for _ in range(n_mut):  # network bit flips (weights change: decoding stays well defined)
            badn = bytearray(nn)
            badn[int(rng.integers(len(badn) // 2, len(badn)))] ^= 1 << int(rng.integers(0, 8))
            cases.append((hdr, bytes(badn), lat, True))
        for _ in range(n_mut):  # ... several flips at once in the upsampling / synthesis half: larger damage to the float stages
            badn = bytearray(nn)
            for pos in rng.integers(len(badn) // 2, len(badn), size=int(rng.integers(2, 9))):
                badn[pos] ^= 1 << int(rng.integers(0, 8))
            cases.append((hdr, bytes(badn), lat, True))
        cases.append((hdr, nn, lat[: 4 * (len(lat) // 8)], False))  # truncated payload: the coder reads zeros past the end
    cases = [c if len(c) == 4 else (*c, False) for c in cases]
    n_rejected = n_float = n_outside = 0
    for hdr, nn, lat, all_stages in cases:
        # a damaged NETWORK payload goes through every stage (r05): the float stages must equal the oracle's bit for bit too
        # - whatever the flipped weights make of them, incl. overflow (the finite envelope then selects the vector-ALU kernels)
        try:
            ref = oracle.decode_coolchic(hdr, nn, lat, stop_after_entropy=not all_stages)
        except oracle.OracleError:
            ref = None
        b = gpu(0)
        try:
            try:
                b.add(hdr, nn, lat, 8 if all_stages else 0, 0)
            except CcdError:
                assert ref is None, "the device path rejected a stream the oracle decodes"
                n_rejected += 1
                continue
            if all_stages:
                b.run()
            else:
                b.run(stage=0)
            if ref is None:
                with pytest.raises(CcdError):
                    b.wait()
                n_rejected += 1
            else:
                b.wait()
                for g in range(ref["n_grids"]):
                    assert np.array_equal(b.latent(0, g), ref["latent"][g])
                if all_stages and ref["out"].shape[0] >= 3:
                    ok, n_bad = nan_aware_equal(b.output(0), ref["out"])
                    assert ok, f"{n_bad} words of the synthesis output differ from the oracle's"
                    n_float += 1
                    n_outside += bool(b.slot_kernels(0) & 128)
        finally:
            b.close()
    assert n_rejected < len(cases)  # most mutations still decode (to different symbols): both paths must agree on them
    assert n_float >= 10


def test_float_stage_operand_classes(gpu, oracle):
    """The float stages on what a hostile network payload can produce (tests/float_classes.py: overflow to inf, inf - inf = NaN,
    subnormal values, products that underflow, signed zeros, the extremes of the format's quantisation steps, gains stepping up
    to FLT_MAX): synthesis output == the oracle's bit for bit (two NaNs are equal), integer planes equal wherever the output is
    not NaN.  Networks inside the finite envelope run the matrix-core kernel in both its forms AND the vector-ALU path; the
    others must have been routed to the vector-ALU path by the host (slot_kernels bit 7 set, bit 2 clear)."""
    from float_classes import crafted

    cases = crafted(load_golden, oracle)
    seen = {"nan": 0, "inf": 0, "subnormal_dense": 0}
    for name, ((hdr, nn, lat), stream, inside) in cases.items():
        ref = oracle.decode_coolchic(hdr, nn, lat)
        want_planes = oracle.decode_video(stream)[0]["planes"]
        nan_px = np.isnan(ref["out"]).any(axis=0)
        seen["nan"] += int(np.isnan(ref["out"]).any())
        seen["inf"] += int(np.isinf(ref["out"]).any())
        u = ref["dense"].view(np.uint32)
        seen["subnormal_dense"] += int((((u & 0x7F800000) == 0) & ((u & 0x007FFFFF) != 0)).any())
        for mode in (2, 1, 0):
            b = gpu(0, fused_dec=mode)
            try:
                b.add(hdr, nn, lat, 8, 0)
                b.run(); b.wait()
                k = b.slot_kernels(0)
                if inside is not None:
                    assert bool(k & 128) == (not inside), name
                if k & 128 or mode == 0:
                    assert not k & 4, f"{name}: a network outside the finite envelope on the matrix-core kernel"
                elif inside:
                    assert k & 4, name
                for g in range(ref["n_grids"]):
                    assert np.array_equal(b.latent(0, g), ref["latent"][g]), name
                ok, n_bad = nan_aware_equal(b.output(0), ref["out"])
                assert ok, f"{name} (fused_dec={mode}): {n_bad} words of the synthesis output differ from the oracle's"
                if not k & 4:
                    ok, n_bad = nan_aware_equal(b.dense(0), ref["dense"])
                    assert ok, f"{name}: {n_bad} words of the dense pyramid differ"
                for p, w in zip(b.planes(0), want_planes):
                    assert np.array_equal(p[~nan_px], w[~nan_px].astype(p.dtype)), name
            finally:
                b.close()
    assert seen["nan"] and seen["inf"] and seen["subnormal_dense"], seen


def test_rate_model_matches_the_reference(gpu):
    """compute_rate (arm.py:448-485) on the device vs what the REFERENCE's compute_rate returned for 2^16 symbols
    (tests/golden/rate.npz, written by tests/golden/gen/dump_rate.py: symbols at the mode, half-way, tails on the 2^-16 clamp, the
    smallest and the largest scale of the format's table), and vs the same float32 formula in plain PyTorch on 2^20 more.
    Tolerance per symbol: 2e-6 relative + 2e-6 bits + 3e-7 / p bits, p = the symbol's probability.  The last term is the
    formula's own float32 conditioning: p is a difference of two CDF values near 0.5 .. 1, so a few-ulp difference between
    two expm1 implementations (6e-8 each) moves p by ~1e-7 absolute, i.e. the rate by 1e-7 / (p ln 2) bits - 0.02 bits at
    the 2^-16 clamp, 1e-6 bits for likely symbols.  Totals within 1e-5 relative.  Unaligned views and lengths that are not a
    multiple of four take the kernel's scalar tail."""
    import os
    import re

    import torch

    from conftest import GOLDEN, ROOT
    from cool_chic_amd.component.core.arm import compute_rate, total_rate_bits

    def check(got, want):
        p = torch.exp2(-want.double())
        tol = 2e-6 * want.double().abs() + 2e-6 + 3e-7 / p
        err = (got.double() - want.double()).abs()
        assert bool((err <= tol).all()), f"worst {float((err / tol).max()):.2f} x the tolerance"

    # (1) the reference's own numbers
    z = np.load(os.path.join(GOLDEN, "rate.npz"))
    text = open(os.path.join(ROOT, "include", "ccd_scale_table.inc")).read()
    tab = np.array([int(t, 16) for t in re.findall(r"0x([0-9a-f]{8})u", text)], dtype=np.uint32).view(np.float32)
    x = torch.from_numpy(z["x"].astype(np.float32)).cuda()
    mu = torch.from_numpy((z["mu_idx"].astype(np.float64) / 256 - 64).astype(np.float32)).cuda()
    scale = torch.from_numpy(tab[z["scale_idx"].astype(np.int64)]).cuda()
    want = torch.from_numpy(z["rate"])
    got = compute_rate(x, mu, scale).cpu()
    check(got, want)
    # the clamp gives exactly 16 bits on both sides (a symbol whose p sits within float32 noise of 2^-16 may fall on either side)
    assert int((want == 16.0).sum()) > 100 and abs(int((got == 16.0).sum()) - int((want == 16.0).sum())) <= 20
    assert float(got.max()) <= 16.0 and float(got.min()) >= 0.0
    assert abs(total_rate_bits(x, mu, scale) - float(want.double().sum())) <= 1e-5 * float(want.double().sum())
    # unaligned views / ragged lengths: the scalar tail kernel
    for off, cnt in ((1, 1001), (3, 4098), (0, 7), (2, 65531)):
        g2 = compute_rate(x[off:off + cnt], mu[off:off + cnt], scale[off:off + cnt]).cpu()
        check(g2, want[off:off + cnt])
    # (2) 2^20 more symbols against the same float32 formula in PyTorch on the CPU
    g = torch.Generator().manual_seed(11)
    n = 1 << 20
    x = torch.randint(-64, 64, (n,), generator=g).float()
    mu = (torch.rand(n, generator=g) - 0.5) * 40
    scale = torch.exp((torch.rand(n, generator=g) - 0.5) * 10)
    x[:1000] = mu[:1000].round()          # likely symbols
    x[1000:2000] = mu[1000:2000] + 60     # clamped at 16 bits

    def ref(x, mu, scale):
        def cdf(t):
            d = t - mu
            return 0.5 - 0.5 * d.sign() * torch.expm1(-d.abs() / scale)
        return -torch.log2(torch.clamp_min(cdf(x + 0.5) - cdf(x - 0.5), 2 ** -16))

    want = ref(x, mu, scale)
    got = compute_rate(x.cuda(), mu.cuda(), scale.cuda()).cpu()
    check(got, want)
    tot = total_rate_bits(x.cuda(), mu.cuda(), scale.cuda())
    assert abs(tot - float(want.double().sum())) <= 1e-5 * float(want.double().sum())
    shaped = compute_rate(x.view(1, 1, 1024, 1024).cuda(), mu.view(1, 1, 1024, 1024).cuda(), scale.view(1, 1, 1024, 1024).cuda())
    assert shaped.shape == (1, 1, 1024, 1024)
    with pytest.raises(ValueError):
        compute_rate(x, mu, scale)  # host tensors


def _sharded_worker(rank, world, port, path, q, backend="gloo"):
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # gloo: one GPU on the box, both ranks use cuda:0 and the exchange is staged through the host;
    # nccl: one GPU per rank, planes travel over RCCL send / recv
    device = rank if backend == "nccl" else 0
    if backend == "nccl":
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{device}"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from cool_chic_amd.bitstream.decode import decode_video_sharded
    from cool_chic_amd.parallel import EqualSizeGather

    frames = decode_video_sharded(path, device=device)
    out = {}
    for k, fd in frames.items():
        planes = fd.integer_planes()
        out[k] = [np.asarray(p) for p in planes]
    # the fixed-size gather of bench.py --gpus N on the same backend
    dev = f"cuda:{device}" if backend == "nccl" else "cpu"
    mine = [torch.full((16, 8), 10 * rank + p, dtype=torch.uint8, device=dev) for p in range(3)]
    got = EqualSizeGather(3 * 128, dev, dst=0)(mine)
    gather_ok = True
    if rank == 0:
        for r in range(world):
            want = torch.cat([torch.full((128,), 10 * r + p, dtype=torch.uint8) for p in range(3)])
            gather_ok = gather_ok and torch.equal(got[r].cpu(), want)
    q.put((rank, out, gather_ok))
    dist.barrier()
    dist.destroy_process_group()


def _run_sharded(backend, oracle):
    import os
    import socket

    import torch.multiprocessing as mp

    from conftest import GOLDEN

    path = os.path.join(GOLDEN, "vid5.cool")
    bs, _, _ = load_golden("vid5")
    want = {str(fr["display_index"]): fr["planes"] for fr in oracle.decode_video(bs)}
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, path, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, out, gather_ok in results:
        assert gather_ok
        # rank 0 collects the whole sequence; rank 1 holds its own frames and the references it was sent
        if rank == 0:
            assert sorted(out) == sorted(want)
        else:
            assert 0 < len(out) <= len(want)
        for k in out:
            for p, w in zip(out[k], want[k]):
                assert np.array_equal(p.astype(np.uint16), w), (rank, k)
    return want


def test_video_gop_sharded_over_ranks(gpu, oracle):
    """The I/P/B fixture decoded (a) unsharded through decode_video_sharded, (b) by two ranks that each decode the
    cool-chics of their own frames and hand reconstructed planes point to point to the ranks that predict from them:
    both equal the oracle's planes."""
    import os

    from conftest import GOLDEN
    from cool_chic_amd.bitstream.decode import decode_video, decode_video_sharded

    path = os.path.join(GOLDEN, "vid5.cool")
    want = _run_sharded("gloo", oracle)
    single = decode_video_sharded(path, device=0)
    assert sorted(single) == sorted(want)
    for k, fd in single.items():
        for p, w in zip(fd.integer_planes(), want[k]):
            assert np.array_equal(np.asarray(p).astype(np.uint16), w), k
    # decode_video itself: all cool-chics of the GOP in one batch; a partial decode returns None for the frames that
    # were not reached (decode.py:84-89 iterates every display index)
    full = decode_video(path, device=0)
    for k, fd in full.items():
        for p, w in zip(fd.integer_planes(), want[k]):
            assert np.array_equal(np.asarray(p).astype(np.uint16), w), k
    part = decode_video(path, max_decoding_order=1, device=0)  # coding order I0 P4 ...
    assert sorted(part) == sorted(want)
    decoded = {k for k, fd in part.items() if fd is not None}
    assert decoded == {"0", "4"}
    for k in decoded:
        for p, w in zip(part[k].integer_planes(), want[k]):
            assert np.array_equal(np.asarray(p).astype(np.uint16), w), k


def _n_gpus():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: RCCL send / recv and gather between two ranks")
def test_video_gop_sharded_over_two_gpus_nccl(gpu, oracle):
    """The same over the "nccl" backend (RCCL over xGMI), one GPU per rank: the sharded GOP's point-to-point plane
    hand-over and bench.py's EqualSizeGather."""
    _run_sharded("nccl", oracle)


def _nccl_one_rank_worker(port, golden, q):
    """Runs in a spawned process: everything of the N > 1 path that a single rank can execute on the "nccl" backend."""
    import os
    import sys
    import traceback

    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        sys.path.insert(0, os.path.dirname(os.path.dirname(golden)))
        import bench
        from cool_chic_amd import DecodeBatch, synth
        from cool_chic_amd.bitstream.decode import decode_video_sharded
        from cool_chic_amd.parallel import EqualSizeGather, gather_bytes, pack_planes
        from oracle import oracle_py

        res = {"backend": dist.get_backend()}
        # (1) scalar reductions on cuda tensors, as bench.py does with red_dev = cuda
        t = torch.tensor([7], dtype=torch.int64, device="cuda:0")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        td = torch.tensor([0.25], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        res["all_reduce"] = (int(t.item()), float(td.item()))
        # (2) 8-bit planes of a decoded frame, zero-copy views of library memory, through EqualSizeGather: the wire is the device tensor
        with open(os.path.join(golden, "rgb192.cool"), "rb") as f:
            bs8 = f.read()
        fh, ccs = oracle_py.split_stream(bs8)[1][0]
        b = DecodeBatch(0)
        b.add(*ccs[0], fh.bitdepth, fh.frame_data_type)
        b.run(); b.wait()
        planes = [torch.as_tensor(b.plane_device(0, p), device="cuda:0").reshape(-1) for p in range(3)]
        g = EqualSizeGather(sum(int(p.numel()) for p in planes), "cuda:0", dst=0)
        res["staged"] = g.staged
        got = g(planes)
        res["wire_is_cuda"] = bool(got[0].is_cuda)
        res["gather8_ok"] = bool(torch.equal(got[0].cpu(), torch.cat([torch.from_numpy(x.reshape(-1)) for x in b.planes(0)])))
        b.close()
        # (3) 16-bit planes (10-bit 4:2:0 frame): view(torch.uint8) of contiguous uint16 device tensors, variable-size gather
        with open(os.path.join(golden, "yuv420_10b.cool"), "rb") as f:
            bs10 = f.read()
        fh, ccs = oracle_py.split_stream(bs10)[1][0]
        b = DecodeBatch(0)
        b.add(*ccs[0], fh.bitdepth, fh.frame_data_type)
        b.run(); b.wait()
        p16 = [torch.as_tensor(b.plane_device(0, p), device="cuda:0") for p in range(3)]
        res["dtype16"] = str(p16[0].dtype)
        msg = pack_planes(p16)
        back = gather_bytes(msg, dst=0)
        host = [x.astype("<u2").tobytes() for x in b.planes(0)]
        res["gather16_ok"] = bool(back[0].cpu().numpy().tobytes() == b"".join(host))
        g16 = EqualSizeGather(int(msg.numel()), "cuda:0", dst=0)
        res["gather16_fixed_ok"] = bool(g16(p16)[0].cpu().numpy().tobytes() == b"".join(host))
        b.close()
        # (4) the sharded GOP entry point with an initialised nccl group (one rank owns every frame: no send / recv)
        # r06: owner == every consumer, so run_sharded_gop must not issue a single point-to-point call (counted)
        p2p = {"send": 0, "recv": 0}
        real_send, real_recv = dist.send, dist.recv
        def _count_send(*a, **k):
            p2p["send"] += 1
            return real_send(*a, **k)
        def _count_recv(*a, **k):
            p2p["recv"] += 1
            return real_recv(*a, **k)
        dist.send, dist.recv = _count_send, _count_recv
        try:
            frames = decode_video_sharded(os.path.join(golden, "vid5.cool"), device=0)
        finally:
            dist.send, dist.recv = real_send, real_recv
        res["gop_p2p_calls"] = dict(p2p)
        res["gop"] = {k: [np.asarray(p) for p in fd.integer_planes()] for k, fd in frames.items()}
        # (5) bench.py's timed step with the gather inside, exactly as the driver's N > 1 run enqueues it
        streams, sizes = synth.kodak24()
        items = [(*synth.split_image_stream(s_), hw) for s_, hw in zip(streams[:3], sizes[:3])]
        run = bench.timed_set(items, 1, 0, 0, "nccl", "cuda:0", 2, 1, force_gather=True)
        ver = bench.verify_gathered("kodak24", run["gathered"], lambda r: [(i, items[i][3]) for i in range(3)])
        run["batch"].close()
        res["bench_gather"] = {"ok": ver["ok"], "frames": ver["frames_checked"], "dt": run["dt"]}
        # (6) ... and bench.py's from-bytes step: two sets in flight on two side streams, the gather enqueued on them
        fb = bench.timed_from_bytes([(streams[i], sizes[i]) for i in range(3)], 1, 0, 0, "nccl", "cuda:0", 3, 1, force_gather=True)
        ver = bench.verify_gathered("kodak24", fb["gathered"], lambda r: [(i, sizes[i]) for i in range(3)])
        res["bench_from_bytes"] = {"ok": ver["ok"], "frames": ver["frames_checked"], "dt": fb["dt"]}
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", res))
    except Exception:  # noqa: BLE001 - reported to the parent
        q.put(("error", traceback.format_exc()))


def test_nccl_wire_path_on_one_rank(gpu, oracle):
    """RCCL with world_size = 1 on cuda:0: it cannot prove xGMI, it does execute every `staged == False` branch of
    cool_chic_amd/parallel.py (EqualSizeGather, gather_bytes with device tensors on the wire, 16-bit planes as
    view(torch.uint8)), the cuda-scalar reductions and bench.py's timed step with the gather of decoded planes inside - the
    code a single-GPU box never ran before.  Point-to-point send / recv needs a second rank:
    test_video_gop_sharded_over_two_gpus_nccl."""
    import socket

    import torch.multiprocessing as mp

    from conftest import GOLDEN

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_nccl_one_rank_worker, args=(port, GOLDEN, q))
    proc.start()
    status, res = q.get(timeout=600)
    proc.join(timeout=120)
    assert status == "ok", res
    assert proc.exitcode == 0
    assert res["backend"] == "nccl" and res["all_reduce"] == (7, 0.25)
    assert res["staged"] is False and res["wire_is_cuda"] and res["gather8_ok"]
    assert res["dtype16"] == "torch.uint16" and res["gather16_ok"] and res["gather16_fixed_ok"]
    bs, _, _ = load_golden("vid5")
    want = {str(fr["display_index"]): fr["planes"] for fr in oracle.decode_video(bs)}
    assert res["gop_p2p_calls"] == {"send": 0, "recv": 0}, res["gop_p2p_calls"]
    assert sorted(res["gop"]) == sorted(want)
    for k in want:
        for p, w in zip(res["gop"][k], want[k]):
            assert np.array_equal(p.astype(np.uint16), w), k
    assert res["bench_gather"]["ok"] is True and res["bench_gather"]["frames"] == 3
    assert res["bench_from_bytes"]["ok"] is True and res["bench_from_bytes"]["frames"] == 3


def test_prepare_then_run_behind_a_stream_wait(gpu, oracle):
    """ccd_batch_prepare uploads the launch tables without launching; a run behind it (on a stream that first waits for other
    work, as bench.py's two sets in flight do) gives the planes of a plain run, and slots added after a prepare are picked up."""
    import torch

    bs, _, _ = load_golden("rgb192")
    fh, ccs = oracle.split_stream(bs)[1][0]
    want = oracle.decode_video(bs)[0]["planes"]
    side = torch.cuda.Stream(device="cuda:0")
    ev = torch.cuda.Event()
    b = gpu(0, keep_float=False)
    try:
        b.add(*ccs[0], fh.bitdepth, fh.frame_data_type)
        b.prepare(side.cuda_stream)
        torch.cuda._sleep(2_000_000)          # work on the default stream the side stream has to wait for
        ev.record(torch.cuda.current_stream(0))
        side.wait_event(ev)
        b.run(side.cuda_stream)
        b.wait(side.cuda_stream)
        for p, w in zip(b.planes(0), want):
            assert np.array_equal(p.astype(np.uint16), w)
        b.add(*ccs[0], fh.bitdepth, fh.frame_data_type)  # a slot behind the prepare: the next run uploads its tables itself
        b.run(side.cuda_stream)
        b.wait(side.cuda_stream)
        for p, w in zip(b.planes(1), want):
            assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()


def test_streams_the_reference_cannot_decode_are_rejected(gpu, oracle):
    """Headers that parse but that the reference's decoder raises on (or that would read out of bounds here) give
    CCD_ERR_VALUE instead of garbage: latent / hyperlatent ranges that do not touch (torch.cat of grids two levels
    apart), a P / B frame whose reference has another sample
    layout, a frame type that contradicts the coding structure, odd-sized 4:2:0 frames."""
    import ctypes as C

    from cool_chic_amd import writer
    from cool_chic_amd._lib import CcdError, Video, lib

    bs, z, _ = load_golden("rgb192")
    fh, ccs = oracle.split_stream(bs)[1][0]
    hdr, nn, lat = ccs[0]
    arch = writer.parse_cc_header(hdr)
    # (1) latent levels 0..1 and hyperlatent levels 4..6: consecutive grids three levels apart
    bad = writer.derive_arch(arch)
    bad.latent_resolution[0], bad.latent_resolution[1] = 0, 1
    bad.hyperlatent_resolution[0], bad.hyperlatent_resolution[1] = 4, 6
    bad.n_latent_grids = 5
    b = gpu(0)
    try:
        with pytest.raises(CcdError) as e:
            b.add(writer.cc_header_bytes(bad), nn, lat, 8, 0)
        assert e.value.code == -2
        # (2) a transmitted grid count that is not the length of the size list: the reference recomputes the count from the
        # resolutions and never reads the field (header.py:354-377, component/core/coolchic.py:185), so such a stream
        # decodes, and decodes like the untouched one
        odd = writer.derive_arch(arch)
        odd.n_latent_grids = arch.n_latent_grids - 1
        assert b.add(writer.cc_header_bytes(odd), nn, lat, 8, 0) == 0
        assert b.add(hdr, nn, lat, 8, 0) == 1
        b.run(); b.wait()
        for p0, p1 in zip(b.planes(0), b.planes(1)):
            assert np.array_equal(p0, p1)
    finally:
        b.close()
    # (3)-(5) video level, through ccd_decode_video
    vbs, _, _ = load_golden("vid5")
    vh, frames = oracle.split_stream(vbs)

    def rebuild(edit):
        out = [writer.video_header_bytes(vh.n_frames, list(vh.intra_pos[:vh.n_intras]), list(vh.p_pos[:vh.n_p_frames]))]
        for k, (f, cc) in enumerate(frames):
            di, fdt = f.display_index, f.frame_data_type
            di, fdt = edit(k, di, fdt)
            out.append(writer.frame_header_bytes(di, "IPB"[f.frame_type], fdt, f.bitdepth, list(f.index_references[:f.n_refs]),
                                                 list(f.global_flow[:2 * f.n_refs]), f.warp_filter_size))
            for h_, n_, l_ in cc:
                out.append(h_ + n_ + l_)
        return b"".join(out)

    def decode(stream, planes=None):
        v = Video()
        rc = lib().ccd_decode_video(stream, len(stream), 0, C.byref(v))
        if rc == 0:
            if planes is not None:
                for i in range(v.n_frames):
                    fr = v.frames[i]
                    planes.append([np.ctypeslib.as_array(fr.plane[p], shape=((fr.h, fr.w) if p == 0 else (fr.ch, fr.cw))).copy() for p in range(3)])
            lib().ccd_video_free(C.byref(v))
        return rc

    plain, odd = [], []
    assert decode(rebuild(lambda k, di, fdt: (di, fdt)), plain) == 0
    assert decode(rebuild(lambda k, di, fdt: (di, 2 if k == 0 else fdt))) == -2   # I frame yuv444, its P / B users yuv420
    # decode.py:67-75 never reads a frame header's display_index (nor its index_references): order and references are the
    # coding structure's, so a stream whose frame header carries another display index decodes like the untouched one
    assert decode(rebuild(lambda k, di, fdt: (0 if k == 4 else di, fdt)), odd) == 0
    assert len(plain) == len(odd) == 5
    for fa, fb in zip(plain, odd):
        for pa, pb in zip(fa, fb):
            assert np.array_equal(pa, pb)
    # a frame TYPE that contradicts the structure is what the reference cannot decode (its reconstruction indexes references
    # it was not given): frame 1 in coding order is the P frame of vid5, relabelled B with a second reference
    def retype(k, f):
        return ("B", [0, 0], [0, 0, 0, 0]) if k == 1 else ("IPB"[f.frame_type], list(f.index_references[:f.n_refs]), list(f.global_flow[:2 * f.n_refs]))
    out = [writer.video_header_bytes(vh.n_frames, list(vh.intra_pos[:vh.n_intras]), list(vh.p_pos[:vh.n_p_frames]))]
    for k, (f, cc) in enumerate(frames):
        t, refs, gf = retype(k, f)
        out.append(writer.frame_header_bytes(f.display_index, t, f.frame_data_type, f.bitdepth, refs, gf, f.warp_filter_size))
        for h_, n_, l_ in cc:
            out.append(h_ + n_ + l_)
    assert decode(b"".join(out)) == -2
    # ... but a header type that needs FEWER references than the structure gives decodes, by the header's type (decode.py:119-128,
    # 156-189; apply_global_translation zips references with flows): the B frame at coding index 2 relabelled P predicts from
    # the structure's first reference; the P frame at coding index 1 replaced by an I header + the I frame's cool-chic is plain
    # intra (and the B frames then predict from that).  The oracle follows the frame headers, which here agree with the structure.
    def relabel(kind):
        out = [writer.video_header_bytes(vh.n_frames, list(vh.intra_pos[:vh.n_intras]), list(vh.p_pos[:vh.n_p_frames]))]
        for k, (f, cc) in enumerate(frames):
            t, refs, gf, body = "IPB"[f.frame_type], list(f.index_references[:f.n_refs]), list(f.global_flow[:2 * f.n_refs]), cc
            if kind == "B_as_P" and k == 2:
                assert t == "B"
                t, refs, gf = "P", refs[:1], gf[:2]
            if kind == "P_as_I" and k == 1:
                assert t == "P"
                t, refs, gf, body = "I", [], [], frames[0][1]
            out.append(writer.frame_header_bytes(f.display_index, t, f.frame_data_type, f.bitdepth, refs, gf, f.warp_filter_size))
            for h_, n_, l_ in body:
                out.append(h_ + n_ + l_)
        return b"".join(out)

    for kind in ("B_as_P", "P_as_I"):
        stream, got = relabel(kind), []
        assert decode(stream, got) == 0, kind
        want = oracle.decode_video(stream)
        assert len(got) == len(want) == 5
        for fa, fb in zip(got, want):
            for pa, pb in zip(fa, fb["planes"]):
                assert np.array_equal(pa, pb), kind
        assert any(not np.array_equal(a, b) for fa, fb in zip(got, plain) for a, b in zip(fa, fb)), kind  # the relabelling matters


@pytest.mark.parametrize("name", ["kodak24", "kodak24_wide_envelope", "kodak24_hq", "clic41", "clic41_alt", "uhd4k", "gop1080p33"])
def test_workloads_match_the_oracle(gpu, name):
    """EVERY stream of EVERY benchmark workload (cool_chic_amd/synth.py: BASELINE.json configs[1..4] at full size - 24
    Kodak frames, the 41 CLIC sizes, the 4K frame, the 33-frame depth-5 hierarchical 1080p GOP with its 64 cool-chics):
    the integer planes the GPU decodes equal the CPU oracle's, frame by frame.  The oracle's side was computed in the build
    container (tests/golden/gen/hash_workloads.py -> workload_hashes.json: 260 Mpx on one core per stream would take
    minutes here); the streams are re-manufactured here and must hash to what the oracle saw."""
    import ctypes as C
    import hashlib
    import json
    import os

    from cool_chic_amd import synth
    from cool_chic_amd._lib import Video, check, lib
    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "workload_hashes.json")) as f:
        exp = json.load(f)[name]
    wl = synth.workload(name)
    assert [hashlib.sha256(s).hexdigest() for s in wl["streams"]] == exp["streams_sha256"], "the manufactured streams differ"
    if wl["video"]:
        bs = wl["streams"][0]
        v = Video()
        check(lib().ccd_decode_video(bs, len(bs), 0, C.byref(v)), "ccd_decode_video")
        try:
            assert v.n_frames == len(exp["planes_sha256"][0]) == 33
            for i in range(v.n_frames):
                f = v.frames[i]
                planes = [np.ctypeslib.as_array(f.plane[p], shape=s) for p, s in enumerate([(f.h, f.w), (f.ch, f.cw), (f.ch, f.cw)])]
                assert synth.planes_sha256(planes) == exp["planes_sha256"][0][i], f"frame {i} (display order) vs oracle"
        finally:
            lib().ccd_video_free(C.byref(v))
        return
    b = _decode(gpu, [synth.split_image_stream(s) for s in wl["streams"]], 8, 0, keep_float=False)
    try:
        for i in range(len(wl["streams"])):
            assert b.slot_status(i) == 0
            assert b.slot_kernels(i) & 5 == 5, "benchmark streams run the pipelined entropy kernel and the fused float kernel"
            assert synth.planes_sha256(b.planes(i)) == exp["planes_sha256"][i][0], f"stream {i} ({wl['sizes'][i]}) vs oracle"
    finally:
        b.close()


def _run_bench_two_ranks(extra):
    import json
    import os
    import socket
    import subprocess
    import sys

    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_two_ranks_on_one_gpu_gloo():
    """bench.py --gpus 2 with its DEFAULTS (what the driver's scaling runs launch): BASELINE's kodak24 sharded round-robin over
    the ranks (strong), per-rank batch, gather of the planes to rank 0, max-over-ranks timing - and, beside the metric, the two
    collective legs `clic41_sharded` (BASELINE configs[2] round-robin) and `throughput_regime` (256 streams per rank) - executed
    with two ranks sharing this box's GPU and the host-staged "gloo" backend: every gathered set must hash to the oracle's
    planes.  (The RCCL transport itself needs one GPU per rank: the driver's scaling runs.)"""
    res = _run_bench_two_ranks([])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["steps"] == 2
    assert res["config"]["workload"] == "kodak24" and res["expected_scaling"].startswith("flat")
    assert res["verified"]["ok"] is True and res["verified"]["frames_checked"] == 12  # rank 0's half of the 24 frames
    g = res["verified"]["gathered"]
    assert g["ok"] is True and g["frames_checked"] == 24, g
    fb = res["from_bytes"]  # the same steps from the stream bytes, two sets in flight, every rank's planes in rank 0's pinned memory
    assert fb["verified"]["ok"] is True and fb["verified"]["frames_checked"] == 24 and fb["value"] > 0, fb
    c = res["clic41_sharded"]
    assert c["scaling"] == "strong" and c["frames"] == 41 and c["frames_on_rank0"] == 21
    assert c["verified"]["ok"] is True and c["verified"]["frames_checked"] == 41, c["verified"]
    t = res["throughput_regime"]
    assert t["scaling"] == "weak" and t["streams_per_gpu"] == 256
    assert t["verified"]["ok"] is True and t["verified"]["frames_checked"] == 512, t["verified"]


def test_bench_two_ranks_weak_scaling_gloo():
    """--scaling weak stays available: every rank its own copy of kodak24, all 48 gathered frames verified."""
    res = _run_bench_two_ranks(["--scaling", "weak", "--legs", "none"])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    assert res["verified"]["ok"] is True and res["verified"]["frames_checked"] == 24
    assert res["verified"]["gathered"]["ok"] is True and res["verified"]["gathered"]["frames_checked"] == 48


def test_pooled_blocks_are_recycled_and_results_stay_exact(gpu, oracle):
    """A batch per image set is the normal use: arenas and pinned staging blocks of destroyed batches serve the next one
    (BlockPool, ccd_api.cpp).  Thirty create / add / run / all_planes / destroy cycles over streams of different sizes and
    kinds, a failed decode (its blocks must come back too) and ccd_pool_trim in between: planes always equal the oracle's,
    device memory does not grow, and a batch may be grown and re-run after a first run."""
    import torch

    from cool_chic_amd._lib import CcdError, lib

    names = ["rgb192", "kodim14", "yuv420_8b", "bicubic190", "yuv444_10b", "cr192"]
    cases = []
    for name in names:
        bs, z, j = load_golden(name)
        fh, ccs = oracle.split_stream(bs)[1][0]
        cases.append((ccs[0], fh.bitdepth, fh.frame_data_type, oracle.decode_video(bs)[0]["planes"]))

    def cycle(sel):
        b = gpu(0, keep_float=False)
        try:
            for i in sel:
                b.add(*cases[i][0], cases[i][1], cases[i][2])
            b.run()
            for planes, i in zip(b.all_planes(), sel):
                for p, w in zip(planes, cases[i][3]):
                    assert np.array_equal(p.astype(np.uint16), w)
        finally:
            b.close()

    cycle([0, 1, 2])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(30):
        cycle([(k + d) % len(cases) for d in range(1 + k % 4)])
        if k == 10:  # a truncated payload: the slot fails, wait raises, everything still goes back to the pool
            b = gpu(0)
            hdr, nn, lat = cases[1][0]
            b.add(hdr, nn, lat[: len(lat) // 2 // 4 * 4], 8, 0)
            b.run()
            with pytest.raises(CcdError):
                b.wait()
            with pytest.raises(CcdError):
                b.planes(0)  # a failed slot's planes are never handed out
            b.close()
        if k == 20:
            lib().ccd_pool_trim(0)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 256 << 20, f"device memory grew by {(free0 - free1) >> 20} MiB over 30 batches"
    # a batch that grows after its first run: tables are rebuilt, earlier slots decode as before
    b = gpu(0)
    try:
        b.add(*cases[0][0], cases[0][1], cases[0][2])
        b.run(); b.wait()
        b.add(*cases[2][0], cases[2][1], cases[2][2])
        b.run(); b.wait()
        for slot, i in ((0, 0), (1, 2)):
            for p, w in zip(b.planes(slot), cases[i][3]):
                assert np.array_equal(p.astype(np.uint16), w)
    finally:
        b.close()
    lib().ccd_pool_trim(0)


def test_two_host_threads_two_batches_one_gpu(gpu, oracle):
    """Different batches may be driven by different host threads on one GPU (include/ccd.h, "Threading and global state"): the
    fork / join events of the entropy launches belong to the batch (r03 kept one set per device, so that thread A's side stream
    could wait on thread B's record).  Two threads, each on its own torch stream, run batches that need SEVERAL kernel
    instantiations (streams with 8, 14, 20 and 26 ARM inputs: the launches fork over the shared side streams) 12 times each,
    create / add / run / all_planes / destroy; every result must hash to the single-threaded one."""
    import hashlib
    import threading

    import torch

    names = ["rgb192", "mop192", "vhop192", "yuv444_10b", "kodim14"]
    triples = []
    for n in names:
        bs, _, _ = load_golden(n)
        fh, ccs = oracle.split_stream(bs)[1][0]
        triples.append((ccs[0], fh.bitdepth, fh.frame_data_type))

    def run_once(stream_handle):
        b = gpu(0)
        try:
            for (hdr, nn, lat), bd, fdt in triples:
                b.add(hdr, nn, lat, bd, fdt)
            b.run(stream_handle)
            planes = b.all_planes(stream_handle)
            h = hashlib.sha256()
            for fr in planes:
                for p in fr:
                    h.update(np.ascontiguousarray(p).tobytes())
            return h.hexdigest()
        finally:
            b.close()

    want = run_once(0)
    results, errors = {0: [], 1: []}, []

    def worker(k):
        try:
            st = torch.cuda.Stream(device=0)
            with torch.cuda.stream(st):
                for _ in range(12):
                    results[k].append(run_once(st.cuda_stream))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    assert all(len(v) == 12 for v in results.values())
    assert all(x == want for v in results.values() for x in v), "a concurrent batch decoded differently"


def test_pool_trim_returns_the_cache(gpu, oracle):
    """cool_chic_amd.pool_trim (ccd_pool_trim): the blocks destroyed batches left in the per-device cache go back to the runtime -
    free device memory grows by about what the batches used - and decoding afterwards still works."""
    import torch

    import cool_chic_amd

    bs, z, _ = load_golden("kodim14")
    fh, ccs = oracle.split_stream(bs)[1][0]
    for _ in range(2):
        b = _decode(gpu, [ccs[0]] * 8, fh.bitdepth, fh.frame_data_type)
        b.close()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    cool_chic_amd.pool_trim(0)
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 - free0 > 30 * 2 ** 20, (free0, free1)  # eight Kodak arenas of ~10 MB
    b = _decode(gpu, ccs[:1], fh.bitdepth, fh.frame_data_type)
    try:
        assert np.array_equal(b.latent(0, 0), z["cc0.latent0"])
    finally:
        b.close()
