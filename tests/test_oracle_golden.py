"""The CPU oracle against golden vectors captured from the reference decoder (tests/golden/gen).

Integer stages (NN integers, fixed-point ARM, every latent grid, (mu, scale) indices) must be
bit-exact; float stages are compared with the tolerances of SURVEY.md section 8c: the reference's
own output moves by +-1 LSB on ~1e-5 of the samples between thread counts."""
import numpy as np
import pytest

from conftest import IMAGE_STREAMS, VIDEO_STREAMS, load_golden, reference_planes


@pytest.mark.parametrize("name", IMAGE_STREAMS)
def test_integer_stages_bit_exact(oracle, name):
    bs, z, j = load_golden(name)
    _, frames = oracle.split_stream(bs)
    hdr, nn, lat = frames[0][1][0]
    r = oracle.decode_coolchic(hdr, nn, lat, stop_after_entropy=True)
    cc = j["cc"][0]
    assert r["grid_hw"] == [tuple(s) for s in cc["size_per_latent"]]
    assert r["is_hyper"] == cc["flag_is_hyperlatent"]
    assert r["input_features_ifce"] == cc["input_features_ifce"]
    assert np.array_equal(r["nn_ints"], z["cc0.nn_ints"])
    n_layers = len(r["arm_w"])
    for l in range(n_layers):
        assert np.array_equal(r["arm_w"][l], z[f"cc0.fp.arm.w{l}"])
        assert np.array_equal(r["arm_b"][l], z[f"cc0.fp.arm.b{l}"])
    assert np.array_equal(r["arm_ws"], z["cc0.fp.arm.ws"])
    assert np.array_equal(r["arm_bs"], z["cc0.fp.arm.bs"])
    for g in range(r["n_grids"]):
        assert np.array_equal(r["latent"][g], z[f"cc0.latent{g}"]), f"latent grid {g}"
        # the fixture holds the indices as the ARM produced them; RangeCoder.decode clips them into the tables
        # (`take(..., mode="clip")`, rangecoder.py:98-99: odd18x65 has a log-scale index of -128), the oracle reports them clipped
        head = np.clip(z[f"cc0.mu_scale_idx{g}.head"], 0, [32767, 2560])
        assert np.array_equal(r["mu_scale_idx"][g][: len(head)], head), f"(mu, scale) indices grid {g}"
        key = f"cc0.ctx_ifce{g}"
        if key in z.files:
            assert np.array_equal(r["ctx_ifce"][g], z[key]), f"IFCE features grid {g}"
        elif key + ".crop" in z.files:
            assert np.array_equal(r["ctx_ifce"][g][:, :32, :48], z[key + ".crop"])
    assert r["n_symbols"] == sum(h * w for h, w in r["grid_hw"])
    # the decoder consumed the payload exactly (one read past the end on the last renormalisation at most)
    assert len(lat) // 4 <= r["words_consumed"] <= len(lat) // 4 + 1


@pytest.mark.parametrize("name", IMAGE_STREAMS)
def test_float_stages_close_to_reference(oracle, name):
    bs, z, j = load_golden(name)
    _, frames = oracle.split_stream(bs)
    r = oracle.decode_coolchic(*frames[0][1][0])
    d, s = r["dense"], r["syn_out"]
    H, W = d.shape[1:]
    ch, cw = min(H, 40), min(W, 48)
    for arr, key in ((d, "dense"), (s, "syn")):
        assert np.abs(arr[:, :ch, :cw] - z[f"cc0.{key}.tl"]).max() < 2e-5
        assert np.abs(arr[:, H - ch:, W - cw:] - z[f"cc0.{key}.br"]).max() < 2e-5
        assert np.abs(arr[:, H // 2:H // 2 + ch, W // 2:W // 2 + cw] - z[f"cc0.{key}.mid"]).max() < 2e-5


@pytest.mark.parametrize("name", IMAGE_STREAMS)
def test_integer_planes_vs_reference(oracle, name):
    bs, z, j = load_golden(name)
    fr = oracle.decode_video(bs)[0]
    ref = reference_planes(z, j)
    assert fr["bitdepth"] == j["frames"]["0"]["bitdepth"]
    assert fr["frame_data_type"] == j["frames"]["0"]["frame_data_type"]
    n_diff = n_tot = 0
    for a, b in zip(fr["planes"], ref):
        assert a.shape == b.shape
        diff = np.abs(a.astype(np.int64) - b.astype(np.int64))
        assert diff.max() <= 1, "more than 1 LSB away from the reference decoder"
        n_diff += int((diff != 0).sum())
        n_tot += diff.size
    # reference noise floor: 8 / 1 179 648 samples between thread counts
    # (on the 70-thousand-sample fixtures one rounding tie is already 1.4e-5: allow three samples there)
    assert n_diff <= max(3, 2e-5 * n_tot), f"{n_diff} of {n_tot} samples differ"


def test_arm_sweep_oracle_matches_reference(oracle):
    """Every ARM shape of the sweep fixture (3 .. 32 inputs, 0 .. 3 and 6 / 7 hidden layers, with and without IFCE): the oracle
    decodes the latent grids the REFERENCE decoder decoded (sha256 per grid, tests/golden/gen/make_arm_sweep.py)."""
    import hashlib

    from conftest import load_arm_sweep

    sweep = load_arm_sweep()
    assert len(sweep) == 91
    for name, (stream, want) in sweep.items():
        hdr, nn, lat = oracle.split_stream(stream)[1][0][1][0]
        r = oracle.decode_coolchic(hdr, nn, lat, stop_after_entropy=True)
        got = [hashlib.sha256(np.ascontiguousarray(r["latent"][g]).tobytes()).hexdigest() for g in range(r["n_grids"])]
        assert got == want, name


def test_laplace_known_answers(oracle):
    # (mu_idx, scale_idx, s) -> (left, right)
    kat = [((16384, 1280, 0), (5087973, 11689243)), ((16421, 300, -1), (63, 64)), ((16000, 2560, 5), (8720978, 8775079)),
           ((20000, 0, 14), (78, 16777167)), ((0, 0, -64), (0, 16777089)), ((32767, 2560, 63), (8304534, 16777216))]
    for args, want in kat:
        assert oracle.laplace_bounds(*args) == want


def test_range_encoder_reproduces_shipped_payload(oracle):
    """Known-answer test in the encode direction: re-encoding kodim14's symbols gives its payload."""
    bs, z, j = load_golden("kodim14")
    _, frames = oracle.split_stream(bs)
    hdr, nn, lat = frames[0][1][0]
    r = oracle.decode_coolchic(hdr, nn, lat, stop_after_entropy=True)
    syms, mus, scs = [], [], []
    for g in range(r["n_grids"] - 1, -1, -1):
        h, w = r["grid_hw"][g]
        ys, xs = np.mgrid[0:h, 0:w]
        order = (ys * w + xs if w <= 9 else xs + 10 * ys).ravel()
        idx = np.lexsort((ys.ravel(), order))
        syms.append(r["latent"][g].ravel()[idx])
        mus.append(r["mu_scale_idx"][g][:, 0])
        scs.append(r["mu_scale_idx"][g][:, 1])
    payload = oracle.rc_encode(np.concatenate(syms), np.concatenate(mus), np.concatenate(scs))
    assert payload == lat


@pytest.mark.parametrize("name", VIDEO_STREAMS)
def test_video_ipb_vs_reference(oracle, name):
    """I/P/B video (global translation, alpha/beta blending, 4:2:0) with the sinc-8 warp (vid5) and with the Warper's
    native grid_sample paths, 2 taps = bilinear and 4 taps = bicubic (vid5_w2 / vid5_w4: the same cool-chics, only the
    frame headers' warp_filter_size differs): oracle vs the frames the reference decoder produced. Bar: <= 1 LSB,
    <= 1e-4 of the samples (the reference's float pipeline is not bit-reproducible across torch builds; 5 / 0 / 2 of
    215 040 samples differ here).  vid3_*: 3-frame I / B / P streams with the decoder presets the 5-frame one does not use
    (intra vhop / mop, residue hop / mop / vlop, motion mop)."""
    bs, z, j = load_golden(name)
    frames = oracle.decode_video(bs)
    assert [f["frame_type"] for f in frames] == (["I", "B", "B", "B", "P"] if name.startswith("vid5") else ["I", "P", "P"] if name == "vid3_ldp" else ["I", "B", "P"])
    n_diff = n_tot = 0
    for i, f in enumerate(frames):
        for p, name in enumerate("yuv"):
            d = np.abs(f["planes"][p].astype(np.int64) - z[f"frame{i}.{name}"].astype(np.int64))
            assert d.max() <= 1
            n_diff += int((d != 0).sum())
            n_tot += d.size
    assert n_diff / n_tot <= 1e-4


def test_float_stages_full_plane_block_sums(oracle):
    """Every sample of the float stages against the reference, not only the three crops: float64 sums of all 8 x 8 blocks
    of the dense planes and of the synthesis output of kodim14 (tests/golden/gen/dump_reference.py --block-sums).  A
    regression anywhere in a plane moves its block's sum; the count of blocks that differ by more than what 64 samples
    of f32 noise can explain must be zero."""
    import os

    from conftest import GOLDEN

    bs, z, j = load_golden("kodim14")
    ref = np.load(os.path.join(GOLDEN, "kodim14_blocks.npz"))
    r = oracle.decode_coolchic(*oracle.split_stream(bs)[1][0][1][0])

    def blocks(a):
        c, h, w = a.shape
        hp, wp = -(-h // 8) * 8, -(-w // 8) * 8
        b = np.zeros((c, hp, wp), np.float64)
        b[:, :h, :w] = a
        return b.reshape(c, hp // 8, 8, wp // 8, 8).sum(axis=(2, 4))

    for key, arr, per_sample in (("dense", r["dense"], 2e-6), ("syn_out", r["syn_out"], 2e-6)):
        got, want = blocks(arr), ref[key]
        assert got.shape == want.shape
        diff = np.abs(got - want)
        # oracle vs PyTorch: <= 5e-7 per sample measured; a block holds 64 samples
        n_bad = int((diff > 64 * per_sample).sum())
        assert n_bad == 0, f"{key}: {n_bad} of {diff.size} blocks differ, worst {diff.max():.3e}"


def test_quantise_shortcut():
    """ccd_fused.hip replaces the reference's literal chain for integer samples (decode.py:191-206 then png.py:57 /
    yuv.py:152-160: round(maxv x) / maxv, clamp to [0, 1], round(. maxv) / maxv, round(. maxv)) by rint + clamp.  Same
    result for every float on a dense sweep around every rounding tie and clamp edge, all bit depths."""
    f = np.float32
    for bd in (8, 10, 12, 16):
        maxv = f(2 ** bd - 1)
        ks = np.arange(-3, int(maxv) + 4, max(1, int(maxv) // 4096), dtype=np.float64)
        xs = []
        for off in (0.0, 0.5, 0.25, 0.75):
            centre = ((ks + off) / float(maxv)).astype(f)
            for d in range(-3, 4):  # neighbouring floats around every tie / grid point
                xs.append(np.nextafter(centre, f(np.inf if d > 0 else -np.inf)) if abs(d) == 1 else centre + f(d) * np.spacing(centre))
        rng = np.random.default_rng(bd)
        xs.append(rng.uniform(-0.2, 1.2, 200000).astype(f))
        xs.append(np.array([-1e9, -1.0, -0.0, 0.0, 1.0, 1.0000001, 2.0, 1e9, 1e-30, -1e-30], dtype=f))
        x = np.concatenate(xs).astype(f)
        q = np.rint(maxv * x).astype(f) / maxv               # decode.py:191
        q = np.clip(q, f(0), f(1))                           # decode.py:204
        q = np.rint(q * maxv).astype(f) / maxv               # decode.py:206
        literal = np.rint(q * maxv).astype(np.int64)         # png.py:57 / yuv.py:152-160
        short = np.clip(np.rint(maxv * x), 0, maxv).astype(np.int64)
        assert np.array_equal(literal, short), bd
