"""CPU-side checks of the product library: the C ABI loads and exports every symbol include/ccd.h
declares, the host parsers agree with the reference fixtures, and the bitstream writer reproduces
the reference's shipped file byte-for-byte. No compute entry point is called (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import IMAGE_STREAMS, ROOT, VIDEO_STREAMS, load_golden


def test_every_declared_symbol_is_exported():
    from cool_chic_amd import _lib

    text = open(os.path.join(ROOT, "include", "ccd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ccd_[a-z0-9_]+)\s*\(", text))
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libccd.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with include/ccd.h"
    assert b"gfx950" in L.ccd_version()


@pytest.mark.parametrize("name", IMAGE_STREAMS + ["vid5", "vid3_hop", "vid3_mop", "vid3_vlop", "vid3_ldp"])
def test_headers_match_reference(name):
    from cool_chic_amd.bitstream.header import CoolChicHeader, FrameHeader, VideoHeader

    bs, z, j = load_golden(name)
    vh = VideoHeader()
    rest = vh.read_header(bs)
    for k, v in j["video_header"].items():
        assert vh.get_value(k) == v, k
    n_cc = 0
    for f in range(vh.get_value("n_frames")):
        fh = FrameHeader()
        rest = fh.read_header(rest)
        for k, v in j["headers"][f]["frame"].items():
            assert fh.get_value(k) == v, k
        for _ in range(2 if fh.get_value("frame_type") in ("P", "B") else 1):
            ch = CoolChicHeader()
            rest = ch.read_header(rest)
            want = j["cc"][n_cc]
            for k, v in want["header"].items():
                got = ch.get_value(k)
                if isinstance(v, dict):
                    assert {a: float(b) for a, b in got.items()} == v, k
                else:
                    assert got == v, k
            assert ch.size_per_latent() == [tuple(s) for s in want["size_per_latent"]]
            assert [bool(x) for x in ch.c.is_hyperlatent[: ch.c.n_grids]] == want["flag_is_hyperlatent"]
            assert list(ch.c.input_features_ifce[: ch.c.n_grids]) == want["input_features_ifce"]
            rest = rest[ch.get_value("nn_n_bytes") + ch.get_value("n_bytes_latent"):]
            n_cc += 1
    assert rest == b""


def test_header_errors():
    from cool_chic_amd import CcdError
    from cool_chic_amd.bitstream.header import CoolChicHeader, FrameHeader, VideoHeader

    bs, _, _ = load_golden("kodim14")
    with pytest.raises(CcdError):
        VideoHeader().read_header(bs[:3])  # truncated
    rest = VideoHeader().read_header(bs)
    bad = bytearray(rest)
    bad[1] |= 0x0C  # frame_type index 3 does not exist
    with pytest.raises(CcdError):
        FrameHeader().read_header(bytes(bad))
    rest = FrameHeader().read_header(rest)
    with pytest.raises(CcdError):
        CoolChicHeader().read_header(rest[:20])


def test_writer_reproduces_the_shipped_bitstream(oracle):
    from cool_chic_amd import writer

    bs, z, j = load_golden("kodim14")
    _, frames = oracle.split_stream(bs)
    hdr, nn, lat = frames[0][1][0]
    latents = [z[f"cc0.latent{g}"] for g in range(10)]  # the REFERENCE's decoded latents
    assert writer.encode_stream(hdr, nn, latents) == bs


def test_writer_variants_round_trip_through_the_oracle(oracle):
    from cool_chic_amd import writer

    bs, z, j = load_golden("rgb192")
    _, frames = oracle.split_stream(bs)
    hdr, nn, lat = frames[0][1][0]
    latents = [z[f"cc0.latent{g}"] for g in range(10)]
    sizes, levels = writer.grid_sizes((128, 192), hdr)
    assert sizes == [a.shape for a in latents]
    for seed, transpose in ((1, False), (2, True)):
        v = writer.variant_latents(latents, levels, seed, transpose)
        s = writer.encode_stream(hdr, nn, v, img_size=(192, 128) if transpose else (128, 192))
        r = oracle.decode_coolchic(*oracle.split_stream(s)[1][0][1][0], stop_after_entropy=True)
        assert all(np.array_equal(a, b) for a, b in zip(v, r["latent"]))


def test_range_encode_matches_oracle_encoder(oracle):
    from cool_chic_amd import writer

    rng = np.random.default_rng(0)
    n = 20000
    mu = rng.integers(0, 32768, n)
    sc = rng.integers(0, 2561, n)
    sym = np.clip(np.round((mu / 256.0 - 64) + rng.laplace(size=n) * 2), -64, 63).astype(np.int8)
    assert writer.range_encode(sym, mu, sc) == oracle.rc_encode(sym, mu, sc)
    assert writer.range_encode(sym[:0], mu[:0], sc[:0]) == b""


def test_writer_reproduces_reference_streams(oracle):
    """Known-answer tests for the bitstream writer: re-serialising every header, re-coding the
    NN integers (Exp-Golomb) and re-encoding the latents (host ARM walk + range encoder) of the reference-ENCODED
    fixtures gives back the files byte for byte - image streams and the 5-frame I/P/B video alike."""
    from cool_chic_amd import writer

    for name in ["kodim14", "rgb192", "yuv420_8b", "yuv420_10b", "yuv444_10b", "vid5"]:
        bs, z, _ = load_golden(name)
        vh, frames = oracle.split_stream(bs)
        out = [writer.video_header_bytes(vh.n_frames, list(vh.intra_pos[:vh.n_intras]), list(vh.p_pos[:vh.n_p_frames]))]
        cc_idx = 0
        for fh, ccs in frames:
            out.append(writer.frame_header_bytes(fh.display_index, "IPB"[fh.frame_type], fh.frame_data_type, fh.bitdepth,
                                                 list(fh.index_references[:fh.n_refs]), list(fh.global_flow[:2 * fh.n_refs]),
                                                 fh.warp_filter_size))
            for hdr, nn, _lat in ccs:
                arch = writer.parse_cc_header(hdr)
                assert writer.cc_header_bytes(arch) == hdr
                if f"cc{cc_idx}.nn_ints" in z.files:
                    assert writer.encode_network(arch, z[f"cc{cc_idx}.nn_ints"]) == nn
                lat = [z[f"cc{cc_idx}.latent{g}"] for g in range(arch.n_grids)]
                out.append(writer.encode_coolchic(arch, nn, lat))
                cc_idx += 1
        assert b"".join(out) == bs, name


def test_random_headers_round_trip_and_agree_with_the_oracle(oracle):
    """Seeded sweep over the header space: every transmitted cool-chic field drawn at random (architectures, picture
    sizes, latent / hyperlatent / IFCE ranges, quantisation-step and Exp-Golomb indices, synthesis layers), frame
    headers of all three types with signed global flows, video headers with intra / P position lists.  Written with the
    C ABI writers, parsed back by the C ABI readers (all fields equal) and by the oracle (same fields, same derived
    geometry: grid sizes, hyperlatent flags, IFCE inputs, synthesis inputs)."""
    import ctypes as C

    from cool_chic_amd import writer
    from cool_chic_amd._lib import CCHeader, FrameHeader, VideoHeader, check, lib

    rng = np.random.default_rng(7)
    q_first = [-8, -16, -8, -16, -12, 0, -12, -24]
    q_count = [9, 17, 9, 17, 13, 1, 13, 25]
    for _ in range(300):
        a = CCHeader()
        a.linear_stabiliser_synth = int(rng.integers(0, 2)); a.n_layer_synthesis = int(rng.integers(1, 6))
        a.ups_k_size = int(rng.choice([4, 6, 8])); a.ups_preconcat_k_size = int(rng.choice([3, 5, 7]))
        a.spatial_context_arm = int(rng.integers(1, 25)); a.output_feature_ifce = int(rng.integers(0, 8))
        a.linear_stabiliser_arm = int(rng.integers(0, 2)); a.n_hidden_layers_arm = int(rng.integers(0, 4))
        a.img_size[0], a.img_size[1] = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
        lo = int(rng.integers(0, 4)); hi = int(rng.integers(lo, 9))
        a.latent_resolution[0], a.latent_resolution[1] = lo, hi
        a.flag_hyperlatent = int(rng.integers(0, 2))
        if a.flag_hyperlatent:
            hl = int(rng.integers(0, 9)); a.hyperlatent_resolution[0], a.hyperlatent_resolution[1] = hl, int(rng.integers(hl, 9))
        a.n_latent_grids = (hi - lo + 1) + ((a.hyperlatent_resolution[1] - a.hyperlatent_resolution[0] + 1) if a.flag_hyperlatent else 0)
        a.flag_common_randomness = int(rng.integers(0, 2)); a.final_upsampling_type = int(rng.integers(0, 3))
        for i in range(8):
            a.nn_q_step_log2[i] = q_first[i] + int(rng.integers(0, q_count[i]))
            a.nn_expgol_cnt[i] = int(rng.integers(0, 13))  # POSSIBLE_EXP_GOL_COUNT: 0..12 (nnquant/expgolomb.py:20-37)
        a.nn_n_bytes, a.nn_n_bit_pad, a.n_bytes_latent = int(rng.integers(0, 16384)), int(rng.integers(0, 8)), 4 * int(rng.integers(0, 1 << 20))
        if a.output_feature_ifce > 0:
            il = int(rng.integers(0, 9)); a.ifce_resolution[0], a.ifce_resolution[1] = il, int(rng.integers(il, 9))
        c_in = (hi - lo + 1) * (2 if a.flag_common_randomness else 1)
        for l in range(a.n_layer_synthesis):
            mode = int(rng.integers(0, 2))
            a.syn_layer[l].out_ft = c_in if mode else int(rng.integers(1, 64))
            a.syn_layer[l].k_size = int(rng.choice([1, 3, 5])); a.syn_layer[l].mode = mode
            a.syn_layer[l].non_linearity = int(rng.integers(0, 2))
            c_in = a.syn_layer[l].out_ft
        raw = writer.cc_header_bytes(a)
        b = writer.parse_cc_header(raw)
        assert b.n_bytes_header == len(raw)
        for name, ctype in CCHeader._fields_:
            if name in ("n_bytes_header", "has_ifce_resolution") or name in ("n_grids", "grid_h", "grid_w", "is_hyperlatent",
                                                                               "input_features_ifce", "input_feature_synthesis",
                                                                               "total_context_arm", "out_channels", "n_symbols"):
                continue
            va, vb = getattr(a, name), getattr(b, name)
            if name == "syn_layer":
                for l in range(a.n_layer_synthesis):
                    assert (va[l].out_ft, va[l].k_size, va[l].mode, va[l].non_linearity) == (vb[l].out_ft, vb[l].k_size, vb[l].mode, vb[l].non_linearity)
            elif name in ("ifce_resolution", "hyperlatent_resolution") and not (a.output_feature_ifce if name[0] == "i" else a.flag_hyperlatent):
                continue
            elif hasattr(va, "__len__"):
                assert list(va) == list(vb), name
            else:
                assert va == vb, name
        oh, og = oracle.read_cc_header(raw)
        assert (oh.img_size[0], oh.img_size[1], oh.n_bytes_header, oh.n_bytes_latent) == (a.img_size[0], a.img_size[1], len(raw), a.n_bytes_latent)
        assert og.n_grids == b.n_grids
        assert list(og.grid_h[: og.n_grids]) == list(b.grid_h[: b.n_grids]) and list(og.grid_w[: og.n_grids]) == list(b.grid_w[: b.n_grids])
        assert list(og.is_hyper[: og.n_grids]) == list(b.is_hyperlatent[: b.n_grids])
        assert list(og.input_features_ifce[: og.n_grids]) == list(b.input_features_ifce[: b.n_grids])
        assert og.input_feature_synthesis == b.input_feature_synthesis and og.total_context_arm == b.total_context_arm
    # out-of-table indices raise like the reference (element.py:289-292)
    from cool_chic_amd import CcdError
    a.nn_expgol_cnt[3] = 13
    with pytest.raises(CcdError):
        writer.parse_cc_header(writer.cc_header_bytes(a))
    for _ in range(200):
        ft = int(rng.integers(0, 3))
        refs = [int(x) for x in rng.integers(0, 4096, size=ft)]
        flows = [int(x) for x in rng.integers(-8191, 8192, size=2 * ft)]
        raw = writer.frame_header_bytes(int(rng.integers(0, 4096)), "IPB"[ft], int(rng.integers(0, 4)), int(rng.integers(8, 17)), refs,
                                        flows, int(rng.integers(0, 16)))
        f = FrameHeader()
        assert check(lib().ccd_read_frame_header(raw, len(raw), C.byref(f)), "read") == len(raw)
        assert f.frame_type == ft and list(f.index_references[:ft]) == refs and list(f.global_flow[: 2 * ft]) == flows
    for _ in range(50):
        n_i, n_p = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        ip, pp = [int(x) for x in rng.integers(0, 4096, size=n_i)], [int(x) for x in rng.integers(0, 4096, size=n_p)]
        raw = writer.video_header_bytes(int(rng.integers(1, 4096)), ip, pp)
        v = VideoHeader()
        assert check(lib().ccd_read_video_header(raw, len(raw), C.byref(v)), "read") == len(raw)
        assert list(v.intra_pos[:n_i]) == ip and list(v.p_pos[:n_p]) == pp


def test_header_readers_survive_garbage():
    """Random bytes and every truncation of a real stream's head: the three readers return a byte count within the
    buffer or a negative CCD_ERR_* code - never read past the buffer, never crash."""
    import ctypes as C

    from cool_chic_amd._lib import CCHeader, FrameHeader, VideoHeader, lib

    rng = np.random.default_rng(3)
    bs, _, _ = load_golden("vid5")
    bufs = [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.integers(0, 80, size=300)]
    bufs += [bs[:n] for n in range(0, 120)]
    for raw in bufs:
        for fn, st in ((lib().ccd_read_video_header, VideoHeader()), (lib().ccd_read_frame_header, FrameHeader()),
                       (lib().ccd_read_cc_header, CCHeader())):
            rc = fn(raw, len(raw), C.byref(st))
            assert rc < 0 or 0 < rc <= len(raw)


def test_fixtures_reach_every_entropy_instantiation(oracle):
    """ccd_network_kernel_class (host only) over every cool-chic of every fixture: which `entropy_pipe_kernel<NV, MF, DYN>` the
    `-m gpu` parity tests execute (DESIGN.md section 2 holds the table).  The reference ENCODER's presets give NV = 2 (lop, motion
    lop / mop, residue lop / vlop), 3 (residue mop), 4 (intra mop, residue hop), 5 (intra hop = kodim14), 7 (intra vhop); the
    sweep fixture gives every NV = 1..8 with 1..4, 7 and 8 layers, and test_arm_sweep_every_instantiation forces both variants."""
    from conftest import load_arm_sweep
    from cool_chic_amd._lib import lib

    def classes(stream):
        out = []
        for _fh, ccs in oracle.split_stream(stream)[1]:
            for hdr, nn, _lat in ccs:
                k = lib().ccd_network_kernel_class(hdr, len(hdr), nn, len(nn))
                assert k > 0 and k & 1, "every fixture network fits the pipelined kernel"
                out.append(((k >> 8) & 15, (k >> 12) & 15, bool(k & 16)))
        return out

    ref_encoded = {}
    for name in IMAGE_STREAMS + VIDEO_STREAMS:
        ref_encoded[name] = classes(load_golden(name)[0])
    assert ref_encoded["kodim14"] == [(5, 3, False)] and ref_encoded["mop192"] == [(4, 3, False)] and ref_encoded["vhop192"] == [(7, 3, False)]
    # vid3_hop: I = vhop (26 inputs), then per inter frame residue hop (14 inputs) + motion mop (8 inputs)
    assert [c[:2] for c in ref_encoded["vid3_hop"]] == [(7, 3), (4, 3), (2, 3), (4, 3), (2, 3)]
    # vid3_mop: I = mop (14), residue mop (12) + motion lop (8 inputs, one hidden layer)
    assert [c[:2] for c in ref_encoded["vid3_mop"]] == [(4, 3), (3, 3), (2, 2), (3, 3), (2, 2)]
    # vid3_vlop: I = lop (8), residue vlop (6 inputs, one hidden layer) + motion mop (8)
    assert [c[:2] for c in ref_encoded["vid3_vlop"]] == [(2, 3), (2, 2), (2, 3), (2, 2), (2, 3)]
    assert [c[:2] for c in ref_encoded["vid3_ldp"]] == [(2, 3), (2, 3), (2, 2), (2, 3), (2, 2)]
    nv_ref = {c[0] for v in ref_encoded.values() for c in v}
    assert nv_ref == {2, 3, 4, 5, 7}
    sweep = {n: classes(s)[0] for n, (s, _) in load_arm_sweep().items()}
    assert {c[0] for c in sweep.values()} == set(range(1, 9))
    for nv in range(1, 9):  # every width with 1 .. 4 layers, statically without the feature check at least once
        assert {c[1] for c in sweep.values() if c[0] == nv} >= {1, 2, 3, 4}
        assert any(c[0] == nv and not c[2] for c in sweep.values())
    assert {c[1] for c in sweep.values()} >= {7, 8}


def test_chain_groups_plan():
    """ccd_debug_chain_groups = the pure planning function behind ccd_batch_run's chain groups: which slots share an
    entropy launch.  kodak24's shape (18 + 6), the caps by concurrent streams and instantiations, and the XCD rule: a split must leave
    every workgroup a CU - sum over launches of ceil(n / 8) <= CUs / 8 - so 63 + 193 streams stay ONE launch (33 workgroups would
    land on one 32-CU XCD and the 33rd would wait a whole chain: measured 68 ms against 36.6), 56 + 168 are split."""
    from cool_chic_amd._lib import lib

    def plan(est, inst, n_conc=4, n_cu=256):
        e = np.asarray(est, np.float64)
        k = np.asarray(inst, np.int32)
        cg = np.full(len(e), -7, np.int32)
        g = lib().ccd_debug_chain_groups(e.ctypes.data, k.ctypes.data, len(e), n_conc, n_cu, cg.ctypes.data)
        return g, cg.tolist()

    # 6 portrait streams 9 % longer than 18 landscape ones, one instantiation
    g, cg = plan([1.09] * 6 + [1.0] * 18, [0] * 24)
    assert g == 2 and cg == [0] * 6 + [1] * 18
    # three size classes -> three groups; with two concurrent streams only two; with one (or under a serialising profiler) one
    est = [1.0, 0.99, 0.9, 0.85, 0.5, 0.2]
    assert plan(est, [0] * 6) == (3, [0, 0, 1, 1, 2, 2])
    assert plan(est, [0] * 6, n_conc=2) == (2, [0, 0, 1, 1, 1, 1])
    assert plan(est, [0] * 6, n_conc=1) == (1, [0] * 6)
    # two instantiations share four streams: two groups each; four instantiations: one each; the generic launch (-1) is never split
    assert plan(est, [0, 1, 0, 1, 0, 1])[0] == 2
    assert plan(est, [0, 1, 2, 3, 0, 1]) == (1, [0] * 6)
    assert plan(est, [-1] * 6) == (1, [0] * 6)
    # the XCD rule at a full chip: 63 + 193 would put 8 + 25 = 33 on one XCD - the boundary moves to a multiple of 8 (the slowest
    # landscape stream joins the portrait group): 64 + 192 = 8 + 24; where no rounding helps, fewer groups
    g, cg = plan([1.09] * 63 + [1.0 - 1e-6 * i for i in range(193)], [0] * 256)
    assert g == 2 and cg.count(0) == 64 and cg[63] == 0 and cg[64] == 1
    g, cg = plan([1.09] * 63 + [1.0] * 193 + [0.5] * 3, [0] * 259)    # 259 streams: more workgroups than CUs whatever the split
    assert g == 1 and set(cg) == {0}
    g, cg = plan([1.09] * 60 + [1.0] * 100 + [0.5] * 96, [0] * 256)   # three groups 60 / 100 / 96 -> 64 / 96 / 96
    assert g == 3 and [cg.count(k) for k in range(3)] == [64, 96, 96]
    g, cg = plan([1.09] * 64 + [1.0] * 192, [0] * 256)   # 8 + 24 = 32 per XCD: fits exactly
    assert g == 2
    g, cg = plan([1.09] * 56 + [1.0] * 168, [0] * 224)
    assert g == 2 and cg.count(0) == 56
    # uniform batches are not split at all
    assert plan([1.0] * 24, [0] * 24) == (1, [0] * 24)
    assert lib().ccd_debug_chain_groups(None, None, 0, 4, 256, None) < 0


def test_symbol_ring_width_limit(oracle):
    """The pipelined entropy kernel keeps the recent symbols of W / 10 + 6 picture rows in a 512-row LDS ring: pictures up to
    5 069 columns (documented as 5 060) (include/ccd.h "envelope"); the format's 14-bit img_size allows 16 383 (header.py:244-307), and a
    wider picture is served by the generic kernel (priced in bench.py's fallback_cliffs; GPU test
    test_picture_wider_than_the_symbol_ring).  Host only: ccd_network_fits_fast_path at both sides of the limit."""
    from cool_chic_amd import writer
    from cool_chic_amd._lib import lib

    bs, z, _ = load_golden("kodim14")
    hdr, _, _ = oracle.split_stream(bs)[1][0][1][0]
    donor = writer.parse_cc_header(hdr)
    for w, want in ((5060, 1), (5069, 1), (5070, 0), (7680, 0), (16383, 0)):  # W / 10 + 6 <= 512 rows
        arch = writer.derive_arch(donor, img_size=(64, w))
        nn = writer.encode_network(arch, writer.adapt_network(donor, z["cc0.nn_ints"], arch))
        h = writer.cc_header_bytes(arch)
        assert lib().ccd_network_fits_fast_path(h, len(h), nn, len(nn)) == want, w
        assert lib().ccd_network_kernel_class(h, len(h), nn, len(nn)) & 1 == want, w


@pytest.mark.parametrize("name", IMAGE_STREAMS + VIDEO_STREAMS)
def test_every_reference_network_fits_the_pipelined_kernel(oracle, name):
    """ccd_network_fits_fast_path (host only): the static envelope of the pipelined entropy kernel is on the WEIGHTS (int32);
    what depends on the data is checked on the device.  Every network the reference encoder produced in the build container
    is inside it - the r02 worst-case envelope sent rgb192 / cr192, yuv444_10b and vid5's I frame to the generic kernel."""
    from cool_chic_amd import writer

    bs, z, j = load_golden(name)
    _, frames = oracle.split_stream(bs)
    n = 0
    for fh, ccs in frames:
        for hdr, nn, lat in ccs:
            assert writer.fits_fast_path(hdr, nn), f"{name}: cool-chic {n}"
            n += 1
    assert n == (9 if name.startswith("vid5") else 5 if name.startswith("vid3") else 1)


def test_coding_structure_matches_reference():
    """ccd_get_coding_structure (ccd_format.cpp::coding_structure) against what the reference's CodingStructure builds
    (utils/codingstructure.py:267-436; dumped by tests/golden/gen/dump_coding_structures.py): coding order, frame types,
    references and depths for ten (n_frames, intra_pos, p_pos) cases incl. the 33-frame GOP of BASELINE configs[3], the
    reference-encoded 5-frame fixture and the docstring's "peculiar" structure."""
    import json

    from cool_chic_amd import writer
    from cool_chic_amd.bitstream.header import VideoHeader

    with open(os.path.join(ROOT, "tests", "golden", "coding_structures.json")) as f:
        cases = json.load(f)
    assert "gop33" in cases and "vid5" in cases
    for name, c in cases.items():
        vh = VideoHeader()
        vh.read_header(writer.video_header_bytes(c["n_frames"], c["intra_pos"], c["p_pos"]))
        assert vh.get_coding_structure() == c["coding_order"], name
    # where the reference asserts (codingstructure.py:230-263)
    for n, intra, p in ((4, [1, 3], []), (4, [0], []), (4, [0], [2]), (5, [0, 4], [4])):
        vh = VideoHeader()
        vh.read_header(writer.video_header_bytes(n, intra, p))
        with pytest.raises(ValueError):
            vh.get_coding_structure()


def test_fixture_frame_headers_follow_the_coding_structure(oracle):
    """The frame headers of the reference-encoded video agree with the structure its video header implies - the property
    ccd_decode_video / decode_video now enforce."""
    from cool_chic_amd.bitstream.header import VideoHeader

    bs, z, j = load_golden("vid5")
    vh = VideoHeader()
    vh.read_header(bs)
    cs = vh.get_coding_structure()
    _, frames = oracle.split_stream(bs)
    assert [(f["display_order"], "IPB".index(f["frame_type"]), f["index_references"]) for f in cs] == \
           [(fh.display_index, fh.frame_type, list(fh.index_references[: fh.n_refs])) for fh, _ in frames]


@pytest.mark.parametrize("bitdepth", [8, 10, 16])
def test_ppm_writer_matches_reference(tmp_path, bitdepth):
    """save_frame_data_to_file(.ppm) (io/io.py:84-90 -> io/format/ppm.py:161-203: "P6\\nW H\\nmax\\n", interleaved RGB,
    two-byte samples MSB first above 8 bits) against the bytes the reference wrote for the same FrameData
    (tests/golden/gen/dump_ppm.py): ranges' ends, 255 / 256 / 257 around the byte boundary, random samples."""
    import torch

    from cool_chic_amd.io import FrameData, save_frame_data_to_file

    planes = np.load(os.path.join(ROOT, "tests", "golden", "ppm_planes.npz"))[f"ppm{bitdepth}"]
    data = torch.from_numpy(planes.astype(np.float32) / np.float32(2 ** bitdepth - 1))[None]
    out = tmp_path / "x.ppm"
    save_frame_data_to_file(FrameData(bitdepth, "rgb", data), str(out))
    with open(os.path.join(ROOT, "tests", "golden", f"ppm{bitdepth}.ppm"), "rb") as f:
        want = f.read()
    assert out.read_bytes() == want


def test_generated_kernel_tables_are_current(tmp_path, monkeypatch):
    """The generated pieces of the entropy kernel (the decoder's unrolled symbol blocks, the table of the device exp) are what
    their generators write: nobody edited one side only."""
    import importlib.util
    import shutil
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "cool_chic_amd", "csrc")
    names = ["ccd_dec_block16.inc", "ccd_dec_block16p.inc", "ccd_dec_block16pm.inc", "ccd_dec_tramp16p.inc", "ccd_dec_parts8.inc", "ccd_dec_parts4.inc", "ccd_dec_parts4_nc.inc", "ccd_dec_block32.inc", "ccd_dec_tramp16.inc",
             "ccd_dec_tramp32.inc", "ccd_exp_table.inc"]
    # run the generators on a copy of the tree layout (they write next to the sources)
    fake = tmp_path / "repo"
    (fake / "tools").mkdir(parents=True)
    (fake / "cool_chic_amd" / "csrc").mkdir(parents=True)
    for tool, argv in (("gen_decoder_block.py", []), ("gen_exp_table.py", ["7"])):
        dst = fake / "tools" / tool
        shutil.copy(os.path.join(root, "tools", tool), dst)
        monkeypatch.setattr(sys, "argv", [str(dst)] + argv)
        spec = importlib.util.spec_from_file_location("gen_" + tool[:-3], dst)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if hasattr(mod, "main"):
            mod.main()
    for n in names:
        assert (fake / "cool_chic_amd" / "csrc" / n).read_text() == open(os.path.join(csrc, n)).read(), n


def test_float_envelope_of_fixtures_and_crafted_networks(oracle):
    """ccd::float_path_stays_finite through ccd_network_kernel_class bit 7 (host only): every network the reference encoder
    produced and every synthetic workload's network is INSIDE the finite envelope (they run the matrix-core float kernel);
    the crafted networks of tests/float_classes.py fall on the side they were built for, the oracle's outputs for them really
    contain the classes (NaN, inf, subnormals) the GPU test compares on, and a network inside the envelope never produces a
    non-finite value."""
    from float_classes import crafted
    from cool_chic_amd import synth
    from cool_chic_amd._lib import lib

    for name in IMAGE_STREAMS + VIDEO_STREAMS:
        for _fh, ccs in oracle.split_stream(load_golden(name)[0])[1]:
            for hdr, nn, _lat in ccs:
                k = lib().ccd_network_kernel_class(hdr, len(hdr), nn, len(nn))
                assert k > 0 and not k & 128, name
    for stream in (synth.image_stream(1365, 2048, 0), synth.image_stream(2160, 3840, 0)):  # the grown 8- and 9-level networks
        hdr, nn, _ = synth.split_image_stream(stream)
        assert not lib().ccd_network_kernel_class(hdr, len(hdr), nn, len(nn)) & 128
    n_nonfinite = n_sub = 0
    for name, ((hdr, nn, lat), _stream, inside) in crafted(load_golden, oracle).items():
        k = lib().ccd_network_kernel_class(hdr, len(hdr), nn, len(nn))
        assert k > 0
        if inside is not None:
            assert bool(k & 128) == (not inside), name
        ref = oracle.decode_coolchic(hdr, nn, lat)
        finite = np.isfinite(ref["out"]).all() and np.isfinite(ref["dense"]).all() and np.isfinite(ref["syn_out"]).all()
        if not k & 128:
            assert finite, f"{name}: inside the envelope, yet the oracle met a non-finite value"
        n_nonfinite += not finite
        u = ref["dense"].view(np.uint32)
        n_sub += bool((((u & 0x7F800000) == 0) & ((u & 0x007FFFFF) != 0)).any())
    assert n_nonfinite >= 2 and n_sub >= 1


def test_bench_refuses_a_variant_library_and_build_cleans_variants(tmp_path):
    """Housekeeping that protects the numbers: bench.py does not time a library CCD_LIB names unless told
    to, and _build.clean_variants removes variant libraries / object trees of earlier profiling runs (they ship to the GPU box
    with every push), keeping the ones asked for."""
    import subprocess
    import sys

    from cool_chic_amd import _build

    env = dict(os.environ, CCD_LIB=os.path.join(ROOT, "cool_chic_amd", "libccd_does_not_exist.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "CCD_LIB" in r.stderr and "--allow-variant" in r.stderr
    # on a scratch copy of the package layout: the real directory may hold variants somebody is about to use (r05 advisor)
    here = str(tmp_path)
    os.makedirs(os.path.join(here, "csrc", "_obj"))
    open(os.path.join(here, "libccd.so"), "wb").close()
    lib_a, lib_b = os.path.join(here, "libccd_tmpa.so"), os.path.join(here, "libccd_tmpb.so")
    obj_a = os.path.join(here, "csrc", "_obj_tmpa")
    for f in (lib_a, lib_b):
        open(f, "wb").close()
    os.makedirs(obj_a, exist_ok=True)
    assert set(_build.list_variants(root=here)) == {lib_a, lib_b, obj_a}
    gone = _build.clean_variants(keep=("tmpb",), root=here)
    assert lib_a in gone and obj_a in gone and lib_b not in gone
    assert not os.path.exists(lib_a) and not os.path.exists(obj_a) and os.path.exists(lib_b)
    assert os.path.exists(os.path.join(here, "libccd.so")) and os.path.isdir(os.path.join(here, "csrc", "_obj"))  # the product itself is never touched
