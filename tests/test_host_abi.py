"""CPU-side checks of the product library: the C ABI loads and exports every symbol include/ccd.h
declares, the host parsers agree with the reference fixtures, and the bitstream writer reproduces
the reference's shipped file byte-for-byte. No compute entry point is called (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import IMAGE_STREAMS, ROOT, load_golden


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


@pytest.mark.parametrize("name", IMAGE_STREAMS + ["vid5"])
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
