#!/usr/bin/env python3
"""vid5 with the warp filter of its P / B frames switched to 2 taps (grid_sample bilinear) and 4 taps (bicubic):
only the 4-bit `warp_filter_size` of the frame headers changes (header.py:217), every cool-chic stays as the reference
encoder wrote it.  The expected planes then come from the REFERENCE decoder:

    python tests/golden/gen/make_warp_variants.py
    for n in vid5_w2 vid5_w4; do python tests/golden/gen/dump_reference.py tests/golden/$n.cool $n --planes-only; done
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden  # noqa: E402
from cool_chic_amd import writer  # noqa: E402
from oracle import oracle_py  # noqa: E402


def main():
    bs, _, _ = load_golden("vid5")
    vh, frames = oracle_py.split_stream(bs)
    n_video_header = vh.n_bytes_header
    for taps in (2, 4):
        out = bytearray(bs[:n_video_header])
        for fh, ccs in frames:
            n_refs = fh.n_refs
            out += writer.frame_header_bytes(fh.display_index, "IPB"[fh.frame_type], fh.frame_data_type, fh.bitdepth,
                                             list(fh.index_references)[:n_refs], list(fh.global_flow)[:2 * n_refs],
                                             warp_filter_size=taps if n_refs else 8)
            for hdr, nn, lat in ccs:
                out += hdr + nn + lat
        assert len(out) == len(bs)
        path = os.path.join(ROOT, "tests", "golden", f"vid5_w{taps}.cool")
        with open(path, "wb") as f:
            f.write(bytes(out))
        print(path, len(out), "bytes,", sum(a != b for a, b in zip(out, bs)), "bytes differ from vid5.cool")


if __name__ == "__main__":
    main()
