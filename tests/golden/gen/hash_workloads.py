#!/usr/bin/env python3
"""Writes tests/golden/workload_hashes.json: for every stream of every benchmark workload that cool_chic_amd/synth.py
manufactures (kodak24, clic41, uhd4k, gop1080p33 = BASELINE.json configs[1..4] at FULL size), the sha256 of the stream
and the sha256 of the integer planes the CPU ORACLE decodes from it.  bench.py compares what it timed with these hashes
(`verified`), and tests/test_gpu_parity.py::test_workloads_match_the_oracle does the same on the GPU box - where running
the single-thread oracle on 260 Mpx would take minutes, and where /root/reference does not exist anyway.

Run in the build container (needs libccd.so's host-side writer and the oracle; no GPU):
    python tests/golden/gen/hash_workloads.py [workload ...]          (~2 min on 8 cores)
"""
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "workload_hashes.json")
WORKLOADS = ["kodak24", "kodak24_wide_envelope", "kodak24_hq", "clic41", "clic41_alt", "uhd4k", "gop1080p33"]


def planes_sha256(planes) -> str:
    """sha256 over the planes in order, each as little-endian uint16 rows (8-bit planes widened): one definition for the
    oracle, bench.py and the tests (cool_chic_amd.synth.planes_sha256 is the same function)."""
    from cool_chic_amd.synth import planes_sha256 as f

    return f(planes)


def _decode(stream: bytes):
    from oracle import oracle_py

    return [planes_sha256(fr["planes"]) for fr in oracle_py.decode_video(stream)]


def main():
    from cool_chic_amd import synth
    from oracle import oracle_py

    oracle_py.build()
    names = sys.argv[1:] or WORKLOADS
    try:
        with open(OUT) as f:
            out = json.load(f)
    except OSError:
        out = {}
    for name in names:
        t0 = time.time()
        wl = synth.workload(name)
        streams = wl["streams"]
        order = sorted(range(len(streams)), key=lambda i: -len(streams[i]))  # longest first
        with ProcessPoolExecutor(max_workers=min(len(streams), os.cpu_count() or 1)) as ex:
            res = dict(zip(order, ex.map(_decode, [streams[i] for i in order])))
        out[name] = {"streams_sha256": [hashlib.sha256(s).hexdigest() for s in streams],
                     "planes_sha256": [res[i] for i in range(len(streams))],  # per stream: one hash per frame, display order
                     "frames": sum(len(res[i]) for i in range(len(streams))),
                     "decoded_by": "oracle/cc_oracle.c (single-thread CPU restatement of the reference decoder)"}
        print(name, len(streams), "streams,", out[name]["frames"], "frames, %.0f s" % (time.time() - t0), flush=True)
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
