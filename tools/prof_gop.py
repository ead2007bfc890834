#!/usr/bin/env python3
"""ccd_decode_video of the 33-frame 1080p GOP a few times (for rocprofv3 --kernel-trace --stats: where its time beyond the
cool-chic decodes goes).    python tools/prof_gop.py [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from cool_chic_amd import synth  # noqa: E402
from cool_chic_amd._lib import Video, check, lib  # noqa: E402

bs = synth.workload("gop1080p33")["streams"][0]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    v = Video()
    check(lib().ccd_decode_video(bs, len(bs), 0, C.byref(v)), "ccd_decode_video")
    t1 = time.perf_counter()
    lib().ccd_video_free(C.byref(v))
    print("ccd_decode_video %.1f ms" % ((t1 - t0) * 1e3))
