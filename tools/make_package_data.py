#!/usr/bin/env python3
"""Builds cool_chic_amd/data/donors.npz: what cool_chic_amd/synth.py needs to manufacture benchmark streams (the
reference-encoded donor streams, their decoded latent grids and their network integers), extracted from the golden
fixtures of tests/golden (which tests/golden/gen/dump_reference.py dumped from the imported reference).  The package
then never reads the test tree.      python tools/make_package_data.py"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "cool_chic_amd", "data", "donors.npz")


def main():
    out = {}
    # kodim14 (HOP) and vid5 (I/P/B, LOP): BASELINE's configurations; hq192 (LOP at 2.5 bpp: wide windows), mop192 / vhop192 (the
    # MOP and VHOP decoders): the r05 workloads that vary the statistics and the network (synth.kodak24_hq, synth.clic41_alt)
    for name in ("kodim14", "vid5", "hq192", "mop192", "vhop192"):
        with open(os.path.join(GOLDEN, name + ".cool"), "rb") as f:
            out[name + ".cool"] = np.frombuffer(f.read(), dtype=np.uint8)
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        for k in z.files:
            if k.startswith("cc") and (".latent" in k or k.endswith(".nn_ints")):
                out[f"{name}.{k}"] = z[k]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
