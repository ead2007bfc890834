#!/usr/bin/env python3
"""Golden coding structures (build container only): what the REFERENCE's CodingStructure (utils/codingstructure.py:267-436,
used by decode_video through VideoHeader.get_coding_structure, header.py:161) derives from (n_frames, intra_pos, p_pos),
dumped as data for tests/test_host_abi.py::test_coding_structure_matches_reference.

    python tests/golden/gen/dump_coding_structures.py     -> tests/golden/coding_structures.json
"""
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(1, "/root/reference")

from coolchic.utils.codingstructure import CodingStructure  # noqa: E402

CASES = [
    ("image", 1, [0], []),
    ("vid5", 5, [0], [4]),                      # the reference-encoded fixture: I0 P4 B2 B1 B3
    ("gop33", 33, [0, 32], []),                 # BASELINE configs[3]: intra period 32, hierarchical B
    ("low_delay_p", 9, [0], [1, 2, 3, 4, 5, 6, 7, 8]),
    ("open_gop", 9, [0, 8], []),
    ("peculiar", 9, [0], [6, 8]),               # the docstring's example (codingstructure.py:196-202)
    ("two_gops", 17, [0, 8, 16], []),
    ("odd_gaps", 12, [0, 11], [5]),
    ("p_between_intras", 14, [0, 13], [3, 7]),
    ("all_intra", 4, [0, 1, 2, 3], []),
]


def main():
    out = {}
    for name, n, intra, p in CASES:
        cs = CodingStructure(n_frames=n, intra_pos=list(intra), p_pos=list(p))
        frames = sorted(cs.frames, key=lambda f: f.coding_order)
        out[name] = {"n_frames": n, "intra_pos": intra, "p_pos": p,
                     "coding_order": [{"display_order": f.display_order, "frame_type": f.frame_type, "index_references": list(f.index_references),
                                       "depth": f.depth} for f in frames]}
    path = os.path.abspath(os.path.join(HERE, "..", "coding_structures.json"))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, {k: len(v["coding_order"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
