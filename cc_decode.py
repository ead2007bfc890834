#!/usr/bin/env python3
"""Drop-in for the reference's cc_decode.py (cc_decode.py:12-20): same flags, MI355X decode."""
import argparse

from cool_chic_amd.bitstream.decode import decode_video

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--input", "-i", type=str, help="Bitstream path.")
    parser.add_argument("--output", "-o", type=str, help="Decoded file path.")
    parser.add_argument("--verbosity", type=int, help="Verbosity level.", default=0)
    parser.add_argument("--device", type=int, help="GPU index.", default=0)
    args = parser.parse_args()
    decode_video(args.input, decoded_path=args.output, verbosity=args.verbosity, device=args.device)
