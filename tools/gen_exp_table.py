"""Writes cool_chic_amd/csrc/ccd_exp_table.inc: 2^(j / N), j = 0 .. N - 1, correctly rounded to float64 (decimal, 80 digits).
    python tools/gen_exp_table.py [log2 N = 7]"""
import sys
from decimal import Decimal, getcontext
from pathlib import Path

getcontext().prec = 80
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 1 << log_n
ln2 = Decimal(2).ln()
vals = [float((ln2 * Decimal(j) / Decimal(n)).exp()) for j in range(n)]  # float(Decimal) rounds correctly
out = [f"/* 2^(j / {n}), j = 0 .. {n - 1}, correctly rounded to float64 (tools/gen_exp_table.py: Python decimal, 80 digits).  Table of the entropy",
       " * kernel's exp (ccd_entropy_pipe.hip::exp_nonpos).  Generated - do not edit. */"]
for i in range(0, n, 4):
    out.append("    " + ", ".join(v.hex() for v in vals[i:i + 4]) + ",")
Path(__file__).resolve().parents[1].joinpath("cool_chic_amd/csrc/ccd_exp_table.inc").write_text("\n".join(out) + "\n")
