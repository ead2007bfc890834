#!/usr/bin/env python3
"""256 streams (kodak24 repeated) in one batch: entropy stage and whole run with and without chain groups; under
rocprofv3 --kernel-trace the start / end of every kernel of two runs (why did two launches of 64 + 192 workgroups take 2 x one?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cool_chic_amd import DecodeBatch

items, _ = bench.build_kodak24(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
st = torch.cuda.current_stream(0); sh = st.cuda_stream
for overlap in ((True, False) if len(sys.argv) < 4 else (sys.argv[3] == "1",)):
    b = DecodeBatch(0, overlap=overlap)
    for i in range(n):
        b.add(*items[i % 24][:3], 8, 0)
    b.run(sh); b.wait(sh)
    def ms(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    print(f"{n} streams overlap={overlap}: launches {b.entropy_launches()}  entropy stage {ms(lambda: b.run(sh, stage=0)):.2f} ms  whole run {ms(lambda: b.run(sh)):.2f} ms", flush=True)
    b.close()
