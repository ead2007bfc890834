import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from cool_chic_amd import DecodeBatch
items, streams = bench.build_kodak24(0)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    b = DecodeBatch(0, keep_float=False)
    t1 = time.perf_counter()
    for hdr, nn, lat, _ in items: b.add(hdr, nn, lat, 8, 0)
    t2 = time.perf_counter()
    b.run()
    t3 = time.perf_counter()
    out = b.all_planes()
    t4 = time.perf_counter()
    b.close()
    t5 = time.perf_counter()
    print("create %.2f add %.2f launch %.2f wait+planes %.2f close %.2f total %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)))
