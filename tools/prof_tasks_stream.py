"""Level-2 profile (-DCCD_PIPE_PROFILE=2 build) of producer 0's tasks on ONE stream of kodak24 (index 0 = landscape kodim14,
3 = the first portrait stream): ticks per task idle before the early wait / early work / late wait / late work / between tasks.
    CCD_LIB=cool_chic_amd/libccd_prof2.so python tools/prof_tasks_stream.py 3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
items, _ = bench.build_kodak24(0)
b = DecodeBatch(0)
b.add(*items[idx][:3], 8, 0)
for _ in range(2):
    b.run(stage=0); b.wait()
st = np.zeros(64, np.int32); lib().ccd_batch_slot_stats(b._h, 0, st.ctypes.data)
u = st[4:24].view(np.uint64); e = st[40:50].view(np.uint64)
nsym = int(b.header(0).n_symbols); nt = max(int(e[4]), 1)
print("stream %d (%s, kernels %d): %d symbols, decoder %.0f ticks / symbol; producer 0 (all grids): %d tasks, per task: idle %.0f  early work %.0f  late wait %.0f  late work %.0f  between %.0f" %
      (idx, "x".join(map(str, items[idx][3])), b.slot_kernels(0), nsym, float(u[0]) / nsym, nt, u[6] / nt, u[7] / nt, u[8] / nt, u[9] / nt, e[0] / nt))
