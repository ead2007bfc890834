#!/usr/bin/env python3
"""Time line of producer tasks on one grid of a kodak24 stream (trace build):
    python -c "from cool_chic_amd import _build; _build.build_variant('trace', '-DCCD_PIPE_PROFILE=2 -DCCD_PIPE_TRACE')"
    CCD_LIB=cool_chic_amd/libccd_trace.so python tools/trace_tasks.py [stream = 3] [grid = 1] [first step = 1500] [steps = 12]
Per task: producer, pixels, and relative to the step's reference time: start, early wait over, early work done (= late wait
begins), left neighbours seen (= the decoder published them + the poll), ready bit set.  Then per step the chain
"published -> late work -> ready -> decoder -> published"."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 1
step0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
items, _ = bench.build_kodak24(0)
b = DecodeBatch(0)
b.add(*items[idx][:3], 8, 0)
b.run(stage=0); b.wait()
h, w = b.latent(0, grid).shape[-2:]
L = lib()
L.ccd_debug_trace_config.argtypes = [ctypes.c_uint, ctypes.c_uint]
L.ccd_debug_trace_read.argtypes = [ctypes.c_void_p]
n_max = min(h, (w - 1) // 10 + 1)
tp = 8 if n_max >= 25 else (4 if n_max >= 9 else 2)
total_steps = w + 10 * (h - 1)
assert L.ccd_debug_trace_config(w, total_steps - 1 - step0) == 0
b.run(stage=0); b.wait()
buf = np.zeros(512 * 8, np.uint32)
assert L.ccd_debug_trace_read(buf.ctypes.data) == 0
rec = buf.reshape(-1, 8)
assert b.slot_kernels(0) & 1, 'the pipelined kernel did not serve the stream'
rows = []
for r in rec:
    if r[5] == 0: continue
    left = int(r[0] >> 12); task = int((r[0] >> 8) & 0xf); pw = int((r[0] >> 4) & 0xf); cnt = int(r[0] & 0xf)
    step = total_steps - 1 - left
    base = int(r[7]) << 32
    ts = [base | int(x) for x in (r[1], r[2], r[3], r[4], r[5])]
    for k in range(1, 5):  # the low words may wrap once within a task
        while ts[k] < ts[k - 1]: ts[k] += 1 << 32
    need = int(np.int32(r[6]))
    rows.append((step, task, pw, cnt, ts, need, 0))
rows.sort()
t0 = rows[0][4][0]
print(f"stream {idx} grid {grid} {h}x{w}: {tp}-pixel tasks; times in ticks relative to the first record")
last_ready0 = None
seen = {}
for step, task, pw, cnt, t, need, pix0 in rows:
    if step >= step0 + nsteps: break
    a, bq, c_, d, e = [x - t0 for x in t]
    print(f"step {step:5d} task {task} prod {pw} px {cnt}: start {a:7d}  early wait over {bq:7d} (+{bq - a:5d})  late wait from {c_:7d} (+{c_ - bq:5d})  "
          f"left seen {d:7d} (+{d - c_:5d})  ready {e:7d} (+{e - d:5d})   need_px {need - pix0:4d} of prev")
    seen[(step, task)] = (d, e)
print("chains (task 0 of successive steps): late work, then ready -> next step's task 0 sees its left neighbours")
steps = sorted({s for s, _ in seen})
for k in range(4):
    line = []
    for s in steps:
        if (s, k) in seen and (s + 1, k) in seen:
            d, e = seen[(s, k)]; d2, _ = seen[(s + 1, k)]
            line.append(f"{e - d}/{d2 - e}")
    if line: print(f"  task {k}: late work / ready->next seen: " + "  ".join(line))
per = [seen[(s + 1, 0)][0] - seen[(s, 0)][0] for s in steps if (s, 0) in seen and (s + 1, 0) in seen]
if per: print("  step period (task 0 seen -> next task 0 seen): mean %.0f" % (sum(per) / len(per)), per)
# per producer: the phases of its tasks in the window, and the time between the ready bit of one task and the start of the next
import collections
by_prod = collections.defaultdict(list)
for step, task, pw, cnt, t, need, _ in rows:
    by_prod[pw].append((t[0], t, cnt))
ph = collections.defaultdict(list)
for pw, lst in by_prod.items():
    lst.sort()
    for k, (_, t, cnt) in enumerate(lst):
        ph["early wait"].append(t[1] - t[0]); ph["early work"].append(t[2] - t[1]); ph["late wait"].append(t[3] - t[2]); ph["late work"].append(t[4] - t[3])
        if k: ph["between"].append(t[0] - lst[k - 1][1][4])
print("all producers, %d tasks: " % len(rows) + "  ".join("%s %.0f" % (k, sum(v) / len(v)) for k, v in ph.items()))
span = max(r[4][4] for r in rows) - min(r[4][0] for r in rows)
nst = len({r[0] for r in rows})
print("window: %d steps in %d ticks = %.0f per step; producers busy (early work + late work + between) %.0f %% of 7 waves" %
      (nst, span, span / nst, 100.0 * (sum(ph["early work"]) + sum(ph["late work"]) + sum(ph["between"])) / (7.0 * span)))
# per producer (is one of them slower - e.g. the decoder's SIMD neighbour?)
for pw in sorted(by_prod):
    lst = [t for _, t, cnt in by_prod[pw] if cnt == tp]
    if not lst: continue
    k = len(lst)
    print("  producer %d: %3d full tasks, early work %5.0f  late wait %5.0f  late work %5.0f" %
          (pw, k, sum(t[2] - t[1] for t in lst) / k, sum(t[3] - t[2] for t in lst) / k, sum(t[4] - t[3] for t in lst) / k))
