#!/usr/bin/env python3
"""Benchmark of the MI355X Cool-chic decoder on BASELINE.json's metric: decoded Mpixel/s.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak|throughput] [--legs a,b,..] [--no-cpu-baseline]
    (N > 1: launched by torch.distributed.run, one rank per GPU)

`value` = BASELINE configs[1] (config.workload = "kodak24"): the Kodak-24 set - 24 RGB 8-bit 512x768 frames (18
landscape, 6 portrait), HOP decoder, one frame per batch slot, all in flight on one MI355X.  A "step" = one full decode
of the set: entropy decode (integer ARM / IFCE + range decoder), then ONE fused kernel for latent-pyramid upsampling +
synthesis + integer samples; inputs (payload words, network parameters) are resident in HBM before the timed region,
outputs stay in HBM.  Only kodim14.cool is a real bitstream; every other input is manufactured by cool_chic_amd/synth.py
(SURVEY.md section 8d, H6).  `from_bytes` repeats the same K steps from the stream BYTES in host memory to integer planes in
pinned host memory - batch creation, parsing, uploads, decode, planes back, nothing cached - with two sets in flight, so that
everything but the decode hides behind the decode of the set before (timed_from_bytes).

Beside the metric (never as `value`), rank 0 reports one leg per other BASELINE configuration - `clic41` (configs[2]),
`gop1080p33` (configs[3]), `uhd4k` (configs[4]) - each with Mpixel/s, ms, Msymbol/s and its own CPU sample, plus
`more_frames_in_flight` (256 streams: the chip-filling regime), `with_png_packing`, `end_to_end_from_bytes` (one set at a time),
`kodak24_hq` / `clic41_alt` (other statistics, other decoders), `fallback_cliffs` (the generic kernels on the metric's set), the
per-orientation entropy times and the float-stage roofline.

N GPUs: --scaling strong (default) = what BASELINE.json names: the 24 frames of kodak24 sharded round-robin over the ranks
(frame i -> rank i mod N), rank 0 gathers the decoded planes over RCCL inside the timed region.  BASELINE's own sets (24 /
41 frames, 66 cool-chics per GOP) keep 9-26 % of ONE GPU's CUs busy - a stream is one serial range-decoder chain on one
CU, and the step time is the slowest stream's - so they CANNOT strong-scale: the expected curve is flat (`expected_scaling`
in the line says so).  Beside the metric an N > 1 run measures, with all ranks, `clic41_sharded` (BASELINE configs[2]: the
41 pictures round-robin, strong) and `throughput_regime` (every rank its own 256 streams = kodak24 repeated, one stream
per CU: the only regime that scales; also available as the metric with --scaling throughput).  --scaling weak: every
rank its own copy of kodak24.

`roofline` = the dominant kernel (entropy, a latency chain: its HBM fraction only says how far from that roof it sits);
`roofline_float_stages` = the fused float kernel against the fp32 peak (it is compute-bound: 4.33 B/px, ~1.7 kflop/px)
and as algorithmic GB/s; `rate_model` = the one genuinely HBM-bound kernel of the build (16 B / symbol) against the HBM
peak; `cpu_baseline` = the CPU oracle (C port of the reference's algorithm) MEASURED on the host cores: every stream of
the set at once, one oracle call per core (`cores` = threads used), with the one-core figure beside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
SYMBOL_FLOOR_TICKS = 111.0  # bare unrolled symbol block of the entropy kernel's decoder (paired tests), ticks per symbol (tools/ubench/dcycle.hip)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PROFILE_DIRS = ["profiles/r06", "profiles/r05", "profiles/r04", "profiles/r03", "profiles/r02", "profiles/r01"]


def build_kodak24(device: int = 0):
    """Kept for tools/: 24 (cc_header, bytes_nn, bytes_latent, (H, W)) tuples + the raw stream bytes."""
    from cool_chic_amd import synth

    streams, sizes = synth.kodak24()
    return [(*synth.split_image_stream(s), hw) for s, hw in zip(streams, sizes)], streams


def cpu_sample(streams, px_each, budget_s: float, what: str, parallel: bool = True):
    """The oracle (plain-C restatement of the reference algorithm, one thread per stream) timed on this host:
    (a) ONE core on a bounded sample of `streams` (the first streams until `budget_s` is spent);
    (b) `parallel`: EVERY stream of the set at once, one oracle call per stream on min(cores, streams) threads (the calls are
        ctypes calls into libcc_oracle.so: they release the GIL and run concurrently; threads, not processes, because this
        process holds a HIP context).  Streams are independent, so this is the CPU's whole-set throughput, measured.
    `value` / `cores` are (b) when it ran, else (a)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle_py

    oracle_py.build()
    t0 = time.perf_counter()
    px = n = 0
    for s, p in zip(streams, px_each):
        oracle_py.decode_video(s)
        px += p
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    one = {"value": px / dt / 1e6, "unit": "Mpixel/s", "cores": 1,
           "sample": f"first {n} of {len(streams)} {what}, full decode to integer planes, {dt:.1f} s"}
    out = {"value": one["value"], "unit": "Mpixel/s", "cores": 1, "kind": "port", "sample": one["sample"], "host_cores_available": cores}
    n_thr = min(cores, len(streams))
    if parallel and n_thr > 1:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=n_thr) as ex:
            list(ex.map(oracle_py.decode_video, streams))
        dt_all = time.perf_counter() - t0
        out.update({"value": sum(px_each) / dt_all / 1e6, "cores": n_thr,
                    "sample": f"all {len(streams)} {what} at once, one oracle call per stream on {n_thr} threads (one per core, "
                              f"{cores} cores available), full decode to integer planes, {dt_all:.1f} s wall",
                    "one_core": one})
    return out


def event_ms(stream, fn, reps: int, device: int) -> float:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / reps


def wall_ms(fn, reps: int, device: int) -> float:
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / reps * 1e3


HASHES_PATH = os.path.join(ROOT, "tests", "golden", "workload_hashes.json")
_hashes = None


def expected_hashes(name):
    """Per stream of workload `name`: the sha256 of the integer planes the CPU oracle decodes from it (one per frame),
    generated in the build container by tests/golden/gen/hash_workloads.py; None when the file is missing."""
    global _hashes
    if _hashes is None:
        try:
            with open(HASHES_PATH) as f:
                _hashes = json.load(f)
        except (OSError, ValueError):
            _hashes = {}
    return _hashes.get(name)


def verify_frames(name, frames, stream_of=lambda i: i, video=False, streams=None):
    """What was timed is what the oracle decodes: `frames` = [planes of frame i] (copied to the host AFTER the timed
    region) hashed like tests/golden/gen/hash_workloads.py hashes the oracle's planes.  Image workloads: frame i is
    stream stream_of(i); a video workload is one stream whose frames come in display order."""
    import hashlib

    from cool_chic_amd.synth import planes_sha256

    exp = expected_hashes(name)
    got = [planes_sha256(p) for p in frames]
    out = {"frames_checked": len(got), "planes_sha256_of_set": hashlib.sha256("".join(got).encode()).hexdigest()[:32],
           "against": "tests/golden/workload_hashes.json: integer planes of the CPU oracle for the same streams (hash_workloads.py)"}
    if exp is None:
        out.update({"ok": None, "note": "no expected hashes for this workload"})
        return out
    want = list(exp["planes_sha256"][0]) if video else [exp["planes_sha256"][stream_of(i)][0] for i in range(len(got))]
    bad = [i for i in range(min(len(got), len(want))) if got[i] != want[i]]
    out.update({"ok": not bad and len(got) == len(want), "mismatching_frames": bad[:8]})
    if streams is not None:  # the inputs themselves are the ones the oracle saw
        out["streams_identical"] = all(hashlib.sha256(st).hexdigest() == exp["streams_sha256"][0 if video else stream_of(i)]
                                       for i, st in enumerate(streams))
    return out


def measure_traffic_live(timeout_s: float = 150.0):
    """HBM traffic per launch of the workload's kernels from the PMC counters, collected the way MI355X_MICROARCH.md prescribes
    (one rocprofv3 --pmc pass per counter, never combined with other traces) on `tools/prof_workload.py kodak24 2`, with the
    gfx950 corrections measured by tools/pmc_calibrate.sh.  {} when rocprofv3 is missing or fails (the caller falls back to the
    tracked profile)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}
    out = {}
    names = {"entropy_pipe_kernel": "entropy_pipe_kernel", "decode_fused_kernel": "decode_fused_kernel"}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "tools", "prof_workload.py"), "kodak24", "2", "keep_float"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            except (subprocess.TimeoutExpired, OSError):
                return {}
            if r.returncode != 0:
                return {}
            vals = defaultdict(list)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter:
                            vals[row["Kernel_Name"]].append(float(row["Counter_Value"]) * 1024.0 * factor)
            for short, sub in names.items():
                # every kernel of the step whose name holds `sub` (the fused float path is two launches: pyramid + tiles; r06: per
                # chain group): mean over the launches of each (first dropped) x its launches per step, summed over the kernels
                key = "fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"
                tot = 0.0
                for k, v in vals.items():
                    if sub in k and len(v) > 1:
                        per_step = max(1, round(len(v) / 3))  # prof_workload runs 1 warm-up + 2 timed steps
                        kb = sum(v[per_step:]) / max(1, len(v[per_step:])) * per_step
                        tot += kb
                        out.setdefault(short, {}).setdefault("per_kernel", {}).setdefault(k.replace("ccd::(anonymous namespace)::", "")[:70], {})[key] = kb
                if tot:
                    out.setdefault(short, {})[key] = tot
    return out if all("fetch_bytes" in v and "write_bytes" in v for v in out.values()) and out else {}


def n_symbols(batch, n):
    return int(sum(batch.header(s).n_symbols for s in range(n)))


def image_leg(name, device, sh, stream, steps, cpu_budget, want_cpu):
    """One image-set configuration decoded in one batch on this GPU (inputs resident, planes left in HBM)."""
    from cool_chic_amd import DecodeBatch, synth

    t0 = time.perf_counter()
    wl = synth.workload(name)
    t_build = time.perf_counter() - t0
    triples = [synth.split_image_stream(s) for s in wl["streams"]]
    px = [h * w for h, w in wl["sizes"]]
    b = DecodeBatch(device)
    for hdr, nn, lat in triples:
        b.add(hdr, nn, lat, 8, 0)
    b.run(sh)
    b.wait(sh)
    kernels = [b.slot_kernels(s) for s in range(len(triples))]
    launches = b.entropy_launches()
    ms = wall_ms(lambda: b.run(sh), steps, device)
    ms_entropy = event_ms(stream, lambda: b.run(sh, stage=0), max(1, steps // 2), device)
    ms_float = event_ms(stream, lambda: (b.run(sh, stage=1), b.run(sh, stage=2)), max(1, steps // 2), device)
    b.wait(sh)
    nsym = n_symbols(b, len(triples))
    stats = [b.slot_stats(s) for s in range(len(triples))]
    verified = verify_frames(name, [b.planes(i) for i in range(len(triples))], streams=wl["streams"])
    hdrs = [b.header(i) for i in range(len(triples))]
    b.close()
    leg = {"frames": len(triples), "mpixels": sum(px) / 1e6, "verified": verified, "value": sum(px) / ms / 1e3, "unit": "Mpixel/s", "n_gpus": 1, "steps": steps,
           "ms_per_step": ms, "entropy_ms": ms_entropy, "float_ms": ms_float,
           # r06: a run puts the float launches of the streams that finish early behind THEIR entropy launch (chain groups): what
           # of the float path is still exposed behind the slowest entropy launch
           "float_ms_exposed": ms - ms_entropy, "entropy_launches": launches, "symbols": nsym,
           "entropy_msym_per_s": nsym / ms_entropy / 1e3, "largest_frame_mpx": max(px) / 1e6,
           "slots_on_generic_entropy_kernel": sum(1 for k in kernels if not k & 1),
           "slots_on_unfused_float_path": sum(1 for k in kernels if not k & 4),
           "entropy_kernel_widths_nv": sorted({(b_.total_context_arm + 3) // 4 for b_ in hdrs}),
           "bpp": 8.0 * sum(len(t[2]) for t in triples) / sum(px),
           # symbols that left the decoder's common path (window misses / sentinels), and the full 128-way searches among them
           "rare_path_symbols": int(sum(int(st[62]) for st in stats)), "full_searches": int(sum(int(st[63]) for st in stats)),
           "stream_bytes": int(sum(len(s) for s in wl["streams"])), "build_s": round(t_build, 1),
           "note": "one batch, one workgroup (one CU) per stream: the step time is the slowest stream's serial chain"}
    if want_cpu:
        order = np.argsort(px)[::-1]  # the largest picture first: the sample is then representative of the set's bulk
        leg["cpu_baseline"] = cpu_sample([wl["streams"][i] for i in order], [px[i] for i in order], cpu_budget, f"{name} streams")
    return leg


def gop_leg(device, sh, stream, steps, cpu_budget, want_cpu):
    """BASELINE configs[3]: the 33-frame 1080p GOP.  (a) every cool-chic (66) resident in one batch: entropy + float
    stages; (b) ccd_decode_video from the stream bytes: parse + upload + decode + reconstruction in coding order + planes
    back on the host."""
    from cool_chic_amd import DecodeBatch, synth
    from cool_chic_amd._lib import Video, check, lib
    from cool_chic_amd.bitstream.decode import _split_frame
    from cool_chic_amd.bitstream.header import VideoHeader

    t0 = time.perf_counter()
    wl = synth.workload("gop1080p33")
    t_build = time.perf_counter() - t0
    bs = wl["streams"][0]
    n_frames = wl["info"]["frames"]
    H, W = wl["info"]["size"]
    px = n_frames * H * W
    vh = VideoHeader()
    rest = vh.read_header(bs)
    b = DecodeBatch(device)
    n_cc = 0
    for _ in range(n_frames):
        fh, ccs, rest = _split_frame(rest)
        intra = fh.get_value("frame_type") == "I"
        for ch, nn, lat in ccs:
            b.add(ch.raw, nn, lat, fh.get_value("bitdepth") if intra else 0, 1 if intra else 0)
            n_cc += 1
    b.run(sh)
    b.wait(sh)
    kernels = [b.slot_kernels(s) for s in range(n_cc)]
    ms_cc = wall_ms(lambda: b.run(sh), steps, device)
    ms_entropy = event_ms(stream, lambda: b.run(sh, stage=0), max(1, steps // 2), device)
    ms_float = event_ms(stream, lambda: (b.run(sh, stage=1), b.run(sh, stage=2)), max(1, steps // 2), device)
    b.wait(sh)
    nsym = n_symbols(b, n_cc)
    b.close()

    def whole():
        v = Video()
        check(lib().ccd_decode_video(bs, len(bs), device, C.byref(v)), "ccd_decode_video")
        lib().ccd_video_free(C.byref(v))

    whole()
    ms_e2e = wall_ms(whole, max(3, steps), device)  # (0.18 s each; one sample showed 190 against 178 ms: a hiccup of a single call)
    # what ccd_decode_video returns, against the oracle's planes for the same stream
    v = Video()
    check(lib().ccd_decode_video(bs, len(bs), device, C.byref(v)), "ccd_decode_video")
    frames_v = []
    for i in range(v.n_frames):
        f = v.frames[i]
        shapes = [(f.h, f.w), (f.ch, f.cw), (f.ch, f.cw)]
        frames_v.append([np.ctypeslib.as_array(f.plane[p], shape=shapes[p]).copy() for p in range(3)])
    lib().ccd_video_free(C.byref(v))
    verified = verify_frames("gop1080p33", frames_v, video=True, streams=[bs])
    leg = {"frames": n_frames, "cool_chics": n_cc, "verified": verified, "mpixels": px / 1e6, "value": px / ms_e2e / 1e3, "unit": "Mpixel/s", "n_gpus": 1,
           "ms_per_step": ms_e2e, "what": "ccd_decode_video from the stream bytes: parse, upload, all cool-chics in one batch, "
           "reconstruction in coding order, integer planes back on the host",
           "resident_coolchics_ms": ms_cc, "resident_coolchics_mpx_per_s": px / ms_cc / 1e3, "entropy_ms": ms_entropy, "float_ms": ms_float,
           "symbols": nsym, "entropy_msym_per_s": nsym / ms_entropy / 1e3,
           "slots_on_generic_entropy_kernel": sum(1 for k in kernels if not k & 1),
           "slots_on_unfused_float_path": sum(1 for k in kernels if not k & 4),
           "stream_bytes": len(bs), "build_s": round(t_build, 1), "coding_order": wl["info"]["coding_order"][:9] + ["..."]}
    if want_cpu:
        # a 3-frame GOP of the same cool-chics (I0 I2 B1) is the bounded CPU sample
        small, info = synth.gop1080p(2)
        leg["cpu_baseline"] = cpu_sample([small], [info["frames"] * H * W], cpu_budget, "3-frame 1080p GOP (I0 I2 B1)", parallel=False)
    return leg


def timed_set(mine, world, rank, local_rank, backend, red_dev, steps, warmup, force_gather=False):
    """One image set, this rank's share of it: `mine` = [(cc_header, bytes_nn, bytes_latent, (H, W))].  Inputs go to HBM, then
    `warmup` untimed and `steps` timed steps; a step = decode of every frame of the share + (N > 1) the gather of the decoded
    integer planes to rank 0 inside the timed region.  Barrier + device synchronisation on both sides, MAX over ranks.
    Returns {"dt", "batch" (still open), "gathered" (rank 0: per-rank byte messages of the last step), "stream", "sh"}."""
    from cool_chic_amd import DecodeBatch
    from cool_chic_amd.parallel import EqualSizeGather

    dev = f"cuda:{local_rank}"
    batch = DecodeBatch(local_rank)
    for hdr, nn, lat, _ in mine:
        batch.add(hdr, nn, lat, 8, 0)
    batch.time_launches()  # HIP events around every entropy launch ON THE STREAM IT RUNS ON (two records per launch): the roofline's ms_per_launch
    stream = torch.cuda.current_stream(local_rank)
    sh = stream.cuda_stream
    n = len(mine)
    planes = [torch.as_tensor(batch.plane_device(s_, p), device=dev).reshape(-1) for s_ in range(n) for p in range(3)]
    n_bytes = sum(int(p.numel()) * p.element_size() for p in planes)
    gatherer = None
    if world > 1 or force_gather:  # equal message sizes: pad to the largest share (force_gather: the one-rank nccl test of the wire path)
        t = torch.tensor([n_bytes], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pad = int(t.item()) - n_bytes
        if pad or not planes:
            planes.append(torch.zeros(pad, dtype=torch.uint8, device=dev))
        gatherer = EqualSizeGather(int(t.item()), dev, dst=0)
    gathered = [None]

    def step():
        if n:
            batch.run(sh)
        if gatherer is not None:  # decoded integer planes of this rank's frames -> writer rank (RCCL over xGMI), inside the timed region
            gathered[0] = gatherer(planes)

    def fence():
        if world > 1 or force_gather:
            dist.barrier()
        torch.cuda.synchronize(local_rank)

    for _ in range(warmup):
        step()
    batch.wait(sh)  # raises on decode errors
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1 or force_gather:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    batch.wait(sh)
    return {"dt": dt, "batch": batch, "gathered": gathered[0], "stream": stream, "sh": sh,
            "launch_ms": batch.launch_ms() if n else []}  # the entropy launches of the LAST TIMED step


def timed_from_bytes(mine, world, rank, local_rank, backend, red_dev, steps, warmup, force_gather=False):
    """The metric's step, from the stream BYTES: `mine` = [(.cool bytes of one picture in host memory, (H, W))] -> integer planes
    in pinned host memory (on rank 0 for every rank's frames when world > 1).  A step = batch creation, per-stream header /
    network parsing + fixed-point conversion + staged asynchronous uploads (ccd_batch_add), the decode, and the planes' way
    back: one device -> host copy per frame (N = 1), or the gather of every rank's planes to rank 0 over RCCL and their copy
    to rank 0's pinned memory (N > 1).  Nothing is cached between steps: every step parses the same bytes again.

    TWO steps are in flight: while the GPU decodes set k, the host creates and fills the batch of set k + 1 (its uploads
    travel on the library's upload stream) and set k - 1's planes are still on their way to the host.  The DECODES do not
    overlap - set k + 1's kernels wait for an event behind set k's (two sets decoding at once would be the 48-streams-in-flight
    regime, not this configuration) - so the step time is the resident decode + what does not hide behind it.  The two sets
    alternate between two HIP streams because ccd_batch_destroy drains the streams its batch used.
    Returns {"dt", "planes" (N = 1: [frame][plane] host arrays of the last step), "gathered" (rank 0, N > 1: per-rank byte
    messages of the last step)}."""
    from cool_chic_amd import DecodeBatch, synth
    from cool_chic_amd.parallel import EqualSizeGather

    dev = f"cuda:{local_rank}"
    lanes = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    gather = world > 1 or force_gather
    n = len(mine)
    n_bytes = sum(3 * h * w for _, (h, w) in mine)  # 8-bit RGB planes
    msg, pad, gatherers = n_bytes, None, [None, None]
    if gather:  # equal message sizes: pad to the largest share
        t = torch.tensor([n_bytes], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        msg = max(int(t.item()), 1)
        if msg > n_bytes:
            pad = torch.zeros(msg - n_bytes, dtype=torch.uint8, device=dev)
        gatherers = [EqualSizeGather(msg, dev, dst=0) for _ in lanes]
    host = [None, None]       # pinned destination per lane
    prev_decoded = [None]     # event behind the decode kernels of the step launched last

    def launch(k):
        lane = k & 1
        st = lanes[lane]
        b = DecodeBatch(local_rank, keep_float=False)
        for s_, _ in mine:
            b.add(*synth.split_image_stream(s_), 8, 0)
        b.prepare(st.cuda_stream)           # the launch tables' upload: not behind the wait below
        if prev_decoded[0] is not None:
            st.wait_event(prev_decoded[0])  # decodes one after the other; everything else overlaps
        if n:
            b.run(st.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(st)
        prev_decoded[0] = ev
        layout = None
        if gather:
            with torch.cuda.stream(st):
                planes = [torch.as_tensor(b.plane_device(s_, p), device=dev).reshape(-1) for s_ in range(n) for p in range(3)]
                if pad is not None:
                    planes.append(pad)
                bucket = gatherers[lane](planes)  # decoded planes of this rank's frames -> writer rank (RCCL over xGMI)
                if bucket is not None:
                    if host[lane] is None:
                        host[lane] = torch.empty(world * msg, dtype=torch.uint8, pin_memory=True)
                    for r, bk in enumerate(bucket):
                        host[lane][r * msg:(r + 1) * msg].copy_(bk, non_blocking=True)
        elif n:
            total, layout = b.planes_layout()
            if host[lane] is None or host[lane].numel() < total:
                host[lane] = torch.empty(total, dtype=torch.uint8, pin_memory=True)
            b.copy_planes_async(host[lane].data_ptr(), layout, st.cuda_stream)
        return b, lane, layout

    def finish(job, close=True):
        b, lane, _ = job
        b.wait(lanes[lane].cuda_stream)  # this lane only: the decode, the copies; raises on a decode error
        lanes[lane].synchronize()
        if close:
            b.close()                     # arenas back to the pool (drains this lane's stream only)

    def fence():
        if gather:
            dist.barrier()
        torch.cuda.synchronize(local_rank)

    k, pending = 0, None
    for _ in range(warmup):
        cur = launch(k); k += 1
        if pending is not None:
            finish(pending)
        pending = cur
    if pending is not None:
        finish(pending)
        pending = None
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        cur = launch(k); k += 1
        if pending is not None:
            finish(pending)
        pending = cur
    finish(pending, close=False)
    fence()
    dt = time.perf_counter() - t0
    if gather:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    b, lane, layout = pending
    out = {"dt": dt, "planes": None, "gathered": None}
    if gather:
        if host[lane] is not None:
            out["gathered"] = [host[lane][r * msg:(r + 1) * msg].clone() for r in range(world)]
    elif n:
        out["planes"] = [[p.copy() for p in fr] for fr in b.plane_views(host[lane].numpy(), layout)]
    b.close()
    return out


def verify_gathered(name, gathered, frames_of_rank):
    """Rank 0: the planes of EVERY rank's frames as they arrived in the gather of the last timed step, against the oracle's
    hashes.  frames_of_rank(r) = [(stream index in the workload, (H, W))] in the order rank r packed them (8-bit RGB)."""
    frames_g, index_g = [], []
    for r, msg in enumerate(gathered):
        host = msg.cpu().numpy()
        off = 0
        for idx, (h, w) in frames_of_rank(r):
            frames_g.append([host[off + p * h * w: off + (p + 1) * h * w].reshape(h, w) for p in range(3)])
            index_g.append(idx)
            off += 3 * h * w
    return verify_frames(name, frames_g, stream_of=lambda i: index_g[i])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["strong", "weak", "throughput"], default="strong",
                    help="strong (default): BASELINE's kodak24, its 24 frames round-robin over the ranks; weak: every rank its own "
                         "kodak24; throughput: every rank its own 256 streams (kodak24 repeated: one per CU)")
    ap.add_argument("--legs", default="all", help="comma list of clic41,gop1080p33,uhd4k,wide,png,e2e,float,envelope,rate,kodak24_hq,clic41_alt,cliffs (rank 0, N = 1) and sharded (N > 1: "
                    "clic41_sharded + throughput_regime with all ranks), or all / none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes (roofline.traffic then "
                    "comes from the tracked profile)")
    ap.add_argument("--allow-variant", action="store_true", help="time the library CCD_LIB names (a profiling / experiment build of "
                    "_build.build_variant) instead of refusing: the line then says which one")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: nccl = RCCL over xGMI (one GPU per rank); gloo = host-staged exchange, ranks may share a GPU "
                         "(smoke run of the multi-rank path on a single-GPU box)")
    args = ap.parse_args()

    if os.environ.get("CCD_LIB") and not args.allow_variant:
        sys.exit("bench.py: CCD_LIB=%s selects a variant build of the library; the benchmark times cool_chic_amd/libccd.so. "
                 "Unset it, or pass --allow-variant to time the variant on purpose." % os.environ["CCD_LIB"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X: there is no CPU fallback"
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # ranks may share a GPU: there is no device-to-device collective to collide
    if args.backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
        # a launcher that narrows the visibility per rank (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES: every rank sees ONE GPU as
        # device 0): take what is visible; that the ranks really sit on N different GPUs is checked by their UUIDs below
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)  # BEFORE the communicator exists: a rank on the wrong GPU hangs at the first collective
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL P2P buffers)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
    red_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"  # where the scalar reductions live
    # the run is what the command line says: --gpus N ranks (the driver computes scaling from N), each on its own GPU
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (tools/run_8gpu.sh)"
    gpus_active = 1
    if world > 1:
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank
        # (host, device) of every rank: N distinct GPUs - two ranks on one device would halve the numbers silently
        import socket
        who = [None] * world
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        # the physical GPU, whatever its index here (uuid + PCI address; a torch without either falls back to the rank: no check)
        gpu_id = f"{getattr(props, 'uuid', '')}/{getattr(props, 'pci_domain_id', '')}:{getattr(props, 'pci_bus_id', '')}:{getattr(props, 'pci_device_id', '')}"
        if gpu_id == "/::":
            gpu_id = f"rank{rank}"
        dist.all_gather_object(who, (socket.gethostname(), gpu_id if args.backend == "nccl" else rank))
        gpus_active = len(set(who))
        assert gpus_active == world, f"{world} ranks on {gpus_active} distinct GPUs: {who}"
    all_legs = ["clic41", "gop1080p33", "uhd4k", "wide", "png", "e2e", "float", "envelope", "rate", "kodak24_hq", "clic41_alt", "cliffs"]
    legs = all_legs if args.legs == "all" else ([] if args.legs == "none" else args.legs.split(","))
    if world > 1 and args.legs == "all":
        legs = ["sharded"]  # the other configurations are single-GPU legs of the N = 1 run; with N ranks: the sharded sets
    want_cpu = not args.no_cpu_baseline

    from cool_chic_amd import DecodeBatch, synth
    from cool_chic_amd._lib import lib as lib_
    from cool_chic_amd.parallel import EqualSizeGather, shard_indices

    items, streams = build_kodak24(local_rank)
    n_kodak = len(items)
    kodak_px = sum(h * w for *_, (h, w) in items)
    THROUGHPUT_STREAMS = 256  # one stream per CU: a stream is one 512-thread workgroup that owns its CU (139 KB of LDS, 230 VGPRs)
    t_ids = [i % n_kodak for i in range(THROUGHPUT_STREAMS)]
    t_px = sum(items[i][3][0] * items[i][3][1] for i in t_ids)
    if args.scaling == "strong":      # BASELINE's set itself, sharded: frame i -> rank i mod N
        ids_of = lambda r: shard_indices(n_kodak, r, world)
        px_per_step = kodak_px
    elif args.scaling == "weak":      # every rank its own copy of the set
        ids_of = lambda r: list(range(n_kodak))
        px_per_step = world * kodak_px
    else:                             # throughput: enough streams per GPU to give every CU one
        ids_of = lambda r: t_ids
        px_per_step = world * t_px
    mine = [items[i] for i in ids_of(rank)]
    n_frames = len(mine)
    dev = f"cuda:{local_rank}"
    # ---- the metric (the bench contract: inputs resident in HBM when the timed region starts, planes left in HBM; N > 1: gathered
    # to rank 0's HBM); its batch serves the per-stage timing below
    run = timed_set(mine, world, rank, local_rank, args.backend, red_dev, args.steps, args.warmup)
    dt, batch, stream, sh = run["dt"], run["batch"], run["stream"], run["sh"]
    launch_ms = run["launch_ms"]
    gathered = [run["gathered"]]
    # ---- beside it, same steps / warmup, all ranks: from the .cool BYTES in host memory to integer planes in pinned host memory,
    # two sets in flight (`from_bytes`: what a caller who holds files gets; the gap to `value` is what does not hide behind the decode)
    fb = timed_from_bytes([(streams[i], items[i][3]) for i in ids_of(rank)], world, rank, local_rank, args.backend, red_dev, args.steps, args.warmup)

    # ---- N > 1: the sets that BASELINE shards, and the regime that scales, measured with all ranks (collective: every rank runs this)
    sharded = {}
    if world > 1 and "sharded" in legs:
        # (a) BASELINE configs[2]: the 41 CLIC pictures round-robin over the ranks; each rank manufactures only its own streams
        c_ids = shard_indices(len(synth.CLIC41_SIZES), rank, world)
        c_streams = synth.clic41_subset(c_ids)
        c_mine = [(*synth.split_image_stream(st_), synth.CLIC41_SIZES[i]) for st_, i in zip(c_streams, c_ids)]
        n_c = 3
        rc = timed_set(c_mine, world, rank, local_rank, args.backend, red_dev, n_c, 1)
        c_px = sum(h * w for h, w in synth.CLIC41_SIZES)
        if rank == 0:
            sharded["clic41_sharded"] = {
                "value": c_px * n_c / rc["dt"] / 1e6, "unit": "Mpixel/s", "n_gpus": world, "steps": n_c, "ms_per_step": rc["dt"] / n_c * 1e3,
                "scaling": "strong", "frames": len(synth.CLIC41_SIZES), "frames_on_rank0": len(c_mine),
                "expected_scaling": "flat beyond ~2 GPUs: the step is the serial chain of the largest (2.8 Mpx) stream on one CU; 41 streams occupy 16 % of ONE GPU",
                "verified": verify_gathered("clic41", rc["gathered"], lambda r: [(i, synth.CLIC41_SIZES[i]) for i in shard_indices(len(synth.CLIC41_SIZES), r, world)])}
        rc["batch"].close()
        # (b) the regime that scales: every rank its own 256 streams (weak)
        t_mine = [items[i] for i in t_ids]
        n_t = 3
        rt = timed_set(t_mine, world, rank, local_rank, args.backend, red_dev, n_t, 1)
        if rank == 0:
            sharded["throughput_regime"] = {
                "value": world * t_px * n_t / rt["dt"] / 1e6, "unit": "Mpixel/s", "n_gpus": world, "steps": n_t,
                "ms_per_step": rt["dt"] / n_t * 1e3, "scaling": "weak", "streams_per_gpu": len(t_mine),
                "note": "every rank decodes its own 256 streams (kodak24 repeated: one stream per CU) and rank 0 gathers all planes inside the timed region: "
                        "the only regime of this format that scales over GPUs (expected ~N x)",
                "verified": verify_gathered("kodak24", rt["gathered"], lambda r: [(i, items[i][3]) for i in t_ids])}
        rt["batch"].close()

    res = None
    if rank == 0:
        # ---- what was timed, against the CPU oracle (hashes of its integer planes for the same 24 streams): this rank's own
        # frames from its batch, and - N > 1 - every rank's frames as they arrived in the gather of the last timed step
        my_ids = ids_of(rank)
        verified = verify_frames("kodak24", [batch.planes(s_) for s_ in range(n_frames)], stream_of=lambda k: my_ids[k],
                                 streams=streams if n_frames == n_kodak else None)
        if world > 1 and gathered[0] is not None:
            verified["gathered"] = verify_gathered("kodak24", gathered[0], lambda r: [(i, items[i][3]) for i in ids_of(r)])
        # the planes of the LAST TIMED from-bytes step as they lie in pinned host memory: rank 0's own (N = 1), every rank's (N > 1)
        if fb["planes"] is not None:
            fb_verified = verify_frames("kodak24", fb["planes"], stream_of=lambda k: my_ids[k])
        else:
            fb_verified = verify_gathered("kodak24", fb["gathered"], lambda r: [(i, items[i][3]) for i in ids_of(r)])
        # ---- per-stage timing with HIP events on the launch stream (roofline evidence); stage 1 (per-level upsampling) is
        # empty on the fused path
        stage_ms = {name: event_ms(stream, lambda st=st: batch.run(sh, stage=st), args.steps, local_rank)
                    for st, name in ((0, "entropy"), (1, "pyramid_launch"), (2, "fused_float"))}
        kernels = [batch.slot_kernels(s) for s in range(n_frames)]
        hdr0 = batch.header(0)
        nsym = n_symbols(batch, n_frames)
        n_payload = sum(len(lat) for _, _, lat, _ in mine)
        L, c_out = hdr0.input_feature_synthesis, hdr0.out_channels
        rank_px = sum(h * w for *_, (h, w) in mine)
        def latent_px(b_, n_):  # samples of the latent grids that reach the upsampling (hyperlatents do not)
            tot = 0
            for s_ in range(n_):
                h_ = b_.header(s_)
                tot += sum(h_.grid_h[g] * h_.grid_w[g] for g in range(h_.n_grids) if not h_.is_hyperlatent[g])
            return tot

        n_lat_px = latent_px(batch, n_frames)

        pmc, pmc_from = {}, None
        if args.scaling != "throughput" and world == 1 and not args.no_live_traffic:
            pmc = measure_traffic_live()  # two short rocprofv3 --pmc passes of the same workload, outside every timed region
            if pmc:
                pmc_from = "measured in this run"
        for d in ([] if pmc else PROFILE_DIRS):
            try:
                with open(os.path.join(ROOT, d, "kodak24_pmc_traffic.json")) as f:
                    pmc, pmc_from = json.load(f)["kernels"], d + "/kodak24_pmc_traffic.json"
                break
            except (OSError, ValueError, KeyError):
                continue

        def traffic(name):
            k = pmc.get(name, {})
            return k["fetch_bytes"] + k["write_bytes"] if "fetch_bytes" in k and "write_bytes" in k else None

        ent_bytes = n_payload + nsym  # payload in + one byte per symbol out, all streams of the step
        # r06: a step's streams are decoded by SEVERAL concurrent launches of the kernel (chain groups on side streams): the roofline
        # is per launch - algorithmic bytes of a launch's streams over that launch's duration, HIP events on its own stream inside
        # the timed region (last timed step), mean over the step's launches (= what rocprofv3 --stats averages)
        n_l = max(1, len(launch_ms))
        ent_ms_launch = sum(m for m, _ in launch_ms) / n_l if launch_ms else stage_ms["entropy"]
        ent_ach = ent_bytes / n_l / (ent_ms_launch / 1e3) / 1e9
        # fused float kernel: algorithmic bytes = int8 latents in + (4 C f32 +) C integer samples out; flops as executed by
        # the reference's 2-D kernels: synthesis 2 x 672 (HOP) + upsampling ~380 per pixel
        flop_px = 2.0 * 672 + 380.0
        ff_bytes = n_lat_px + (4 * c_out + c_out) * rank_px
        ff_ms = stage_ms["pyramid_launch"] + stage_ms["fused_float"]  # both launches of the float path
        float_lines = [{
            "kernel": f"decode_fused_kernel<{L - 1},2,pyr> (latent levels >= 1, once per frame) + decode_fused_kernel<{L},{c_out},pre> "
                      f"(level 0 + synthesis + integer samples), {n_frames} frames, two launches",
            "ms_pyramid_launch": stage_ms["pyramid_launch"], "ms_fused_kernel": stage_ms["fused_float"],
            "bound": "fp32", "achieved": flop_px * rank_px / ff_ms / 1e9, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop_px * rank_px / ff_ms / 1e9 / FP32_PEAK_TFLOPS, "ms_per_launch": ff_ms,
            "algorithmic_gbs": ff_bytes / ff_ms / 1e6, "frac_of_hbm_peak": ff_bytes / ff_ms / 1e6 / HBM_PEAK_GBS,
            "algorithmic_bytes": ff_bytes, "traffic": traffic("decode_fused_kernel"),
            "traffic_by_kernel": pmc.get("decode_fused_kernel", {}).get("per_kernel"),
            "note": "compute-bound by construction (%.1f B/px, ~%.0f flop/px): priced against the fp32 peak; exact fmaf chains on "
                    "v_mfma_f32_4x4x1 (bitwise the oracle's order)" % (ff_bytes / rank_px, flop_px)}]
        # r06: the step launches the float path once per chain group (two smaller launches each on kodak24, so that the 18 landscape
        # frames are synthesised while the 6 portrait streams still decode): the line above prices THOSE launches; beside it the same
        # kernels as ONE launch each over all frames (overlap off) - the kernel's own efficiency, comparable with earlier rounds
        if world == 1 and n_frames:
            b1 = DecodeBatch(local_rank, overlap=False)
            for hdr, nn, lat, _ in mine:
                b1.add(hdr, nn, lat, 8, 0)
            b1.run(sh); b1.wait(sh)
            one = {name: event_ms(stream, lambda st=st: b1.run(sh, stage=st), args.steps, local_rank) for st, name in ((1, "pyramid_launch"), (2, "fused_float"))}
            b1.close()
            ff1 = one["pyramid_launch"] + one["fused_float"]
            float_lines[0]["note"] += "; launched per chain group (%d groups): see the next entry for one launch over all frames" % batch.entropy_launches()
            float_lines.append({"kernel": float_lines[0]["kernel"].replace("two launches", "ONE launch per kernel over all frames (CCD_OPT_OVERLAP = 0)"),
                                "ms_pyramid_launch": one["pyramid_launch"], "ms_fused_kernel": one["fused_float"], "bound": "fp32",
                                "achieved": flop_px * rank_px / ff1 / 1e9, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": flop_px * rank_px / ff1 / 1e9 / FP32_PEAK_TFLOPS, "ms_per_launch": ff1,
                                "algorithmic_gbs": ff_bytes / ff1 / 1e6, "frac_of_hbm_peak": ff_bytes / ff1 / 1e6 / HBM_PEAK_GBS})
        res = {
            "metric": "decoded Mpixel/s", "value": px_per_step * args.steps / dt / 1e6, "unit": "Mpixel/s",
            "n_gpus": world, "gpus_active": gpus_active, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            **({"library_variant": os.environ["CCD_LIB"]} if os.environ.get("CCD_LIB") else {}),
            "higher_is_better": True, "scaling": "strong" if args.scaling == "strong" else "weak", "vs_baseline": None,
            "dtype": "int64+f64 entropy / f32 synthesis",
            "data": "synthetic (kodim14.cool real + 23 streams re-encoded from rolled/transposed kodim14 latents)",
            "config": {"workload": {"strong": "kodak24", "weak": "kodak24 (every rank its own copy)",
                                    "throughput": f"{THROUGHPUT_STREAMS} streams in flight per GPU (kodak24 repeated: one per CU)"}[args.scaling],
                       "frames_on_rank0": n_frames, "frame": "512x768 RGB 8-bit, HOP decoder", "symbols_per_step_rank0": nsym,
                       "parallelism": (f"frame i -> rank i mod {world} (the set's 24 frames sharded), gather of planes to rank 0" if args.scaling == "strong"
                                       else f"every rank its own frames x{world}, gather of planes to rank 0")},
            # BASELINE's sets cannot strong-scale, by the format: a stream is ONE serial range-decoder chain on one CU and the step is the
            # slowest stream's, so 24 frames on N GPUs take as long as on one (they occupy 9 % of one GPU's CUs); see throughput_regime
            "expected_scaling": ("flat: one serial chain per stream - the step time is the slowest stream's on any number of GPUs (DESIGN.md 6)"
                                 if args.scaling == "strong" else "~N x: every rank decodes its own frames"),
            "parity": "integer stages bit-exact vs reference fixtures; float stages bit-exact vs CPU oracle; integer planes <=1 LSB on "
                      "<=2e-5 of samples vs the reference decoder's output = within the reference's own thread-count noise floor "
                      "(tests/test_gpu_parity.py)",
            "verified": verified,
            # the same K steps from the stream BYTES: what a caller who holds .cool files gets (the contract keeps `value` on resident
            # inputs; r04's single-shot figure of this was 5.4 % below it)
            "from_bytes": {"value": px_per_step * args.steps / fb["dt"] / 1e6, "unit": "Mpixel/s", "ms_per_step": fb["dt"] / args.steps * 1e3,
                           "ratio_to_value": dt / fb["dt"], "steps": args.steps, "verified": fb_verified,
                           "what": "every step: .cool bytes in host memory -> batch creation, header / network parsing, fixed-point conversion, "
                                   "staged uploads, decode, integer planes into pinned host memory (N > 1: gathered to rank 0 over RCCL, then "
                                   "into its pinned memory); nothing cached between steps; TWO sets in flight - parse + upload of set k+1 and "
                                   "the copy-back of set k-1 hide behind the decode of set k, the decodes themselves do not overlap"},
            "stage_ms_per_step": stage_ms,
            # r06: the step minus its entropy stage (all entropy launches, forked and joined): the part of the float path that does NOT
            # hide behind the longest chains since ccd_batch_run launches each chain group's frames behind that group
            "float_ms_exposed": dt / args.steps * 1e3 - stage_ms["entropy"] if world == 1 else None,
            "entropy_launches": batch.entropy_launches(),
            # side streams of the library that were measured to run kernels concurrently (HIP's hardware queues: tools/ubench/queues.hip)
            "concurrent_streams": lib_().ccd_concurrent_streams(local_rank),
            "slots_on_generic_entropy_kernel": sum(1 for k in kernels if not k & 1),
            "slots_on_unfused_float_path": sum(1 for k in kernels if not k & 4),
            "entropy_msym_per_s": nsym / (stage_ms["entropy"] / 1e3) / 1e6,
            # what actually bounds the dominant kernel: every stream is ONE serial range-decoder recurrence; its bare symbol
            # block, fully unrolled, runs at 111.0 ticks of the 2.4 GHz shader clock (tools/ubench/dcycle.hip: the generated
            # 16-symbol block of decoder_grid with paired tests + the least a loop around it needs; the production cycle with its
            # hand-over runs at 122.3 there), so n streams cannot exceed n * 2.4e9 / 111 symbols/s however many CUs idle
            "serial_chain_bound": {"achieved": nsym / (stage_ms["entropy"] / 1e3) / 1e6, "peak": n_frames * 2.4e9 / SYMBOL_FLOOR_TICKS / 1e6,
                                   "unit": "Msymbol/s", "frac": (nsym / (stage_ms["entropy"] / 1e3)) / (n_frames * 2.4e9 / SYMBOL_FLOOR_TICKS),
                                   "streams": n_frames, "ticks_per_symbol_floor": SYMBOL_FLOOR_TICKS},
            "roofline": {"bound": "hbm", "kernel": f"entropy_pipe_kernel<5, false, false, ShapeFix<20, 3, 14>> ({n_frames} streams, one workgroup each, in {n_l} concurrent launches)", "achieved": ent_ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ent_ach / HBM_PEAK_GBS, "traffic": traffic("entropy_pipe_kernel"),
                         "algorithmic_bytes": ent_bytes / n_l, "ms_per_launch": ent_ms_launch,
                         "launches_per_step": [{"ms": m, "streams": k} for m, k in launch_ms],
                         "stage_ms": stage_ms["entropy"],
                         "note": "latency-bound serial chain (one range decoder per stream): see entropy_msym_per_s, serial_chain_bound "
                                 "and DESIGN.md 4.1"},
            "roofline_float_stages": float_lines,
            "traffic_from": (("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python tools/prof_workload.py kodak24 2 keep_float` (the metric's batch), "
                              "launched by this run after the timed region; mean over launches, first launch dropped; FETCH_SIZE x 2, "
                              "WRITE_SIZE x 1 (profiles/r03/pmc_calibration.json)") if pmc_from == "measured in this run" else
                             f"tracked profile {pmc_from} (rocprofv3 not usable in this run)") if pmc_from else None,
        }
        res.update(sharded)
        # ---- per-orientation entropy time: the 6 portrait streams (more, shorter wavefront steps + a network trained on a
        # landscape picture) set the step time
        if "float" in legs:
            for label, sel in (("landscape", lambda hw: hw[0] < hw[1]), ("portrait", lambda hw: hw[0] > hw[1])):
                sub = [it for it in items if sel(it[3])]
                b = DecodeBatch(local_rank)
                for hdr, nn, lat, _ in sub:
                    b.add(hdr, nn, lat, 8, 0)
                b.run(sh); b.wait(sh)
                res.setdefault("entropy_ms_by_orientation", {})[label] = {
                    "streams": len(sub), "ms": event_ms(stream, lambda: b.run(sh, stage=0), max(2, args.steps // 2), local_rank),
                    "bpp": 8.0 * sum(len(l) for _, _, l, _ in sub) / sum(h * w for *_, (h, w) in sub)}
                b.close()
        # ---- the float stages at a chip-filling size, with and without the f32 output, and the unfused path for reference
        if "float" in legs:
            rows = []
            for copies_f in (1, 8):
                for label, opts in (("pyramid launch + fused kernel, integer planes only", dict(fused_dec=2, keep_float=False)),
                                    ("pyramid launch + fused kernel, integer planes + f32 output", dict(fused_dec=2, keep_float=True)),
                                    ("fused kernel alone (whole pyramid per tile), integer planes + f32 output", dict(fused_dec=1, keep_float=True)),
                                    ("unfused (6 upsampling launches + synthesis kernel)", dict(fused_dec=False))):
                    b = DecodeBatch(local_rank, overlap=False, **opts)  # (one launch per kernel: the kernels' own times)
                    for _ in range(copies_f):
                        for hdr, nn, lat, _ in items:
                            b.add(hdr, nn, lat, 8, 0)
                    b.run(sh); b.wait(sh)
                    ms = event_ms(stream, lambda: (b.run(sh, stage=1), b.run(sh, stage=2)), 5, local_rank)
                    px = copies_f * sum(h * w for *_, (h, w) in items)
                    out_b = c_out if "only" in label else 5 * c_out
                    rows.append({"frames": copies_f * len(items), "path": label, "ms": ms, "gpx_per_s": px / ms / 1e6,
                                 "tflops_algorithmic": flop_px * px / ms / 1e9, "frac_of_fp32_peak": flop_px * px / ms / 1e9 / FP32_PEAK_TFLOPS,
                                 "algorithmic_gbs": (latent_px(b, copies_f * len(items)) + out_b * px) / ms / 1e6})
                    b.close()
            res["float_stages_sweep"] = rows
        # ---- decode + PNG packing of every frame on the device (ccd_png_*)
        if "png" in legs:
            from cool_chic_amd.io.png import PngPacker

            packer = PngPacker(local_rank)
            files = [torch.empty(PngPacker.bound(h, w) + 4, dtype=torch.uint8, device=dev) for *_, (h, w) in mine]
            addr = [[batch.plane_device(s, p).__cuda_array_interface__["data"][0] for p in range(3)] for s in range(n_frames)]
            png_items = [(addr[s_][0], addr[s_][1], addr[s_][2], h, w, files[s_]) for s_, (*_, (h, w)) in enumerate(mine)]

            def step_png():
                batch.run(sh)
                packer.pack_batch_async(png_items, sh)  # all deflate blocks of all frames: one set of five launches

            step_png()
            ms_png = wall_ms(step_png, args.steps, local_rank)
            sizes = packer.finish_batch(sh)
            res["with_png_packing"] = {"value": rank_px / ms_png / 1e3, "unit": "Mpixel/s", "n_gpus": 1, "ms_per_step": ms_png,
                                       "png_bytes_per_step": int(sum(sizes)),
                                       "note": "decode + on-device PNG packing of all frames (files left in HBM); rank 0 alone"}
            packer.close()
        # ---- from .cool bytes: header parsing, Exp-Golomb network decode, fixed-point conversion, uploads, decode, planes back
        if "e2e" in legs:
            def from_bytes():
                b = DecodeBatch(local_rank, keep_float=False)
                for s in streams:
                    b.add(*synth.split_image_stream(s), 8, 0)
                b.run(sh)
                out = b.all_planes(sh)  # one copy per frame into pinned memory, one wait (which also collects the decode status)
                b.close()
                return out

            e2e_verified = verify_frames("kodak24", from_bytes())
            ms_e2e = wall_ms(from_bytes, 5, local_rank)
            res["end_to_end_from_bytes"] = {"value": sum(h * w for *_, (h, w) in items) / ms_e2e / 1e3, "unit": "Mpixel/s", "ms": ms_e2e,
                                            "verified": e2e_verified,
                                            "what": "ONE set at a time (the latency of a set; `value` keeps two in flight): 24 .cool files in host memory -> "
                                                    "integer planes in host memory: batch creation, per-stream parsing + staged asynchronous uploads "
                                                    "(ccd_batch_add), decode, one plane-block copy per frame into pinned memory"}
            # ---- the surface users call: cc_decode.py -i kodim14.cool -o x.png = decode_video(path, decoded_path), in-process:
            # read + parse + upload + decode + PNG packed on the device + file written (cc_decode.py:12-20, decode.py:26-91)
            import contextlib
            import io
            import tempfile

            from cool_chic_amd.bitstream.decode import decode_video

            with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
                src, dst = os.path.join(tmp, "kodim14.cool"), os.path.join(tmp, "kodim14.png")
                with open(src, "wb") as f:
                    f.write(streams[0])

                def file_to_file():
                    with contextlib.redirect_stdout(io.StringIO()):
                        decode_video(src, dst, device=local_rank)

                file_to_file()
                ms_f2f = wall_ms(file_to_file, 5, local_rank)
                from PIL import Image

                got_png = np.asarray(Image.open(dst)).transpose(2, 0, 1)
                png_ok = verify_frames("kodak24", [[got_png[0], got_png[1], got_png[2]]])

                def one_from_bytes():
                    b1 = DecodeBatch(local_rank, keep_float=False)
                    b1.add(*synth.split_image_stream(streams[0]), 8, 0)
                    b1.run(sh)
                    out1 = b1.all_planes(sh)
                    b1.close()
                    return out1

                one_from_bytes()
                ms_one = wall_ms(one_from_bytes, 5, local_rank)
                res["cc_decode_file_to_png"] = {
                    "ms": ms_f2f, "value": 512 * 768 / ms_f2f / 1e3, "unit": "Mpixel/s", "png_bytes": os.path.getsize(dst),
                    "verified_png_readback": png_ok, "same_stream_bytes_to_host_planes_ms": ms_one,
                    "what": "decode_video('kodim14.cool', 'kodim14.png') in-process = cc_decode.py's work for ONE picture: file read, parse, "
                            "upload, decode (its serial chain: ~40 ms), PNG packed on the device, file written; PIL reads the file back"}
        # ---- the rate model (arm.py:448-485): the one genuinely HBM-bound kernel of the build, 16 B per symbol
        if "rate" in legs:
            from cool_chic_amd.component.core.arm import compute_rate, total_rate_bits  # noqa: F401

            n_sym = 1 << 26
            g_ = torch.Generator(device=dev).manual_seed(5)
            xr = torch.randint(-64, 64, (n_sym,), generator=g_, device=dev).float()
            mur = xr + torch.randn(n_sym, generator=g_, device=dev) * 1.5
            scr = torch.exp(torch.rand(n_sym, generator=g_, device=dev) * 4 - 2)
            out_r = torch.empty_like(xr)
            from cool_chic_amd._lib import check as _check, lib as _lib_

            def rate_once():
                _check(_lib_().ccd_compute_rate(local_rank, C.c_void_p(sh or None), C.c_void_p(xr.data_ptr()), C.c_void_p(mur.data_ptr()),
                                                C.c_void_p(scr.data_ptr()), n_sym, C.c_void_p(out_r.data_ptr()), None), "ccd_compute_rate")

            rate_once()
            ms_r = event_ms(stream, rate_once, 20, local_rank)
            gbs = 16.0 * n_sym / ms_r / 1e6
            res["rate_model"] = {"kernel": "rate_kernel_v4 (compute_rate, arm.py:448-485)", "symbols": n_sym, "ms_per_launch": ms_r,
                                 "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes": 16 * n_sym, "gsymbols_per_s": n_sym / ms_r / 1e6,
                                 "note": "12 B in (x, mu, scale f32) + 4 B out per symbol; HIP events on the launch stream, 20 launches"}
            del xr, mur, scr, out_r
        # ---- the metric's set re-encoded against a network OUTSIDE the r02 static envelope of the pipelined entropy kernel
        # (worst-case IFCE feature >= 2^15, like three of the six networks the reference encoder produced in the build
        # container): same kernel, features checked on the device, no pixel redone
        if "envelope" in legs:
            wl_e = synth.workload("kodak24_wide_envelope")
            be = DecodeBatch(local_rank)
            for st_ in wl_e["streams"]:
                be.add(*synth.split_image_stream(st_), 8, 0)
            be.run(sh); be.wait(sh)
            ms_e = event_ms(stream, lambda: be.run(sh, stage=0), max(2, args.steps // 2), local_rank)
            be.wait(sh)
            k_e = [be.slot_kernels(s_) for s_ in range(len(wl_e["streams"]))]
            res["wide_envelope_network"] = {
                "workload": "kodak24 re-encoded with kodim14's network pushed outside the r02 envelope (synth.kodak24_wide_envelope)",
                "entropy_ms": ms_e, "ratio_to_kodak24_entropy_ms": ms_e / stage_ms["entropy"],
                "symbols": n_symbols(be, len(k_e)), "stream_bytes": int(sum(len(x) for x in wl_e["streams"])),
                "slots_on_generic_entropy_kernel": sum(1 for k in k_e if not k & 1),
                "pixels_redone_in_int64": int(sum(int(be.slot_stats(s_)[39]) for s_ in range(len(k_e)))),
                "verified": verify_frames("kodak24_wide_envelope", [be.planes(s_) for s_ in range(len(k_e))], streams=wl_e["streams"])}
            be.close()
        # ---- 256 streams (the same 24 repeated) in ONE batch: a stream occupies one CU for its serial chain, so
        # kodak24 keeps 24 of the 256 CUs busy; this is what the chip does when an image set is large enough to fill it
        if "wide" in legs:
            wide = DecodeBatch(local_rank)
            for i in t_ids:
                wide.add(*items[i][:3], 8, 0)
            wide.run(sh); wide.wait(sh)
            n_wide = max(2, min(args.steps, 4))
            ms_w = wall_ms(lambda: wide.run(sh), n_wide, local_rank)
            wide.wait(sh)
            wide_verified = verify_frames("kodak24", [wide.planes(s_) for s_ in range(len(t_ids))], stream_of=lambda i: t_ids[i])
            res["more_frames_in_flight"] = {"frames_in_flight": len(t_ids), "value": t_px / ms_w / 1e3,
                                            "unit": "Mpixel/s", "n_gpus": 1, "steps": n_wide, "ms_per_step": ms_w, "verified": wide_verified,
                                            "note": f"{THROUGHPUT_STREAMS} streams (kodak24 repeated) in one batch on rank 0, one per CU: not the metric's "
                                                    "configuration; the per-GPU figure of the throughput regime (--scaling throughput)"}
            wide.close()
        # ---- the other BASELINE configurations, each on this one GPU
        extra = {}
        n_leg = max(2, min(args.steps, 3))
        for name in ("clic41", "uhd4k"):
            if name in legs:
                extra[name] = image_leg(name, local_rank, sh, stream, n_leg, 6.0 if name == "clic41" else 16.0, want_cpu)
        if "gop1080p33" in legs:
            extra["gop1080p33"] = gop_leg(local_rank, sh, stream, n_leg, 12.0, want_cpu)
        # ---- what other content does to the chain (r05): Kodak geometry at 2.4 bpp with 12-15 % wide windows (LOP network of
        # the reference-encoded hq192), and the CLIC sizes with the VHOP / MOP decoders (two kernel instantiations in one batch)
        for name in ("kodak24_hq", "clic41_alt"):
            if name in legs:
                extra[name] = image_leg(name, local_rank, sh, stream, n_leg, 6.0, False)
        if "kodak24_hq" in extra:
            extra["kodak24_hq"]["entropy_ms_ratio_to_kodak24"] = extra["kodak24_hq"]["entropy_ms"] / stage_ms["entropy"]
        if extra:
            res["baseline_configs"] = extra
        # ---- the fallback cliffs, priced on the metric's own set (r05): CCD_FORCE_GENERIC sends every slot to the generic
        # int64 entropy kernel (the only path for ARM weights >= 2^31, which the format allows: neuralnet.py:184-188) and to the
        # per-layer float kernels (architectures outside the presets); the vector-ALU float path alone is fused_dec = 0 above
        if "cliffs" in legs:
            os.environ["CCD_FORCE_GENERIC"] = "1"
            try:
                bg = DecodeBatch(local_rank)
            finally:
                del os.environ["CCD_FORCE_GENERIC"]
            for hdr, nn, lat, _ in items:
                bg.add(hdr, nn, lat, 8, 0)
            bg.run(sh); bg.wait(sh)
            k_g = [bg.slot_kernels(s_) for s_ in range(n_kodak)]
            ms_g_ent = event_ms(stream, lambda: bg.run(sh, stage=0), 2, local_rank)
            ms_g_flt = event_ms(stream, lambda: (bg.run(sh, stage=1), bg.run(sh, stage=2)), 2, local_rank)
            bg.wait(sh)
            res["fallback_cliffs"] = {
                "workload": "kodak24 with CCD_FORCE_GENERIC=1 (every slot on the generic entropy kernel and the per-layer float kernels)",
                "generic_entropy_ms": ms_g_ent, "ratio_to_pipelined_entropy": ms_g_ent / stage_ms["entropy"],
                "generic_float_ms": ms_g_flt, "ratio_to_fused_float": ms_g_flt / (stage_ms["pyramid_launch"] + stage_ms["fused_float"]),
                "value": kodak_px / (ms_g_ent + ms_g_flt) / 1e3, "unit": "Mpixel/s",
                "slots_on_generic_entropy_kernel": sum(1 for k in k_g if not k & 1), "slots_on_unfused_float_path": sum(1 for k in k_g if not k & 4),
                "verified": verify_frames("kodak24", [bg.planes(s_) for s_ in range(n_kodak)])}
            bg.close()
            # r06: the one cliff a picture's SIZE decides: wider than the 5 069 columns the pipelined kernel's 512-row symbol ring
            # holds (the format allows 16 383, header.py:244-307) -> the generic entropy kernel.  Two pictures of the same network
            # and ~1.97 Mpx, one each side of the limit (parity: tests/test_gpu_parity.py::test_picture_wider_than_the_symbol_ring)
            rows = {}
            for label, (h_, w_) in (("5056x388 (pipelined kernel)", (388, 5056)), ("7680x256 (wider than the ring: generic kernel)", (256, 7680))):
                bw = DecodeBatch(local_rank)
                bw.add(*synth.split_image_stream(synth.image_stream(h_, w_, 0)), 8, 0)
                bw.run(sh); bw.wait(sh)
                ms_w_ = event_ms(stream, lambda: bw.run(sh, stage=0), 2, local_rank)
                bw.wait(sh)
                rows[label] = {"entropy_ms": ms_w_, "symbols": n_symbols(bw, 1), "ns_per_symbol": ms_w_ * 1e6 / n_symbols(bw, 1),
                               "on_pipelined_kernel": bool(bw.slot_kernels(0) & 1), "status": int(bw.slot_status(0))}
                bw.close()
            ks = list(rows)
            res["fallback_cliffs"]["picture_wider_than_the_symbol_ring"] = {
                **rows, "ratio_ns_per_symbol": rows[ks[1]]["ns_per_symbol"] / rows[ks[0]]["ns_per_symbol"],
                "limit": "pictures up to 5 069 columns run the pipelined kernel (W / 10 + 6 <= 512 ring rows x 64 B = 32 KB of its 139 KB of LDS; "
                         "16 383 columns would need 1 645 rows = 105 KB next to 66 KB of window tables: not in 160 KB)"}
        if want_cpu:
            res["cpu_baseline"] = cpu_sample(streams, [h * w for *_, (h, w) in items], 8.0, "kodak24 streams")
            # measured once in the build container (8-core Xeon 2.1 GHz, torch 2.10 CPU): the reference's own PyTorch decode of
            # kodim14.cool with the C range coder behind the constriction shim (tools/ref_baseline.py) - see BASELINE.md section 3
            ref_path = os.path.join(ROOT, "profiles", "r02", "reference_pytorch_container.json")
            if os.path.exists(ref_path):
                with open(ref_path) as f:
                    res["reference_pytorch_container"] = json.load(f)
    if world > 1:
        dist.barrier()
    if rank == 0:
        # one line per regime on stderr (N > 1: strong kodak24, sharded clic41, throughput); the contract's ONE JSON line on stdout
        for label, leg in (("kodak24 (strong)" if args.scaling == "strong" else args.scaling, res), ("clic41_sharded (strong)", res.get("clic41_sharded")),
                           ("throughput_regime (weak)", res.get("throughput_regime"))):
            if leg:
                print(f"[bench] {label:28s} n_gpus {world}  {leg['value']:9.1f} Mpixel/s  {leg['ms_per_step']:8.2f} ms per step  verified {leg['verified'].get('ok')}", file=sys.stderr)
        print(json.dumps(res))
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
