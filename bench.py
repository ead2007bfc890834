#!/usr/bin/env python3
"""Benchmark of the MI355X Cool-chic decoder on BASELINE.json's metric: decoded Mpixel/s.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--legs a,b,..] [--no-cpu-baseline]
    (N > 1: launched by torch.distributed.run, one rank per GPU)

`value` = BASELINE configs[1] (config.workload = "kodak24"): the Kodak-24 set - 24 RGB 8-bit 512x768 frames (18
landscape, 6 portrait), HOP decoder, one frame per batch slot, all in flight on one MI355X.  A "step" = one full decode
of the set: entropy decode (integer ARM / IFCE + range decoder), then ONE fused kernel for latent-pyramid upsampling +
synthesis + integer samples; inputs (payload words, network parameters) are resident in HBM before the timed region,
outputs stay in HBM.  Only kodim14.cool is a real bitstream; every other input is manufactured by cool_chic_amd/synth.py
(SURVEY.md section 8d, H6).

Beside the metric (never as `value`), rank 0 reports one leg per other BASELINE configuration - `clic41` (configs[2]),
`gop1080p33` (configs[3]), `uhd4k` (configs[4]) - each with Mpixel/s, ms, Msymbol/s and its own CPU sample, plus
`more_frames_in_flight` (kodak24 x 8: the chip-filling regime), `with_png_packing`, `end_to_end_from_bytes`, the
per-orientation entropy times and the float-stage roofline.

N GPUs: --scaling weak (default): every rank decodes its own 24 frames; --scaling strong: ONE fixed set (kodak24 x 8 =
192 frames) is split round-robin over the ranks.  Either way rank 0 gathers the decoded planes over RCCL inside the
timed region.  BASELINE's own sets (24 / 41 frames, 66 cool-chics per GOP) keep 9-26 % of ONE GPU's CUs busy - a stream
is one serial range-decoder chain on one CU - so they do not strong-scale: their time is the slowest stream's.

`roofline` = the dominant kernel (entropy, a latency chain: its HBM fraction only says how far from that roof it sits);
`roofline_float_stages` = the fused float kernel against the fp32 peak (it is compute-bound: 4.33 B/px, ~1.7 kflop/px)
and as algorithmic GB/s; `cpu_baseline` = the CPU oracle (single-thread C port of the reference's algorithm) on a
bounded sample.
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
SYMBOL_FLOOR_TICKS = 117.5  # bare symbol loop of the entropy kernel's decoder, ticks per symbol (tools/ubench/dcycle.hip)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PROFILE_DIRS = ["profiles/r03", "profiles/r02", "profiles/r01"]


def build_kodak24(device: int = 0):
    """Kept for tools/: 24 (cc_header, bytes_nn, bytes_latent, (H, W)) tuples + the raw stream bytes."""
    from cool_chic_amd import synth

    streams, sizes = synth.kodak24()
    return [(*synth.split_image_stream(s), hw) for s, hw in zip(streams, sizes)], streams


def cpu_sample(streams, px_each, budget_s: float, what: str):
    """The oracle (single-thread C restatement of the reference algorithm) on a bounded sample of `streams`."""
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
    return {"value": px / dt / 1e6, "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": f"first {n} of {len(streams)} {what}, full decode to integer planes, {dt:.1f} s",
            "host_cores_available": cores,
            # streams are independent: one oracle process per host core is the honest CPU ceiling for a set of streams
            "extrapolated_all_cores": {"value": px / dt / 1e6 * min(cores, len(streams)), "unit": "Mpixel/s",
                                       "note": f"one stream per core, min(cores, streams) = {min(cores, len(streams))} at once (not run)"}}


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
                for k, v in vals.items():
                    if sub in k and len(v) > 1:
                        out.setdefault(short, {})["fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"] = sum(v[1:]) / len(v[1:])
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
    ms = wall_ms(lambda: b.run(sh), steps, device)
    ms_entropy = event_ms(stream, lambda: b.run(sh, stage=0), max(1, steps // 2), device)
    ms_float = event_ms(stream, lambda: (b.run(sh, stage=1), b.run(sh, stage=2)), max(1, steps // 2), device)
    b.wait(sh)
    nsym = n_symbols(b, len(triples))
    verified = verify_frames(name, [b.planes(i) for i in range(len(triples))], streams=wl["streams"])
    b.close()
    leg = {"frames": len(triples), "mpixels": sum(px) / 1e6, "verified": verified, "value": sum(px) / ms / 1e3, "unit": "Mpixel/s", "n_gpus": 1, "steps": steps,
           "ms_per_step": ms, "entropy_ms": ms_entropy, "float_ms": ms_float, "symbols": nsym,
           "entropy_msym_per_s": nsym / ms_entropy / 1e3, "largest_frame_mpx": max(px) / 1e6,
           "slots_on_generic_entropy_kernel": sum(1 for k in kernels if not k & 1),
           "slots_on_unfused_float_path": sum(1 for k in kernels if not k & 4),
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
    ms_e2e = wall_ms(whole, max(1, steps // 2), device)
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
        leg["cpu_baseline"] = cpu_sample([small], [info["frames"] * H * W], cpu_budget, "3-frame 1080p GOP (I0 I2 B1)")
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--legs", default="all", help="comma list of clic41,gop1080p33,uhd4k,wide,png,e2e,float,envelope or all / none (rank 0, beside the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes (roofline.traffic then "
                    "comes from the tracked profile)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: nccl = RCCL over xGMI (one GPU per rank); gloo = host-staged exchange, ranks may share a GPU "
                         "(smoke run of the multi-rank path on a single-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X: there is no CPU fallback"
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # ranks may share a GPU: there is no device-to-device collective to collide
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
    red_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"  # where the scalar reductions live
    all_legs = ["clic41", "gop1080p33", "uhd4k", "wide", "png", "e2e", "float", "envelope"]
    legs = all_legs if args.legs == "all" else ([] if args.legs == "none" else args.legs.split(","))
    if world > 1 and args.legs == "all":
        legs = []  # the scaling runs measure the metric; the other configurations are single-GPU legs of the N = 1 run
    want_cpu = not args.no_cpu_baseline

    from cool_chic_amd import DecodeBatch, synth
    from cool_chic_amd.parallel import EqualSizeGather, shard_indices

    items, streams = build_kodak24(local_rank)
    copies = 8 if args.scaling == "strong" else 1
    mine = list(items) * copies
    if args.scaling == "strong":
        mine = [mine[i] for i in shard_indices(len(mine), rank, world)]
        px_per_step = copies * sum(h * w for *_, (h, w) in items)           # the whole fixed set
    else:
        px_per_step = world * sum(h * w for *_, (h, w) in items)           # every rank its own set
    n_frames = len(mine)
    batch = DecodeBatch(local_rank)
    for hdr, nn, lat, _ in mine:
        batch.add(hdr, nn, lat, 8, 0)
    stream = torch.cuda.current_stream(local_rank)
    sh = stream.cuda_stream
    dev = f"cuda:{local_rank}"

    planes = [torch.as_tensor(batch.plane_device(s, p), device=dev).reshape(-1) for s in range(n_frames) for p in range(3)]
    n_bytes = sum(int(p.numel()) * p.element_size() for p in planes)
    if world > 1:  # equal message sizes: pad to the largest share
        t = torch.tensor([n_bytes], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pad = int(t.item()) - n_bytes
        if pad:
            planes.append(torch.zeros(pad, dtype=torch.uint8, device=dev))
        gatherer = EqualSizeGather(int(t.item()), dev, dst=0)
    else:
        gatherer = None

    gathered = [None]

    def step():
        batch.run(sh)
        if gatherer is not None:  # decoded integer planes of this rank's frames -> writer rank (RCCL over xGMI), inside the timed region
            gathered[0] = gatherer(planes)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(local_rank)

    for _ in range(args.warmup):
        step()
    batch.wait(sh)  # raises on decode errors
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    batch.wait(sh)

    res = None
    if rank == 0:
        # ---- what was timed, against the CPU oracle (hashes of its integer planes for the same 24 streams): this rank's own
        # frames from its batch, and - N > 1 - every rank's frames as they arrived in the gather of the last timed step
        n_kodak = len(items)
        own_index = (lambda k: (rank + k * world) % n_kodak) if args.scaling == "strong" else (lambda k: k % n_kodak)
        verified = verify_frames("kodak24", [batch.planes(s) for s in range(n_frames)], stream_of=own_index,
                                 streams=streams if args.scaling == "weak" else None)
        if world > 1 and gathered[0] is not None:
            fb = 3 * 512 * 768  # bytes of one frame's planes in a rank's message (all 8-bit 512 x 768 RGB)
            frames_g, index_g = [], []
            for r, msg in enumerate(gathered[0]):
                host = msg.cpu().numpy()
                n_r = len(shard_indices(copies * n_kodak, r, world)) if args.scaling == "strong" else n_kodak
                for k in range(n_r):
                    frames_g.append([host[k * fb + p * (fb // 3): k * fb + (p + 1) * (fb // 3)] for p in range(3)])
                    index_g.append(((r + k * world) if args.scaling == "strong" else k) % n_kodak)
            verified["gathered"] = verify_frames("kodak24", frames_g, stream_of=lambda i: index_g[i])
        # ---- per-stage timing with HIP events on the launch stream (roofline evidence); stage 1 (per-level upsampling) is
        # empty on the fused path
        stage_ms = {name: event_ms(stream, lambda st=st: batch.run(sh, stage=st), args.steps, local_rank)
                    for st, name in ((0, "entropy"), (1, "upsampling_unfused_only"), (2, "fused_float"))}
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
        if args.scaling == "weak" and world == 1 and not args.no_live_traffic:
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

        ent_bytes = n_payload + nsym  # payload in + one byte per symbol out
        ent_ach = ent_bytes / (stage_ms["entropy"] / 1e3) / 1e9
        # fused float kernel: algorithmic bytes = int8 latents in + (4 C f32 +) C integer samples out; flops as executed by
        # the reference's 2-D kernels: synthesis 2 x 672 (HOP) + upsampling ~380 per pixel
        flop_px = 2.0 * 672 + 380.0
        ff_bytes = n_lat_px + (4 * c_out + c_out) * rank_px
        ff_ms = stage_ms["fused_float"]
        float_lines = [{
            "kernel": f"decode_fused_kernel<{L},{c_out}> (pyramid + synthesis + integer samples, {n_frames} frames, one launch)",
            "bound": "fp32", "achieved": flop_px * rank_px / ff_ms / 1e9, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop_px * rank_px / ff_ms / 1e9 / FP32_PEAK_TFLOPS, "ms_per_launch": ff_ms,
            "algorithmic_gbs": ff_bytes / ff_ms / 1e6, "frac_of_hbm_peak": ff_bytes / ff_ms / 1e6 / HBM_PEAK_GBS,
            "algorithmic_bytes": ff_bytes, "traffic": traffic("decode_fused_kernel"),
            "note": "compute-bound by construction (%.1f B/px, ~%.0f flop/px): priced against the fp32 peak; exact fmaf chains on "
                    "v_mfma_f32_4x4x1 (bitwise the oracle's order)" % (ff_bytes / rank_px, flop_px)}]
        res = {
            "metric": "decoded Mpixel/s", "value": px_per_step * args.steps / dt / 1e6, "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int64+f64 entropy / f32 synthesis",
            "data": "synthetic (kodim14.cool real + 23 streams re-encoded from rolled/transposed kodim14 latents)",
            "config": {"workload": "kodak24" if args.scaling == "weak" else "kodak24 x 8 (192 frames, one fixed set split over the ranks)",
                       "frames_per_gpu": n_frames, "frame": "512x768 RGB 8-bit, HOP decoder", "symbols_per_step_rank0": nsym,
                       "parallelism": f"frames x{world} (round-robin, gather of planes to rank 0)"},
            "parity": "integer stages bit-exact vs reference fixtures; float stages bit-exact vs CPU oracle; integer planes <=1 LSB on "
                      "<=2e-5 of samples vs the reference decoder's output = within the reference's own thread-count noise floor "
                      "(tests/test_gpu_parity.py)",
            "verified": verified,
            "stage_ms_per_step": stage_ms,
            "slots_on_generic_entropy_kernel": sum(1 for k in kernels if not k & 1),
            "slots_on_unfused_float_path": sum(1 for k in kernels if not k & 4),
            "entropy_msym_per_s": nsym / (stage_ms["entropy"] / 1e3) / 1e6,
            # what actually bounds the dominant kernel: every stream is ONE serial range-decoder recurrence; its bare symbol
            # loop, fully unrolled, runs at 117.5 ticks of the 2.4 GHz shader clock (tools/ubench/dcycle.hip variant 0: the generated
            # 16-symbol block of decoder_grid + the least a loop around it needs; the production cycle with its hand-over runs at
            # 127.5 there), so n streams cannot exceed n * 2.4e9 / 117.5 symbols/s however many CUs idle
            "serial_chain_bound": {"achieved": nsym / (stage_ms["entropy"] / 1e3) / 1e6, "peak": n_frames * 2.4e9 / SYMBOL_FLOOR_TICKS / 1e6,
                                   "unit": "Msymbol/s", "frac": (nsym / (stage_ms["entropy"] / 1e3)) / (n_frames * 2.4e9 / SYMBOL_FLOOR_TICKS),
                                   "streams": n_frames, "ticks_per_symbol_floor": SYMBOL_FLOOR_TICKS},
            "roofline": {"bound": "hbm", "kernel": f"entropy_pipe_kernel<5, false> ({n_frames} streams, one workgroup each)", "achieved": ent_ach,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ent_ach / HBM_PEAK_GBS, "traffic": traffic("entropy_pipe_kernel"),
                         "algorithmic_bytes": ent_bytes, "ms_per_launch": stage_ms["entropy"],
                         "note": "latency-bound serial chain (one range decoder per stream): see entropy_msym_per_s, serial_chain_bound "
                                 "and DESIGN.md 4.1"},
            "roofline_float_stages": float_lines,
            "traffic_from": (("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python tools/prof_workload.py kodak24 2 keep_float` (the metric's batch), "
                              "launched by this run after the timed region; mean over launches, first launch dropped; FETCH_SIZE x 2, "
                              "WRITE_SIZE x 1 (profiles/r03/pmc_calibration.json)") if pmc_from == "measured in this run" else
                             f"tracked profile {pmc_from} (rocprofv3 not usable in this run)") if pmc_from else None,
        }
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
                for label, opts in (("fused, integer planes only", dict(fused_dec=True, keep_float=False)),
                                    ("fused, integer planes + f32 output", dict(fused_dec=True, keep_float=True)),
                                    ("unfused (6 upsampling launches + synthesis kernel)", dict(fused_dec=False))):
                    b = DecodeBatch(local_rank, **opts)
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
                                            "what": "24 .cool files in host memory -> integer planes in host memory: batch creation, per-stream "
                                                    "parsing + staged asynchronous uploads (ccd_batch_add), decode, one plane-block copy per frame "
                                                    "into pinned memory; comparable with cpu_baseline"}
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
        # ---- the same 24 streams eight times over in ONE batch: a stream occupies one CU for its serial chain, so kodak24 keeps
        # 24 of the 256 CUs busy; this is what the chip does when an image set is large enough to fill it
        if "wide" in legs:
            wide = DecodeBatch(local_rank)
            for _ in range(8):
                for hdr, nn, lat, _ in items:
                    wide.add(hdr, nn, lat, 8, 0)
            wide.run(sh); wide.wait(sh)
            n_wide = max(2, min(args.steps, 4))
            ms_w = wall_ms(lambda: wide.run(sh), n_wide, local_rank)
            wide.wait(sh)
            wide_verified = verify_frames("kodak24", [wide.planes(s_) for s_ in range(8 * len(items))], stream_of=lambda i: i % len(items))
            res["more_frames_in_flight"] = {"frames_in_flight": 8 * len(items), "value": 8 * sum(h * w for *_, (h, w) in items) / ms_w / 1e3,
                                            "unit": "Mpixel/s", "n_gpus": 1, "steps": n_wide, "ms_per_step": ms_w, "verified": wide_verified,
                                            "note": "kodak24 x 8 in one batch on rank 0: not the metric's configuration, shown for occupancy"}
            wide.close()
        # ---- the other BASELINE configurations, each on this one GPU
        extra = {}
        n_leg = max(2, min(args.steps, 3))
        for name in ("clic41", "uhd4k"):
            if name in legs:
                extra[name] = image_leg(name, local_rank, sh, stream, n_leg, 6.0 if name == "clic41" else 16.0, want_cpu)
        if "gop1080p33" in legs:
            extra["gop1080p33"] = gop_leg(local_rank, sh, stream, n_leg, 12.0, want_cpu)
        if extra:
            res["baseline_configs"] = extra
        if want_cpu:
            res["cpu_baseline"] = cpu_sample(streams, [h * w for *_, (h, w) in items], 12.0, "kodak24 streams")
            # measured once in the build container (8-core Xeon 2.1 GHz, torch 2.10 CPU): the reference's own PyTorch decode of
            # kodim14.cool with the C range coder behind the constriction shim (tools/ref_baseline.py) - see BASELINE.md section 3
            ref_path = os.path.join(ROOT, "profiles", "r02", "reference_pytorch_container.json")
            if os.path.exists(ref_path):
                with open(ref_path) as f:
                    res["reference_pytorch_container"] = json.load(f)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(res))
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
