#!/usr/bin/env python3
"""Benchmark of the MI355X Cool-chic decoder on BASELINE.json's metric: decoded Mpixel/s.

    python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)

Workload (config.workload = "kodak24"): the Kodak-24 set of BASELINE configs[1] - 24 RGB 8-bit
512x768 frames (18 landscape, 6 portrait) with the HOP decoder architecture, one frame per batch
slot, all in flight on one MI355X. Only kodim14.cool is a real bitstream (shipped fixture); the
other 23 are written by the build's own bitstream writer from spatial rolls / transpositions of
kodim14's decoded latents (real symbol statistics, 0.67-0.9 bpp), with kodim14's network payload
(SURVEY.md section 8d, H6). A "step" = one full decode of the 24 frames: entropy decode (integer
ARM/IFCE + range decoder), upsampling, synthesis, integer planes; inputs (payload words, network
parameters) are resident in HBM before the timed region, outputs stay in HBM.

With N GPUs every rank decodes its own 24 frames (weak scaling, frames are independent units) and
rank 0 gathers the decoded planes over RCCL inside the timed region.

The JSON line carries `roofline` for the dominant HBM-bound kernel (synthesis) measured with HIP
events on the launch stream, and `cpu_baseline` = the CPU oracle (a single-thread C port of the
reference's algorithm) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def build_kodak24(device: int):
    """Returns 24 (cc_header, bytes_nn, bytes_latent, (H, W)) tuples + the raw stream bytes."""
    from cool_chic_amd import DecodeBatch, writer
    from cool_chic_amd.bitstream.header import CoolChicHeader, FrameHeader, VideoHeader

    with open(os.path.join(ROOT, "tests", "golden", "kodim14.cool"), "rb") as f:
        real = f.read()

    def split(bs):
        rest = VideoHeader().read_header(bs)
        rest = FrameHeader().read_header(rest)
        ch = CoolChicHeader()
        rest = ch.read_header(rest)
        n_nn = ch.get_value("nn_n_bytes")
        return ch, rest[:n_nn], rest[n_nn:n_nn + ch.get_value("n_bytes_latent")]

    ch, nn, lat = split(real)
    # kodim14's latents, decoded by the product path itself
    b = DecodeBatch(device)
    b.add(ch.raw, nn, lat, 8, 0)
    b.run(stage=0)
    b.wait()
    latents = [b.latent(0, g) for g in range(ch.c.n_grids)]
    b.close()
    _, levels = writer.grid_sizes((512, 768), ch.raw)
    # Kodak has 18 landscape + 6 portrait images
    jobs = [(1000 + i, i in (3, 8, 9, 16, 17, 18)) for i in range(1, 24)]

    def make(job):
        seed, portrait = job
        v = writer.variant_latents(latents, levels, seed, portrait)
        return writer.encode_stream(ch.raw, nn, v, img_size=(768, 512) if portrait else (512, 768))

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        streams = [real] + list(ex.map(make, jobs))
    out = []
    for s in streams:
        c, n, l = split(s)
        out.append((c.raw, n, l, (c.c.img_size[0], c.c.img_size[1])))
    return out, streams


def cpu_baseline(streams, budget_s: float = 12.0):
    """The oracle (single-thread C restatement of the reference algorithm) on a bounded sample."""
    from oracle import oracle_py

    oracle_py.build()
    t0 = time.perf_counter()
    px = 0
    n = 0
    for s in streams:
        oracle_py.decode_video(s)
        px += 512 * 768
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": px / dt / 1e6, "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": f"first {n} of the 24 kodak24 streams, full decode to integer planes, {dt:.1f} s",
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X: there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    from cool_chic_amd import DecodeBatch
    from cool_chic_amd.parallel import EqualSizeGather

    items, streams = build_kodak24(local_rank)
    n_frames = len(items)
    px_per_step = sum(h * w for *_, (h, w) in items)

    batch = DecodeBatch(local_rank)
    for hdr, nn, lat, _ in items:
        batch.add(hdr, nn, lat, 8, 0)
    stream = torch.cuda.current_stream(local_rank)
    sh = stream.cuda_stream
    dev = f"cuda:{local_rank}"

    planes = [torch.as_tensor(batch.plane_device(s, p), device=dev).reshape(-1) for s in range(n_frames) for p in range(3)]
    gatherer = EqualSizeGather(sum(int(p.numel()) * p.element_size() for p in planes), dev, dst=0) if world > 1 else None

    def gather_planes():
        if world > 1:  # decoded integer planes of this rank's frames -> writer rank (RCCL over xGMI), inside the timed region
            gatherer(planes)

    def step():
        batch.run(sh)
        gather_planes()

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
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    batch.wait(sh)

    # ---- per-stage timing with HIP events on the launch stream (roofline evidence) ---------------
    stage_ms = {}
    if rank == 0:
        for stage, name in ((0, "entropy"), (1, "upsampling"), (2, "synthesis")):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(local_rank)
            e0.record(stream)
            for _ in range(args.steps):
                batch.run(sh, stage=stage)
            e1.record(stream)
            torch.cuda.synchronize(local_rank)
            stage_ms[name] = e0.elapsed_time(e1) / args.steps
    # ---- second leg (reported beside the metric, never as `value`): the same step followed by PNG packing of every
    # decoded frame on the device (ccd_png_*), packs spread over side streams
    png_leg = None
    if rank == 0:
        from cool_chic_amd.io.png import PngPacker

        packer = PngPacker(local_rank)
        files = [torch.empty(PngPacker.bound(h, w) + 4, dtype=torch.uint8, device=dev) for *_, (h, w) in items]
        addr = [[batch.plane_device(s, p).__cuda_array_interface__["data"][0] for p in range(3)] for s in range(n_frames)]
        png_items = [(addr[s_][0], addr[s_][1], addr[s_][2], h, w, files[s_]) for s_, (*_, (h, w)) in enumerate(items)]

        def step_png():
            batch.run(sh)
            packer.pack_batch_async(png_items, sh)  # all deflate blocks of all frames: one set of five launches

        step_png()
        torch.cuda.synchronize(local_rank)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_png()
        torch.cuda.synchronize(local_rank)
        dt_png = time.perf_counter() - t1
        sizes = packer.finish_batch(sh)
        png_leg = {"value": px_per_step * args.steps / dt_png / 1e6, "unit": "Mpixel/s", "n_gpus": 1,
                   "ms_per_step": dt_png / args.steps * 1e3, "png_bytes_per_step": int(sum(sizes)),
                   "note": "decode + on-device PNG packing of all frames (files left in HBM); rank 0 alone"}
        packer.close()
    # ---- third leg (reported beside the metric): the same 24 streams eight times over in ONE batch.  A stream occupies one
    # compute unit for its whole serial chain, so kodak24 keeps 24 of the 256 CUs busy; this shows what the chip does when
    # an image set is large enough to fill it.
    wide_leg = None
    if rank == 0:
        copies = 8
        wide = DecodeBatch(local_rank)
        for _ in range(copies):
            for hdr, nn, lat, _ in items:
                wide.add(hdr, nn, lat, 8, 0)
        wide.run(sh)
        wide.wait(sh)
        torch.cuda.synchronize(local_rank)
        n_wide = max(2, min(args.steps, 4))
        t2 = time.perf_counter()
        for _ in range(n_wide):
            wide.run(sh)
        torch.cuda.synchronize(local_rank)
        dt_wide = time.perf_counter() - t2
        wide.wait(sh)
        wide_leg = {"frames_in_flight": copies * n_frames, "value": copies * px_per_step * n_wide / dt_wide / 1e6, "unit": "Mpixel/s",
                    "n_gpus": 1, "steps": n_wide, "ms_per_step": dt_wide / n_wide * 1e3,
                    "note": "kodak24 x 8 in one batch on rank 0: not the metric's configuration, shown for occupancy"}
        wide.close()
    if world > 1:
        dist.barrier()

    if rank == 0:
        hdr0 = batch.header(0)
        n_sym = sum(batch.header(s).n_symbols for s in range(n_frames))
        n_payload = sum(len(lat) for _, _, lat, _ in items)
        L = hdr0.input_feature_synthesis
        c_out = hdr0.out_channels
        # PMC traffic of the same command, collected under rocprofv3 (tools/collect_profiles.sh: one run per counter)
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "kodak24_pmc_traffic.json")) as f:
                pmc = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            pmc = {}

        def traffic(name, suffix=""):
            k = pmc.get(name, {})
            if "fetch_bytes" + suffix in k and "write_bytes" + suffix in k:
                return k["fetch_bytes" + suffix] + k["write_bytes" + suffix]
            return None

        def line(kernel, algo_bytes, ms, tr, note):
            ach = algo_bytes / (ms / 1e3) / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": tr, "algorithmic_bytes": algo_bytes, "ms_per_launch": ms, "note": note}

        # ALGORITHMIC bytes per launch:
        #   entropy: payload in + one byte per symbol out; upsampling: S + 4 L B/px; synthesis: 4 L + 4 C + C B/px
        ent_bytes = n_payload + n_sym
        syn_bytes = (4 * L + 4 * c_out + c_out) * px_per_step
        # latent planes only (hyperlatents do not reach the upsampling): S = sum of the latent grids of every frame
        n_lat_px = 0
        for s_ in range(n_frames):
            h_ = batch.header(s_)
            n_lat_px += sum(h_.grid_h[g] * h_.grid_w[g] for g in range(h_.n_grids) if not h_.is_hyperlatent[g])
        ups_bytes = n_lat_px + 4 * L * px_per_step
        res = {
            "metric": "decoded Mpixel/s", "value": world * px_per_step * args.steps / dt / 1e6, "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64+f64 entropy / f32 synthesis",
            "data": "synthetic (kodim14.cool real + 23 streams re-encoded from rolled/transposed kodim14 latents)",
            "config": {"workload": "kodak24", "frames_per_gpu": n_frames, "frame": "512x768 RGB 8-bit, HOP decoder",
                       "symbols_per_step": int(n_sym), "parallelism": f"frames x{world} (round-robin, gather of planes)"},
            "parity": "bit-exact vs CPU oracle (tests/test_gpu_parity.py); <=1 LSB on <=2e-5 of samples vs reference fixture",
            "stage_ms_per_step": stage_ms,
            "entropy_msym_per_s": n_sym / (stage_ms["entropy"] / 1e3) / 1e6,
            # what actually bounds the dominant kernel: every stream is ONE serial range-decoder recurrence; its bare symbol
            # loop runs at 164 ticks of the 2.4 GHz shader clock (tools/ubench/dloop.hip), so n streams cannot
            # exceed n * 2.4e9 / 164 symbols/s however many CUs idle.  The gap is per-batch hand-over and the producer
            # latency chain on the short wavefront steps of the coarse grids.
            "serial_chain_bound": {"achieved": n_sym / (stage_ms["entropy"] / 1e3) / 1e6,
                                   "peak": n_frames * 2.4e9 / 164.0 / 1e6, "unit": "Msymbol/s",
                                   "frac": (n_sym / (stage_ms["entropy"] / 1e3)) / (n_frames * 2.4e9 / 164.0),
                                   "streams": n_frames, "ticks_per_symbol_floor": 164},
            # the dominant kernel (98 % of the step) is the serial range-decoder chain: one workgroup per stream, bound by
            # dependent-instruction latency, not by HBM or MFMA - its HBM fraction only shows how far from that roof it sits
            "roofline": line("entropy_pipe_kernel<5> (24 streams, one workgroup each)", ent_bytes, stage_ms["entropy"],
                             traffic("entropy_pipe_kernel"),
                             "latency-bound serial chain (one range decoder per stream): see entropy_msym_per_s and DESIGN.md 4.1"),
            "roofline_float_stages": [
                line("upsample_step_kernel x6 (whole pyramid, 24 frames)", ups_bytes, stage_ms["upsampling"],
                     traffic("upsample_step_kernel", "_per_step"), "HBM-bound; traffic includes the intermediate stacks"),
                line("syn_fused_kernel<8,3> (all layers + integer planes, 24 frames)", syn_bytes, stage_ms["synthesis"],
                     traffic("syn_fused_kernel"), "above the fp32 ridge for the HOP network: 1344 flop/px over 43 B/px"),
            ],
            "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, raw KiB counters summed "
                              "(profiles/r01/kodak24_pmc_traffic.json; FETCH_SIZE uncalibrated for 4-byte accesses on gfx950)",
        }
        res["with_png_packing"] = png_leg
        res["more_frames_in_flight"] = wide_leg
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(streams)
        print(json.dumps(res))
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
