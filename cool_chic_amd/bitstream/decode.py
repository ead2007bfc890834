"""decode_video / decode_frame on the MI355X - mirror of coolchic/bitstream/decode.py:26-212.

All-intra streams (images, image sets) take the batched path: every frame's cool-chic is added to
one DecodeBatch, all of them decode concurrently, and the reference's rounding / clamping / 4:2:0
chain (decode.py:191-206) runs in the HIP epilogue that produces the integer planes."""
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..batch import DecodeBatch
from ..io import FrameData, save_frame_data_to_file
from .component.coolchic import encode_decode_coolchic
from .header import CoolChicHeader, FrameHeader, VideoHeader

_FDT_INDEX = {"rgb": 0, "yuv420": 1, "yuv444": 2, "flow": 3}


def _planes_to_frame_data(planes: List[torch.Tensor], bitdepth: int, frame_data_type: str) -> FrameData:
    maxv = float(2 ** bitdepth - 1)
    f = [p.to(torch.float32).div(maxv)[None, None] for p in planes]
    if frame_data_type == "yuv420":
        return FrameData(bitdepth, frame_data_type, {"y": f[0], "u": f[1], "v": f[2]})
    return FrameData(bitdepth, frame_data_type, torch.cat(f, dim=1))


def _own_planes(batch: DecodeBatch, slot: int, bitdepth: int, device) -> List[torch.Tensor]:
    """The slot's integer planes in memory the caller owns: ONE device-to-device copy of the block the three planes share
    (the arena goes back to the pool when the batch closes), then three views of the copy."""
    block, layout = batch.planes_block_device(slot)
    own = torch.as_tensor(block, device=device).clone()
    dt, sb = (torch.uint8, 1) if bitdepth == 8 else (torch.uint16, 2)
    return [own[off: off + h * w * sb].view(dt).view(h, w) for off, (h, w) in layout]


def _split_frame(bitstream_bytes: bytes):
    """decode.py:115-143: frame header, then per cool-chic header + NN bytes + latent bytes."""
    fh = FrameHeader()
    rest = fh.read_header(bitstream_bytes)
    ccs = []
    for _name in (["residue"] + (["motion"] if fh.get_value("frame_type") in ("P", "B") else [])):
        ch = CoolChicHeader()
        rest = ch.read_header(rest)
        n_nn, n_lat = ch.get_value("nn_n_bytes"), ch.get_value("n_bytes_latent")
        if len(rest) < n_nn + n_lat:
            raise ValueError("bitstream truncated")
        ccs.append((ch, rest[:n_nn], rest[n_nn:n_nn + n_lat]))
        rest = rest[n_nn + n_lat:]
    return fh, ccs, rest


@torch.no_grad()
def decode_frame(bitstream_bytes: bytes, reference_frames: List[FrameData], verbosity: int = 0,
                 device: int = 0) -> Tuple[FrameData, bytes]:
    """decode.py:96-212. Returns the decoded FrameData and the remaining bytes."""
    fh, ccs, rest = _split_frame(bitstream_bytes)
    if verbosity:
        print(fh.pretty_string())
    frame_type, bitdepth, fdt = fh.get_value("frame_type"), fh.get_value("bitdepth"), fh.get_value("frame_data_type")
    if frame_type != "I":
        from .intercoding import reconstruct_inter_frame  # P / B frames

        outs = [encode_decode_coolchic(ch, nn, "decode", dec_bytes_latent=lat, verbosity=verbosity,
                                       device=f"cuda:{device}")[0] for ch, nn, lat in ccs]
        return reconstruct_inter_frame(fh, outs[0], outs[1], reference_frames), rest
    ch, nn, lat = ccs[0]
    batch = DecodeBatch(device)
    try:
        batch.add(ch.raw, nn, lat, bitdepth, _FDT_INDEX[fdt])
        stream = torch.cuda.current_stream(device).cuda_stream
        batch.run(stream)
        batch.wait(stream)
        planes = _own_planes(batch, 0, bitdepth, f"cuda:{device}")
    finally:
        batch.close()
    return _planes_to_frame_data(planes, bitdepth, fdt), rest


@torch.no_grad()
def decode_video(bitstream_path: str, decoded_path: Optional[str] = None, max_decoding_order: int = -1,
                 verbosity: int = 0, device: int = 0) -> Dict[str, FrameData]:
    """decode.py:26-91: decode a .cool file; returns {display index as str: FrameData}."""
    with open(bitstream_path, "rb") as f:
        bitstream_bytes = f.read()
    vh = VideoHeader()
    bitstream_bytes = vh.read_header(bitstream_bytes)
    if verbosity:
        print(vh.pretty_string())
    n_frames = vh.get_value("n_frames")
    all_intra = vh.get_value("n_intras") == n_frames
    if max_decoding_order == -1:
        max_decoding_order = n_frames - 1
    frames: Dict[int, FrameData] = {}
    if all_intra:
        # every frame is independent: one batch, all cool-chics in flight together
        start = time.time()
        batch = DecodeBatch(device)
        try:
            meta = []
            structure = vh.get_coding_structure()
            for k in range(max_decoding_order + 1):
                fh, ccs, bitstream_bytes = _split_frame(bitstream_bytes)
                if fh.get_value("frame_type") != "I":
                    # a P / B header needs references an all-intra structure does not give (decode.py:160: IndexError)
                    raise ValueError(f"frame {k} (coding order): header says {fh.get_value('frame_type')}, the coding structure I")
                ch, nn, lat = ccs[0]
                bd, fdt = fh.get_value("bitdepth"), fh.get_value("frame_data_type")
                batch.add(ch.raw, nn, lat, bd, _FDT_INDEX[fdt])
                meta.append((structure[k]["display_order"], bd, fdt))  # decode.py:67: the structure says which frame this is
            stream = torch.cuda.current_stream(device).cuda_stream
            batch.run(stream)
            batch.wait(stream)
            for slot, (di, bd, fdt) in enumerate(meta):
                frames[di] = _planes_to_frame_data(_own_planes(batch, slot, bd, f"cuda:{device}"), bd, fdt)
        finally:
            batch.close()
        print(f"Decoding {len(frames)} intra frame(s) time = {time.time() - start:6.2f} seconds.")
    else:
        # P / B frames: every cool-chic of every frame decodes in ONE batch (they do not depend on other frames,
        # decode.py:132-153); the frames are then reconstructed in coding order (ccd_decode_video does the same)
        start = time.time()
        frames = _decode_gop(bitstream_bytes, max_decoding_order + 1, device, None, verbosity, sharded=False,
                             structure=vh.get_coding_structure())
        print(f"Decoding {len(frames)} frame(s) time = {time.time() - start:6.2f} seconds.")
    # decode.py:84-89: one entry per display index of the sequence; frames beyond max_decoding_order stay None
    all_frames = {}
    for display_idx in range(n_frames):
        all_frames[str(display_idx)] = frames.get(display_idx)
        if decoded_path is not None and display_idx in frames:
            save_frame_data_to_file(frames[display_idx], decoded_path, append=display_idx != 0)
    return all_frames


def _decode_gop(rest: bytes, n_decode: int, device: int, group, verbosity: int = 0, collect: Optional[int] = 0,
                sharded: bool = True, structure: Optional[list] = None) -> Dict[int, FrameData]:
    """The first `n_decode` frames (coding order) of the frame payload `rest`: {display index: FrameData}.

    With a process group, frame at coding index k belongs to rank k mod world_size: every rank decodes ALL cool-chics
    of its own frames in one DecodeBatch (residue and motion networks in flight together), then the frames are
    reconstructed in coding order on their owner; a frame's integer planes go point to point to the ranks whose frames
    predict from it and to rank `collect`, which returns the whole sequence (other ranks: what they produced or used)."""
    import torch.distributed as dist

    from ..parallel import gop_owner, run_sharded_gop
    from .intercoding import _integer_planes, reconstruct_inter_frame

    initialised = sharded and dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    rank = dist.get_rank(group) if initialised else 0
    dev = torch.device(f"cuda:{device}")
    parsed = []
    for _ in range(n_decode):
        fh, ccs, rest = _split_frame(rest)
        if verbosity:
            print(fh.pretty_string())
        parsed.append((fh, ccs))
    # decode.py:67-75: which frame sits at coding index k and what it predicts from come from the VIDEO header's coding
    # structure; the frame headers' display_index / index_references are never read there.  The header's frame_type alone
    # decides the cool-chics per frame and the reconstruction (decode.py:119-128, 156-189): an "I" header at a P / B position
    # is plain intra, a "P" header at a B position predicts from the structure's first reference.  Rejected is what the
    # reference raises on (IndexError): a header type that needs more references than the structure gives.  Without a
    # structure (decode_frame-style callers) the headers are followed.
    if structure is not None:
        for k, (fh, _) in enumerate(parsed):
            if "IPB".index(fh.get_value("frame_type")) > len(structure[k]["index_references"]):
                raise ValueError(f"frame {k} (coding order): header says {fh.get_value('frame_type')}, the coding structure "
                                 f"{structure[k]['frame_type']}{structure[k]['display_order']} gives it "
                                 f"{len(structure[k]['index_references'])} reference(s)")
        display = [structure[k]["display_order"] for k in range(n_decode)]
        ref_display = [list(structure[k]["index_references"]) for k in range(n_decode)]
    else:
        display = [fh.get_value("display_index") for fh, _ in parsed]
        ref_display = [list(fh.get_value("index_references")) for fh, _ in parsed]
    if len(set(display)) != len(display):
        raise ValueError("two frames share a display index")
    coding_of_display = {d: k for k, d in enumerate(display)}
    references = []
    for k in range(n_decode):
        refs = [coding_of_display.get(r, n_decode) for r in ref_display[k]]
        if any(r >= k for r in refs):
            raise ValueError("a frame references a frame that is not decoded before it")
        references.append(refs)
    specs = []
    for fh, ccs in parsed:
        h, w = ccs[0][0].c.img_size[0], ccs[0][0].c.img_size[1]
        bd, fdt = fh.get_value("bitdepth"), fh.get_value("frame_data_type")
        dt = torch.uint8 if bd == 8 else torch.uint16
        chroma = (h // 2, w // 2) if fdt == "yuv420" else (h, w)
        specs.append([((h, w), dt), (chroma, dt), (chroma, dt)])
    batch = DecodeBatch(device)
    try:
        slots = {}
        for k, (fh, ccs) in enumerate(parsed):
            if gop_owner(k, world) != rank:
                continue
            intra = fh.get_value("frame_type") == "I"
            bd, fdt = fh.get_value("bitdepth"), fh.get_value("frame_data_type")
            slots[k] = [batch.add(ch.raw, nn, lat, bd if intra else 0, _FDT_INDEX[fdt] if intra else 0) for ch, nn, lat in ccs]
        stream = torch.cuda.current_stream(device).cuda_stream
        if slots:
            batch.run(stream)
            batch.wait(stream)

        def to_frame_data(k, planes):
            fh = parsed[k][0]
            return _planes_to_frame_data(planes, fh.get_value("bitdepth"), fh.get_value("frame_data_type"))

        def produce(k, refs):
            fh = parsed[k][0]
            if fh.get_value("frame_type") == "I":
                return _own_planes(batch, slots[k][0], fh.get_value("bitdepth"), dev)
            outs = [torch.as_tensor(batch.output_device(s), device=dev) for s in slots[k]]
            ref_fd = [to_frame_data(r, pl) for r, pl in zip(references[k], refs)]
            return _integer_planes(reconstruct_inter_frame(fh, outs[0], outs[1], ref_fd), dev)

        if initialised:
            done = run_sharded_gop(n_decode, specs, references, produce, device=dev, group=group, collect=collect)
        else:  # one process: plain coding-order loop
            done = {}
            for k in range(n_decode):
                done[k] = list(produce(k, [done[r] for r in references[k]]))
    finally:
        batch.close()
    return {display[k]: to_frame_data(k, done[k]) for k in done}


@torch.no_grad()
def decode_video_sharded(bitstream_path: str, device: int = 0, group=None, collect: Optional[int] = 0) -> Dict[str, FrameData]:
    """decode_video for one process per GPU (torch.distributed initialised by the caller; works unsharded without).
    Rank `collect` of the group returns the whole sequence like decode_video; the other ranks return the frames they
    produced or received as references (see _decode_gop)."""
    with open(bitstream_path, "rb") as f:
        rest = f.read()
    vh = VideoHeader()
    rest = vh.read_header(rest)
    frames = _decode_gop(rest, vh.get_value("n_frames"), device, group, collect=collect, structure=vh.get_coding_structure())
    return {str(d): frames[d] for d in sorted(frames)}
