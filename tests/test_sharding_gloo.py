"""Multi-process (world_size 2, gloo, CPU) test of the frame sharding + plane gather used by
bench.py --gpus N and decode of image sets."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cool_chic_amd.parallel import (EqualSizeGather, gather_bytes, gop_consumers, gop_owner, pack_planes, run_sharded_gop, shard_indices,
                                    unshard)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame(i):  # deterministic fake decoded frame of a size that depends on i
    rng = np.random.default_rng(i)
    h, w = 8 + i, 12 + 2 * i
    return [torch.from_numpy(rng.integers(0, 256, (h, w), dtype=np.uint8)) for _ in range(3)]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_frames, rank, world)
    local = [pack_planes(_frame(i)) for i in mine]
    lens = torch.tensor([t.numel() for t in local], dtype=torch.int64)
    blob = torch.cat(local) if local else torch.empty(0, dtype=torch.uint8)
    got = gather_bytes(blob, dst=0)
    got_lens = gather_bytes(lens.view(torch.uint8).reshape(-1), dst=0)
    if rank == 0:
        per_rank = []
        for b, l in zip(got, got_lens):
            l = l.view(torch.int64)
            per_rank.append(list(torch.split(b, [int(x) for x in l])))
        frames = unshard(per_rank, n_frames)
        ok = all(torch.equal(f, pack_planes(_frame(i))) for i, f in enumerate(frames))
    # the fixed-size, persistent-buffer gather bench.py uses at N > 1 (same geometry on every rank), called twice
    planes = [torch.full((4, 6), 10 * rank + p, dtype=torch.uint8) for p in range(3)] + [torch.full((2, 3), 1000 + rank, dtype=torch.int16).view(torch.uint16)]
    g = EqualSizeGather(sum(p.numel() * p.element_size() for p in planes), "cpu", dst=0)
    for rep in range(2):
        got2 = g(planes)
        if rank == 0:
            for r in range(world):
                want = torch.cat([torch.full((24,), 10 * r + p, dtype=torch.uint8) for p in range(3)]
                                 + [torch.full((6,), 1000 + r, dtype=torch.int16).view(torch.uint8)])
                ok = ok and torch.equal(got2[r], want)
        else:
            assert got2 is None
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_and_gather_world2():
    assert shard_indices(5, 0, 2) == [0, 2, 4] and shard_indices(5, 1, 2) == [1, 3]
    assert shard_indices(1, 1, 2) == []
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok


# ---- a GOP sharded over ranks: coding-order schedule + point-to-point hand-over of reference planes -------------------
# hierarchical GOP like the reference's (I0 P4 B2 B1 B3 I8 B6 ...): references by coding index
_GOP_REFS = [[], [0], [0, 1], [0, 2], [2, 1], [], [1, 5], [1, 6], [6, 5]]


def _gop_specs(k):
    h, w = 6 + (k % 3), 10
    dt = torch.uint8 if k % 2 == 0 else torch.uint16
    return [((h, w), dt), ((h // 2, w // 2), dt), ((h // 2, w // 2), dt)]


def _gop_produce(k, refs):
    """Fake reconstruction: every sample depends on k and on the content of every reference plane."""
    mix = sum(int(p.to(torch.int64).sum()) * (i + 1) for i, r in enumerate(refs) for p in r)
    out = []
    for j, (shape, dt) in enumerate(_gop_specs(k)):
        base = torch.arange(shape[0] * shape[1], dtype=torch.int64).reshape(shape)
        out.append(((base * (k + 3) + 7 * j + mix) % (256 if dt == torch.uint8 else 1024)).to(torch.int32).to(dt))
    return out


def _gop_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def produce(k, refs):
        calls.append(k)
        return _gop_produce(k, refs)

    n = len(_GOP_REFS)
    done = run_sharded_gop(n, [_gop_specs(k) for k in range(n)], _GOP_REFS, produce, device="cpu")
    q.put((rank, calls, {k: [p.to(torch.int32).numpy() for p in v] for k, v in done.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gop_schedule_world2():
    n = len(_GOP_REFS)
    serial = run_sharded_gop(n, [_gop_specs(k) for k in range(n)], _GOP_REFS, _gop_produce)  # no process group: one rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    consumers = gop_consumers(n, _GOP_REFS, 2, collect=0)
    for rank, calls, done in results:
        assert calls == [k for k in range(n) if gop_owner(k, 2) == rank]  # only its own frames, in coding order
        # a rank holds what it produced and what it was sent: everything on the collector (rank 0), only the
        # references of its own frames elsewhere - nothing is broadcast
        assert sorted(done) == [k for k in range(n) if gop_owner(k, 2) == rank or rank in consumers[k]]
        if rank == 0:
            assert sorted(done) == list(range(n))
        else:
            assert len(done) < n
        for k in sorted(done):
            for got, want, (shape, dt) in zip(done[k], serial[k], _gop_specs(k)):
                assert got.shape == tuple(shape)
                assert np.array_equal(got, want.to(torch.int32).numpy()), (rank, k)


def test_gop_consumers_are_only_the_predicting_ranks():
    """Hierarchical GOP of 9 frames on 4 ranks: a frame goes to the owners of the frames that predict from it (+ collector)."""
    cons = gop_consumers(len(_GOP_REFS), _GOP_REFS, 4, collect=None)
    for k, need in enumerate(cons):
        want = sorted({gop_owner(j, 4) for j, refs in enumerate(_GOP_REFS) if k in refs} - {gop_owner(k, 4)})
        assert need == want
    assert max(len(c) for c in cons) <= 3 and cons[8] == []
    with_collector = gop_consumers(len(_GOP_REFS), _GOP_REFS, 4, collect=0)
    assert all((0 in c) or gop_owner(k, 4) == 0 for k, c in enumerate(with_collector))
