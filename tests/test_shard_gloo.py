"""CPU, world_size=2, gloo: the multi-rank path of bench.py (shard bounds + the single all_gather of records)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffspectra_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.shard_bounds(total, rank, world)
        N = 5
        ids = torch.arange(lo, hi, dtype=torch.float32)
        pos = ids.view(-1, 1, 1).expand(-1, N, 3) + 0.25
        atom = (ids.view(-1, 1).expand(-1, N) % 5).long()
        fc = -(ids.view(-1, 1).expand(-1, N) % 2).long()
        et = ids.view(-1, 1, 1).expand(-1, N, N) % 4
        rec = shard.pack_records(pos, atom, fc, et)
        counts = [shard.shard_bounds(total, r, world)[1] - shard.shard_bounds(total, r, world)[0] for r in range(world)]
        allrec = shard.gather_records(rec, counts)
        p2, a2, f2, e2 = shard.unpack_records(allrec, N)
        want = torch.arange(total, dtype=torch.float32)
        ok = (allrec.shape[0] == total and torch.equal(p2[:, 0, 0], want + 0.25) and torch.equal(a2[:, 0], (want % 5).long())
              and torch.equal(f2[:, 0], -(want % 2).long()) and torch.equal(e2[:, 0, 0], want % 4))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, total = 2, 7                      # uneven split: 4 + 3
    assert [shard.shard_bounds(total, r, world) for r in range(world)] == [(0, 4), (4, 7)]
    assert [shard.shard_bounds(10000, r, 8) for r in (0, 7)] == [(0, 1250), (8750, 10000)]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}
