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


def _worker_u8(rank, world, port, n_atoms, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert shard.world_info() == (rank, world)
        mine = shard.assign_slots(n_atoms, rank, world)
        N = 6
        ids = mine.float()
        pos = ids.view(-1, 1, 1).expand(-1, N, 3) * 0.5 - 1.25
        atom = (mine.view(-1, 1).expand(-1, N) % 5)
        fc = (mine.view(-1, 1).expand(-1, N) % 5) - 2
        et = (mine.view(-1, 1, 1).expand(-1, N, N) % 4).float()
        rec = shard.pack_records_u8(pos, atom, fc, et)
        counts = [shard.assign_slots(n_atoms, r, world).numel() for r in range(world)]
        allrec = shard.gather_records(rec, counts)
        order = torch.cat([shard.assign_slots(n_atoms, r, world) for r in range(world)])
        by_slot = torch.empty_like(allrec)
        by_slot[order] = allrec
        p2, a2, f2, e2 = shard.unpack_records_u8(by_slot)
        k = torch.arange(len(n_atoms))
        ok = (torch.equal(p2[:, 0, 0], k.float() * 0.5 - 1.25) and torch.equal(a2[:, 0], k % 5) and torch.equal(f2[:, 3], k % 5 - 2)
              and torch.equal(e2[:, 1, 2], (k % 4).float()) and float(p2[:, N:].abs().max()) == 0.0 and int(e2[:, N:].abs().max()) == 0)
        perm = shard.broadcast_from_rank0(torch.randperm(11, generator=torch.Generator().manual_seed(100 + rank)), "cpu")
        ret[rank] = (bool(ok), perm.tolist())
    finally:
        dist.destroy_process_group()


def test_u8_records_and_slot_assignment():
    n_atoms = [9, 29, 3, 18, 18, 12, 21, 5, 16]
    parts = [shard.assign_slots(n_atoms, r, 3) for r in range(3)]
    assert sorted(torch.cat(parts).tolist()) == list(range(9))                      # a partition of the slots
    for part in parts:
        sizes = [n_atoms[i] for i in part.tolist()]
        assert sizes == sorted(sizes, reverse=True)                                 # n-bucketed inside a rank
    cost = [sum(n_atoms[i] * (n_atoms[i] - 1) for i in p.tolist()) for p in parts]
    assert max(cost) - min(cost) <= 29 * 28                                          # balanced to within one molecule
    assert shard.assign_slots(n_atoms, 0, 1).tolist() == torch.argsort(-torch.tensor(n_atoms), stable=True).tolist()
    assert shard.world_info() == (0, 1)
    g = torch.Generator().manual_seed(0)
    B, N = 5, 7
    pos = torch.randn(B, N, 3, generator=g)
    atom = torch.randint(0, 5, (B, N), generator=g)
    fc = torch.randint(-3, 4, (B, N, 1), generator=g)
    et = torch.randint(0, 4, (B, N, N), generator=g).float()
    rec = shard.pack_records_u8(pos, atom, fc, et)
    assert rec.shape == (B, shard.RECORD_BYTES) and rec.dtype == torch.uint8 and shard.RECORD_BYTES == 1248
    p2, a2, f2, e2 = shard.unpack_records_u8(rec)
    assert torch.equal(p2[:, :N], pos) and torch.equal(a2[:, :N], atom) and torch.equal(f2[:, :N], fc.squeeze(-1))
    assert torch.equal(e2[:, :N, :N], et) and float(p2[:, N:].abs().max()) == 0.0
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_u8, args=(r, 2, port, n_atoms, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == ret[1][1] == torch.randperm(11, generator=torch.Generator().manual_seed(100)).tolist()


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


def _worker_8(rank, world, port, ret):
    """Eight gloo ranks, three evaluations through the product's closing collective (shard.gather_by_slot): an uneven share
    (10 001 sample slots: 1 251 + 7 x 1 250), ranks without a molecule (5 slots on 8 ranks), Top-K slots (K consecutive slots per
    spectrum, sampling.py's `perm[:n].repeat_interleave(K)`)."""
    import numpy as np
    from diffspectra_amd import filler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}

        def run(n_atoms, tag):
            n_atoms = [int(v) for v in n_atoms]
            mine = shard.assign_slots(n_atoms, rank, world)
            N = 29
            # a record whose every field is a function of the SLOT: position = slot / 8, atom type = slot % 5, charge = slot % 7 - 3, bond = slot % 4
            pos = (mine.float() / 8.0).view(-1, 1, 1).expand(-1, N, 3)
            atom = (mine % 5).view(-1, 1).expand(-1, N)
            fc = (mine % 7 - 3).view(-1, 1).expand(-1, N)
            et = (mine % 4).float().view(-1, 1, 1).expand(-1, N, N)
            rec = shard.pack_records_u8(pos, atom, fc, et)
            by_slot = shard.gather_by_slot(rec, n_atoms)
            p2, a2, f2, e2 = shard.unpack_records_u8(by_slot)
            k = torch.arange(len(n_atoms))
            ok = (by_slot.shape[0] == len(n_atoms) and torch.equal(p2[:, 3, 1], k.float() / 8.0) and torch.equal(a2[:, 0], k % 5)
                  and torch.equal(f2[:, 28], k % 7 - 3) and torch.equal(e2[:, 2, 5], (k % 4).float()))
            sizes = [n_atoms[i] for i in mine.tolist()]
            out[tag] = (bool(ok), int(mine.numel()), sizes == sorted(sizes, reverse=True),
                        int(sum(v * (v - 1) for v in sizes)))
        run(filler.sample_n_atoms(10001, seed=0), "uneven")
        run([9, 29, 3, 18, 12], "empty_ranks")
        K = 10
        base = filler.sample_n_atoms(125, seed=3)
        run(np.repeat(base, K), "topk")
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_eight_ranks_uneven_empty_and_topk():
    """World size 8 (the node the SCALE driver uses) on CPU: per-rank counts, size-sorted shares, cost balance, and the gathered records
    in slot order on every rank - with 10 001 slots, with three ranks that own nothing, and with Top-K slots."""
    world = 8
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_8, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(ret)
    assert sorted(res) == list(range(world))
    for tag, counts in (("uneven", [1251] + [1250] * 7), ("empty_ranks", [1, 1, 1, 1, 1, 0, 0, 0]), ("topk", [157, 157] + [156] * 6)):
        assert [res[r][tag][1] for r in range(world)] == counts, tag
        assert all(res[r][tag][0] for r in range(world)), tag            # every rank sees every slot's record in slot order
        assert all(res[r][tag][2] for r in range(world)), tag            # n-bucketed inside a rank
    cost = [res[r]["uneven"][3] for r in range(world)]
    assert max(cost) - min(cost) <= 29 * 28 + 28 * 27                    # round-robin over the size-sorted slots: within ~one large molecule
    cost = [res[r]["topk"][3] for r in range(world)]
    assert max(cost) - min(cost) <= 2 * 29 * 28
