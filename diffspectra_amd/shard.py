"""Molecule sharding across ranks (one process per GPU) and the single end-of-sampling collective.

Sampling is embarrassingly parallel over molecules: rank r of R owns a contiguous block of the
evaluation set, no data-path collective runs during the 1000 steps, and the fixed-size result records are gathered
once with ``all_gather`` (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) block of ``total`` molecules for ``rank`` (first ``total % world`` ranks get +1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_records(pos, atom_type, fc, edge_type) -> torch.Tensor:
    """One fixed-size fp32 record per molecule: pos [N*3] | atom type [N] | charge [N] | bond order [N*N]."""
    B = pos.shape[0]
    return torch.cat([pos.reshape(B, -1).float(), atom_type.reshape(B, -1).float(), fc.reshape(B, -1).float(),
                      edge_type.reshape(B, -1).float()], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, N: int):
    B = rec.shape[0]
    o = 0
    pos = rec[:, o:o + 3 * N].reshape(B, N, 3); o += 3 * N
    atom = rec[:, o:o + N].long(); o += N
    fc = rec[:, o:o + N].long(); o += N
    et = rec[:, o:o + N * N].reshape(B, N, N)
    return pos, atom, fc, et


RECORD_ATOMS = 29          # DS_MAX_ATOMS: records have one width whatever the padded width of the batch they came from
RECORD_BYTES = 1248        # 29*3 f32 (348) + 29 u8 atom types + 29 i8 charges + 29*29 u8 bond orders (841) = 1247, padded to 16


def pack_records_u8(pos, atom_type, fc, edge_type) -> torch.Tensor:
    """One 1 248-byte record per molecule: positions stay fp32, the integer outputs travel as bytes.
    ``pos [B,N,3]`` f32, ``atom_type [B,N]`` in 0..4, ``fc [B,N]`` (or ``[B,N,1]``) in -128..127, ``edge_type [B,N,N]`` in 0..3;
    N <= 29, padded entries zero."""
    B, N = pos.shape[0], pos.shape[1]
    if N > RECORD_ATOMS:
        raise ValueError(f"records hold at most {RECORD_ATOMS} atoms")
    W, dev = RECORD_ATOMS, pos.device
    p = torch.zeros(B, W, 3, dtype=torch.float32, device=dev)
    p[:, :N] = pos.float()
    a = torch.zeros(B, W, dtype=torch.uint8, device=dev)
    a[:, :N] = atom_type.reshape(B, N).to(torch.uint8)
    c = torch.zeros(B, W, dtype=torch.int8, device=dev)
    c[:, :N] = fc.reshape(B, N).to(torch.int8)
    e = torch.zeros(B, W, W, dtype=torch.uint8, device=dev)
    e[:, :N, :N] = edge_type.to(torch.uint8)
    rec = torch.zeros(B, RECORD_BYTES, dtype=torch.uint8, device=dev)
    rec[:, :348] = p.reshape(B, W * 3).view(torch.uint8)            # (explicit widths: a rank without molecules packs B = 0 rows)
    rec[:, 348:377] = a
    rec[:, 377:406] = c.view(torch.uint8)
    rec[:, 406:1247] = e.reshape(B, W * W)
    return rec


def unpack_records_u8(rec: torch.Tensor):
    """→ ``(pos [M,29,3] f32, atom_type [M,29] i64, fc [M,29] i64, edge_type [M,29,29] f32)``."""
    M = rec.shape[0]
    pos = rec[:, :348].contiguous().view(torch.float32).reshape(M, RECORD_ATOMS, 3)
    atom = rec[:, 348:377].long()
    fc = rec[:, 377:406].contiguous().view(torch.int8).long()
    et = rec[:, 406:1247].reshape(M, RECORD_ATOMS, RECORD_ATOMS).float()
    return pos, atom, fc, et


def assign_slots(n_atoms, rank: int, world: int):
    """Sample slots owned by ``rank``: slots sorted by molecule size (largest first, stable) and dealt round-robin, so every
    rank gets the same mix of sizes (cost grows as n(n-1)) and its own list is size-sorted, i.e. its micro-batches are
    n-bucketed.  Returns an int64 tensor of slot indices."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    n = torch.as_tensor(n_atoms, dtype=torch.int64).reshape(-1)
    order = torch.argsort(-n, stable=True)
    return order[rank::world]


# Rehearsal switch: with a process group of ONE rank the collectives below are identities and are skipped;
# force_collectives(True) (or DIFFSPECTRA_FORCE_COLLECTIVES=1) runs them anyway, so that a single GPU with a world-size-1 `nccl`
# group exercises every RCCL call site of the path (dtype / contiguity / aliasing errors show up before an 8-GPU run does).
import os as _os
_FORCE = _os.environ.get("DIFFSPECTRA_FORCE_COLLECTIVES", "0") == "1"


def force_collectives(on: bool = True) -> None:
    global _FORCE
    _FORCE = bool(on)


def collectives_on() -> bool:
    """True when the collectives of this module must really run: a process group with more than one rank, or the rehearsal switch."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _FORCE


def world_info():
    """(rank, world) of the initialised default process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_from_rank0(t: torch.Tensor, device) -> torch.Tensor:
    """Every rank gets rank 0's ``t`` (a small host tensor: permutations, seeds); identity without a process group."""
    if not collectives_on():
        return t
    buf = t.clone() if dist.get_backend() == "gloo" else t.to(device)
    dist.broadcast(buf, src=0)
    return buf.cpu()


def all_gather_counts(count: torch.Tensor, device):
    """Every rank's value of the one-element int64 host tensor ``count`` (list of ints, rank order)."""
    if not collectives_on():
        return [int(count.item())]
    world = dist.get_world_size()
    buf = count.clone() if dist.get_backend() == "gloo" else count.to(device)
    out = torch.empty(world, dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    return out.cpu().tolist()


def gather_records(rec: torch.Tensor, counts=None) -> torch.Tensor:
    """all_gather of per-rank record blocks → [sum(counts), record] on every rank, in rank order.

    Ranks may hold different molecule counts (``counts``); blocks are padded to the largest for the collective.
    """
    if not collectives_on():
        return rec
    world = dist.get_world_size()
    if counts is None:
        counts = [rec.shape[0]] * world
    m = max(counts)
    buf = rec
    if rec.shape[0] < m:
        buf = torch.cat([rec, rec.new_zeros(m - rec.shape[0], rec.shape[1])], 0)
    dev = rec.device
    if dist.get_backend() == "gloo" and dev.type != "cpu":
        buf = buf.cpu()                          # rehearsal path: gloo gathers host tensors; RCCL gathers in HBM
    out = torch.empty(world * m, rec.shape[1], dtype=rec.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous())
    out = out.to(dev)
    return torch.cat([out[r * m:r * m + counts[r]] for r in range(world)], 0)


def gather_by_slot(rec: torch.Tensor, n_atoms) -> torch.Tensor:
    """The closing collective of a sharded evaluation: this rank's records (rows in the order of ``assign_slots(n_atoms, rank, world)``)
    -> the records of ALL sample slots in slot order, on every rank.  One ``all_gather_into_tensor`` (``gather_records``); ranks may own
    different numbers of slots, or none."""
    rank, world = world_info()
    owners = [assign_slots(n_atoms, r, world) for r in range(world)]
    if rec.shape[0] != owners[rank].numel():
        raise ValueError(f"rank {rank} holds {rec.shape[0]} records for {owners[rank].numel()} slots")
    allrec = gather_records(rec, [o.numel() for o in owners])
    by_slot = torch.empty_like(allrec)
    by_slot[torch.cat(owners).to(allrec.device)] = allrec
    return by_slot
