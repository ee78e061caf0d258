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


def gather_records(rec: torch.Tensor, counts=None) -> torch.Tensor:
    """all_gather of per-rank record blocks → [sum(counts), record] on every rank, in rank order.

    Ranks may hold different molecule counts (``counts``); blocks are padded to the largest for the collective.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
