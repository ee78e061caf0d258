"""Procedural (hash-seeded) weights and synthetic inputs.

There is no network for the published checkpoints, so tests, golden vectors and
``bench.py`` all use random-init-scale weights that are a pure function of the
tensor *name and shape* (SURVEY §8c): the golden generator (which imports the
reference) and the GPU box (which never sees the reference) regenerate the same
weights bit for bit and the fixtures only have to store outputs.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

from .config import QM9_SECOND_HALF_N_NODES, SPECTRUM_LENGTHS, used_spectra


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()), salt])


def _uniform(name, shape, lo, hi, salt=0):
    r = _rng(name, salt).random(size=tuple(shape), dtype=np.float32)
    return (lo + (hi - lo) * r).astype(np.float32)


def fill_tensor(name: str, shape, like: torch.Tensor | None = None, salt: int = 0) -> torch.Tensor:
    """Deterministic value for the state-dict entry ``name`` of shape ``shape``."""
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.tensor(7, dtype=torch.long)
    if leaf == "running_mean":
        arr = _uniform(name, shape, -0.2, 0.2, salt)
    elif leaf == "running_var":
        arr = _uniform(name, shape, 0.5, 1.5, salt)
    elif "norm" in name and leaf == "weight" and len(shape) == 1:      # BatchNorm / LayerNorm gain
        arr = _uniform(name, shape, 0.8, 1.2, salt)
    elif "norm" in name and leaf == "bias":
        arr = _uniform(name, shape, -0.1, 0.1, salt)
    elif leaf == "scale" and "sdp_attn" in name:                        # frozen d_k^-0.5 (specformer.py:382)
        arr = np.full(shape, 8.0 ** -0.5, dtype=np.float32)
    elif leaf == "scale" and "coord_norm" in name:                      # CoorsNorm scale, init 1e-2
        arr = _uniform(name, shape, 0.5e-2, 2e-2, salt)
    elif ("means" in name or "stds" in name) and leaf == "weight":      # RBF tables, init U(0,3)
        arr = _uniform(name, shape, 0.05, 3.0, salt)
    elif leaf == "weights":                                             # learned sinusoid frequencies
        arr = _uniform(name, shape, -1.5, 1.5, salt)
    elif leaf.startswith("W_pos"):
        arr = _uniform(name, shape, -0.02, 0.02, salt)
    elif leaf == "weight" and len(shape) == 2:
        bound = 1.0 / np.sqrt(shape[1])
        arr = _uniform(name, shape, -bound, bound, salt)
    elif leaf == "bias":
        arr = _uniform(name, shape, -0.05, 0.05, salt)
    else:
        raise KeyError(f"filler has no rule for state-dict entry {name!r} {shape}")
    t = torch.from_numpy(arr)
    if like is not None:
        t = t.to(dtype=like.dtype)
    return t


def fill_state_dict(template: "OrderedDict[str, torch.Tensor]", salt: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Procedural state dict with the names/shapes/dtypes of ``template`` (``module.`` prefix ignored for seeding)."""
    out = OrderedDict()
    for k, v in template.items():
        base = k[len("module."):] if k.startswith("module.") else k
        out[k] = fill_tensor(base, v.shape, like=v, salt=salt)
    return out


def fill_module_(module: torch.nn.Module, salt: int = 0) -> torch.nn.Module:
    module.load_state_dict(fill_state_dict(module.state_dict(), salt), strict=True)
    return module


# ----------------------------------------------------------------------------- synthetic inputs

def normal(name: str, shape, salt: int = 0) -> torch.Tensor:
    return torch.from_numpy(_rng(name, salt).standard_normal(size=tuple(shape), dtype=np.float32))


def uniform(name: str, shape, lo=0.0, hi=1.0, salt: int = 0) -> torch.Tensor:
    return torch.from_numpy(_uniform(name, shape, lo, hi, salt))


def sample_n_atoms(count: int, seed: int = 0) -> np.ndarray:
    """n_atoms drawn from the qm9_second_half histogram (default_rng(0))."""
    sizes = np.array(sorted(QM9_SECOND_HALF_N_NODES), dtype=np.int64)
    w = np.array([QM9_SECOND_HALF_N_NODES[int(s)] for s in sizes], dtype=np.float64)
    return np.random.default_rng(seed).choice(sizes, size=count, p=w / w.sum())


def synthetic_spectra(batch: int, spectra_version: str, seed: int = 1):
    """log10(1+u)-style non-negative spectra (reference build_dataset.py:142-148): u~U[0,1)."""
    rng = np.random.default_rng(seed)
    specs = [torch.from_numpy(np.log10(1.0 + rng.random((batch, 1, L), dtype=np.float32)).astype(np.float32))
             for L in SPECTRUM_LENGTHS]
    idx = used_spectra(spectra_version)
    if spectra_version == "allspectra":
        return specs                      # list [uv, ir, raman] as in sampling.py:427
    return specs[idx[0]]                  # single tensor [B,1,L]


def masks_from_n_atoms(n_atoms, n_max: int | None = None):
    """node_mask [B,N,1], edge_mask [B*N*N,1] exactly as sampling.py:432-439 builds them."""
    n_atoms = [int(n) for n in n_atoms]
    B = len(n_atoms)
    N = int(max(n_atoms)) if n_max is None else int(n_max)
    node_mask = torch.zeros(B, N)
    for i, n in enumerate(n_atoms):
        node_mask[i, :n] = 1
    edge_mask = node_mask.unsqueeze(1) * node_mask.unsqueeze(2)
    edge_mask = edge_mask * (~torch.eye(N, dtype=torch.bool)).unsqueeze(0)
    return node_mask.unsqueeze(2), edge_mask.reshape(B * N * N, 1)


def synthetic_state(n_atoms, name: str = "state", n_max: int | None = None, salt: int = 0):
    """A masked, CoM-free, edge-symmetric sampler state (xh [B,N,9], edge_x [B,N,N,2])."""
    node_mask, edge_mask = masks_from_n_atoms(n_atoms, n_max)
    B, N, _ = node_mask.shape
    pos = normal(name + ".pos", (B, N, 3), salt) * node_mask
    cnt = node_mask.sum(1, keepdim=True)
    pos = pos - (pos.sum(1, keepdim=True) / cnt) * node_mask
    feat = normal(name + ".feat", (B, N, 6), salt) * node_mask
    e = normal(name + ".edge", (B, 2, N, N), salt)
    e = torch.tril(e, -1)
    e = (e + e.transpose(-1, -2)).permute(0, 2, 3, 1) * edge_mask.reshape(B, N, N, 1)
    return torch.cat([pos, feat], dim=2), e.contiguous(), node_mask, edge_mask
