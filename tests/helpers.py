"""Shared test helpers: procedural state dicts without instantiating the GPU model."""
from __future__ import annotations

import functools

import torch

from diffspectra_amd import filler
from diffspectra_amd.config import qm9s_config
from diffspectra_amd.params import build_dmt_tree, Holder


@functools.lru_cache(maxsize=4)
def procedural_state_dict(version: str):
    """Reference-named state dict (no ``module.`` prefix) with procedural weights."""
    cfg = qm9s_config(spectra_version=version)
    tree = Holder()
    build_dmt_tree(tree, cfg)
    return cfg, filler.fill_state_dict(tree.state_dict())


def max_abs_diff(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().max())
