"""Batched 3-D stability check for generated molecules.

The reference decides bond orders from inter-atomic distances and checks every atom's valence one molecule, one atom
pair at a time in Python (``evaluation/stability.py:17-73`` calling ``evaluation/bond_analyze.py:108-133``): ~400 pair
iterations per molecule, minutes for a 10 000-sample evaluation.  Here the same decision runs for a whole batch as
one HIP kernel (``ds_check_stability``) on the device the sampler left its output on; RDKit molecule construction (the other half of
``check_stability``) stays with the reference's host code.

QM9 atom set only (decoder order H, C, N, O, F - ``datasets/datasets_config.py:4``); distances are in Angstrom and are
compared in picometres exactly as ``get_bond_order`` does: single if ``100 d < L1 + 10``; then double if a double-bond
length exists for the pair and ``100 d < L2 + 5``; then triple if one exists and ``100 d < L3 + 3``.
"""
from __future__ import annotations

import torch

ATOMS = ("H", "C", "N", "O", "F")


def check_stability_batch(pos: torch.Tensor, atom_type: torch.Tensor, node_mask: torch.Tensor, engine=None):
    """-> ``(molecule_stable [B] bool, nr_stable_atoms [B], n_atoms [B], order [B,N,N])`` - the first three values of
    ``check_stability`` (``stability.py:58-73``) for every molecule of the batch, decided by the HIP library
    (``ds_check_stability``, one wave per molecule) on the tensors the sampler left on the GPU.  ``engine`` =
    ``model.engine()`` is required: there is no PyTorch implementation of the decision in this package."""
    if engine is None:
        raise RuntimeError("check_stability_batch runs in the HIP library: pass engine=model.engine()")
    L, _ = engine.layout_for(node_mask)
    return engine.check_stability(L, pos, atom_type)
