"""Batched 3-D stability check for generated molecules.

The reference decides bond orders from inter-atomic distances and checks every atom's valence one molecule, one atom
pair at a time in Python (``evaluation/stability.py:17-73`` calling ``evaluation/bond_analyze.py:108-133``): ~400 pair
iterations per molecule, minutes for a 10 000-sample evaluation.  Here the same decision runs for a whole batch as
tensor operations on the device the sampler left its output on; RDKit molecule construction (the other half of
``check_stability``) stays with the reference's host code.

QM9 atom set only (decoder order H, C, N, O, F - ``datasets/datasets_config.py:4``); distances are in Angstrom and are
compared in picometres exactly as ``get_bond_order`` does: single if ``100 d < L1 + 10``; then double if a double-bond
length exists for the pair and ``100 d < L2 + 5``; then triple if one exists and ``100 d < L3 + 3``.
"""
from __future__ import annotations

import torch

ATOMS = ("H", "C", "N", "O", "F")
_NONE = -1.0e9          # "no such bond for this pair": the threshold test can never pass
#                H      C      N      O      F
_SINGLE = [[74.0, 109.0, 101.0, 96.0, 92.0],
           [109.0, 154.0, 147.0, 143.0, 135.0],
           [101.0, 147.0, 145.0, 140.0, 136.0],
           [96.0, 143.0, 140.0, 148.0, 142.0],
           [92.0, 135.0, 136.0, 142.0, 142.0]]
_DOUBLE = [[_NONE] * 5,
           [_NONE, 134.0, 129.0, 120.0, _NONE],
           [_NONE, 129.0, 125.0, 121.0, _NONE],
           [_NONE, 120.0, 121.0, 121.0, _NONE],
           [_NONE] * 5]
_TRIPLE = [[_NONE] * 5,
           [_NONE, 120.0, 116.0, 113.0, _NONE],
           [_NONE, 116.0, 110.0, _NONE, _NONE],
           [_NONE, 113.0, _NONE, _NONE, _NONE],
           [_NONE] * 5]
_MARGINS = (10.0, 5.0, 3.0)
_VALENCE = (1, 4, 3, 2, 1)          # allowed_bonds for H, C, N, O, F (bond_analyze.py:90)


def bond_orders(pos: torch.Tensor, atom_type: torch.Tensor, node_mask: torch.Tensor) -> torch.Tensor:
    """``pos [B,N,3]`` (Angstrom), ``atom_type [B,N]`` (0..4), ``node_mask [B,N]`` or ``[B,N,1]`` -> ``[B,N,N]`` int64 in
    {0,1,2,3}; zero on the diagonal and wherever either atom is padding."""
    dev = pos.device
    t = atom_type.long()
    mask = node_mask.reshape(pos.shape[0], pos.shape[1]).bool()
    l1, l2, l3 = (torch.tensor(tab, dtype=torch.float32, device=dev)[t.unsqueeze(2), t.unsqueeze(1)]
                  for tab in (_SINGLE, _DOUBLE, _TRIPLE))
    diff = pos.float().unsqueeze(2) - pos.float().unsqueeze(1)
    d = torch.sqrt((diff * diff).sum(-1)) * 100.0
    single = d < l1 + _MARGINS[0]
    double = single & (d < l2 + _MARGINS[1])
    triple = double & (d < l3 + _MARGINS[2])
    order = single.long() + double.long() + triple.long()
    pair_ok = mask.unsqueeze(2) & mask.unsqueeze(1) & ~torch.eye(pos.shape[1], dtype=torch.bool, device=dev).unsqueeze(0)
    return order * pair_ok.long()


def check_stability_batch(pos: torch.Tensor, atom_type: torch.Tensor, node_mask: torch.Tensor, engine=None):
    """-> ``(molecule_stable [B] bool, nr_stable_atoms [B], n_atoms [B], order [B,N,N])`` - the first three values of
    ``check_stability`` (``stability.py:58-73``) for every molecule of the batch.

    With ``engine`` (``model.engine()``) the decision runs in the HIP library (``ds_check_stability``, one workgroup per
    molecule) on the tensors the sampler left on the GPU; without it, as tensor operations on whatever device the inputs
    live on (host-side analytics, where the reference runs its Python loops)."""
    if engine is not None:
        L, _ = engine.layout_for(node_mask)
        return engine.check_stability(L, pos, atom_type)
    order = bond_orders(pos, atom_type, node_mask)
    mask = node_mask.reshape(pos.shape[0], pos.shape[1]).bool()
    valence = torch.tensor(_VALENCE, dtype=torch.long, device=pos.device)[atom_type.long()]
    stable_atom = (order.sum(-1) == valence) & mask
    n_atoms = mask.sum(-1)
    nr_stable = stable_atom.sum(-1)
    return nr_stable == n_atoms, nr_stable, n_atoms, order
