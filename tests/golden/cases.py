"""Shared definitions of the golden cases: inputs are regenerated procedurally on both sides
(the generator that imports the reference, and the tests that never see it), so the committed
fixtures hold expected OUTPUTS only."""
from __future__ import annotations

import os

import numpy as np
import torch

from diffspectra_amd import filler
from diffspectra_amd.config import qm9s_config, SPECTRUM_LENGTHS

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
RAGGED = [3, 9, 18, 29]          # n_atoms of the B=4 parity batch
FULL_LENGTH_ATOMS = [4, 7, 11]   # molecules of the 1000-step golden trajectory (G7)


def fixture_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name)


def spectra_for(version: str, batch: int, salt: int = 0):
    """Non-negative log10(1+u)-style spectra, names keyed so every case is reproducible."""
    specs = [torch.log10(1.0 + filler.uniform(f"spectra.{n}", (batch, 1, L), salt=salt))
             for n, L in zip(("uv", "ir", "raman"), SPECTRUM_LENGTHS)]
    if version == "allspectra":
        return specs
    return specs[{"uv": 0, "ir": 1, "raman": 2}[version]]


def forward_inputs(version: str, first_step: bool, n_atoms=RAGGED, salt: int = 0):
    """Inputs of one ``model(...)`` call (dmt.py:306-321) for the G4 cases."""
    xh, edge_x, node_mask, edge_mask = filler.synthetic_state(n_atoms, "g4.x", salt=salt)
    B = len(n_atoms)
    noise_level = filler.uniform("g4.noise_level", (B,), -6.0, 6.0, salt=salt)
    if first_step:
        cond_x = cond_edge_x = None
    else:
        cond_x, cond_edge_x, _, _ = filler.synthetic_state(n_atoms, "g4.cond", salt=salt)
        cond_x = cond_x * 0.7
        cond_edge_x = cond_edge_x * 0.5
    return dict(xh=xh, node_mask=node_mask, edge_mask=edge_mask, edge_x=edge_x, noise_level=noise_level,
                cond_x=cond_x, cond_edge_x=cond_edge_x, context=spectra_for(version, B, salt))


def trajectory_inputs(version: str, steps: int, n_atoms=RAGGED, salt: int = 0):
    """Initial state + per-step raw randn draws (order pos, feat, edge — sampling.py:611-612,623-624)."""
    B, N = len(n_atoms), max(n_atoms)
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    raw0 = (filler.normal("g5.init.pos", (B, N, 3), salt), filler.normal("g5.init.feat", (B, N, 6), salt),
            filler.normal("g5.init.edge", (B, 2, N, N), salt))
    raws = [(filler.normal(f"g5.step{i}.pos", (B, N, 3), salt), filler.normal(f"g5.step{i}.feat", (B, N, 6), salt),
             filler.normal(f"g5.step{i}.edge", (B, 2, N, N), salt)) for i in range(steps)]
    return dict(node_mask=node_mask, edge_mask=edge_mask, raw0=raw0, raws=raws,
                context=spectra_for(version, B, salt), n_atoms=list(n_atoms))


def block_inputs(n_atoms=RAGGED, salt: int = 0):
    """Inputs of one EquivariantMixBlock call (dmt.py:122) for the G3 component cases."""
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    B, N, _ = node_mask.shape
    nm = node_mask.reshape(-1, 1)
    pos = (filler.normal("g3.pos", (B * N, 3), salt) * 1.5) * nm
    h = filler.normal("g3.h", (B * N, 256), salt) * nm
    adj = edge_mask.reshape(B, N, N)
    b, i, j = adj.nonzero(as_tuple=True)
    edge_index = torch.stack([b * N + i, b * N + j])
    # symmetric edge features / adjacency heads, as every caller on the path provides
    e_dense = filler.normal("g3.e", (B, N, N, 64), salt)
    e_dense = 0.5 * (e_dense + e_dense.transpose(1, 2))
    a_dense = (filler.uniform("g3.adj", (B, N, N, 2), salt=salt) > 0.5).float()
    a_dense = torch.maximum(a_dense, a_dense.transpose(1, 2))
    temb = filler.normal("g3.temb", (B, 1024), salt)
    return dict(pos=pos, h=h, edge_attr=e_dense[b, i, j], edge_index=edge_index, node_mask=nm,
                extra_heads=a_dense[b, i, j], node_time_emb=temb.repeat_interleave(N, 0), edge_time_emb=temb[b],
                temb=temb, n_atoms=list(n_atoms), N=N, B=B, dense_index=(b, i, j))


READOUT_GAIN_KEYS = ("node_pred_mlp.4.", "edge_type_mlp.4.", "edge_exist_mlp.4.")


def readout_gain(state_dict, gain: float = 8.0):
    """G8 only: multiply the last readout layers by ``gain`` so that predictions leave the self-conditioning clamp
    range (random-init readouts stay inside it and the clamp would never act).  Works with or without ``module.``."""
    out = dict(state_dict)
    for k, v in state_dict.items():
        if any(key in k for key in READOUT_GAIN_KEYS):
            out[k] = v * gain
    return out


def config_for(version: str, steps: int = 1000):
    return qm9s_config(spectra_version=version, steps=steps)


def save_npz(name: str, **arrays):
    np.savez_compressed(fixture_path(name), **{k: np.asarray(v) for k, v in arrays.items()})


def load_npz(name: str):
    with np.load(fixture_path(name)) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}
