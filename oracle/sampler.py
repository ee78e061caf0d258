"""Oracle: ancestral reverse-diffusion loop, noise samplers and post-processing (test infrastructure).

Follows reference ``sampling.py:565-631`` (``AncestralSampler.sampling``, data-prediction +
self-conditioning branch, the only one the shipped configs take), ``models/utils.py:67-106``
(noise samplers), ``sampling.py:53-97`` (``post_process``, ``compress_edge=True`` branch),
``sampling.py:12-32`` (``mol_process``) and ``utils.py:71-105`` (inverse scaler, centered=True,
normalize_factors '1, 4, 4, 1').
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .schedule import ancestral_coefficients


def _remove_mean_with_mask(x, node_mask):
    N = node_mask.sum(1, keepdims=True)
    return x - (torch.sum(x, dim=1, keepdim=True) / N) * node_mask


def combined_noise(raw_pos, raw_feat, node_mask):
    """models/utils.py:67-97: mask, CoM-project positions, concat."""
    z_x = _remove_mean_with_mask(raw_pos * node_mask, node_mask)
    return torch.cat([z_x, raw_feat * node_mask], dim=2)


def symmetric_edge_noise(raw, edge_mask):
    """models/utils.py:100-106: raw [B, ch, N, N] → tril(-1) + transpose → [B,N,N,ch] * edge_mask."""
    B, _, N, _ = raw.shape
    z = torch.tril(raw, -1)
    z = z + z.transpose(-1, -2)
    return z.permute(0, 2, 3, 1) * edge_mask.reshape(B, N, N, 1)


def self_cond_ori(pred, edge_pred):
    """utils.py:135-136."""
    return pred, edge_pred


def self_cond_clamp(pred, edge_pred, atom_types=5, norms=(1, 4, 4, 1), fc_scale=(-1.0, 1.0), centered=True):
    """utils.py:113-148: clamp the predicted type and charge channels IN PLACE (so the clamped prediction also enters
    the posterior mean that follows, sampling.py:604-606) and return a clamped copy of the edge prediction."""
    lo, hi = (-1.0, 1.0) if centered else (0.0, 1.0)
    pred[:, :, 3:3 + atom_types] = pred[:, :, 3:3 + atom_types].clamp(lo / norms[1], hi / norms[1])
    pred[:, :, -1:] = pred[:, :, -1:].clamp(fc_scale[0] / norms[2], fc_scale[1] / norms[2])
    return pred, edge_pred.clamp(lo / norms[3], hi / norms[3])


@torch.no_grad()
def ancestral_sampling(model_fn, z_T, node_mask, edge_mask, edge_z_T, steps, noise_fn, temperature=1.0,
                       eps=1e-3, cond_process_fn=self_cond_ori):
    """sampling.py:565-631.  ``model_fn(x, edge_x, noise_level[B], cond_x, cond_edge_x) -> (pred, edge_pred)``;
    ``noise_fn(i) -> (raw_pos[B,N,3], raw_feat[B,N,6], raw_edge[B,2,N,N])`` replays the three randn draws of
    step ``i`` in the reference's order (``:611-612,623-624``)."""
    co = ancestral_coefficients(steps, eps)
    x, edge_x = z_T, edge_z_T
    bs = z_T.shape[0]
    cond_x = cond_edge_x = None
    x_mean = edge_x_mean = None
    for i in range(steps):
        noise_level = torch.ones(bs) * co["noise_level"][i]
        pred, edge_pred = model_fn(x, edge_x, noise_level, cond_x, cond_edge_x)
        cond_x, cond_edge_x = cond_process_fn(pred, edge_pred)             # sampling.py:590
        raw_pos, raw_feat, raw_edge = noise_fn(i)
        x_mean = co["c_x"][i] * x + co["c_pred"][i] * pred
        x = x_mean + co["sigma"][i] * combined_noise(raw_pos, raw_feat, node_mask) * temperature
        edge_x_mean = co["c_x"][i] * edge_x + co["c_pred"][i] * edge_pred
        edge_x = edge_x_mean + co["sigma"][i] * symmetric_edge_noise(raw_edge, edge_mask) * temperature
    return x_mean, edge_x_mean


def inverse_scale(pos, atom_type, fc_charge, node_mask, edge_type, edge_mask, norms=(1, 4, 4, 1)):
    """utils.py:88-103 with centered=True."""
    pos_norm, atom_norm, fc_norm, edge_norm = norms
    pos = pos * pos_norm * node_mask
    atom_type = atom_type * atom_norm
    fc_charge = fc_charge * fc_norm * node_mask
    atom_type = (atom_type + 1.0) / 2.0 * node_mask
    edge_type = edge_type * edge_norm
    edge_type = (edge_type + 1.0) / 2.0
    B, N = node_mask.shape[0], node_mask.shape[1]
    edge_type = edge_type * edge_mask.reshape(B, N, N, 1)
    return pos, atom_type, fc_charge, edge_type


def post_process(xh, node_mask, edge_x, edge_mask, atom_types=5):
    """sampling.py:53-97 (include_charge=True, compress_edge=True, 2 edge channels)."""
    pos, h_cat, h_int = xh[:, :, :3], xh[:, :, 3:-1], xh[:, :, -1:]
    pos, h_cat, h_int, h_edge = inverse_scale(pos, h_cat, h_int, node_mask, edge_x, edge_mask)
    one_hot = F.one_hot(torch.argmax(h_cat, dim=2), atom_types) * node_mask
    fc = torch.round(h_int).long() * node_mask
    edge_exist = (h_edge[:, :, :, 0] >= 0.5).to(h_edge.dtype)
    t = h_edge[:, :, :, 1] * 3.0
    edge_type = torch.zeros_like(t)
    edge_type[t >= 0.5] = 1.0
    edge_type[t >= 1.5] = 2.0
    edge_type[t >= 2.5] = 3.0
    return pos, one_hot, fc, edge_exist * edge_type


def mol_process(one_hot, x, formal_charges, n_nodes, edge_types):
    """sampling.py:12-32 → list of (pos[n,3], atom_type[n] i64, edge_type[n,n] f32, fc[n] i64)."""
    out = []
    for i in range(one_hot.shape[0]):
        n = int(n_nodes[i])
        out.append((x[i][:n].cpu(), one_hot[i].argmax(1)[:n].cpu(), edge_types[i][:n, :n].cpu(),
                    formal_charges[i][:n, 0].long().cpu()))
    return out
