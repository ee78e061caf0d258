"""Oracle: DMT denoiser forward, eval mode (test infrastructure).

A functional restatement over a state dict (reference parameter names, no ``module.``
prefix) of reference ``models/dmt.py:306-412`` with ``EquivariantMixBlock.forward``
``:122-174``, ``MultiCondEquiUpdate.forward`` ``:37-60``, ``TransMixLayer`` ``models/layers.py:131-186``,
``CondGaussianLayer``/``gaussian`` ``layers.py:291-295,328-334``, ``CoorsNorm`` ``layers.py:344-347``,
``LearnedSinusodialposEmb`` ``layers.py:283-288`` and the helpers of ``models/utils.py:38-45,118-144``.

The three PyG / torch_scatter primitives are restated with plain index ops
(semantics of the pinned versions): ``dense_to_sparse`` = row-major nonzero of
the [B,N,N] mask; ``propagate`` gathers ``*_i`` by ``edge_index[1]`` (target) and ``*_j`` by
``edge_index[0]`` (source) and scatter-sums messages onto the target; ``softmax`` is the
max-shifted segment softmax with ``+1e-16`` in the denominator; ``scatter(reduce='add')`` is
``index_add_``.  Like the reference, the per-molecule time/adaLN MLPs are evaluated per edge
and per node (``dmt.py:355-357``) — this is the reference's CPU cost, kept for the baseline.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .specformer import specformer_forward

PI_TRUNC = 3.14159  # layers.py:293 — truncated pi is part of the reference's arithmetic


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)  # elementwise_affine=False, eps=1e-6 (dmt.py:86)


def _modulate(x, shift, scale):
    return x * (1 + scale) + shift  # dmt.py:13-14


def _gaussian(x, mean, std):
    a = (2 * PI_TRUNC) ** 0.5
    return torch.exp(-0.5 * (((x - mean) / std) ** 2)) / (a * std)  # layers.py:291-295


def _cond_gaussian(sd, name, x, time_emb):
    """layers.py:328-334: x[E,1], time_emb[E,1024] → [E, 64]."""
    scale, shift = _lin(sd, name + ".time_mlp.1", F.silu(time_emb)).chunk(2, dim=1)
    x = x * (scale + 1) + shift
    mean = sd[name + ".means.weight"].float().view(-1)
    std = sd[name + ".stds.weight"].float().view(-1).abs() + 1e-5
    return torch.cat([x, _gaussian(x, mean, std)], dim=-1)


def _segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax (2.4.0): shift by segment max, exp, / (segment sum + 1e-16)."""
    idx = index.view(-1, 1).expand_as(src)
    mx = torch.full((num_nodes, src.shape[1]), float("-inf")).scatter_reduce(0, idx, src, "amax", include_self=True)
    out = (src - mx.index_select(0, index)).exp()
    den = torch.zeros(num_nodes, src.shape[1]).index_add_(0, index, out) + 1e-16
    return out / den.index_select(0, index)


def _trans_mix(sd, name, x, edge_index, edge_attr, extra_heads, heads=16, n_extra=2):
    """layers.py:131-186 with propagate(flow=source_to_target, aggr='add')."""
    C = x.shape[1] // heads
    sub_heads = heads - n_extra
    sub_ch = (heads * C) // sub_heads
    src, tgt = edge_index[0], edge_index[1]
    query = _lin(sd, name + ".lin_query", x).reshape(-1, sub_heads, sub_ch)
    key = _lin(sd, name + ".lin_key", x).reshape(-1, sub_heads, sub_ch)
    value = _lin(sd, name + ".lin_value", x).reshape(-1, heads, C)
    q_i, k_j, v_j = query.index_select(0, tgt), key.index_select(0, src), value.index_select(0, src)
    edge_attn = torch.tanh(_lin(sd, name + ".lin_edge0", edge_attr).view(-1, sub_heads, sub_ch))
    alpha = (q_i * k_j * edge_attn).sum(dim=-1) / math.sqrt(C)
    extra = extra_heads.clone()
    extra[extra == 0.0] = -1e10                                            # layers.py:171-174
    alpha = torch.cat([extra, alpha], dim=-1)
    alpha = _segment_softmax(alpha, tgt, x.shape[0])
    msg = v_j * torch.tanh(_lin(sd, name + ".lin_edge1", edge_attr).view(-1, heads, C))
    msg = msg * alpha.view(-1, heads, 1)
    out = torch.zeros(x.shape[0], heads, C).index_add_(0, tgt, msg)
    return out.view(-1, heads * C)


def _equi_update(sd, name, h, pos, edge_index, edge_attr, dist, time_emb, adj_extra):
    """dmt.py:37-60."""
    row, col = edge_index
    h_input = torch.cat([h[row], h[col], edge_attr, dist], dim=1)
    coord_diff = pos[row] - pos[col]
    norm = coord_diff.norm(dim=-1, keepdim=True)
    coord_diff = coord_diff / norm.clamp(min=1e-8) * sd[name + ".coord_norm.scale"]    # layers.py:344-347
    shift, scale = _lin(sd, name + ".time_mlp.1", F.silu(time_emb)).chunk(2, dim=1)
    inv = _modulate(_ln(_lin(sd, name + ".input_lin", h_input)), shift, scale)
    inv = _lin(sd, name + ".coord_mlp.2", F.silu(_lin(sd, name + ".coord_mlp.0", inv)))
    inv = torch.tanh(inv)
    adjs = torch.cat([torch.ones(adj_extra.size(0), 1), adj_extra], dim=-1)
    inv = (inv * adjs).mean(-1, keepdim=True)
    agg = torch.zeros_like(pos).index_add_(0, row, coord_diff * inv)
    return pos + agg


def _mix_block(sd, name, pos, h, edge_attr, edge_index, node_mask, extra_heads, node_t, edge_t, drop=None):
    """dmt.py:122-174 (cond_time=True, dist_gbf=True).  ``drop(site, x)`` applies the FF dropout of dmt.py:114-120 (site 0 / 1 =
    node hidden / output, 2 / 3 = edge hidden / output) with an injected mask; None = eval mode (identity)."""
    if drop is None:
        drop = lambda site, x: x
    h_in_node, h_in_edge = h, edge_attr
    row, col = edge_index
    d = pos[row] - pos[col]
    distance = torch.sum(d ** 2, 1).unsqueeze(1)                           # models/utils.py:129-133
    distance = _cond_gaussian(sd, name + ".dist_layer", distance, edge_t)
    edge_attr = _lin(sd, name + ".edge_emb", torch.cat([distance, edge_attr], dim=-1))
    n_sh_a, n_sc_a, n_g_a, n_sh_m, n_sc_m, n_g_m = _lin(sd, name + ".node_time_mlp.1", F.silu(node_t)).chunk(6, dim=1)
    e_sh_a, e_sc_a, e_g_a, e_sh_m, e_sc_m, e_g_m = _lin(sd, name + ".edge_time_mlp.1", F.silu(edge_t)).chunk(6, dim=1)
    h = _modulate(_ln(h), n_sh_a, n_sc_a)
    edge_attr = _modulate(_ln(edge_attr), e_sh_a, e_sc_a)
    h_node = _trans_mix(sd, name + ".attn_mpnn", h, edge_index, edge_attr, extra_heads)
    h_edge = _lin(sd, name + ".node2edge_lin", h_node[row] + h_node[col])
    h_node = h_in_node + n_g_a * h_node
    h_node = _modulate(_ln(h_node), n_sh_m, n_sc_m) * node_mask
    ff = drop(1, _lin(sd, name + ".ff_linear2", drop(0, F.silu(_lin(sd, name + ".ff_linear1", h_node)))))
    h_out = (h_node + n_g_m * ff) * node_mask
    h_edge = h_in_edge + e_g_a * h_edge
    h_edge = _modulate(_ln(h_edge), e_sh_m, e_sc_m)
    ffe = drop(3, _lin(sd, name + ".ff_linear4", drop(2, F.silu(_lin(sd, name + ".ff_linear3", h_edge)))))
    h_edge_out = h_edge + e_g_m * ffe
    pos = _equi_update(sd, name + ".equi_update", h_out, pos, edge_index, h_edge_out, distance, edge_t, extra_heads)
    return h_out, h_edge_out, pos


def _remove_mean_with_mask(x, node_mask):
    N = node_mask.sum(1, keepdims=True)
    mean = torch.sum(x, dim=1, keepdim=True) / N
    return x - mean * node_mask                                            # models/utils.py:38-45


def _mlp3(sd, name, x):
    x = F.silu(_lin(sd, name + ".0", x))
    x = F.silu(_lin(sd, name + ".2", x))
    return _lin(sd, name + ".4", x)


def time_embedding(sd, noise_level):
    """dmt.py:249-257 + layers.py:283-288: [B] → [B,1024]."""
    x = noise_level.unsqueeze(-1)
    freqs = x * sd["time_mlp.0.weights"].unsqueeze(0) * 2 * math.pi
    f = torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)
    return _lin(sd, "time_mlp.3", F.gelu(_lin(sd, "time_mlp.1", f)))


def context_embedding(sd, context, cfg):
    """dmt.py:348-350: cond_lin(SpecFormer(context)) → [B,1024]."""
    z = specformer_forward(sd, context, cfg.data.spectra_version, cfg.model.patch_len, cfg.model.stride)
    return _lin(sd, "cond_lin", z)


@torch.no_grad()
def dmt_forward(sd, cfg, xh, node_mask, edge_mask, edge_x, noise_level, cond_x=None, cond_edge_x=None,
                context=None, context_emb=None, return_debug=False, dropout_masks=None):
    """models/dmt.py:306-412.  ``context_emb`` (a precomputed ``context_embedding``) may replace ``context``.
    ``dropout_masks(block, site, x) -> x * mask`` = training-mode FF dropout with injected masks (golden G17); None = eval."""
    n_layers = cfg.model.n_layers
    bs, n_nodes, _ = xh.shape
    pos = xh[:, :, 0:3].clone().reshape(bs * n_nodes, -1)
    h = xh[:, :, 3:].clone().reshape(bs * n_nodes, -1)
    adj_mask = edge_mask.reshape(bs, n_nodes, n_nodes)
    dense_index = adj_mask.nonzero(as_tuple=True)
    b_i, r_i, c_i = dense_index
    edge_index = torch.stack([b_i * n_nodes + r_i, b_i * n_nodes + c_i])  # dense_to_sparse (PyG 2.4.0)
    if cond_x is None:
        cond_x = torch.zeros_like(xh)
        cond_edge_x = torch.zeros_like(edge_x)
        cond_adj_2d = torch.ones((edge_index.size(1), 1))
    else:
        cond_adj_2d = cond_edge_x[dense_index][:, 0:1].clone()
        ge = cond_adj_2d >= cfg.model.edge_quan_th
        cond_adj_2d = ge.to(cond_adj_2d.dtype)                             # dmt.py:338-340
    cond_pos = cond_x[:, :, 0:3].clone().reshape(bs * n_nodes, -1)
    cond_h = cond_x[:, :, 3:].clone().reshape(bs * n_nodes, -1)
    h = torch.cat([h, cond_h], dim=-1)
    if context_emb is None:
        context_emb = context_embedding(sd, context, cfg)
    time_emb = time_embedding(sd, noise_level) + context_emb               # dmt.py:354
    node_t = time_emb.unsqueeze(1).expand(-1, n_nodes, -1).reshape(bs * n_nodes, -1)
    edge_t = time_emb[torch.div(edge_index[0], n_nodes, rounding_mode="floor")]
    row, col = edge_index
    cd = cond_pos[row] - cond_pos[col]
    distances = torch.sum(cd ** 2, 1).unsqueeze(1)                         # models/utils.py:118-126
    cond_adj_spatial = (distances <= cfg.model.spatial_cut_off).to(distances.dtype)
    if distances.sum() == 0:
        distances = distances.repeat(1, cfg.model.nf // 4)                 # dmt.py:364-365
    else:
        distances = _cond_gaussian(sd, "dist_layer", distances, edge_t)
    extra_adj = torch.cat([cond_adj_2d, cond_adj_spatial], dim=-1)
    edge_attr = torch.cat([edge_x[dense_index], cond_edge_x[dense_index], distances], dim=-1)
    h = _lin(sd, "node_emb", h)
    edge_attr = _lin(sd, "edge_emb", edge_attr)
    atom_hids, edge_hids = [h], [edge_attr]
    nm_flat = node_mask.reshape(-1, 1)
    dbg = {}
    for i in range(n_layers):
        drop = None if dropout_masks is None else (lambda site, x, _i=i: dropout_masks(_i, site, x))
        h, edge_attr, pos = _mix_block(sd, "e_block_%d" % i, pos, h, edge_attr, edge_index, nm_flat,
                                       extra_adj, node_t, edge_t, drop)
        pos = _remove_mean_with_mask(pos.reshape(bs, n_nodes, -1), node_mask).reshape(bs * n_nodes, -1)
        atom_hids.append(_lin(sd, "node_%d" % i, h))
        edge_hids.append(_lin(sd, "edge_%d" % i, edge_attr))
        if return_debug:
            dbg["h_%d" % i], dbg["e_%d" % i], dbg["pos_%d" % i] = h.clone(), edge_attr.clone(), pos.clone()
    atom_hids = torch.cat(atom_hids, dim=-1)
    edge_hids = torch.cat(edge_hids, dim=-1)
    atom_pred = _mlp3(sd, "node_pred_mlp", atom_hids).reshape(bs, n_nodes, -1) * node_mask
    edge_pred = torch.cat([_mlp3(sd, "edge_exist_mlp", edge_hids), _mlp3(sd, "edge_type_mlp", edge_hids)], dim=-1)
    edge_final = torch.zeros(bs, n_nodes, n_nodes, edge_pred.shape[-1])
    edge_final[dense_index] = edge_pred                                     # to_dense_edge_attr (unique indices)
    edge_final = 0.5 * (edge_final + edge_final.permute(0, 2, 1, 3))
    pos = pos * nm_flat
    if torch.any(torch.isnan(pos)):
        pos = torch.zeros_like(pos)                                         # dmt.py:407-409
    pos = _remove_mean_with_mask(pos.reshape(bs, n_nodes, -1), node_mask)
    out = torch.cat([pos, atom_pred], dim=2), edge_final
    if return_debug:
        dbg["edge_index"] = edge_index
        return out + (dbg,)
    return out
