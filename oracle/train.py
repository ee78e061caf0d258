"""Oracle: the training loss of the DMT (TEST INFRASTRUCTURE ONLY) - a torch-autograd restatement of reference
``losses.get_sde_graph_loss_fn`` (``losses.py:286-396``, the pred_data / self_cond / noise_align branch every shipped config
takes), ``process_edge_batch`` (``:498-529``), ``get_data_scaler`` (``utils.py:33-68``), ``get_align_position`` /
``kabsch_batch`` (``losses.py:414-452``) and the noise samplers (``models/utils.py:67-106``), over the functional forward of
``oracle/dmt.py`` with the SpecFormer encoder in TRAINING mode (BatchNorm batch statistics, ``specformer.py:247,260``).
FF dropout (dmt.py:114-120) is the identity by default (golden G13 pins the p = 0 arithmetic) or, with ``dropout=(p, seeds)``, a
multiplication by the masks of ``dropout_masks`` - the Philox masks of the HIP kernels laid out on the reference's tensors, which is
what golden G17 injects into the reference's own ``nn.Dropout`` (the reference's torch-RNG masks cannot be reproduced on a GPU).
Gradients come from ``torch.autograd`` on this restatement; it is pinned by goldens G13 / G17 (the reference's own loss and gradients).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dmt as odmt
from .schedule import marginal_prob
from .sampler import combined_noise, symmetric_edge_noise
from .specformer import _USED, _lin


def _bn_train(sd, name, x, stats):
    """BatchNorm1d over d_model in TRAINING mode on [B, L, D]: batch statistics over (B, L); records the running-stat update
    (momentum 0.1, unbiased variance) in ``stats`` instead of mutating ``sd``."""
    xt = x.transpose(1, 2)
    mean = xt.mean(dim=(0, 2))
    var_b = xt.var(dim=(0, 2), unbiased=False)
    n = xt.shape[0] * xt.shape[2]
    stats[name + ".running_mean"] = (0.9 * sd[name + ".running_mean"] + 0.1 * mean).detach()
    stats[name + ".running_var"] = (0.9 * sd[name + ".running_var"] + 0.1 * var_b * n / (n - 1)).detach()
    y = (xt - mean[None, :, None]) / torch.sqrt(var_b[None, :, None] + 1e-5)
    return (y * sd[name + ".weight"][None, :, None] + sd[name + ".bias"][None, :, None]).transpose(1, 2)


def specformer_forward_train(sd, spectra, spectra_version="allspectra", patch_len=(20, 50, 50), stride=(10, 25, 25),
                             prefix="cond_encoder.", n_layers=3, n_heads=16):
    """``oracle.specformer.specformer_forward`` with training-mode BatchNorm; returns (z [B,256], running-stat updates)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    used = _USED[spectra_version]
    specs = list(spectra) if spectra_version == "allspectra" else [spectra]
    tokens, stats = [], {}
    for slot, (ti, spec) in enumerate(zip(used, specs)):
        p = spec.reshape(spec.shape[0], -1).unfold(-1, patch_len[ti], stride[ti])
        z = _lin(sd, f"backbone.W_P.{slot}", p)
        z = z + (sd["backbone." + ("W_pos_uv", "W_pos_ir", "W_pos_raman")[slot]] if spectra_version == "allspectra" else sd["backbone.W_pos"])
        tokens.append(z)
    z = torch.cat(tokens, dim=1)
    B, L, D = z.shape
    prev = None
    for l in range(n_layers):
        base = f"backbone.encoder.layers.{l}."
        dk = D // n_heads
        q = _lin(sd, base + "self_attn.W_Q", z).view(B, L, n_heads, dk).transpose(1, 2)
        k = _lin(sd, base + "self_attn.W_K", z).view(B, L, n_heads, dk).permute(0, 2, 3, 1)
        v = _lin(sd, base + "self_attn.W_V", z).view(B, L, n_heads, dk).transpose(1, 2)
        scores = torch.matmul(q, k) * sd[base + "self_attn.sdp_attn.scale"]
        if prev is not None:
            scores = scores + prev
        o = torch.matmul(F.softmax(scores, dim=-1), v).transpose(1, 2).contiguous().view(B, L, n_heads * dk)
        o = _lin(sd, base + "self_attn.to_out.0", o)
        prev = scores
        z = _bn_train(sd, base + "norm_attn.1", z + o, stats)
        f = _lin(sd, base + "ff.3", F.gelu(_lin(sd, base + "ff.0", z)))
        z = _bn_train(sd, base + "norm_ffn.1", z + f, stats)
    z = _lin(sd, "head.linear", z.reshape(B, L * D))
    z = F.layer_norm(z, (z.shape[-1],), sd["out_norm.weight"], sd["out_norm.bias"], 1e-5)
    return z, {prefix + k: v for k, v in stats.items()}


def scale_batch(batch, normalize_factors=(1.0, 4.0, 4.0, 1.0)):
    """process_edge_batch (losses.py:498-529, model 'DMT') + get_data_scaler (utils.py:33-68, centered): xh [B,N,9],
    edge_x [B,N,N,2], node_mask [B,N,1], edge_mask."""
    node_mask = batch["atom_mask"].unsqueeze(2)
    edge_mask = batch["edge_mask"]
    pos = odmt._remove_mean_with_mask(batch["positions"], node_mask)
    pn, an, fn, en = normalize_factors
    atom = (batch["atom_one_hot"] * 2.0 - 1.0) / an * node_mask
    fc = batch["formal_charges"] / fn * node_mask
    pos = pos / pn * node_mask
    B, N = node_mask.shape[:2]
    edge = (batch["edge_one_hot"] * 2.0 - 1.0) / en * edge_mask.reshape(B, N, N, 1)
    return torch.cat([pos, atom, fc], dim=2), edge, node_mask, edge_mask


@torch.no_grad()
def kabsch_batch(coords_pred, coords_tar):
    """losses.py:441-452."""
    A = torch.einsum("...ki, ...kj -> ...ij", coords_pred, coords_tar)
    U, S, Vt = torch.linalg.svd(A)
    diag = torch.ones((A.size(0), 3))
    diag[:, -1] = torch.sign(torch.det(A))
    return torch.einsum("...ij, ...jk, ...kl -> ...il", U, torch.diag_embed(diag), Vt)


@torch.no_grad()
def align_position(z_t, xh):
    """losses.py:414-422."""
    rot = kabsch_batch(z_t[:, :, :3], xh[:, :, :3])
    return torch.einsum("...ki, ...ji -> ...jk", rot, xh[:, :, :3])


def loss_from_predictions(pred, edge_pred, xh, edge_x, align_pos, alpha_t, sigma_t, loss_weights=(1.0, 0.25, 0.1)):
    """losses.py:359-394 (pred_data, reduce_mean False): the three MSE terms, weighted, times sqrt(alpha_t / sigma_t), batch mean."""
    B = xh.shape[0]
    l_pos = torch.square(pred[:, :, :3] - align_pos).mean(-1).sum(-1)
    l_type = torch.square(pred[:, :, 3:] - xh[:, :, 3:]).mean(-1).sum(-1)
    l_edge = torch.square(edge_x - edge_pred).mean(-1).reshape(B, -1).sum(-1)
    losses = loss_weights[0] * l_pos + loss_weights[1] * l_type + loss_weights[2] * l_edge
    return (torch.sqrt(alpha_t / sigma_t) * losses).mean()


def forward_with_context(sd, cfg, z_t, node_mask, edge_mask, edge_z_t, noise_level, ctx_emb, cond_x=None, cond_edge_x=None):
    """The DMT forward with gradients enabled and the conditioning embedding supplied (used to check the DMT-side backward alone)."""
    return odmt.dmt_forward.__wrapped__(sd, cfg, z_t, node_mask, edge_mask, edge_z_t, noise_level, cond_x, cond_edge_x, context_emb=ctx_emb)


def dropout_masks(n_atoms, n_max, p, seed):
    """FF-dropout masks of ONE DMT evaluation for the reference's tensor shapes: ``fn(block, site, x) -> x * mask``.

    The HIP training path draws the mask of element c of packed node row r (valid atoms, molecule-major) / packed pair row q
    (unordered pairs a < b, molecule-major, a-major) from Philox stream ``4 * block + site`` at flat index ``r * C + c`` /
    ``q * C + c`` (``philox.dropout_keep``).  The reference holds node tensors densely, [B * N, C] with padded atoms, and edge tensors
    per DIRECTED edge in ``dense_to_sparse`` order; its masks here are those values gathered: a padded atom's row is kept (it is
    multiplied by node_mask anyway), and both directions of a pair get the pair's mask - the edge features stay pair-symmetric,
    which the reference's own independent per-edge masks would not (the documented deviation)."""
    import numpy as np
    from . import philox
    n_atoms = [int(n) for n in n_atoms]
    B, N = len(n_atoms), int(n_max)
    node_row = np.full(B * N, -1, dtype=np.int64)
    edge_row = []
    r = q = 0
    for b, n in enumerate(n_atoms):
        node_row[b * N:b * N + n] = np.arange(r, r + n)
        r += n
        tri = np.zeros((n, n), dtype=np.int64)
        iu = np.triu_indices(n, 1)
        tri[iu] = np.arange(q, q + len(iu[0]))
        tri = tri + tri.T
        q += len(iu[0])
        ii, jj = np.nonzero(~np.eye(n, dtype=bool))                 # row-major (i, j), i != j: dense_to_sparse order
        edge_row.append(tri[ii, jj])
    edge_row = np.concatenate(edge_row) if edge_row else np.zeros(0, dtype=np.int64)
    Nn, Pp = r, q
    scale = float(philox.dropout_scale(p))
    cache = {}

    def fn(block, site, x):
        C = x.shape[1]
        key = (block, site)
        if key not in cache:
            rows = Nn if site < 2 else Pp
            keep = philox.dropout_keep(seed, 4 * block + site, rows * C, p).reshape(rows, C)
            if site < 2:
                dense = np.ones((B * N, C), dtype=bool)
                dense[node_row >= 0] = keep[node_row[node_row >= 0]]
            else:
                dense = keep[edge_row]
            cache[key] = torch.from_numpy(dense.astype(np.float32) * np.float32(scale))
        m = cache[key]
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m

    return fn


def training_loss(sd, cfg, batch, t_raw, randn, self_cond_coin: bool, loss_weights=(1.0, 0.25, 0.1), dropout=None):
    """One ``loss_fn(model, batch)`` call (losses.py:301-394) with its random draws passed in: ``t_raw`` = the ``torch.rand(B)``
    draw, ``randn`` = the three noise draws (pos [B,N,3], feat [B,N,6], edge [B,2,N,N]), ``self_cond_coin`` = ``random() < 0.5``.
    ``sd`` tensors that require grad receive gradients from ``loss.backward()``.  ``dropout = (p, (seed_selfcond, seed_main))``
    turns the FF dropout on with the injected masks of ``dropout_masks`` (golden G17).  Returns (loss, dict of intermediates)."""
    xh, edge_x, node_mask, edge_mask = scale_batch(batch)
    B = xh.shape[0]
    t = t_raw * (1.0 - 1e-5) + 1e-5
    alpha_t, sigma_t = marginal_prob(t)
    noise = combined_noise(randn[0], randn[1], node_mask)
    edge_noise = symmetric_edge_noise(randn[2], edge_mask)
    ex = lambda v, like: v.reshape(-1, *([1] * (like.dim() - 1)))
    z_t = ex(alpha_t, xh) * xh + ex(sigma_t, xh) * noise
    edge_z_t = ex(alpha_t, edge_x) * edge_x + ex(sigma_t, edge_x) * edge_noise
    align_pos = align_position(z_t, xh)
    noise_level = torch.log(alpha_t ** 2 / sigma_t ** 2)
    fwd = odmt.dmt_forward.__wrapped__                                       # the same forward, gradients enabled

    n_at = node_mask.reshape(B, -1).sum(1).long().tolist()

    def model(sd, cond_x, cond_edge_x, seed_index):
        z, stats = specformer_forward_train(sd, batch["context"], cfg.data.spectra_version, cfg.model.patch_len, cfg.model.stride)
        ctx = odmt._lin(sd, "cond_lin", z)
        masks = None if dropout is None else dropout_masks(n_at, node_mask.shape[1], dropout[0], dropout[1][seed_index])
        return fwd(sd, cfg, z_t, node_mask, edge_mask, edge_z_t, noise_level, cond_x, cond_edge_x, context_emb=ctx, dropout_masks=masks), stats

    cond_x = cond_edge_x = None
    info = {}
    if self_cond_coin:
        with torch.no_grad():
            (cond_x, cond_edge_x), stats0 = model(sd, None, None, 0)  # also a training-mode forward: BatchNorm stats move twice
            sd = dict(sd)
            sd.update(stats0)
    (pred, edge_pred), stats = model(sd, cond_x, cond_edge_x, 1)
    loss = loss_from_predictions(pred, edge_pred, xh, edge_x, align_pos, alpha_t, sigma_t, loss_weights)
    info.update(xh=xh, edge_x=edge_x, z_t=z_t, edge_z_t=edge_z_t, alpha_t=alpha_t, sigma_t=sigma_t, noise_level=noise_level,
                align_pos=align_pos, pred=pred, edge_pred=edge_pred, cond_x=cond_x, cond_edge_x=cond_edge_x, bn=stats,
                node_mask=node_mask, edge_mask=edge_mask)
    return loss, info
