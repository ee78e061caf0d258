"""Oracle: SpecFormer conditioning encoder, eval mode (test infrastructure).

Functional restatement over a state dict (keys relative to ``prefix``), following
reference ``models/specformer.py``: patching ``:88-107``, per-spectrum projection +
learned positions + concat ``:167-194``, 3 post-norm encoder layers ``:279-309`` with
residual attention scores ``:401-404,418-424`` and eval-mode BatchNorm1d ``:247,260``,
flatten head ``:467-469`` and output LayerNorm ``:119``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SPECTRUM_LENGTHS = (701, 3501, 3501)
_USED = {"uv": [0], "ir": [1], "raman": [2], "allspectra": [0, 1, 2]}


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _bn_eval(sd, name, x):
    """BatchNorm1d over d_model in eval mode on [B, L, D] (Transpose, BN, Transpose: :247)."""
    return F.batch_norm(x.transpose(1, 2), sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=1e-5).transpose(1, 2)


def specformer_forward(sd, spectra, spectra_version="allspectra", patch_len=(20, 50, 50),
                       stride=(10, 25, 25), prefix="cond_encoder.", n_layers=3, n_heads=16):
    """spectra: list [uv, ir, raman] of [B,1,L] (allspectra) or one [B,1,L] tensor → [B, 256]."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    used = _USED[spectra_version]
    specs = list(spectra) if spectra_version == "allspectra" else [spectra]
    tokens = []
    for slot, (ti, spec) in enumerate(zip(used, specs)):
        spec = spec.reshape(spec.shape[0], -1)                              # :101-104 squeeze to [B, L]
        p = spec.unfold(-1, patch_len[ti], stride[ti])                      # :105 [B, patch_num, patch_len]
        z = _lin(sd, f"backbone.W_P.{slot}", p)                             # :181
        if spectra_version == "allspectra":
            z = z + sd["backbone." + ("W_pos_uv", "W_pos_ir", "W_pos_raman")[slot]]   # :183-188
        else:
            z = z + sd["backbone.W_pos"]                                    # :176
        tokens.append(z)
    z = torch.cat(tokens, dim=1)                                            # :194 [B, L, 128]
    B, L, D = z.shape
    prev = None
    for l in range(n_layers):
        base = f"backbone.encoder.layers.{l}."
        dk = D // n_heads
        q = _lin(sd, base + "self_attn.W_Q", z).view(B, L, n_heads, dk).transpose(1, 2)        # :353
        k = _lin(sd, base + "self_attn.W_K", z).view(B, L, n_heads, dk).permute(0, 2, 3, 1)    # :354
        v = _lin(sd, base + "self_attn.W_V", z).view(B, L, n_heads, dk).transpose(1, 2)        # :355
        scores = torch.matmul(q, k) * sd[base + "self_attn.sdp_attn.scale"]                    # :401
        if prev is not None:
            scores = scores + prev                                                              # :404
        attn = F.softmax(scores, dim=-1)                                                        # :418
        o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, L, n_heads * dk)         # :422,365
        o = _lin(sd, base + "self_attn.to_out.0", o)                                            # :366
        prev = scores
        z = _bn_eval(sd, base + "norm_attn.1", z + o)                                           # :292-294
        f = _lin(sd, base + "ff.3", F.gelu(_lin(sd, base + "ff.0", z)))                         # :300
        z = _bn_eval(sd, base + "norm_ffn.1", z + f)                                            # :302-304
    z = _lin(sd, "head.linear", z.reshape(B, L * D))                                            # :467-468
    return F.layer_norm(z, (z.shape[-1],), sd["out_norm.weight"], sd["out_norm.bias"], 1e-5)   # :119
