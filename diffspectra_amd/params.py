"""Parameter trees of DMT and SpecFormer — names, shapes and registration order only.

The drop-in contract for weights is that ``state_dict()`` has the
reference's 435 entries (names / shapes / order) so ``load_state_dict(strict=True)``
of a reference checkpoint and ``ExponentialMovingAverage.copy_to(model.parameters())``
both work.  The modules here therefore carry the same attribute names as reference
``models/dmt.py:182-262``, ``models/layers.py:98-120,277-281,316-326,338-342`` and
``models/specformer.py:53-67,139-156,210-262,326-335,444-446``; they hold *no*
arithmetic — the forward pass lives in the HIP library (``csrc/``) and is driven by
``diffspectra_amd.dmt.DMT.forward``.
"""
from __future__ import annotations

import torch
from torch import nn

from .config import SPECTRUM_LENGTHS, used_spectra


class Holder(nn.Module):
    """Pure container: parameters/buffers/children, never called."""

    def forward(self, *a, **k):  # pragma: no cover - structural module
        raise RuntimeError("structural parameter holder; compute happens in the HIP engine")


class Marker(Holder):
    """Parameter-free placeholder that keeps ``nn.Sequential`` indices aligned with the reference."""


def _seq(*mods) -> nn.Sequential:
    return nn.Sequential(*mods)


class SinusoidWeights(Holder):
    """``time_mlp.0`` of the reference: one ``weights[half_dim]`` parameter (layers.py:277-281)."""

    def __init__(self, dim: int):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))


class CondGaussianParams(Holder):
    """means/stds tables [1, K-1] + ``time_mlp.1`` Linear(time_dim, 2) (layers.py:316-326)."""

    def __init__(self, K: int, time_dim: int):
        super().__init__()
        self.K = K - 1
        self.means = nn.Embedding(1, self.K)
        self.stds = nn.Embedding(1, self.K)
        self.time_mlp = _seq(Marker(), nn.Linear(time_dim, 2))
        nn.init.uniform_(self.means.weight, 0, 3)
        nn.init.uniform_(self.stds.weight, 0, 3)


class CoorsNormParams(Holder):
    def __init__(self, scale_init: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(1).fill_(scale_init))


class TransMixParams(Holder):
    """q/k: 14 heads x 18 channels; v / edge gate: 16 x 16 (layers.py:111-120)."""

    def __init__(self, x_channels: int, out_channels: int, extra_heads: int, heads: int, edge_dim: int):
        super().__init__()
        self.heads, self.extra_heads, self.out_channels = heads, extra_heads, out_channels
        self.sub_heads = heads - extra_heads
        self.sub_channels = (heads * out_channels) // self.sub_heads
        qk = self.sub_heads * self.sub_channels
        self.lin_key = nn.Linear(x_channels, qk)
        self.lin_query = nn.Linear(x_channels, qk)
        self.lin_value = nn.Linear(x_channels, heads * out_channels)
        self.lin_edge0 = nn.Linear(edge_dim, qk, bias=False)
        self.lin_edge1 = nn.Linear(edge_dim, heads * out_channels, bias=False)


class EquiUpdateParams(Holder):
    def __init__(self, hidden_dim: int, edge_dim: int, dist_dim: int, time_dim: int, extra_heads: int):
        super().__init__()
        self.coord_norm = CoorsNormParams(scale_init=1e-2)
        self.time_mlp = _seq(Marker(), nn.Linear(time_dim, hidden_dim * 2))
        self.input_lin = nn.Linear(hidden_dim * 2 + edge_dim + dist_dim, hidden_dim)
        self.ln = Marker()
        self.coord_mlp = _seq(nn.Linear(hidden_dim, hidden_dim), Marker(),
                              nn.Linear(hidden_dim, 1 + extra_heads, bias=False))


class MixBlockParams(Holder):
    """One equivariant block; child order as registered in dmt.py:66-112."""

    def __init__(self, node_dim: int, edge_dim: int, time_dim: int, extra_heads: int, heads: int,
                 mlp_ratio: int):
        super().__init__()
        dist_dim = edge_dim
        self.edge_emb = nn.Linear(edge_dim + dist_dim, edge_dim)
        self.node2edge_lin = nn.Linear(node_dim, edge_dim)
        self.attn_mpnn = TransMixParams(node_dim, node_dim // heads, extra_heads, heads, edge_dim)
        self.ff_linear1 = nn.Linear(node_dim, node_dim * mlp_ratio)
        self.ff_linear2 = nn.Linear(node_dim * mlp_ratio, node_dim)
        self.ff_linear3 = nn.Linear(edge_dim, edge_dim * mlp_ratio)
        self.ff_linear4 = nn.Linear(edge_dim * mlp_ratio, edge_dim)
        self.equi_update = EquiUpdateParams(node_dim, edge_dim, dist_dim, time_dim, extra_heads)
        self.node_time_mlp = _seq(Marker(), nn.Linear(time_dim, node_dim * 6))
        self.edge_time_mlp = _seq(Marker(), nn.Linear(time_dim, edge_dim * 6))
        self.dist_layer = CondGaussianParams(dist_dim, time_dim)


class MHAParams(Holder):
    def __init__(self, d_model: int, n_heads: int):
        super().__init__()
        d_k = d_model // n_heads
        self.W_Q = nn.Linear(d_model, d_k * n_heads)
        self.W_K = nn.Linear(d_model, d_k * n_heads)
        self.W_V = nn.Linear(d_model, d_k * n_heads)
        sdp = Holder()
        sdp.scale = nn.Parameter(torch.tensor(d_k ** -0.5), requires_grad=False)
        self.sdp_attn = sdp
        self.to_out = _seq(nn.Linear(n_heads * d_k, d_model), Marker())


class EncoderLayerParams(Holder):
    def __init__(self, d_model: int, n_heads: int, d_ff: int):
        super().__init__()
        self.self_attn = MHAParams(d_model, n_heads)
        self.norm_attn = _seq(Marker(), nn.BatchNorm1d(d_model), Marker())
        self.ff = _seq(nn.Linear(d_model, d_ff), Marker(), Marker(), nn.Linear(d_ff, d_model))
        self.norm_ffn = _seq(Marker(), nn.BatchNorm1d(d_model), Marker())


class SpecFormerParams(Holder):
    """Parameter tree of the spectral encoder (specformer.py:14-67,123-156)."""

    def __init__(self, patch_len, stride, output_dim: int = 256, spectra_version: str = "ir",
                 n_layers: int = 3, d_model: int = 128, n_heads: int = 16, d_ff: int = 256):
        super().__init__()
        self.patch_len, self.stride = list(patch_len), list(stride)
        self.spectra_version = spectra_version
        self.used_spectra_type = used_spectra(spectra_version)
        self.d_model, self.n_heads, self.d_ff, self.n_layers = d_model, n_heads, d_ff, n_layers
        self.output_dim = output_dim
        self.patch_nums = [int((SPECTRUM_LENGTHS[i] - patch_len[i]) / stride[i] + 1)
                           for i in self.used_spectra_type]
        backbone = Holder()
        backbone.W_P = nn.ModuleList([nn.Linear(patch_len[i], d_model) for i in self.used_spectra_type])

        def pos(q_len):
            w = torch.empty((q_len, d_model))
            nn.init.uniform_(w, -0.02, 0.02)
            return nn.Parameter(w)

        if spectra_version == "allspectra":
            backbone.W_pos_uv = pos(self.patch_nums[0])
            backbone.W_pos_ir = pos(self.patch_nums[1])
            backbone.W_pos_raman = pos(self.patch_nums[2])
        else:
            backbone.W_pos = pos(self.patch_nums[0])
        encoder = Holder()
        encoder.layers = nn.ModuleList([EncoderLayerParams(d_model, n_heads, d_ff) for _ in range(n_layers)])
        backbone.encoder = encoder
        self.backbone = backbone
        self.head_nf = d_model * sum(self.patch_nums)
        head = Holder()
        head.linear = nn.Linear(self.head_nf, output_dim)
        self.head = head
        self.out_norm = nn.LayerNorm(output_dim)


def build_dmt_tree(module: nn.Module, config) -> None:
    """Attach the DMT parameter tree to ``module`` in the reference's registration order."""
    in_node_dim = config.data.atom_types + int(config.model.include_fc_charge)
    hidden = config.model.nf
    edge_hidden = hidden // 4
    n_heads = config.model.n_heads
    n_layers = config.model.n_layers
    time_dim = hidden * 4
    if not (config.model.dist_gbf and config.model.cond_time and config.model.gbf_name == "CondGaussianLayer"):
        raise ValueError("the MI355X path implements the shipped configuration: dist_gbf=True, "
                         "cond_time=True, gbf_name='CondGaussianLayer'")
    in_edge_dim = config.model.edge_ch * 2 + edge_hidden
    module.node_emb = nn.Linear(in_node_dim * 2, hidden)
    module.edge_emb = nn.Linear(in_edge_dim, edge_hidden)
    module.dist_layer = CondGaussianParams(edge_hidden, time_dim)
    cat_node = (hidden * 2) // n_layers
    cat_edge = (edge_hidden * 2) // n_layers
    for i in range(n_layers):
        module.add_module("e_block_%d" % i, MixBlockParams(hidden, edge_hidden, time_dim,
                                                           config.model.n_extra_heads, n_heads,
                                                           config.model.mlp_ratio))
        module.add_module("node_%d" % i, nn.Linear(hidden, cat_node))
        module.add_module("edge_%d" % i, nn.Linear(edge_hidden, cat_edge))
    module.node_pred_mlp = _seq(nn.Linear(cat_node * n_layers + hidden, hidden), Marker(),
                                nn.Linear(hidden, hidden // 2), Marker(),
                                nn.Linear(hidden // 2, in_node_dim))
    module.edge_type_mlp = _seq(nn.Linear(cat_edge * n_layers + edge_hidden, edge_hidden), Marker(),
                                nn.Linear(edge_hidden, edge_hidden // 2), Marker(),
                                nn.Linear(edge_hidden // 2, config.model.edge_ch - 1))
    module.edge_exist_mlp = _seq(nn.Linear(cat_edge * n_layers + edge_hidden, edge_hidden), Marker(),
                                 nn.Linear(edge_hidden, edge_hidden // 2), Marker(),
                                 nn.Linear(edge_hidden // 2, 1))
    module.time_mlp = _seq(SinusoidWeights(16), nn.Linear(17, time_dim), Marker(),
                           nn.Linear(time_dim, time_dim))
    module.cond_encoder = SpecFormerParams(config.model.patch_len, config.model.stride, output_dim=hidden,
                                           spectra_version=config.data.spectra_version)
    module.cond_lin = nn.Linear(hidden, time_dim)
