"""DMT denoiser with the reference's factory / call surface, computed by the HIP library.

Mirror of reference ``models/dmt.py:178-412``: ``@register_model(name='DMT')``, one ``config`` constructor
argument, the same parameter tree (``params.build_dmt_tree``) and
``forward(t, xh, node_mask, edge_mask, context=None, *args, edge_x=, noise_level=, cond_x=, cond_edge_x=)``
→ ``(Tensor[B,N,9], Tensor[B,N,N,2])``.  The forward pass is ``ds_forward`` of ``csrc/ds_kernels.hip``;
there is no PyTorch implementation of the arithmetic in this package.
"""
from __future__ import annotations

import torch
from torch import nn

from .params import build_dmt_tree
from .registry import register_model


@register_model(name="DMT")
class DMT(nn.Module):
    """Conditional Diffusion Molecule Transformer with self-conditioning (inference path, MI355X)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_layers = config.model.n_layers
        self.pred_data = config.model.pred_data
        self.spectra_version = config.data.spectra_version
        if not (config.model.nf == 256 and config.model.n_layers == 8 and config.model.n_heads == 16
                and config.model.n_extra_heads == 2 and config.model.edge_ch == 2 and config.model.mlp_ratio == 2
                and config.model.CoM and config.model.pred_data and config.model.softmax_inf
                and config.model.include_fc_charge and config.data.atom_types == 5):
            raise ValueError("the HIP kernels are specialised to the shipped DMT configuration "
                             "(nf=256, 8 blocks, 16 heads incl. 2 adjacency heads, 2 edge channels, CoM, pred_data)")
        build_dmt_tree(self, config)
        self._engine = None
        self._engine_key = None
        path = getattr(config.model, "pretrained_specformer_path", "")
        if path:
            self.load_pretrained_specformer(path)

    # ------------------------------------------------------------------ weights
    def load_pretrained_specformer(self, ckpt_path):
        """Key mapping of reference dmt.py:268-303 (Lightning-style checkpoint → ``cond_encoder.*``)."""
        ckpt = torch.load(ckpt_path, map_location="cpu")
        if "state_dict" not in ckpt:
            print("Warning: pretrained model does not contain 'state_dict' key. Loading the entire checkpoint.")
            return 0
        return self.load_pretrained_specformer_state(ckpt["state_dict"])

    def load_pretrained_specformer_state(self, state_dict):
        current = self.cond_encoder.state_dict()
        prefix = next((p for p in ("model.representation_spec_model", "model.representation_model")
                       if any(k.startswith(p) for k in state_dict)), None)
        if prefix is None:
            print("Warning: No matching prefix found in the state_dict.")
            return 0
        matched = 0
        for tgt in list(current.keys()):
            src = f"{prefix}.{tgt}"
            if tgt in ("out_norm.weight", "out_norm.bias"):
                src = f"model.representation_model.out_norm.{tgt.split('.')[-1]}"
            if src in state_dict and current[tgt].shape == state_dict[src].shape:
                current[tgt] = state_dict[src]
                matched += 1
        if matched:
            self.cond_encoder.load_state_dict(current, strict=False)
        return matched

    def _weights_key(self):
        """Fingerprint of the current weights: device + a position-sensitive 64-bit checksum of every floating-point parameter
        and buffer (``ds_fingerprint``: one kernel over all tensors + one 8-byte device->host copy).

        ``Tensor._version`` is not enough: both EMA implementations write with ``p.data.copy_(...)`` (reference
        models/ema.py:55,77), which leaves the version counter untouched.  The checksum weights every element by a hash of
        its position, so sign flips, row / head permutations and copies from an equal-norm tensor all change it."""
        import ctypes as C
        from .engine import load_library, _check, _stream
        ts = [t.detach() for t in list(self.parameters()) + list(self.buffers()) if t.is_floating_point()]
        dev = ts[0].device
        if dev.type != "cuda":
            return (str(dev),)
        ptrs = tuple(t.data_ptr() for t in ts)
        tab = getattr(self, "_fp_table", None)
        if tab is None or tab[0] != ptrs:
            if any(t.dtype != torch.float32 or not t.is_contiguous() for t in ts):
                raise RuntimeError("DMT parameters must be contiguous fp32 tensors")
            prefix = [0]
            for t in ts:
                prefix.append(prefix[-1] + t.numel())
            tab = (ptrs, torch.tensor(ptrs, dtype=torch.int64, device=dev), torch.tensor(prefix, dtype=torch.int64, device=dev),
                   torch.zeros(1, dtype=torch.int64, device=dev))
            self._fp_table = tab
        _check(load_library().ds_fingerprint(C.c_void_p(tab[1].data_ptr()), C.c_void_p(tab[2].data_ptr()), C.c_int32(len(ts)),
                                             C.c_void_p(tab[3].data_ptr()), _stream()), "ds_fingerprint")
        return (str(dev), ptrs[:4], int(tab[3].item()))

    def invalidate_engine(self):
        """Drop the packed weights; the next call re-packs from the current parameters."""
        self._engine = None
        self._engine_key = None
        self._ctx_cache = None

    def engine(self):
        """Packed-weight HIP engine for the current parameters (re-packed whenever their values change: ``load_state_dict``,
        ``ema.copy_to`` / ``ema.restore``, in-place edits)."""
        from .engine import DmtEngine
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("DMT runs on an MI355X only (move it to a 'cuda' device); diffspectra_amd has no CPU path")
        key = self._weights_key()
        if self._engine is None or self._engine_key != key:
            self._engine = DmtEngine(self.state_dict(), self.config, dev)
            self._engine_key = key
        return self._engine

    # ------------------------------------------------------------------ score function
    def forward(self, t, xh, node_mask, edge_mask, context=None, *args, **kwargs):
        """Same contract as reference dmt.py:306-321; ``t`` is accepted and unused there as well.

        ``model.eval()``: the sampling kernels (``ds_forward``), no autograd graph.  ``model.train()`` (reference callers:
        ``losses.py:346-357``): the training library - FF dropout active, SpecFormer's BatchNorm on batch statistics with its running
        statistics updated - and, with gradients enabled, outputs that carry a ``grad_fn``: a caller's own loss around ``model(...)``
        fills ``p.grad`` on ``backward()`` (the hand-written backward of ``train_engine`` / ``spec_train`` behind one autograd node)."""
        if self.training:
            return self._forward_train(xh, node_mask, edge_mask, context, kwargs)
        with torch.no_grad():
            return self._forward_eval(xh, node_mask, edge_mask, context, kwargs)

    def _forward_train(self, xh, node_mask, edge_mask, context, kwargs):
        from .losses import _trainer, _deliver_grads
        edge_x, cond_x, cond_edge_x = kwargs["edge_x"], kwargs["cond_x"], kwargs["cond_edge_x"]
        noise_level = kwargs["noise_level"]
        if context is None:
            raise TypeError("DMT.forward needs `context` (spectra)")
        tr = _trainer(self)
        dev = tr.dev
        tr.ops.bf16 = getattr(self.config.training, "precision", "fp32") == "bf16"      # (before graphs(): the per-step weight copies depend on it)
        named, dmt, spec = tr.graphs()
        f32 = lambda v: v.detach().to(dev, torch.float32)
        TL = tr.layout(f32(node_mask).reshape(node_mask.shape[0], -1))
        TL.L.check_edge_mask(edge_mask)
        TL.L.check_edge_symmetry(edge_x, "edge_x")
        z, ez = TL.pack_nodes(f32(xh)), TL.pack_pairs(f32(edge_x))
        cond_n = cond_e = None
        if cond_x is not None:
            TL.L.check_edge_symmetry(cond_edge_x, "cond_edge_x")
            cond_n, cond_e = TL.pack_nodes(f32(cond_x)), TL.pack_pairs(f32(cond_edge_x))
        ctx_in = [f32(c) for c in context] if isinstance(context, (list, tuple)) else f32(context)
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in named.values())
        dmt.dropout_p = float(getattr(self.config.model, "dropout", 0.0))
        dmt.dropout_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if dmt.dropout_p > 0 else 0
        tr.ops.begin()
        try:
            with torch.no_grad():
                ctx = spec.forward(ctx_in, save=want_grad)
                pos, atom, edge = dmt.forward(TL, z, ez, f32(noise_level).contiguous(), ctx, cond_n, cond_e, save=want_grad)
                out_xh = TL.unpack_nodes(torch.cat([pos, atom], dim=1)).to(xh.dtype)
                out_edge = TL.unpack_pairs(edge).to(edge_x.dtype)
        finally:
            tr.ops.end()
        if not want_grad:
            return out_xh, out_edge
        params = list(named.values())
        return _DmtGraph.apply(tr, named, dmt, spec, TL, _deliver_grads, out_xh, out_edge, *params)

    def _forward_eval(self, xh, node_mask, edge_mask, context, kwargs):
        edge_x, cond_x, cond_edge_x = kwargs["edge_x"], kwargs["cond_x"], kwargs["cond_edge_x"]
        noise_level = kwargs["noise_level"]
        eng = self.engine()
        L, ws = eng.layout_for(node_mask, edge_mask, validate=True)
        # the kernels keep edge features once per unordered pair (the sampler's edge tensors are symmetric by
        # construction, models/utils.py:100-106); the reference reads both directed entries, so refuse anything else
        L.check_edge_symmetry(edge_x, "edge_x")
        if cond_edge_x is not None:
            L.check_edge_symmetry(cond_edge_x, "cond_edge_x")
        if context is None:
            # reference: `time_mlp(noise_level) + None` raises TypeError (dmt.py:354); keep that behaviour explicit
            raise TypeError("DMT.forward needs `context` (spectra); pass context_emb to the engine for a zero context")
        # The reference re-encodes the loop-invariant spectra on every call (dmt.py:348-350; 62 % of its forward time).  A caller that
        # passes the SAME context tensors again (its sampler does, 1000 times per round) gets the embedding of the first call: keyed
        # on the tensor objects + version counters and on the weights the engine was packed from.
        ctx_list = context if isinstance(context, (list, tuple)) else [context]
        cached = getattr(self, "_ctx_cache", None)
        # keyed on the tensor OBJECTS (held by the cache, so their addresses cannot be re-issued) + their version counters
        if (cached is not None and cached[0] == self._engine_key and len(cached[1]) == len(ctx_list)
                and all(a is b and v == b._version for (a, v), b in zip(cached[1], ctx_list))):
            ctx = cached[2]
        else:
            ctx = eng.context_embedding(context)
            self._ctx_cache = (self._engine_key, [(t, t._version) for t in ctx_list], ctx)
        out_xh, out_edge = eng.forward(L, ws, xh, edge_x, noise_level, cond_x, cond_edge_x, ctx)
        return out_xh.to(xh.dtype), out_edge.to(edge_x.dtype)


class _DmtGraph(torch.autograd.Function):
    """One autograd node around the training graphs' forward tape: ``backward`` packs the gradients of the two dense outputs, walks
    the tape (``DmtTrainGraph.backward`` + ``SpecTrainGraph.backward``) and hands every parameter its gradient.  The score function's
    tensor inputs (noisy state, self-conditioning, spectra) are treated as constants - no reference caller differentiates them."""

    @staticmethod
    def forward(ctx, tr, named, dmt, spec, TL, deliver, out_xh, out_edge, *params):
        ctx.pack = (tr, named, dmt, spec, TL, deliver)
        ctx.tapes = (dmt.t, spec.t)                        # the graphs are shared per model: keep THIS call's tapes
        return out_xh.clone(), out_edge.clone()

    @staticmethod
    def backward(ctx, d_xh, d_edge):
        tr, named, dmt, spec, TL, deliver = ctx.pack
        if ctx.tapes is None:
            raise RuntimeError("backward through DMT.forward twice: the forward tape was already consumed")
        dmt.t, spec.t = ctx.tapes
        ctx.tapes = None
        if d_xh is None:
            d_xh = torch.zeros(TL.B, TL.N, 9, device=tr.dev)
        if d_edge is None:
            d_edge = torch.zeros(TL.B, TL.N, TL.N, 2, device=tr.dev)
        dn = TL.pack_nodes(d_xh.to(torch.float32))
        dpos, datom = dn[:, 0:3].contiguous(), dn[:, 3:9].contiguous()
        de = d_edge.to(torch.float32).reshape(TL.B * TL.N * TL.N, -1)
        dedge = (de.index_select(0, TL.pair_dense) + de.index_select(0, TL.pair_dense_t)).contiguous()   # both cells of a pair carry its value
        flat, _, offs = tr.stage_begin(named)
        tr.ops.begin()
        try:
            g = dmt.backward(dpos, datom, dedge)
            g.update(spec.backward(g.pop("@ctx_emb")))
        finally:
            tr.ops.end()
        return (None,) * 8 + deliver(tr, named, g, flat, offs, None)
