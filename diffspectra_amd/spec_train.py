"""SpecFormer conditioning encoder in TRAINING mode on the HIP training library: forward tape + hand-written backward.

Mirror of reference ``models/specformer.py:77-120`` (patching, ``TSTiEncoder`` ``:167-200``, three post-norm ``TSTEncoderLayer`` s
``:279-309`` with residual attention scores ``:385-425`` and BatchNorm1d over d_model in training mode ``:247,260``, ``Flatten_Head``
``:467-469``, output LayerNorm ``:119``) followed by ``cond_lin`` (``models/dmt.py:350``).  BatchNorm uses batch statistics and
updates its running statistics in place (momentum 0.1, unbiased variance), as ``nn.BatchNorm1d`` does.  All dropouts of the
encoder are 0 in the reference's constructor defaults.  No PyTorch arithmetic: GEMMs, BatchNorm, attention, GELU and LayerNorm are
``dst_*`` kernels; PyTorch slices, tiles and concatenates buffers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict

import torch

from . import engine as E
from .config import SPECTRUM_LENGTHS, used_spectra
from .train_engine import ADDREF, GELU, Ops, mv

D_MODEL, N_HEADS, D_K, D_FF, N_LAYERS = 128, 16, 8, 256, 3


class SpecTrainGraph:
    def __init__(self, params: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], config, ops: Ops):
        """``params``: reference-named trainable tensors (``cond_encoder.*``, ``cond_lin.*``); ``buffers``: the BatchNorm running
        statistics (updated in place by ``forward``)."""
        self.p, self.buf, self.cfg, self.ops = params, buffers, config, ops
        self.lib, self.dev = ops.lib, ops.dev
        self.version = config.data.spectra_version
        self.used = used_spectra(self.version)
        pl, stv = config.model.patch_len, config.model.stride
        self.patch = [(pl[i], stv[i], int((SPECTRUM_LENGTHS[i] - pl[i]) / stv[i] + 1)) for i in self.used]
        self.L = sum(pn for _, _, pn in self.patch)
        self.pos_names = (["W_pos_uv", "W_pos_ir", "W_pos_raman"] if self.version == "allspectra" else ["W_pos"])
        self.flash = None      # None: the score-free flash attention in bf16 mode, the materialised fp32 kernels otherwise; True / False force one

    def f(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def _bn_fwd(self, x, name, y, stats):
        p, b = self.p, self.buf
        o = self.ops
        R, Cc = x.shape
        E._check(self.lib.dst_bn_fwd(E._ptr(x), C.c_int32(R), C.c_int32(Cc), E._ptr(p[name + ".weight"]), E._ptr(p[name + ".bias"]), C.c_float(1e-5),
                                     E._ptr(y), E._ptr(stats), E._ptr(b[name + ".running_mean"]), E._ptr(b[name + ".running_var"]), E._ptr(o.scratch),
                                     C.c_int64(o.scratch.numel()), self.ops._s()), "dst_bn_fwd")
        b[name + ".num_batches_tracked"] += 1
        self._bn_last.append((name, stats))

    def running_stats_again(self):
        """The buffer updates of one more training-mode forward over the batch of the last ``forward`` (same batch statistics, so nothing
        else of that pass would differ): the reference's self-conditioning step evaluates the encoder twice on one input."""
        b = self.buf
        for name, stats in self._bn_last:
            E._check(self.lib.dst_bn_running_again(E._ptr(stats), C.c_int32(stats.shape[1]), E._ptr(b[name + ".running_mean"]),
                                                   E._ptr(b[name + ".running_var"]), self.ops._s()), "dst_bn_running_again")
            b[name + ".num_batches_tracked"] += 1

    def _bn_bwd(self, dy, x, stats, name, dx, g):
        o = self.ops
        R, Cc = x.shape
        gb = getattr(self, "gbuf", None)
        g[name + ".weight"], g[name + ".bias"] = (gb[name + ".weight"], gb[name + ".bias"]) if gb is not None else (self.f(Cc), self.f(Cc))
        E._check(self.lib.dst_bn_bwd(E._ptr(dy), E._ptr(x), E._ptr(stats), C.c_int32(R), C.c_int32(Cc), E._ptr(self.p[name + ".weight"]), E._ptr(dx),
                                     E._ptr(g[name + ".weight"]), E._ptr(g[name + ".bias"]), E._ptr(o.scratch), C.c_int64(o.scratch.numel()), self.ops._s()),
                 "dst_bn_bwd")

    # ------------------------------------------------------------------ forward
    def forward(self, context, save: bool = True) -> torch.Tensor:
        """``context``: list [uv, ir, raman] of [B,1,L] (allspectra) or one [B,1,L] tensor -> ctx_emb [B,1024] (dmt.py:348-350)."""
        o, p = self.ops, self.p
        pre = "cond_encoder."
        specs = list(context) if self.version == "allspectra" else [context]
        B = specs[0].shape[0]
        L = self.L
        t: Dict[str, object] = dict(B=B)
        self._bn_last = []
        toks, Xs = [], []
        for slot, ((pl, stv, pn), spec) in enumerate(zip(self.patch, specs)):
            X = spec.reshape(B, -1).to(torch.float32).unfold(-1, pl, stv).contiguous().reshape(B * pn, pl)       # specformer.py:105
            z = p[pre + "backbone." + self.pos_names[slot]].unsqueeze(0).expand(B, pn, D_MODEL).contiguous().reshape(B * pn, D_MODEL)
            o.gemm(mv(X), mv(p[pre + f"backbone.W_P.{slot}.weight"]), mv(z), False, True, bias=p[pre + f"backbone.W_P.{slot}.bias"], acc=True)
            Xs.append(X)
            toks.append(z.reshape(B, pn, D_MODEL))
        Z = torch.cat(toks, dim=1).contiguous().reshape(B * L, D_MODEL)                                              # :194
        t["Xs"] = Xs
        layers = []
        prev = None
        scale = float(D_K ** -0.5)
        flash = self.ops.bf16 if self.flash is None else bool(self.flash)
        qkvs = []
        for l in range(N_LAYERS):
            base = pre + f"backbone.encoder.layers.{l}."
            qkv = self.f(B * L, 3 * D_MODEL)
            # W_Q | W_K | W_V read the same tokens: one product from a per-call concatenation of the three weights
            Wc = torch.cat([p[base + f"self_attn.{nm}.weight"] for nm in ("W_Q", "W_K", "W_V")], dim=0)
            bc = torch.cat([p[base + f"self_attn.{nm}.bias"] for nm in ("W_Q", "W_K", "W_V")], dim=0)
            o.lin_fwd(mv(Z), mv(Wc), bc, mv(qkv))
            ast, ao = self.f(B, N_HEADS, L, 2), self.f(B * L, D_MODEL)
            qkvs.append(qkv)
            if flash:                                                  # bf16 mode: scores recomputed from the q | k of layers 0 .. l, never stored
                qp = [E._ptr(q_) for q_ in qkvs] + [None] * (3 - len(qkvs))
                E._check(self.lib.dst_spec_attn_flash_fwd(qp[0], qp[1], qp[2], C.c_int32(l + 1), E._ptr(ast), E._ptr(ao), C.c_int32(B), C.c_int32(L),
                                                          C.c_int32(N_HEADS), C.c_int32(D_K), C.c_float(scale), self.ops._s()), "dst_spec_attn_flash_fwd")
                scores = None
            else:
                Lp = (L + 31) // 32 * 32                               # padded row stride of the [L, L] score matrices
                scores = self.f(B, N_HEADS, L, Lp)
                E._check(self.lib.dst_spec_attn_fwd(E._ptr(qkv), E._ptr(prev), E._ptr(scores), E._ptr(ast), E._ptr(ao), C.c_int32(B), C.c_int32(L),
                                                    C.c_int32(N_HEADS), C.c_int32(D_K), C.c_float(scale), self.ops._s()), "dst_spec_attn_fwd")
            r1 = self.f(B * L, D_MODEL)                               # Z + to_out(attention): the residual is added in the product's epilogue
            o.gemm(mv(ao), mv(p[base + "self_attn.to_out.0.weight"]), mv(r1), False, True, bias=p[base + "self_attn.to_out.0.bias"], dact=ADDREF, ref=mv(Z))
            z1, st1 = self.f(B * L, D_MODEL), self.f(3, D_MODEL)
            self._bn_fwd(r1, base + "norm_attn.1", z1, st1)
            a = self.f(B * L, D_FF)
            ga = self.f(B * L, D_FF)
            o.lin_fwd(mv(z1), mv(p[base + "ff.0.weight"]), p[base + "ff.0.bias"], mv(a), act=GELU, out2=mv(ga))
            r2 = self.f(B * L, D_MODEL)
            o.gemm(mv(ga), mv(p[base + "ff.3.weight"]), mv(r2), False, True, bias=p[base + "ff.3.bias"], dact=ADDREF, ref=mv(z1))
            z2, st2 = self.f(B * L, D_MODEL), self.f(3, D_MODEL)
            self._bn_fwd(r2, base + "norm_ffn.1", z2, st2)
            if save:
                layers.append(dict(Zin=Z, qkv=qkv, Wqkv=Wc, scores=scores, ast=ast, ao=ao, r1=r1, st1=st1, z1=z1, a=a, ga=ga, r2=r2, st2=st2, has_prev=prev is not None))
            prev = scores
            Z = z2
        flat = Z.reshape(B, L * D_MODEL)
        zh = self.f(B, 256)
        o.lin_fwd(mv(flat), mv(p[pre + "head.linear.weight"]), p[pre + "head.linear.bias"], mv(zh))
        zs, st_ln = self.f(B, 256), self.f(B, 2)
        E._check(self.lib.dst_ln_affine_fwd(E._ptr(zh), C.c_int32(B), C.c_int32(256), E._ptr(p[pre + "out_norm.weight"]), E._ptr(p[pre + "out_norm.bias"]),
                                            C.c_float(1e-5), E._ptr(zs), E._ptr(st_ln), self.ops._s()), "dst_ln_affine_fwd")
        ctx = self.f(B, 1024)
        o.lin_fwd(mv(zs), mv(p["cond_lin.weight"]), p["cond_lin.bias"], mv(ctx))
        if save:
            t.update(layers=layers, flat=flat, zh=zh, zs=zs, st_ln=st_ln, flash=flash)
            self.t = t
        return ctx

    # ------------------------------------------------------------------ backward
    def backward(self, dctx: torch.Tensor) -> Dict[str, torch.Tensor]:
        o, p, t = self.ops, self.p, self.t
        pre = "cond_encoder."
        B, L = t["B"], self.L
        g: Dict[str, torch.Tensor] = {}
        o.async_dw = bool(int(os.environ.get("DIFFSPECTRA_ASYNC_DW", "1")))     # weight gradients on the side stream (train_engine.Ops.lin_bwd_w)
        two_streams = bool(int(os.environ.get("DIFFSPECTRA_NODE_STREAM", "1"))) and getattr(o, "main_stream", None) is not None

        gbuf = getattr(self, "gbuf", None)

        def gw(name):
            g[name] = gbuf[name] if gbuf is not None else torch.empty_like(p[name])
            return g[name]

        o.lin_bwd_w(mv(dctx), mv(t["zs"]), mv(gw("cond_lin.weight")), gw("cond_lin.bias"))
        dzs = self.f(B, 256)
        o.lin_bwd_x(mv(dctx), mv(p["cond_lin.weight"]), mv(dzs))
        dzh = self.f(B, 256)
        E._check(self.lib.dst_ln_affine_bwd(E._ptr(dzs), E._ptr(t["zh"]), E._ptr(t["st_ln"]), C.c_int32(B), C.c_int32(256), E._ptr(p[pre + "out_norm.weight"]),
                                            E._ptr(dzh), E._ptr(gw(pre + "out_norm.weight")), E._ptr(gw(pre + "out_norm.bias")), self.ops._s()),
                 "dst_ln_affine_bwd")
        o.lin_bwd_w(mv(dzh), mv(t["flat"]), mv(gw(pre + "head.linear.weight")), gw(pre + "head.linear.bias"))
        dZ = self.f(B * L, D_MODEL)
        o.lin_bwd_x(mv(dzh), mv(p[pre + "head.linear.weight"]), mv(dZ.view(B, L * D_MODEL)))
        dscores_in = None
        scale = float(D_K ** -0.5)
        cat_dst, cat_src = [], []                                      # gradients of the concatenated W_Q | W_K | W_V, scattered after the side stream has joined
        flash = t["flash"]
        if flash:                                                      # every layer's attention backward adds into the q | k columns of the layers below it
            dqkv_all = [self.f(B * L, 3 * D_MODEL) for _ in range(N_LAYERS)]     # (assigned by the last layer's call: accumulate = 0)
            qkv_all = [lt_["qkv"] for lt_ in t["layers"]]
        for l in reversed(range(N_LAYERS)):
            lt = t["layers"][l]
            base = pre + f"backbone.encoder.layers.{l}."
            dr2 = self.f(B * L, D_MODEL)
            self._bn_bwd(dZ, lt["r2"], lt["st2"], base + "norm_ffn.1", dr2, g)
            o.lin_bwd_w(mv(dr2), mv(lt["ga"]), mv(gw(base + "ff.3.weight")), gw(base + "ff.3.bias"))
            da = self.f(B * L, D_FF)
            o.lin_bwd_x(mv(dr2), mv(p[base + "ff.3.weight"]), mv(da), dact=GELU, ref=mv(lt["a"]))
            o.lin_bwd_w(mv(da), mv(lt["z1"]), mv(gw(base + "ff.0.weight")), gw(base + "ff.0.bias"))
            dz1 = self.f(B * L, D_MODEL)                                                         # dz1 = dr2 (residual) + da W0, in a buffer of its own:
            o.lin_bwd_x(mv(da), mv(p[base + "ff.0.weight"]), mv(dz1), dact=ADDREF, ref=mv(dr2))  # dr2 is an operand of a weight-gradient product that
                                                                                                 # may still be running on the side stream
            dr1 = self.f(B * L, D_MODEL)
            self._bn_bwd(dz1, lt["r1"], lt["st1"], base + "norm_attn.1", dr1, g)
            o.lin_bwd_w(mv(dr1), mv(lt["ao"]), mv(gw(base + "self_attn.to_out.0.weight")), gw(base + "self_attn.to_out.0.bias"))
            dao = self.f(B * L, D_MODEL)
            o.lin_bwd_x(mv(dr1), mv(p[base + "self_attn.to_out.0.weight"]), mv(dao))
            if flash:
                qp = [E._ptr(q_) for q_ in qkv_all[:l + 1]] + [None] * (2 - l)
                gp = [E._ptr(q_) for q_ in dqkv_all[:l + 1]] + [None] * (2 - l)
                def flash_bwd(part):
                    E._check(self.lib.dst_spec_attn_flash_bwd(qp[0], qp[1], qp[2], C.c_int32(l + 1), E._ptr(lt["ast"]), E._ptr(lt["ao"]), E._ptr(dao), gp[0], gp[1],
                                                              gp[2], C.c_int32(B), C.c_int32(L), C.c_int32(N_HEADS), C.c_int32(D_K), C.c_float(scale),
                                                              C.c_int32(part), C.c_int32(0 if l == N_LAYERS - 1 else 1), self.ops._s()), "dst_spec_attn_flash_bwd")
                if two_streams:                                        # the key side beside the query side (disjoint columns of dqkv): each kernel
                    with o.node_section():                             # alone keeps two 4-wave workgroups on a CU
                        flash_bwd(2)
                    flash_bwd(1)
                    o.main_wait()
                else:
                    flash_bwd(0)
                dqkv, dscores = dqkv_all[l], None
            else:
                dqkv, dscores = self.f(B * L, 3 * D_MODEL), self.f(B, N_HEADS, L, (L + 31) // 32 * 32)
                E._check(self.lib.dst_spec_attn_bwd(E._ptr(lt["qkv"]), E._ptr(lt["scores"]), E._ptr(lt["ast"]), E._ptr(dao), E._ptr(dscores_in), E._ptr(dqkv), E._ptr(dscores),
                                                    C.c_int32(B), C.c_int32(L), C.c_int32(N_HEADS), C.c_int32(D_K), C.c_float(scale), self.ops._s()),
                         "dst_spec_attn_bwd")
            dzin = self.f(B * L, D_MODEL)                                                        # dZin = dr1 (residual) + dqkv Wqkv (dr1 stays intact, as dr2 above)
            dWc, dbc = self.f(3 * D_MODEL, D_MODEL), self.f(3 * D_MODEL)
            o.lin_bwd_w(mv(dqkv), mv(lt["Zin"]), mv(dWc), dbc)
            o.lin_bwd_x(mv(dqkv), mv(lt["Wqkv"]), mv(dzin), dact=ADDREF, ref=mv(dr1))
            for k, nm in enumerate(("W_Q", "W_K", "W_V")):
                cat_dst += [gw(base + f"self_attn.{nm}.weight"), gw(base + f"self_attn.{nm}.bias")]
                cat_src += [dWc[k * D_MODEL:(k + 1) * D_MODEL], dbc[k * D_MODEL:(k + 1) * D_MODEL]]
            dscores_in = dscores if lt["has_prev"] else None
            dZ = dzin
        dZ3 = dZ.view(B, L, D_MODEL)
        tok0 = 0
        for slot, (pl, stv, pn) in enumerate(self.patch):
            dz = dZ3[:, tok0:tok0 + pn].contiguous().reshape(B * pn, D_MODEL)
            o.lin_bwd_w(mv(dz), mv(t["Xs"][slot]), mv(gw(pre + f"backbone.W_P.{slot}.weight")), gw(pre + f"backbone.W_P.{slot}.bias"))
            o.colsum(mv(dz.view(B, pn * D_MODEL)), gw(pre + "backbone." + self.pos_names[slot]).view(-1))
            tok0 += pn
        o.join_dw()
        o.async_dw = False
        torch._foreach_copy_(cat_dst, cat_src)
        self.t = None
        return g
