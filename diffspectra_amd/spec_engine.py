"""SpecFormer conditioning encoder on the HIP library (K1 of SURVEY §7).

Runs once per molecule (its input, the spectra, is loop-invariant over the 1000 denoising steps —
SURVEY §0.6a), so it is assembled from the library's generic fp32-MFMA GEMM (fused bias / GELU / residual /
eval-BatchNorm epilogues, unfold-view and token-slice addressing), the residual-score attention kernel and
the LayerNorm kernel.  Arithmetic restated from reference ``models/specformer.py:77-120,167-200,279-309,
345-425,457-470``; torch is used for allocation only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from .config import SPECTRUM_LENGTHS, used_spectra


class SpecEngine:
    D, HEADS, DK, DFF, LAYERS = 128, 16, 8, 256, 3
    CHUNK = 128   # molecules per pass: the residual-score buffer is B*16*L*L floats (7.7 MB per molecule at L=347)

    def __init__(self, sd: Dict[str, torch.Tensor], config, device, lib):
        from . import engine as E
        self.E, self.lib, self.device = E, lib, device
        self.version = config.data.spectra_version
        self.used = used_spectra(self.version)
        self.patch_len, self.stride = list(config.model.patch_len), list(config.model.stride)
        self.patch_nums = [int((SPECTRUM_LENGTHS[i] - self.patch_len[i]) / self.stride[i] + 1) for i in self.used]
        self.L = sum(self.patch_nums)
        p = "cond_encoder."
        g = lambda k: sd[p + k].detach().float().cpu()
        t: Dict[str, torch.Tensor] = {}
        for slot in range(len(self.used)):
            t[f"wp{slot}.w"] = E.pack_linear(g(f"backbone.W_P.{slot}.weight"))
            t[f"wp{slot}.b"] = E.pad_vec(g(f"backbone.W_P.{slot}.bias"))
            name = ("W_pos_uv", "W_pos_ir", "W_pos_raman")[slot] if self.version == "allspectra" else "W_pos"
            t[f"pos{slot}"] = g("backbone." + name).contiguous().reshape(-1)
        for l in range(self.LAYERS):
            b = f"backbone.encoder.layers.{l}."
            wqkv = torch.cat([g(b + f"self_attn.W_{n}.weight") for n in "QKV"], 0)
            bqkv = torch.cat([g(b + f"self_attn.W_{n}.bias") for n in "QKV"])
            t[f"l{l}.qkv.w"], t[f"l{l}.qkv.b"] = E.pack_linear(wqkv), E.pad_vec(bqkv)
            t[f"l{l}.out.w"], t[f"l{l}.out.b"] = E.pack_linear(g(b + "self_attn.to_out.0.weight")), E.pad_vec(g(b + "self_attn.to_out.0.bias"))
            t[f"l{l}.ff0.w"], t[f"l{l}.ff0.b"] = E.pack_linear(g(b + "ff.0.weight")), E.pad_vec(g(b + "ff.0.bias"))
            t[f"l{l}.ff3.w"], t[f"l{l}.ff3.b"] = E.pack_linear(g(b + "ff.3.weight")), E.pad_vec(g(b + "ff.3.bias"))
            for nm in ("norm_attn", "norm_ffn"):                          # eval-mode BatchNorm1d as a column affine
                w_, b_ = g(b + nm + ".1.weight"), g(b + nm + ".1.bias")
                rm, rv = g(b + nm + ".1.running_mean"), g(b + nm + ".1.running_var")
                sc = w_ / torch.sqrt(rv + 1e-5)
                t[f"l{l}.{nm}.scale"], t[f"l{l}.{nm}.shift"] = sc, b_ - rm * sc
            t[f"l{l}.scale"] = g(b + "self_attn.sdp_attn.scale").reshape(1)
        t["head.w"], t["head.b"] = E.pack_linear(g("head.linear.weight")), E.pad_vec(g("head.linear.bias"))
        t["norm.g"], t["norm.b"] = g("out_norm.weight"), g("out_norm.bias")
        t["cond.w"], t["cond.b"] = E.pack_linear(sd["cond_lin.weight"].detach().float().cpu()), E.pad_vec(sd["cond_lin.bias"])
        self.scales = [float(t.pop(f"l{l}.scale")[0]) for l in range(self.LAYERS)]
        offs, chunks, cur = {}, [], 0
        for k, v in t.items():
            pad = (-cur) % 64
            if pad:
                chunks.append(torch.zeros(pad)); cur += pad
            offs[k] = cur
            chunks.append(v.reshape(-1).float()); cur += v.numel()
        self.flat = torch.cat(chunks).to(device)
        self.off = offs

    def _w(self, key) -> int:
        return self.flat.data_ptr() + 4 * self.off[key]

    def encode(self, context) -> torch.Tensor:
        """context: list [uv, ir, raman] of [B,1,L] (allspectra) or a single [B,1,L] tensor → [B,1024] on device."""
        specs = list(context) if self.version == "allspectra" else [context]
        specs = [s.detach().to(self.device, torch.float32).reshape(s.shape[0], -1).contiguous() for s in specs]
        B = specs[0].shape[0]
        out = torch.empty(B, 1024, dtype=torch.float32, device=self.device)
        # The encoder layers run in chunks (their residual-score buffer bounds the chunk), each leaving its final tokens in
        # `zall`; the head (a [B, L*D] x [L*D, 256] GEMM, K = 44 416 for all spectra) then runs ONCE over every molecule: per
        # chunk of 128 rows it was a 4-workgroup launch of 5.6 ms - 0.18 s per 4096 molecules, as much as the attention.
        E, lib, L, D = self.E, self.lib, self.L, self.D
        zall = torch.empty(B * L, D, dtype=torch.float32, device=self.device)
        for b0 in range(0, B, self.CHUNK):
            b1 = min(B, b0 + self.CHUNK)
            self._encode_chunk([s[b0:b1] for s in specs], zall.data_ptr() + 4 * b0 * L * D)
        head = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        hn = torch.empty(B, 256, dtype=torch.float32, device=self.device)
        E.gemm(lib, zall, L * D, self._w("head.w"), self._w("head.b"), head, 256, B, L * D, 256)
        E._check(lib.ds_layernorm_affine(E._ptr(head), C.c_void_p(self._w("norm.g")), C.c_void_p(self._w("norm.b")),
                                         E._ptr(hn), C.c_int(B), C.c_int(256), C.c_float(1e-5), E._stream()),
                 "ds_layernorm_affine")
        E.gemm(lib, hn, 256, self._w("cond.w"), self._w("cond.b"), out, 1024, B, 256, 1024)
        return out

    def _encode_chunk(self, specs, z_out: int):
        """Patch embedding + encoder layers of one chunk of molecules; the last layer writes its tokens to ``z_out`` (device pointer)."""
        E, lib, dev = self.E, self.lib, self.device
        B, L, D = specs[0].shape[0], self.L, self.D
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        z, z1 = f(B * L, D), f(B * L, D)
        qkv, o, ff = f(B * L, 3 * D), f(B * L, D), f(B * L, self.DFF)
        scores = f(B, self.HEADS, L, L)
        tok = 0
        for slot, (ti, spec) in enumerate(zip(self.used, specs)):
            npatch, pl, st = self.patch_nums[slot], self.patch_len[ti], self.stride[ti]
            assert spec.shape[1] == SPECTRUM_LENGTHS[ti]
            # A = unfold view (row (b,p) starts at b*len + p*stride); C = token slice; R = positional table
            E.gemm(lib, spec, st, self._w(f"wp{slot}.w"), self._w(f"wp{slot}.b"), z.data_ptr() + 4 * tok * D, D,
                   B * npatch, pl, D, act=0, R=self._w(f"pos{slot}"), ldr=D, r_grp_rows=npatch,
                   a_grp=(npatch, spec.shape[1]), c_grp=(npatch, L * D))
            tok += npatch
        for l in range(self.LAYERS):
            k = f"l{l}."
            E.gemm(lib, z, D, self._w(k + "qkv.w"), self._w(k + "qkv.b"), qkv, 3 * D, B * L, D, 3 * D)
            E._check(lib.ds_spec_attention(E._ptr(qkv), E._ptr(scores), E._ptr(o), C.c_int(B), C.c_int(L),
                                           C.c_int(self.HEADS), C.c_int(self.DK), C.c_float(self.scales[l]),
                                           C.c_int(1 if l > 0 else 0), E._stream()), "ds_spec_attention")
            E.gemm(lib, o, D, self._w(k + "out.w"), self._w(k + "out.b"), z1, D, B * L, D, D, R=z, ldr=D,
                   col_scale=self._w(k + "norm_attn.scale"), col_shift=self._w(k + "norm_attn.shift"))
            E.gemm(lib, z1, D, self._w(k + "ff0.w"), self._w(k + "ff0.b"), ff, self.DFF, B * L, D, self.DFF, act=2)
            E.gemm(lib, ff, self.DFF, self._w(k + "ff3.w"), self._w(k + "ff3.b"), z_out if l == self.LAYERS - 1 else z, D,
                   B * L, self.DFF, D, R=z1, ldr=D,
                   col_scale=self._w(k + "norm_ffn.scale"), col_shift=self._w(k + "norm_ffn.shift"))
