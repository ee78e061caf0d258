"""ctypes binding to ``libdiffspectra_hip.so`` + the host plumbing around it.

Everything arithmetic happens in the HIP library (``csrc/ds_kernels.hip``); this module packs the
reference-named parameters into the library's MFMA-operand layout once, builds the packed-ragged index
tables from ``node_mask``, owns the (torch-allocated) device workspace and issues the C-ABI calls on
torch's current HIP stream.  There is NO fallback: if the library is missing or no GPU is visible the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Optional

import numpy as np
import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("DIFFSPECTRA_HIP_LIB", os.path.join(_PKG, "libdiffspectra_hip.so"))
HEADER_PATH = os.path.join(_ROOT, "include", "diffspectra_hip.h")

_c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())


def _parse_header():
    txt = open(HEADER_PATH).read()
    txt_nc = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)

    def enum_names(name):
        body = re.search(r"enum\s+%s\s*\{(.*?)\}" % name, txt_nc, flags=re.S).group(1)
        return [t.split("=")[0].strip() for t in body.split(",") if t.strip()]

    defs = {}
    for m in re.finditer(r"#define\s+(DS_\w+)\s+(\(?[-\w\s\*\+\(\)]+?\)?)\s*$", txt_nc, flags=re.M):
        defs[m.group(1)] = m.group(2)
    consts: Dict[str, int] = {}
    for _ in range(4):  # resolve nested defines
        for k, v in defs.items():
            if k in consts:
                continue
            try:
                consts[k] = int(eval(v, {"__builtins__": {}}, consts))
            except Exception:
                pass
    blk = enum_names("ds_block_slot")
    glb = enum_names("ds_global_slot")
    exports = re.findall(r"^\s*(?:int|void)\s+(ds_\w+)\s*\(", txt_nc, flags=re.M)
    return consts, blk, glb, exports


CONSTS, BLOCK_SLOTS, GLOBAL_SLOTS, EXPORTS = _parse_header()
NB = CONSTS["DS_NBLOCKS"]
W_BLOCK_SLOTS = len(BLOCK_SLOTS) - 1      # last enumerator is the count
W_GLOBAL_SLOTS = len(GLOBAL_SLOTS) - 1
W_NUM_SLOTS = NB * W_BLOCK_SLOTS + W_GLOBAL_SLOTS
ADA_COLS = CONSTS["DS_ADA_COLS"]
ADA_STRIDE = CONSTS["DS_ADA_BLOCK_STRIDE"]
MAX_ATOMS = CONSTS["DS_MAX_ATOMS"]


class DsWeights(C.Structure):
    _fields_ = [("base", C.c_void_p), ("off_dev", C.c_void_p), ("off", C.c_int64 * W_NUM_SLOTS),
                ("edge_th", C.c_float), ("spatial_cut_off", C.c_float)]


class DsLayout(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("Nn", C.c_int32), ("Pp", C.c_int32),
                ("max_n", C.c_int32), ("_pad", C.c_int32),
                ("node_off", C.c_void_p), ("pair_off", C.c_void_p), ("node_dense", C.c_void_p),
                ("node_mol", C.c_void_p), ("pair_a", C.c_void_p), ("pair_b", C.c_void_p), ("pair_mol", C.c_void_p),
                ("mol_by_size", C.c_void_p)]


_WS_FIELDS = ["pos", "h", "e", "atom_hids", "edge_hids", "tfeat", "tmid", "temb_silu", "ada", "qkv", "ye",
              "dist", "attn", "u", "ac", "ed", "lg", "tr", "adj", "flags"]


class DsWorkspace(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _WS_FIELDS]


class DsGemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("a_grp_rows", C.c_int32), ("_p0", C.c_int32),
                ("a_grp_stride", C.c_int64), ("Wp", C.c_void_p), ("bias", C.c_void_p),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("c_grp_rows", C.c_int32), ("_p1", C.c_int32),
                ("c_grp_stride", C.c_int64), ("M", C.c_int32), ("K", C.c_int32), ("N", C.c_int32), ("act", C.c_int32),
                ("R", C.c_void_p), ("ldr", C.c_int64), ("r_grp_rows", C.c_int32), ("a_silu", C.c_int32),
                ("col_scale", C.c_void_p), ("col_shift", C.c_void_p)]


_lib = None


def load_library() -> C.CDLL:
    """Load the HIP library; fail loudly if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                           "diffspectra_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    sizes = (C.c_int64 * 4)()
    lib.ds_struct_sizes(sizes)
    mine = [C.sizeof(DsWeights), C.sizeof(DsLayout), C.sizeof(DsWorkspace), C.sizeof(DsGemmArgs)]
    if list(sizes) != mine:
        raise RuntimeError(f"C-ABI struct layout mismatch: library {list(sizes)} vs binding {mine}")
    for name in EXPORTS:
        fn = getattr(lib, name)          # AttributeError if the header declares something the .so lacks
        fn.restype = None if name == "ds_struct_sizes" else C.c_int
    _lib = lib
    return lib


def _check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status} "
                           f"({ {-1: 'bad argument', -2: 'HIP launch error'}.get(status, 'unknown')})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


# ----------------------------------------------------------------------------------------- weight packing

def pack_linear(weight: torch.Tensor, n_pad_to: int = 32) -> torch.Tensor:
    """torch Linear weight [N_out, K_in] → MFMA-B packed [Kp/8][2][Np][4] with k = 8*kg + 4*half + s."""
    w = weight.detach().to(torch.float32).cpu()
    N, K = w.shape
    Kp, Np = (K + 7) // 8 * 8, (N + n_pad_to - 1) // n_pad_to * n_pad_to
    b = torch.zeros(Kp, Np)
    b[:K, :N] = w.t()
    return b.view(Kp // 8, 2, 4, Np).permute(0, 1, 3, 2).contiguous().reshape(-1)


SPLIT_SCALE = 2048.0   # 2^11: the low fp16 plane of a split operand is stored scaled so that it keeps 11 significant bits
TANH_PRESCALE = 2.8853900817779268     # 2 log2(e), ds_device.h ds_tanh2_prescaled
F16_MAX = 65504.0      # a split value has fp16's exponent range: |w| must stay below this (activations saturate in the kernels)


def _check_split_range(w: torch.Tensor, what: str) -> None:
    """A weight at or beyond fp16's largest finite value cannot be carried as two fp16 planes (its high plane would be inf and
    every product NaN).  No trained DMT comes near it (random-init |w| < 1); refuse loudly instead of computing garbage."""
    m = float(w.abs().max()) if w.numel() else 0.0
    if not (m < F16_MAX):
        raise ValueError(f"{what}: max |w| = {m:.4g} is outside the split-fp16 range (|w| < {F16_MAX:g}); "
                         "this weight cannot run on the f16 matrix pipe")



def pack_linear_f16_split(weight: torch.Tensor) -> torch.Tensor:
    """torch Linear weight [N_out, K_in] (K % 16 == 0, N % 32 == 0) → two fp16 planes with w ≈ w1 + w2 / 2048 (|error| ≤
    2^-23 |w|), in the A-operand order of v_mfma_f32_32x32x16_f16: halves [plane][K/16][k-half][N][8], k = 16 kb + 8 h + j;
    returned as the fp32 view of those bits (the packed weight buffer is fp32)."""
    w = weight.detach().to(torch.float32).cpu()
    _check_split_range(w, "pack_linear_f16_split")
    if w.shape[0] % 32:                                               # zero rows up to the MFMA tile width
        w = torch.cat([w, torch.zeros(32 - w.shape[0] % 32, w.shape[1])], 0)
    N, K = w.shape
    assert K % 16 == 0
    w1 = w.half()
    w2 = ((w - w1.float()) * SPLIT_SCALE).half()
    planes = torch.stack([w1, w2])                                   # [2, N, K]
    t = planes.view(2, N, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()   # [plane, kb, h, N, 8]
    return t.reshape(-1).view(torch.float32).clone()


def pack_ff4_chain(weight: torch.Tensor) -> torch.Tensor:
    """ff_linear4 [64, 128] for the accumulator-as-operand chain of k_edge_update: the SiLU'd 32 x 32 accumulator tile hc of
    ff_linear3 (lane = row, register i = hidden feature hc*32 + (i&3) + 8(i>>2) + 4h) is used, registers 8s .. 8s+7 at a time,
    as the B fragment of a 32x32x16 f16 MFMA; the A fragment of lane (r, h) must hold, in element j, the weight of output
    ft*32 + r for exactly that hidden feature.  Two fp16 planes (w = w1 + w2/2048); returned as the fp32 view of the bits."""
    w = weight.detach().to(torch.float32).cpu()
    _check_split_range(w, "pack_ff4_chain")
    n_out, k_in = w.shape                                                        # ff_linear4: (64, 128); edge readout .2: (32, 64)
    assert n_out % 32 == 0 and k_in % 32 == 0
    w1 = w.half()
    planes = torch.stack([w1, ((w - w1.float()) * SPLIT_SCALE).half()])          # [2, N, K]
    hc, s_, ft, h, r, j = torch.meshgrid(torch.arange(k_in // 32), torch.arange(2), torch.arange(n_out // 32), torch.arange(2),
                                          torch.arange(32), torch.arange(8), indexing="ij")
    k = hc * 32 + 16 * s_ + 8 * (j >> 2) + 4 * h + (j & 3)
    out = planes[:, ft * 32 + r, k]                                              # [2, K/32, 2, N/32, 2, 32, 8]
    return out.contiguous().reshape(-1).view(torch.float32).clone()


def split_rows_f16(a: torch.Tensor) -> torch.Tensor:
    """fp32 [M, K] → the A-operand layout of ``ds_gemm_split``: halves [M][2][K] (a = a1 + a2/2048), returned as float16."""
    a = a.detach().to(torch.float32)
    fin = torch.isfinite(a)
    sat = lambda t: torch.where(fin, t.clamp(-F16_MAX, F16_MAX), t)                    # the kernels' saturating conversion
    a1 = sat(a).half()
    a2 = sat(((a.double() - a1.double()) * SPLIT_SCALE).float()).half()               # fused on the device: no fp32 overflow in between
    return torch.stack([a1, a2], dim=1).contiguous()


def gemm_split(lib, a_split: torch.Tensor, w_split: torch.Tensor, bias, out: torch.Tensor, M: int, K: int, N: int):
    """C = A W + bias through ``ds_gemm_split`` (operands prepared by ``split_rows_f16`` / ``pack_linear_f16_split``)."""
    _check(lib.ds_gemm_split(_ptr(a_split), _ptr(w_split), _ptr(bias), _ptr(out), C.c_int64(out.stride(0)), C.c_int32(M), C.c_int32(K),
                             C.c_int32(N), _stream()), "ds_gemm_split")


def pad_vec(v: torch.Tensor, to: int = 32) -> torch.Tensor:
    v = v.detach().to(torch.float32).cpu().reshape(-1)
    out = torch.zeros((v.numel() + to - 1) // to * to)
    out[:v.numel()] = v
    return out


def _rbf_tables(sd, name):
    """CondGaussianLayer tables padded 63 → 64 (entry 63 is never read; kept at 1 so it is harmless)."""
    mean = sd[name + ".means.weight"].float().view(-1)
    std = sd[name + ".stds.weight"].float().view(-1).abs() + 1e-5          # layers.py:333
    a = (2 * 3.14159) ** 0.5                                               # layers.py:293-294 (truncated pi)
    astd = a * std                                                         # fp32 product, as torch evaluates it
    one = torch.ones(1)
    return torch.cat([mean, one]), torch.cat([std, one]), torch.cat([astd, one])


def pack_dmt_weights(sd: Dict[str, torch.Tensor]):
    """Reference-named DMT state dict (no ``module.`` prefix) → (flat fp32 tensor, slot offsets)."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items() if not k.startswith("cond_encoder.")}
    chunks: List[torch.Tensor] = []
    offsets = [0] * W_NUM_SLOTS
    cursor = 0

    def put(slot_index, t):
        nonlocal cursor
        pad = (-cursor) % 64                     # 256-byte alignment of every slot
        if pad:
            chunks.append(torch.zeros(pad))
            cursor += pad
        offsets[slot_index] = cursor
        chunks.append(t.reshape(-1).float())
        cursor += t.numel()

    def bslot(b, name):
        return b * W_BLOCK_SLOTS + BLOCK_SLOTS.index(name)

    def gslot(name):
        return NB * W_BLOCK_SLOTS + GLOBAL_SLOTS.index(name)

    def cat_pad_rows(ws, pads):
        rows = []
        for w, p in zip(ws, pads):
            rows.append(w)
            if p > w.shape[0]:
                rows.append(torch.zeros(p - w.shape[0], w.shape[1]))
        return torch.cat(rows, 0)

    ada_rows, ada_bias = [], []
    for b in range(NB):
        p = f"e_block_{b}."
        put(bslot(b, "DS_BW_EDGE_EMB_W"), pack_linear(sd[p + "edge_emb.weight"]))
        put(bslot(b, "DS_BW_EDGE_EMB_B"), pad_vec(sd[p + "edge_emb.bias"]))
        put(bslot(b, "DS_BW_E0_W"), pack_linear(sd[p + "attn_mpnn.lin_edge0.weight"]))
        put(bslot(b, "DS_BW_E1_W"), pack_linear(sd[p + "attn_mpnn.lin_edge1.weight"]))
        wq, wk, wv = (sd[p + f"attn_mpnn.lin_{n}.weight"] for n in ("query", "key", "value"))
        bq, bk, bv = (sd[p + f"attn_mpnn.lin_{n}.bias"] for n in ("query", "key", "value"))
        put(bslot(b, "DS_BW_QKV_W"), pack_linear(cat_pad_rows([wq, wk, wv], [256, 256, 256])))
        put(bslot(b, "DS_BW_QKV_H"), pack_linear_f16_split(cat_pad_rows([wq, wk, wv], [256, 256, 256])))
        put(bslot(b, "DS_BW_QKV_B"), torch.cat([pad_vec(bq, 256), pad_vec(bk, 256), pad_vec(bv, 256)]))
        put(bslot(b, "DS_BW_N2E_W"), pack_linear(sd[p + "node2edge_lin.weight"]))
        put(bslot(b, "DS_BW_N2E_B"), pad_vec(sd[p + "node2edge_lin.bias"]))
        for i, nm in ((1, "FF1"), (2, "FF2"), (3, "FF3"), (4, "FF4")):
            put(bslot(b, f"DS_BW_{nm}_W"), pack_linear(sd[p + f"ff_linear{i}.weight"]))
            put(bslot(b, f"DS_BW_{nm}_B"), pad_vec(sd[p + f"ff_linear{i}.bias"]))
        put(bslot(b, "DS_BW_NODE_RO_W"), pack_linear(sd[f"node_{b}.weight"]))
        put(bslot(b, "DS_BW_NODE_RO_B"), pad_vec(sd[f"node_{b}.bias"]))
        put(bslot(b, "DS_BW_EDGE_RO_W"), pack_linear(sd[f"edge_{b}.weight"]))
        put(bslot(b, "DS_BW_EDGE_RO_B"), pad_vec(sd[f"edge_{b}.bias"]))
        win = sd[p + "equi_update.input_lin.weight"]                       # [256, 640] = [h_row | h_col | e | dist]
        put(bslot(b, "DS_BW_AC_W"), pack_linear(torch.cat([win[:, 0:256], win[:, 256:512]], 0)))
        put(bslot(b, "DS_BW_ED_W"), pack_linear(win[:, 512:640]))
        put(bslot(b, "DS_BW_ED_B"), pad_vec(sd[p + "equi_update.input_lin.bias"]))
        put(bslot(b, "DS_BW_CM0_W"), pack_linear(sd[p + "equi_update.coord_mlp.0.weight"]))
        put(bslot(b, "DS_BW_CM0_B"), pad_vec(sd[p + "equi_update.coord_mlp.0.bias"]))
        put(bslot(b, "DS_BW_CM2_W"), pack_linear(sd[p + "equi_update.coord_mlp.2.weight"]))
        put(bslot(b, "DS_BW_CM0_H"), pack_linear_f16_split(sd[p + "equi_update.coord_mlp.0.weight"]))
        put(bslot(b, "DS_BW_FF3_H"), pack_linear_f16_split(sd[p + "ff_linear3.weight"]))
        put(bslot(b, "DS_BW_FF4_C"), pack_ff4_chain(sd[p + "ff_linear4.weight"]))
        put(bslot(b, "DS_BW_N2E_H"), pack_linear_f16_split(sd[p + "node2edge_lin.weight"]))
        put(bslot(b, "DS_BW_FF1_H"), pack_linear_f16_split(sd[p + "ff_linear1.weight"]))
        put(bslot(b, "DS_BW_FF2_H"), pack_linear_f16_split(sd[p + "ff_linear2.weight"]))
        put(bslot(b, "DS_BW_NODE_RO_H"), pack_linear_f16_split(sd[f"node_{b}.weight"]))
        put(bslot(b, "DS_BW_AC_H"), pack_linear_f16_split(torch.cat([win[:, 0:256], win[:, 256:512]], 0)))
        # k_attn_fused evaluates tanh(x) as 1 - 2 / (1 + exp2(2 log2(e) x)): the factor rides in the packed weights
        put(bslot(b, "DS_BW_E0_H"), pack_linear_f16_split((sd[p + "attn_mpnn.lin_edge0.weight"].double() * TANH_PRESCALE).float()))
        put(bslot(b, "DS_BW_E1_H"), pack_linear_f16_split((sd[p + "attn_mpnn.lin_edge1.weight"].double() * TANH_PRESCALE).float()))
        put(bslot(b, "DS_BW_ED_H"), pack_linear_f16_split(win[:, 512:640]))
        put(bslot(b, "DS_BW_EDGE_EMB_H"), pack_linear_f16_split(sd[p + "edge_emb.weight"]))
        mean, std, astd = _rbf_tables(sd, p + "dist_layer")
        put(bslot(b, "DS_BW_RBF_MEAN"), mean)
        put(bslot(b, "DS_BW_RBF_STD"), std)
        put(bslot(b, "DS_BW_RBF_ASTD"), astd)
        put(bslot(b, "DS_BW_COORD_SCALE"), pad_vec(sd[p + "equi_update.coord_norm.scale"]))
        ws = [sd[p + "node_time_mlp.1.weight"], sd[p + "edge_time_mlp.1.weight"],
              sd[p + "equi_update.time_mlp.1.weight"], sd[p + "dist_layer.time_mlp.1.weight"]]
        bs = [sd[p + "node_time_mlp.1.bias"], sd[p + "edge_time_mlp.1.bias"],
              sd[p + "equi_update.time_mlp.1.bias"], sd[p + "dist_layer.time_mlp.1.bias"]]
        ada_rows.append(cat_pad_rows(ws, [1536, 384, 512, 32]))
        ada_bias.append(torch.cat([pad_vec(x, p_) for x, p_ in zip(bs, [1536, 384, 512, 32])]))
    ada_rows.append(cat_pad_rows([sd["dist_layer.time_mlp.1.weight"]], [32]))
    ada_bias.append(pad_vec(sd["dist_layer.time_mlp.1.bias"], 32))
    ada_w, ada_b = torch.cat(ada_rows, 0), torch.cat(ada_bias)
    assert ada_w.shape == (ADA_COLS, 1024) and ada_b.numel() == ADA_COLS and ada_rows[0].shape[0] == ADA_STRIDE
    put(gslot("DS_GW_SIN_W"), pad_vec(sd["time_mlp.0.weights"]))
    put(gslot("DS_GW_TM1_W"), pack_linear(sd["time_mlp.1.weight"]))
    put(gslot("DS_GW_TM1_B"), pad_vec(sd["time_mlp.1.bias"]))
    put(gslot("DS_GW_TM3_W"), pack_linear(sd["time_mlp.3.weight"]))
    put(gslot("DS_GW_TM3_B"), pad_vec(sd["time_mlp.3.bias"]))
    put(gslot("DS_GW_ADA_W"), pack_linear_f16_split(ada_w))
    put(gslot("DS_GW_ADA_B"), ada_b)
    put(gslot("DS_GW_NODE_EMB_W"), pack_linear(sd["node_emb.weight"]))
    put(gslot("DS_GW_NODE_EMB_B"), pad_vec(sd["node_emb.bias"]))
    put(gslot("DS_GW_EDGE_EMB_W"), pack_linear(sd["edge_emb.weight"]))
    put(gslot("DS_GW_EDGE_EMB_B"), pad_vec(sd["edge_emb.bias"]))
    mean, std, astd = _rbf_tables(sd, "dist_layer")
    put(gslot("DS_GW_RBF_MEAN"), mean)
    put(gslot("DS_GW_RBF_STD"), std)
    put(gslot("DS_GW_RBF_ASTD"), astd)
    for mlp, tag in (("node_pred_mlp", "NP"), ("edge_exist_mlp", "EX"), ("edge_type_mlp", "ET")):
        for i in (0, 2, 4):
            put(gslot(f"DS_GW_{tag}{i}_W"), pack_linear(sd[f"{mlp}.{i}.weight"]))
            put(gslot(f"DS_GW_{tag}{i}_B"), pad_vec(sd[f"{mlp}.{i}.bias"]))
    put(gslot("DS_GW_NP0_H"), pack_linear_f16_split(sd["node_pred_mlp.0.weight"]))
    put(gslot("DS_GW_NP2_H"), pack_linear_f16_split(sd["node_pred_mlp.2.weight"]))
    put(gslot("DS_GW_EX0_H"), pack_linear_f16_split(sd["edge_exist_mlp.0.weight"]))
    put(gslot("DS_GW_ET0_H"), pack_linear_f16_split(sd["edge_type_mlp.0.weight"]))
    put(gslot("DS_GW_EX2_C"), pack_ff4_chain(sd["edge_exist_mlp.2.weight"]))
    put(gslot("DS_GW_ET2_C"), pack_ff4_chain(sd["edge_type_mlp.2.weight"]))
    return torch.cat(chunks), offsets


# ----------------------------------------------------------------------------------------- layout

class Layout:
    """Packed-ragged index tables for one (node_mask) batch structure."""

    def __init__(self, node_mask: torch.Tensor, device):
        nm = node_mask.detach().reshape(node_mask.shape[0], node_mask.shape[1]).to("cpu")
        valid = (nm != 0).numpy()
        B, N = valid.shape
        n_atoms = valid.sum(1).astype(np.int64)
        if n_atoms.max(initial=0) > MAX_ATOMS:
            raise ValueError(f"molecule with {int(n_atoms.max())} atoms exceeds DS_MAX_ATOMS={MAX_ATOMS}")
        node_off = np.zeros(B + 1, np.int64)
        node_off[1:] = np.cumsum(n_atoms)
        pair_cnt = n_atoms * (n_atoms - 1) // 2
        pair_off = np.zeros(B + 1, np.int64)
        pair_off[1:] = np.cumsum(pair_cnt)
        bb, ii = np.nonzero(valid)                                   # row-major → molecule-major, index-ascending
        node_dense = (bb * N + ii).astype(np.int32)
        node_mol = bb.astype(np.int32)
        pa, pb, pm = [], [], []
        tri_cache = {}
        for m in range(B):
            n = int(n_atoms[m])
            if n < 2:
                continue
            if n not in tri_cache:
                tri_cache[n] = np.triu_indices(n, 1)                 # (a asc, b asc): p = a(2n-a-1)/2 + (b-a-1)
            a, b = tri_cache[n]
            pa.append(a + node_off[m]); pb.append(b + node_off[m]); pm.append(np.full(a.shape, m))
        cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.zeros(0, np.int32)
        self.B, self.N, self.Nn, self.Pp = B, N, int(node_off[-1]), int(pair_off[-1])
        self.max_n = int(n_atoms.max(initial=0))
        self.n_atoms = n_atoms
        self.valid = valid
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.t = dict(node_off=dev(node_off.astype(np.int32)), pair_off=dev(pair_off.astype(np.int32)),
                      node_dense=dev(node_dense), node_mol=dev(node_mol), pair_a=dev(cat(pa)), pair_b=dev(cat(pb)),
                      pair_mol=dev(cat(pm)),
                      mol_by_size=dev(np.stack([node_off[:-1], n_atoms, pair_off[:-1], pair_cnt], 1)[
                          np.argsort(-n_atoms.astype(np.int64), kind="stable")].astype(np.int32)))
        self.c = DsLayout(B=B, N=N, Nn=self.Nn, Pp=self.Pp, max_n=self.max_n, _pad=0,
                          **{k: v.data_ptr() for k, v in self.t.items()})

    def check_edge_mask(self, edge_mask: torch.Tensor):
        """The path assumes edge_mask = outer(node_mask) minus the diagonal, as every reference caller builds it."""
        ok = getattr(self, "_mask_ok", None)
        # the SAME tensor object, unmodified since it was checked (a B*N*N device->host copy saved).  The cache holds the tensor
        # itself: an address is not an identity - the caching allocator hands a freed mask's address to the next same-shape mask
        if ok is not None and ok[0] is edge_mask and ok[1] == edge_mask._version:
            return
        v = torch.from_numpy(self.valid)
        want = (v.unsqueeze(1) & v.unsqueeze(2)) & ~torch.eye(self.N, dtype=torch.bool).unsqueeze(0)
        got = edge_mask.detach().reshape(self.B, self.N, self.N).cpu() != 0
        if not torch.equal(want, got):
            raise ValueError("edge_mask is not node_mask ⊗ node_mask minus the diagonal; unsupported graph structure")
        self._mask_ok = (edge_mask, edge_mask._version)

    def check_edge_symmetry(self, edge: torch.Tensor, name: str = "edge_x"):
        """The pair layout stores one value per unordered pair: ``edge[b,i,j,:] == edge[b,j,i,:]`` must hold on valid pairs."""
        ok = getattr(self, "_sym_ok", {}).get(name)
        if ok is not None and ok[0] is edge and ok[1] == edge._version:   # the same tensor object, unmodified (a blocking .any() saved)
            return
        e = edge.detach().reshape(self.B, self.N, self.N, -1)
        v = torch.from_numpy(self.valid).to(e.device)
        m = (v.unsqueeze(1) & v.unsqueeze(2)).unsqueeze(-1)
        if bool(((e != e.transpose(1, 2)) & m).any()):
            raise ValueError(f"{name} is not symmetric in its two atom indices; the MI355X path stores edge features per "
                             "unordered pair and does not support directed edge inputs")
        if not hasattr(self, "_sym_ok"):
            self._sym_ok = {}
        self._sym_ok[name] = (edge, edge._version)


class Workspace:
    def __init__(self, L: Layout, device):
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        Nn, Pp, B = max(L.Nn, 1), max(L.Pp, 1), L.B
        self.t = dict(pos=f(Nn, 4), h=f(Nn, 256), e=f(Pp, 64), atom_hids=f(Nn, 768), edge_hids=f(Pp, 192),
                      tfeat=f(B, 24), tmid=f(B, 1024), temb_silu=f(B, 1024), ada=f(B, ADA_COLS), qkv=f(Nn, 768),
                      ye=f(Pp, 64), dist=f(Pp), attn=f(Nn, 256), u=f(Nn, 64), ac=f(Nn, 512),
                      ed=f(Pp, 256), lg=f(Pp, 32), tr=f(Pp, 8),
                      adj=torch.zeros(Pp, dtype=torch.int32, device=device),
                      flags=torch.zeros(64, dtype=torch.int32, device=device))
        self.c = DsWorkspace(**{k: self.t[k].data_ptr() for k in _WS_FIELDS})


# ----------------------------------------------------------------------------------------- engine

class DmtEngine:
    """Owns packed weights on one GPU and runs the C-ABI stages."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], config, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DmtEngine needs a HIP device (torch device type 'cuda'); there is no CPU path")
        self.lib = load_library()
        self.cfg = config
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        flat, offsets = pack_dmt_weights(sd)
        self.wflat = flat.to(self.device)
        self.woff = torch.tensor(offsets, dtype=torch.int64, device=self.device)
        self.w = DsWeights(base=self.wflat.data_ptr(), off_dev=self.woff.data_ptr(),
                           off=(C.c_int64 * W_NUM_SLOTS)(*offsets), edge_th=float(config.model.edge_quan_th),
                           spatial_cut_off=float(config.model.spatial_cut_off))
        from .spec_engine import SpecEngine
        self.spec = SpecEngine(sd, config, self.device, self.lib)
        self._layouts: Dict[tuple, tuple] = {}

    # layout/workspace cache: keyed on the mask's bytes (cheap: B*N bytes) so equal structures share tables
    def layout_for(self, node_mask: torch.Tensor, edge_mask: Optional[torch.Tensor] = None, validate: bool = False):
        last = getattr(self, "_last_layout", None)
        # the same mask TENSOR again (object identity + version; the cache keeps the tensor alive, so its address cannot be
        # re-issued to another mask while it is the key): no device->host copy
        if last is not None and last[0] is node_mask and last[1] == node_mask._version and last[2] in self._layouts:
            hit = self._layouts[last[2]]
            if validate and edge_mask is not None:
                hit[0].check_edge_mask(edge_mask)
            return hit
        key_t = (node_mask.detach().reshape(node_mask.shape[0], -1) != 0).to("cpu")
        key = (tuple(key_t.shape), key_t.numpy().tobytes())
        self._last_layout = (node_mask, node_mask._version, key)
        hit = self._layouts.get(key)
        if hit is None:
            L = Layout(node_mask, self.device)
            if len(self._layouts) >= 4:
                self._layouts.pop(next(iter(self._layouts)))
            hit = (L, Workspace(L, self.device))
            self._layouts[key] = hit
        if validate and edge_mask is not None:
            hit[0].check_edge_mask(edge_mask)
        return hit

    def context_embedding(self, context) -> Optional[torch.Tensor]:
        """cond_lin(SpecFormer(context)) [B,1024] (dmt.py:348-350)."""
        if context is None:
            return None
        return self.spec.encode(context)

    def forward(self, L: Layout, ws: Workspace, xh, edge_x, noise_level, cond_x, cond_edge_x, ctx_emb,
                out_xh=None, out_edge=None):
        dev = self.device
        xh, edge_x, noise_level = _f32c(xh, dev), _f32c(edge_x, dev), _f32c(noise_level, dev)
        cond_x = None if cond_x is None else _f32c(cond_x, dev)
        cond_edge_x = None if cond_edge_x is None else _f32c(cond_edge_x, dev)
        ctx_emb = None if ctx_emb is None else _f32c(ctx_emb, dev)
        if out_xh is None:
            out_xh = torch.empty(L.B, L.N, 9, dtype=torch.float32, device=dev)
        if out_edge is None:
            out_edge = torch.empty(L.B, L.N, L.N, 2, dtype=torch.float32, device=dev)
        st = self.lib.ds_forward(C.byref(self.w), C.byref(L.c), C.byref(ws.c), _ptr(xh), _ptr(edge_x), _ptr(cond_x),
                                 _ptr(cond_edge_x), _ptr(noise_level), _ptr(ctx_emb), _ptr(out_xh), _ptr(out_edge),
                                 _stream())
        _check(st, "ds_forward")
        return out_xh, out_edge

    # stage-level calls for the parity tests
    def stage_time(self, L, ws, noise_level, ctx_emb):
        _check(self.lib.ds_stage_time(C.byref(self.w), C.byref(L.c), C.byref(ws.c), _ptr(noise_level), _ptr(ctx_emb),
                                      _stream()), "ds_stage_time")

    def stage_init(self, L, ws, xh, edge_x, cond_x, cond_edge_x):
        _check(self.lib.ds_stage_init(C.byref(self.w), C.byref(L.c), C.byref(ws.c), _ptr(xh), _ptr(edge_x),
                                      _ptr(cond_x), _ptr(cond_edge_x), _stream()), "ds_stage_init")

    def stage_block(self, L, ws, blk, last=False):
        _check(self.lib.ds_stage_block(C.byref(self.w), C.byref(L.c), C.byref(ws.c), C.c_int(blk), C.c_int(int(last)),
                                       _stream()), "ds_stage_block")

    def stage_readout(self, L, ws, out_xh, out_edge):
        _check(self.lib.ds_stage_readout(C.byref(self.w), C.byref(L.c), C.byref(ws.c), _ptr(out_xh), _ptr(out_edge),
                                         _stream()), "ds_stage_readout")

    def sampler_step(self, L, c_x, c_pred, sigma, temperature, x, edge_x, pred, edge_pred, raw_pos, raw_feat, raw_edge,
                     x_mean, edge_mean):
        st = self.lib.ds_sampler_step(C.byref(L.c), C.c_float(c_x), C.c_float(c_pred), C.c_float(sigma),
                                      C.c_float(temperature), _ptr(x), _ptr(edge_x), _ptr(pred), _ptr(edge_pred),
                                      _ptr(raw_pos), _ptr(raw_feat), _ptr(raw_edge), _ptr(x_mean), _ptr(edge_mean),
                                      _stream())
        _check(st, "ds_sampler_step")

    def initial_noise_philox(self, L, seed: int, mol_id: torch.Tensor):
        """z_T, edge_z_T of sampling.py:442-447 from the per-molecule Philox streams (``ds_initial_noise``)."""
        dev = self.device
        x = torch.empty(L.B, L.N, 9, dtype=torch.float32, device=dev)
        edge_x = torch.empty(L.B, L.N, L.N, 2, dtype=torch.float32, device=dev)
        _check(self.lib.ds_initial_noise(C.byref(L.c), C.c_uint64(seed), _ptr(mol_id), _ptr(x), _ptr(edge_x), _stream()),
               "ds_initial_noise")
        return x, edge_x

    def sampler_step_philox(self, L, c_x, c_pred, sigma, temperature, seed: int, step: int, mol_id, x, edge_x, pred, edge_pred,
                            x_mean, edge_mean):
        st = self.lib.ds_sampler_step_philox(C.byref(L.c), C.c_float(c_x), C.c_float(c_pred), C.c_float(sigma),
                                             C.c_float(temperature), C.c_uint64(seed), C.c_int32(step), _ptr(mol_id), _ptr(x),
                                             _ptr(edge_x), _ptr(pred), _ptr(edge_pred), _ptr(x_mean), _ptr(edge_mean), _stream())
        _check(st, "ds_sampler_step_philox")

    def step_begin(self, table, n_steps: int, step_dev, B: int, noise_level):
        _check(self.lib.ds_step_begin(_ptr(table), C.c_int32(n_steps), _ptr(step_dev), C.c_int32(B), _ptr(noise_level), _stream()),
               "ds_step_begin")

    def sampler_step_philox_dev(self, L, table, step_dev, temperature, seed: int, mol_id, x, edge_x, pred, edge_pred, x_mean,
                                edge_mean):
        st = self.lib.ds_sampler_step_philox_dev(C.byref(L.c), _ptr(table), _ptr(step_dev), C.c_float(temperature),
                                                 C.c_uint64(seed), _ptr(mol_id), _ptr(x), _ptr(edge_x), _ptr(pred),
                                                 _ptr(edge_pred), _ptr(x_mean), _ptr(edge_mean), _stream())
        _check(st, "ds_sampler_step_philox_dev")

    def check_stability(self, L, pos, atom_type, want_orders: bool = True):
        """``ds_check_stability`` on the tensors ``post_process`` left on the GPU: (mol_stable [B] bool, nr_stable [B],
        n_atoms [B], bond_order [B,N,N] int64 or None)."""
        dev = self.device
        pos = _f32c(pos, dev)
        at = atom_type.detach().to(device=dev, dtype=torch.int32).contiguous()
        order = torch.empty(L.B, L.N, L.N, dtype=torch.int32, device=dev) if want_orders else None
        nr = torch.empty(L.B, dtype=torch.int32, device=dev)
        ok = torch.empty(L.B, dtype=torch.int32, device=dev)
        _check(self.lib.ds_check_stability(C.byref(L.c), _ptr(pos), _ptr(at), _ptr(order), _ptr(nr), _ptr(ok), _stream()),
               "ds_check_stability")
        n_atoms = torch.as_tensor(L.n_atoms, device=dev)
        return ok.bool(), nr.long(), n_atoms, (order.long() if want_orders else None)

    def post_process(self, L, xh, edge_x):
        dev = self.device
        pos = torch.empty(L.B, L.N, 3, dtype=torch.float32, device=dev)
        atom = torch.empty(L.B, L.N, dtype=torch.int32, device=dev)
        fc = torch.empty(L.B, L.N, dtype=torch.int32, device=dev)
        et = torch.empty(L.B, L.N, L.N, dtype=torch.float32, device=dev)
        st = self.lib.ds_post_process(C.byref(L.c), _ptr(_f32c(xh, dev)), _ptr(_f32c(edge_x, dev)), _ptr(pos), _ptr(atom),
                                      _ptr(fc), _ptr(et), _stream())
        _check(st, "ds_post_process")
        return pos, atom, fc, et


def gemm(lib, A, lda, Wp, bias, Cout, ldc, M, K, N, act=0, R=None, ldr=0, r_grp_rows=0, col_scale=None, col_shift=None,
         a_silu=0, a_grp=(0, 0), c_grp=(0, 0)):
    """Raw ds_gemm call; A / Wp / C etc. are tensors or integer device addresses."""
    addr = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
    args = DsGemmArgs(A=addr(A), lda=lda, a_grp_rows=a_grp[0], _p0=0, a_grp_stride=a_grp[1], Wp=addr(Wp), bias=addr(bias),
                      C=addr(Cout), ldc=ldc, c_grp_rows=c_grp[0], _p1=0, c_grp_stride=c_grp[1], M=M, K=K, N=N, act=act,
                      R=addr(R), ldr=ldr, r_grp_rows=r_grp_rows, a_silu=a_silu, col_scale=addr(col_scale),
                      col_shift=addr(col_shift))
    _check(lib.ds_gemm(C.byref(args), _stream()), "ds_gemm")
