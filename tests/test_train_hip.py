"""GPU parity of the TRAINING path (row N1, stage A): every forward / backward kernel of csrc/ds_train.hip against torch autograd of
the same operation, then the whole DMT graph (loss + gradients of every parameter) against the training oracle, which golden G13 pins
to the reference's own ``get_sde_graph_loss_fn`` (losses.py:286-396).  Tolerances: loss rtol 1e-5, gradients rtol 1e-4 (plus a floor of
1e-7 of the total gradient norm for gradients that are sums of cancelling terms)."""
import ctypes as C
import json
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import train as otrain
from oracle import dmt as odmt
from tests.golden import cases
from tests.helpers import procedural_state_dict

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(a, b, tol, what):
    e = relerr(a, b)
    assert e <= tol, f"{what}: max |diff| / max |ref| = {e:.3e} (tol {tol:g})"
    return e


@pytest.fixture(scope="module")
def ops(gpu_device):
    from diffspectra_amd import train_engine as T
    return T, T.Ops(gpu_device)


def layout(gpu_device, n_atoms):
    from diffspectra_amd import filler, train_engine as T
    node_mask, _ = filler.masks_from_n_atoms(n_atoms)
    return T.TrainLayout(node_mask, gpu_device)


def mol_tables(n_atoms):
    """CPU index tables of the packed layout: node -> molecule, pair -> (a, b, molecule), directed edge lists."""
    node_mol, pa, pb, pm = [], [], [], []
    off = 0
    for m, n in enumerate(n_atoms):
        node_mol += [m] * n
        for a in range(n):
            for b in range(a + 1, n):
                pa.append(off + a); pb.append(off + b); pm.append(m)
        off += n
    t = lambda x: torch.tensor(x, dtype=torch.long)
    return t(node_mol), t(pa), t(pb), t(pm)


N_ATOMS = [3, 1, 7, 12, 29, 2]

_KEEP = []


def _dev(t, d):
    """``t`` on the device, kept alive: a temporary passed as ``E._ptr(t.to(d))`` is freed as soon as ``_ptr`` returns, and the caching
    allocator may hand its block to the NEXT temporary of the same argument list before the kernel has read it (seen in round 5, when an
    earlier test had left large cached blocks behind: dst_zbuild_fwd read `ed` where `ac` should have been)."""
    x = t.to(d)
    _KEEP.append(x)
    if len(_KEEP) > 64:
        del _KEEP[:32]
    return x


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,ta,tb,bias,acc", [(70, 96, 40, False, True, True, False), (64, 64, 64, False, False, False, True),
                                                    (252, 64, 5000, True, False, False, False), (1, 32, 3000, True, False, False, True),
                                                    (300, 3, 256, False, True, False, False), (17, 1024, 17, False, True, True, False),
                                                    (256, 256, 70000, True, False, False, False), (5, 7, 0, False, False, True, False)])
def test_train_gemm(ops, gpu_device, M, N, K, ta, tb, bias, acc):
    T, o = ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    b = torch.randn(N, generator=g) if bias else None
    C0 = torch.randn(M, N + 5, generator=g)                                   # a column slice of a wider buffer (ldc > N)
    ref = (A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double()
    if bias:
        ref = ref + b.double()
    if acc:
        ref = ref + C0[:, 2:2 + N].double()
    d = gpu_device
    Ad, Bd, Cd = A.to(d), Bm.to(d), C0.to(d)
    o.gemm(T.mv(Ad), T.mv(Bd), T.mv(Cd, 2, 2 + N), ta, tb, bias=None if b is None else b.to(d), acc=acc)
    out = Cd.cpu()
    scale = float(ref.abs().max()) + 1e-6
    assert float((out[:, 2:2 + N].double() - ref).abs().max()) <= 3e-6 * scale * max(1.0, K ** 0.5 / 8), (M, N, K)
    assert torch.equal(out[:, :2], C0[:, :2]) and torch.equal(out[:, 2 + N:], C0[:, 2 + N:])      # nothing outside the slice is touched



def _bf(t):
    return t.bfloat16().double()


@pytest.mark.parametrize("M,N,K,ta,tb", [
    (300, 256, 128, False, True),        # Linear forward: A k-fast, B k-fast
    (4100, 64, 256, False, False),       # input gradient: B row-fast (transposed in registers), BN = 64
    (256, 512, 9000, True, False),       # weight gradient: both row-fast, split over K with the in-kernel reduction
    (252, 128, 70000, True, False),      # ragged M, long K
    (130, 132, 40, False, True),         # ragged everything, K not a multiple of the k-step
    (33, 768, 256, False, True), (1000, 252, 64, False, True), (96, 68, 4096, True, False)])
def test_train_gemm_bf16_vector_kernel(ops, gpu_device, M, N, K, ta, tb):
    """The config-5 GEMM (k_tr_gemm_bf16: vec4 loads, bf16 rounding at LDS staging, in-register transposes, in-kernel split-K finish)
    against the fp64 product of the bf16-rounded operands, in every operand order the training graph uses; bias, accumulate, a fused
    row sum on the weight-gradient shapes; and the scalar fallback for the same product must agree (same arithmetic, other kernel)."""
    T, _ = ops
    o = T.Ops(gpu_device)
    o.bf16 = True
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    b = torch.randn(N, generator=g)
    d = gpu_device
    ref = _bf(A.t() if ta else A) @ _bf(Bm.t() if tb else Bm)
    tol = 2e-6 * float(ref.abs().max()) * max(1.0, K ** 0.5 / 4)
    Cd = torch.full((M, N + 4), 5.0, device=d)
    use_bias = not ta
    rs = torch.full((M,), 2.0, device=d) if ta else None
    o.gemm(T.mv(A.to(d)), T.mv(Bm.to(d)), T.mv(Cd, 4, 4 + N), ta, tb, bias=b.to(d) if use_bias else None, rowsum=rs)
    want = ref + (b.double() if use_bias else 0.0)
    assert float((Cd[:, 4:].cpu().double() - want).abs().max()) <= tol, (M, N, K)
    assert torch.equal(Cd[:, :4].cpu(), torch.full((M, 4), 5.0))
    if ta:
        rref = _bf(A.t()).sum(1)
        assert float((rs.cpu().double() - rref).abs().max()) <= 2e-6 * float(rref.abs().max()) * max(1.0, K ** 0.5 / 4) + 1e-5
    first = Cd.clone()
    o.gemm(T.mv(A.to(d)), T.mv(Bm.to(d)), T.mv(Cd, 4, 4 + N), ta, tb, bias=b.to(d) if use_bias else None, acc=True, rowsum=rs)
    assert float((Cd[:, 4:].cpu().double() - (first[:, 4:].cpu().double() + want)).abs().max()) <= 2 * tol
    # an operand off the 16-byte grid takes the scalar kernel: same bf16 arithmetic, results within fp32 summation-order noise
    A1 = torch.zeros(A.numel() + 1, device=d)
    A1[1:] = A.reshape(-1).to(d)
    Au = A1[1:].view(A.shape)
    assert Au.data_ptr() % 16 != 0
    C2 = torch.zeros(M, N, device=d)
    o.gemm(T.MV(A1, A.shape[0], A.shape[1], A.shape[1], 1), T.mv(Bm.to(d)), T.mv(C2), ta, tb)
    assert float((C2.cpu().double() - ref).abs().max()) <= tol


@pytest.mark.parametrize("bf16", [True, False])
def test_train_gemm_fused_epilogues(ops, gpu_device, bf16):
    """The fused epilogues of dst_gemm against torch: activation with a second output (pre-activation kept for the backward), Philox
    dropout at the producing GEMM (mask = oracle.philox.dropout_keep at element m * N + n), activation derivative x dropout mask on an
    input-gradient product, tanh epilogue; in both arithmetic modes (the fp32 mode runs the scalar kernel with the same epilogue)."""
    from oracle import philox
    T, _ = ops
    o = T.Ops(gpu_device)
    o.bf16 = bf16
    d = gpu_device
    g = torch.Generator().manual_seed(5)
    r = _bf if bf16 else (lambda t: t.double())
    M, K, N = 777, 64, 128
    X, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.3, torch.randn(N, generator=g)
    seed, stream, p = (1 << 40) + 77, 9, 0.1
    keep = torch.from_numpy(philox.dropout_keep(seed, stream, M * N, p).astype(np.float64)).reshape(M, N) * float(philox.dropout_scale(p))
    pre = r(X) @ r(W).t() + b.double()
    tol = 3e-6 * float(pre.abs().max()) * (4.0 if bf16 else 1.0)
    f1, s1 = torch.zeros(M, N, device=d), torch.zeros(M, N, device=d)
    o.gemm(T.mv(X.to(d)), T.mv(W.to(d)), T.mv(f1), False, True, bias=b.to(d), act=T.SILU, out2=T.mv(s1), drop=(p, seed, stream, N))
    assert float((f1.cpu().double() - pre).abs().max()) <= tol
    assert float((s1.cpu().double() - F.silu(pre) * keep).abs().max()) <= 2 * tol
    assert abs(float((s1 == 0).float().mean()) - p) < 0.01
    t1 = torch.zeros(M, N, device=d)
    o.gemm(T.mv(X.to(d)), T.mv(W.to(d)), T.mv(t1), False, True, act=T.TANH)
    assert float((t1.cpu().double() - torch.tanh(pre - b.double())).abs().max()) <= tol
    y = torch.zeros(M, N, device=d)
    o.gemm(T.mv(X.to(d)), T.mv(W.to(d)), T.mv(y), False, True, bias=b.to(d), drop=(p, seed, stream + 1, N))
    keep2 = torch.from_numpy(philox.dropout_keep(seed, stream + 1, M * N, p).astype(np.float64)).reshape(M, N) * float(philox.dropout_scale(p))
    assert float((y.cpu().double() - pre * keep2).abs().max()) <= tol
    # backward of s1 = drop(silu(f1)): ds -> df1 = (dy W2) * mask * silu'(f1), as one input-gradient product
    W2, dy = torch.randn(96, N, generator=g) * 0.2, torch.randn(M, 96, generator=g)
    df1 = torch.zeros(M, N, device=d)
    o.gemm(T.mv(dy.to(d)), T.mv(W2.to(d)), T.mv(df1), False, False, dact=T.SILU, ref=T.mv(f1), drop=(p, seed, stream, N))
    f1r = f1.cpu().double().requires_grad_(True)
    (F.silu(f1r) * keep * (r(dy) @ r(W2))).sum().backward()
    assert float((df1.cpu().double() - f1r.grad).abs().max()) <= 4e-6 * float(f1r.grad.abs().max()) * (4.0 if bf16 else 1.0)

@pytest.mark.parametrize("M,N,K", [(64, 64, 300), (252, 127, 9000), (6, 1024, 70), (130, 256, 40000)])
def test_train_gemm_fused_rowsum(ops, gpu_device, M, N, K):
    """dW = dY^T X with db = column sums of dY in the same launch (a virtual all-ones column of B), incl. the split-K path and accumulate."""
    T, o = ops
    g = torch.Generator().manual_seed(M + N + K)
    dY, X = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    d = gpu_device
    dW, db = torch.full((M, N), 7.0, device=d), torch.full((M,), -3.0, device=d)
    o.lin_bwd_w(T.mv(dY.to(d)), T.mv(X.to(d)), T.mv(dW), db)
    refW, refb = dY.double().t() @ X.double(), dY.double().sum(0)
    tol = 3e-6 * max(1.0, K ** 0.5 / 8)
    assert float((dW.cpu().double() - refW).abs().max()) <= tol * float(refW.abs().max())
    assert float((db.cpu().double() - refb).abs().max()) <= tol * float(refb.abs().max()) + 1e-5
    o.lin_bwd_w(T.mv(dY.to(d)), T.mv(X.to(d)), T.mv(dW), db, acc=True)
    assert float((dW.cpu().double() - 2 * refW).abs().max()) <= 2 * tol * float(refW.abs().max())
    assert float((db.cpu().double() - 2 * refb).abs().max()) <= 2 * tol * float(refb.abs().max()) + 2e-5


@pytest.mark.parametrize("M,N,K,tb", [(5000, 256, 3, False), (4100, 32, 1, True), (2049, 3, 8, True)])
def test_train_gemm_skinny_k(ops, gpu_device, M, N, K, tb):
    """K <= 8 in the bf16 mode takes k_tr_gemm_skinny (fp32 FMAs, no bf16 rounding): plain, accumulate, and the activation-derivative
    epilogue of coord_mlp.2's input gradient (K = 3 -> 256 columns)."""
    T, _ = ops
    o = T.Ops(gpu_device)
    o.bf16 = True
    d = gpu_device
    g = torch.Generator().manual_seed(M + N + K)
    A, Bm, b = torch.randn(M, K, generator=g), torch.randn((N, K) if tb else (K, N), generator=g), torch.randn(N, generator=g)
    ref = A.double() @ (Bm.t() if tb else Bm).double()
    tol = 3e-6 * (float(ref.abs().max()) + 1.0)
    Cd = torch.full((M, N + 3), 2.0, device=d)
    o.gemm(T.mv(A.to(d)), T.mv(Bm.to(d)), T.mv(Cd, 1, 1 + N), False, tb, bias=b.to(d))
    assert float((Cd[:, 1:1 + N].cpu().double() - ref - b.double()).abs().max()) <= tol
    assert torch.equal(Cd[:, :1].cpu(), torch.full((M, 1), 2.0)) and torch.equal(Cd[:, 1 + N:].cpu(), torch.full((M, 2), 2.0))
    o.gemm(T.mv(A.to(d)), T.mv(Bm.to(d)), T.mv(Cd, 1, 1 + N), False, tb, acc=True)
    assert float((Cd[:, 1:1 + N].cpu().double() - 2 * ref - b.double()).abs().max()) <= 2 * tol
    pre = torch.randn(M, N, generator=g)
    out = torch.zeros(M, N, device=d)
    o.gemm(T.mv(A.to(d)), T.mv(Bm.to(d)), T.mv(out), False, tb, dact=T.SILU, ref=T.mv(pre.to(d)))
    sg = torch.sigmoid(pre.double())
    assert float((out.cpu().double() - ref * (sg * (1 + pre.double() * (1 - sg)))).abs().max()) <= 2 * tol


def test_colsum_and_acts(ops, gpu_device):
    T, o = ops
    d = gpu_device
    X = torch.randn(3000, 70)
    out = torch.zeros(64, device=d)
    o.colsum(T.mv(X.to(d), 3, 67), out)
    check(out, X[:, 3:67].double().sum(0), 2e-6, "colsum")
    o.colsum(T.mv(X.to(d), 3, 67), out, acc=True)
    check(out, 2 * X[:, 3:67].double().sum(0), 2e-6, "colsum accumulate")
    x = torch.randn(5000) * 3
    for kind, fn in ((T.SILU, F.silu), (T.GELU, F.gelu), (T.TANH, torch.tanh)):
        xr = x.clone().double().requires_grad_(True)
        yr = fn(xr)
        dy = torch.randn(5000)
        yr.backward(dy.double())
        xd, yd, dxd = x.to(d), torch.empty(5000, device=d), torch.empty(5000, device=d)
        o.act_fwd(xd, yd, kind)
        o.act_bwd(dy.to(d), yd if kind == T.TANH else xd, dxd, kind)
        check(yd, yr, 2e-6, f"act fwd {kind}")
        check(dxd, xr.grad, 3e-6, f"act bwd {kind}")


# ------------------------------------------------------------------------------------------------ LN + modulate, gated residual
@pytest.mark.parametrize("Cc,kind", [(256, "node"), (64, "pair"), (256, "directed")])
def test_lnmod_and_gate(ops, gpu_device, Cc, kind):
    T, o = ops
    d = gpu_device
    TL = layout(d, N_ATOMS)
    node_mol, pa, pb, pm = mol_tables(N_ATOMS)
    seg, mul, row_mol = {"node": (TL.node_off, 1, node_mol), "pair": (TL.pair_off, 1, pm), "directed": (TL.pair_off, 2, pm.repeat_interleave(2))}[kind]
    R, B = row_mol.numel(), len(N_ATOMS)
    g = torch.Generator().manual_seed(Cc + R)
    x = (torch.randn(R, Cc, generator=g) * 2 + 0.3).double().requires_grad_(True)
    ada = (torch.randn(B, T.ADA, generator=g) * 0.5).double().requires_grad_(True)
    sh, sc, gt = 100, 100 + Cc, 100 + 2 * Cc
    y = F.layer_norm(x, (Cc,), None, None, 1e-6) * (1 + ada[row_mol, sc:sc + Cc]) + ada[row_mol, sh:sh + Cc]
    dy = torch.randn(R, Cc, generator=g)
    y.backward(dy.double())
    xd, adad = x.detach().float().to(d), ada.detach().float().to(d)
    yd, st, dxd, d_ada = torch.empty(R, Cc, device=d), torch.empty(R, 2, device=d), torch.zeros(R, Cc, device=d), torch.zeros(B, T.ADA, device=d)
    o.lnmod_fwd(xd, Cc, seg, mul, B, adad, sh, sc, yd, st)
    o.lnmod_bwd(dy.to(d), xd, st, Cc, seg, mul, B, adad, d_ada, sh, sc, dxd, False)
    check(yd, y, 3e-6, "lnmod fwd")
    check(dxd, x.grad, 2e-5, "lnmod dx")
    check(d_ada[:, sh:sh + Cc], ada.grad[:, sh:sh + Cc], 1e-5, "lnmod dshift")
    check(d_ada[:, sc:sc + Cc], ada.grad[:, sc:sc + Cc], 1e-5, "lnmod dscale")
    o.lnmod_bwd(dy.to(d), xd, st, Cc, seg, mul, B, adad, d_ada, sh, sc, dxd, True)
    check(dxd, 2 * x.grad, 2e-5, "lnmod dx accumulate")
    # gated residual
    r = torch.randn(R, Cc, generator=g).double().requires_grad_(True)
    z = torch.randn(R, Cc, generator=g).double().requires_grad_(True)
    ada2 = ada.detach().clone().requires_grad_(True)
    out = r + ada2[row_mol, gt:gt + Cc] * z
    out.backward(dy.double())
    od, drd, dzd = torch.empty(R, Cc, device=d), torch.empty(R, Cc, device=d), torch.empty(R, Cc, device=d)
    o.gate_add_fwd(r.detach().float().to(d), z.detach().float().to(d), Cc, seg, mul, B, adad, gt, od)
    o.gate_add_bwd(dy.to(d), z.detach().float().to(d), Cc, seg, mul, B, adad, d_ada, gt, drd, False, dzd)
    check(od, out, 2e-6, "gate_add fwd")
    check(drd, r.grad, 1e-6, "gate_add dr")
    check(dzd, z.grad, 2e-6, "gate_add dz")
    check(d_ada[:, gt:gt + Cc], ada2.grad[:, gt:gt + Cc], 1e-5, "gate_add dgate")


# ------------------------------------------------------------------------------------------------ geometry
def test_geom(ops, gpu_device):
    T, o = ops
    d = gpu_device
    TL = layout(d, N_ATOMS)
    node_mol, pa, pb, pm = mol_tables(N_ATOMS)
    Nn, P, B = node_mol.numel(), pa.numel(), len(N_ATOMS)
    g = torch.Generator().manual_seed(5)
    pos = (torch.randn(Nn, 3, generator=g) * 1.2).double().requires_grad_(True)
    ada = (torch.randn(B, T.ADA, generator=g) * 0.3).double().requires_grad_(True)
    means = (torch.rand(1, 63, generator=g) * 3).double().requires_grad_(True)
    stds = ((torch.rand(1, 63, generator=g) * 3 + 0.05) * torch.where(torch.rand(1, 63, generator=g) > 0.8, -1.0, 1.0)).double().requires_grad_(True)
    off = 37
    d2 = ((pos[pa] - pos[pb]) ** 2).sum(1, keepdim=True)
    x = d2 * (ada[pm, off:off + 1] + 1) + ada[pm, off + 1:off + 2]
    sd = stds.view(-1).abs() + 1e-5
    feat = torch.cat([x, torch.exp(-0.5 * ((x - means.view(-1)) / sd) ** 2) / ((2 * 3.14159) ** 0.5 * sd)], dim=1)
    g1, g2 = torch.randn(P, 70, generator=g), torch.randn(P, 64, generator=g)
    (feat * g1[:, 3:67].double()).sum().backward(retain_graph=True)
    (feat * g2.double()).sum().backward()
    p = {"x.means.weight": means.detach().float().to(d), "x.stds.weight": stds.detach().float().to(d)}
    graph = T.DmtTrainGraph.__new__(T.DmtTrainGraph)
    graph.p, graph.lib, graph.dev, graph.ops = p, o.lib, d, o
    X, xs, d2s = torch.zeros(P, 70, device=d), torch.empty(P, device=d), torch.empty(P, device=d)
    posd, adad = pos.detach().float().to(d), ada.detach().float().to(d)
    graph._geom_fwd(TL, posd, adad, off, "x.", X, 70, 3, xs, d2s)
    check(X[:, 3:67], feat, 3e-6, "geom features")
    assert float(X[:, :3].abs().max()) == 0 and float(X[:, 67:].abs().max()) == 0
    d_ada, dms, dd2, dpos = torch.zeros(B, T.ADA, device=d), torch.empty(B, 128, device=d), torch.empty(P, device=d), torch.zeros(Nn, 3, device=d)
    g1d = g1[:, 3:67].contiguous().to(d)
    graph._geom_bwd(TL, posd, adad, d_ada, off, "x.", xs, d2s, g1d, g2.to(d), dms, dd2, dpos)
    check(dpos, pos.grad, 2e-5, "geom dpos")
    check(d_ada[:, off:off + 2], ada.grad[:, off:off + 2], 2e-5, "geom d(scale, shift)")
    check(dms[:, 1:64].sum(0), means.grad.view(-1), 2e-5, "geom dmeans")
    check(dms[:, 65:128].sum(0), stds.grad.view(-1), 2e-5, "geom dstds")


# ------------------------------------------------------------------------------------------------ attention
def torch_attention(qkv, te0, te1, adj, pa, pb, Nn):
    """layers.py:131-186 on explicit directed edges: row 2p = source a -> target b, row 2p+1 = source b -> target a."""
    src = torch.stack([pa, pb], 1).reshape(-1)
    tgt = torch.stack([pb, pa], 1).reshape(-1)
    q, k, v = qkv[:, 0:252].reshape(-1, 14, 18), qkv[:, 256:508].reshape(-1, 14, 18), qkv[:, 512:768].reshape(-1, 16, 16)
    e0 = te0[:, :252].reshape(-1, 14, 18).repeat_interleave(2, 0)
    e1 = te1.reshape(-1, 16, 16).repeat_interleave(2, 0)
    alpha = (q[tgt] * k[src] * e0).sum(-1) / 4.0
    bits = adj.repeat_interleave(2)
    extra = torch.stack([(bits & 1).double(), ((bits >> 1) & 1).double()], 1)
    extra = torch.where(extra == 0, torch.full_like(extra, -1e10), extra)
    alpha = torch.cat([extra, alpha], 1)
    idx = tgt.view(-1, 1).expand_as(alpha)
    mx = torch.full((Nn, 16), float("-inf"), dtype=alpha.dtype).scatter_reduce(0, idx, alpha, "amax", include_self=True)
    ex = (alpha - mx[tgt]).exp()
    den = torch.zeros(Nn, 16, dtype=alpha.dtype).index_add_(0, tgt, ex) + 1e-16
    al = ex / den[tgt]
    out = torch.zeros(Nn, 16, 16, dtype=alpha.dtype).index_add_(0, tgt, v[src] * e1 * al.unsqueeze(-1))
    return out.reshape(Nn, 256), al


def test_attention(ops, gpu_device):
    T, o = ops
    d = gpu_device
    TL = layout(d, N_ATOMS)
    node_mol, pa, pb, pm = mol_tables(N_ATOMS)
    Nn, P = node_mol.numel(), pa.numel()
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(Nn, 768, generator=g)
    qkv[:, 252:256] = 0
    qkv[:, 508:512] = 0
    te0 = torch.tanh(torch.randn(P, 256, generator=g))
    te0[:, 252:] = 0
    te1 = torch.tanh(torch.randn(P, 256, generator=g))
    adj = torch.randint(0, 4, (P,), generator=g).to(torch.int32)
    qr, e0r, e1r = (t.double().requires_grad_(True) for t in (qkv, te0, te1))
    out, al = torch_attention(qr, e0r, e1r, adj.long(), pa, pb, Nn)
    dout = torch.randn(Nn, 256, generator=g)
    out.backward(dout.double())
    outd, ald = torch.empty(Nn, 256, device=d), torch.empty(2 * P, 16, device=d)
    qd, e0d, e1d, adjd = qkv.to(d), te0.to(d), te1.to(d), adj.to(d)
    from diffspectra_amd import engine as E
    E._check(o.lib.dst_attn_fwd(C.byref(TL.c), E._ptr(qd), E._ptr(e0d), E._ptr(e1d), C.c_int64(256), E._ptr(adjd), E._ptr(outd), E._ptr(ald), E._stream()), "attn_fwd")
    check(ald, al, 3e-6, "attention alpha")
    check(outd, out, 3e-6, "attention out")
    dq, de0, de1 = torch.empty(Nn, 768, device=d), torch.empty(P, 256, device=d), torch.empty(P, 256, device=d)
    E._check(o.lib.dst_attn_bwd(C.byref(TL.c), E._ptr(qd), E._ptr(e0d), E._ptr(e1d), C.c_int64(256), E._ptr(ald), E._ptr(_dev(dout, d)), E._ptr(dq), E._ptr(de0), E._ptr(de1),
                                C.c_int32(0), None, C.c_int64(0), E._stream()), "attn_bwd")
    check(dq, qr.grad, 2e-5, "attention dqkv")
    check(de0, e0r.grad, 2e-5, "attention dte0")
    check(de1, e1r.grad, 2e-5, "attention dte1")
    # with scratch the same backward runs as two launches (d logit through global memory, four workgroups per molecule for the gradients):
    # same arithmetic, same summation order -> the same bits; te_is_tanh multiplies the pair gradients by 1 - te^2
    dq2, de02, de12 = torch.zeros_like(dq), torch.zeros_like(de0), torch.zeros_like(de1)
    scr = torch.empty(32 * P + 64, device=d)
    E._check(o.lib.dst_attn_bwd(C.byref(TL.c), E._ptr(qd), E._ptr(e0d), E._ptr(e1d), C.c_int64(256), E._ptr(ald), E._ptr(_dev(dout, d)), E._ptr(dq2), E._ptr(de02),
                                E._ptr(de12), C.c_int32(0), E._ptr(scr), C.c_int64(scr.numel()), E._stream()), "attn_bwd")
    assert torch.equal(dq2, dq) and torch.equal(de02, de0) and torch.equal(de12, de1)
    E._check(o.lib.dst_attn_bwd(C.byref(TL.c), E._ptr(qd), E._ptr(e0d), E._ptr(e1d), C.c_int64(256), E._ptr(ald), E._ptr(_dev(dout, d)), E._ptr(dq2), E._ptr(de02),
                                E._ptr(de12), C.c_int32(1), E._ptr(scr), C.c_int64(scr.numel()), E._stream()), "attn_bwd")
    check(de02, de0.double().cpu() * (1 - te0.double() ** 2), 1e-6, "attention dte0 in front of the tanh")
    check(de12, de1.double().cpu() * (1 - te1.double() ** 2), 1e-6, "attention dte1 in front of the tanh")


# ------------------------------------------------------------------------------------------------ gathers + coordinates
def test_pair_sum_zbuild_coord(ops, gpu_device):
    T, o = ops
    from diffspectra_amd import engine as E
    d = gpu_device
    TL = layout(d, N_ATOMS)
    node_mol, pa, pb, pm = mol_tables(N_ATOMS)
    Nn, P, B = node_mol.numel(), pa.numel(), len(N_ATOMS)
    g = torch.Generator().manual_seed(21)
    s = E._stream
    # pair sum
    u = torch.randn(Nn, 64, generator=g).double().requires_grad_(True)
    bias = torch.randn(64, generator=g)
    ps = u[pa] + u[pb] + bias.double()
    dps = torch.randn(P, 64, generator=g)
    ps.backward(dps.double())
    psd, dud = torch.empty(P, 64, device=d), torch.empty(Nn, 64, device=d)
    E._check(o.lib.dst_pair_sum_fwd(C.byref(TL.c), E._ptr(_dev(u.detach().float(), d)), C.c_int32(64), E._ptr(_dev(bias, d)), E._ptr(psd), s()), "pair_sum_fwd")
    E._check(o.lib.dst_pair_sum_bwd(C.byref(TL.c), E._ptr(_dev(dps, d)), C.c_int32(64), E._ptr(dud), C.c_int32(0), s()), "pair_sum_bwd")
    check(psd, ps, 1e-6, "pair_sum fwd")
    check(dud, u.grad, 3e-6, "pair_sum bwd")
    # zbuild: directed edge 2p = (row a, col b), 2p+1 = (row b, col a)
    row = torch.stack([pa, pb], 1).reshape(-1)
    col = torch.stack([pb, pa], 1).reshape(-1)
    ac = torch.randn(Nn, 512, generator=g).double().requires_grad_(True)
    ed = torch.randn(P, 256, generator=g).double().requires_grad_(True)
    z = ac[row, :256] + ac[col, 256:] + ed.repeat_interleave(2, 0)
    dz = torch.randn(2 * P, 256, generator=g)
    z.backward(dz.double())
    zd, dacd, dedd = torch.empty(2 * P, 256, device=d), torch.empty(Nn, 512, device=d), torch.empty(P, 256, device=d)
    E._check(o.lib.dst_zbuild_fwd(C.byref(TL.c), E._ptr(_dev(ac.detach().float(), d)), E._ptr(_dev(ed.detach().float(), d)), E._ptr(zd), s()), "zbuild_fwd")
    E._check(o.lib.dst_zbuild_bwd(C.byref(TL.c), E._ptr(_dev(dz, d)), E._ptr(dacd), E._ptr(dedd), s()), "zbuild_bwd")
    check(zd, z, 1e-6, "zbuild fwd")
    check(dacd, ac.grad, 3e-6, "zbuild dac")
    check(dedd, ed.grad, 1e-6, "zbuild ded")
    # coordinate update + CoM removal (dmt.py:40-58,385-386)
    pos = (torch.randn(Nn, 3, generator=g) * 1.5).double().requires_grad_(True)
    c2 = torch.randn(2 * P, 3, generator=g).double().requires_grad_(True)
    scale = torch.tensor([0.013], dtype=torch.float64, requires_grad=True)
    adj = torch.randint(0, 4, (P,), generator=g).to(torch.int32)
    bits = adj.long().repeat_interleave(2)
    adjs = torch.stack([torch.ones(2 * P, dtype=torch.float64), (bits & 1).double(), ((bits >> 1) & 1).double()], 1)
    diff = pos[row] - pos[col]
    unit = diff / diff.norm(dim=-1, keepdim=True).clamp(min=1e-8) * scale
    inv = (torch.tanh(c2) * adjs).mean(-1, keepdim=True)
    new = pos + torch.zeros_like(pos).index_add_(0, row, unit * inv)
    cnt = torch.zeros(B, 1, dtype=torch.float64).index_add_(0, node_mol, torch.ones(Nn, 1, dtype=torch.float64))
    mean = torch.zeros(B, 3, dtype=torch.float64).index_add_(0, node_mol, new) / cnt
    out = new - mean[node_mol]
    dout = torch.randn(Nn, 3, generator=g)
    out.backward(dout.double())
    posd, c2d, adjd, scd = pos.detach().float().to(d), c2.detach().float().to(d), adj.to(d), scale.detach().float().to(d)
    outd, dposd, dc2d, dsp = torch.empty(Nn, 3, device=d), torch.empty(Nn, 3, device=d), torch.empty(2 * P, 3, device=d), torch.empty(B, device=d)
    E._check(o.lib.dst_coord_fwd(C.byref(TL.c), E._ptr(posd), E._ptr(c2d), E._ptr(adjd), E._ptr(scd), E._ptr(outd), s()), "coord_fwd")
    E._check(o.lib.dst_coord_bwd(C.byref(TL.c), E._ptr(posd), E._ptr(c2d), E._ptr(adjd), E._ptr(scd), E._ptr(_dev(dout, d)), E._ptr(dposd), E._ptr(dc2d),
                                 E._ptr(dsp), s()), "coord_bwd")
    check(outd, out, 2e-6, "coord fwd")
    check(dposd, pos.grad, 2e-5, "coord dpos")
    check(dc2d, c2.grad, 2e-5, "coord dc2")
    check(dsp.sum().reshape(1), scale.grad, 2e-5, "coord dscale")


# ------------------------------------------------------------------------------------------------ whole graph vs the oracle
def _graph_case(version, coin, gpu_device):
    """Inputs of one training evaluation on the G13 batch: the oracle (CPU, autograd) computes loss + gradients with the conditioning
    embedding as a leaf; the same tensors drive the HIP graph."""
    cfg, sd0 = procedural_state_dict(version)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.startswith("cond_encoder.") else v.clone()) for k, v in sd0.items()}
    batch, draws = cases.training_batch(version), cases.training_draws()
    with torch.no_grad():
        _, info = otrain.training_loss({k: v.detach() for k, v in sd.items()}, cfg, batch, draws["t_raw"], draws["randn"], coin)
        z, _ = otrain.specformer_forward_train({k: v.detach() for k, v in sd.items()}, batch["context"], version, cfg.model.patch_len, cfg.model.stride)
        ctx0 = odmt._lin({k: v.detach() for k, v in sd.items()}, "cond_lin", z)
    ctx = ctx0.clone().requires_grad_(True)
    cond = (info["cond_x"], info["cond_edge_x"]) if coin else (None, None)
    pred, edge_pred = otrain.forward_with_context(sd, cfg, info["z_t"], info["node_mask"], info["edge_mask"], info["edge_z_t"], info["noise_level"], ctx,
                                                  cond[0], cond[1])
    loss = otrain.loss_from_predictions(pred, edge_pred, info["xh"], info["edge_x"], info["align_pos"], info["alpha_t"], info["sigma_t"])
    loss.backward()
    return cfg, sd, info, ctx, cond, pred.detach(), edge_pred.detach(), loss.detach()


@pytest.mark.parametrize("version,coin", [("ir", True), ("ir", False)])
def test_dmt_graph_loss_and_gradients_vs_oracle(gpu_device, version, coin):
    from diffspectra_amd import train_engine as T
    d = gpu_device
    cfg, sd, info, ctx, cond, pred, edge_pred, loss = _graph_case(version, coin, d)
    params = {k: v.detach().to(d).contiguous() for k, v in sd.items() if not k.startswith("cond_encoder.") and v.is_floating_point()}
    graph = T.DmtTrainGraph(params, cfg, d)
    TL = T.TrainLayout(info["node_mask"], d)
    pk_n = lambda t: None if t is None else TL.pack_nodes(t.to(d))
    pk_e = lambda t: None if t is None else TL.pack_pairs(t.to(d))
    pos, atom, edge = graph.forward(TL, pk_n(info["z_t"]), pk_e(info["edge_z_t"]), info["noise_level"].to(d), ctx.detach().to(d),
                                    pk_n(cond[0]), pk_e(cond[1]))
    check(TL.unpack_nodes(torch.cat([pos, atom], 1)), pred, 2e-5, "forward pred")
    check(TL.unpack_pairs(edge), edge_pred, 2e-5, "forward edge_pred")
    B = TL.B
    wm = (torch.sqrt(info["alpha_t"] / info["sigma_t"]) / B).to(d)
    tn = pk_n(torch.cat([info["align_pos"], info["xh"][:, :, 3:]], 2))
    loss_m, dpos, dfeat, dedge = graph.loss(TL, pos, atom, edge, tn[:, :3].contiguous(), tn[:, 3:].contiguous(), pk_e(info["edge_x"]), wm)
    got = float(loss_m.sum())
    assert abs(got - float(loss)) <= 1e-5 * abs(float(loss)), (got, float(loss))
    grads = graph.backward(dpos, dfeat, dedge)
    total = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for k, v in sd.items() if v.requires_grad and v.grad is not None)))
    floor = 1e-7 * total
    worst = ("", 0.0)
    bad = []
    for k, v in sd.items():
        if not v.requires_grad:
            continue
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        if k == "cond_lin.weight" or k == "cond_lin.bias":
            continue                                   # downstream of ctx_emb: covered by the SpecFormer test
        gk = grads[k].cpu().reshape(ref.shape)
        err = float((gk.double() - ref.double()).abs().max())
        lim = 1e-4 * float(ref.abs().max()) + floor
        if err > lim:
            bad.append((k, err, lim))
        if float(ref.abs().max()) > 0 and err / float(ref.abs().max()) > worst[1]:
            worst = (k, err / float(ref.abs().max()))
    gctx = grads["@ctx_emb"].cpu()
    err = float((gctx.double() - ctx.grad.double()).abs().max())
    if err > 1e-4 * float(ctx.grad.abs().max()) + floor:
        bad.append(("@ctx_emb", err, 0.0))
    print(f"[train graph {version} selfcond={coin}] loss {got:.6f} (oracle {float(loss):.6f}); total grad norm {total:.3f}; worst relative deviation {worst}")
    assert not bad, bad[:12]


# ------------------------------------------------------------------------------------------------ SpecFormer (training mode)
@pytest.mark.parametrize("version", ["ir", "allspectra"])
def test_specformer_train_graph_vs_oracle(gpu_device, version):
    """Forward (batch-statistics BatchNorm, residual attention scores) and backward of the conditioning encoder + cond_lin against
    autograd over oracle.train.specformer_forward_train; running statistics after the step as nn.BatchNorm1d leaves them."""
    from diffspectra_amd import train_engine as T
    from diffspectra_amd.spec_train import SpecTrainGraph
    d = gpu_device
    cfg, sd0 = procedural_state_dict(version)
    keys = [k for k in sd0 if k.startswith("cond_encoder.") or k.startswith("cond_lin.")]
    sd = {}
    for k in keys:
        v = sd0[k].clone()
        if v.is_floating_point() and not any(s in k for s in ("running_mean", "running_var", "sdp_attn.scale")):
            v.requires_grad_(True)
        sd[k] = v
    B = 5
    context = cases.spectra_for(version, B, salt=2)
    z, stats = otrain.specformer_forward_train(sd, context, version, cfg.model.patch_len, cfg.model.stride)
    ctx = odmt._lin(sd, "cond_lin", z)
    dctx = torch.randn(B, 1024, generator=torch.Generator().manual_seed(3)) * 0.1
    ctx.backward(dctx)
    params = {k: v.detach().to(d).contiguous() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    bufs = {k: v.detach().clone().to(d) for k, v in sd.items() if "running" in k or "num_batches" in k}
    graph = SpecTrainGraph(params, bufs, cfg, T.Ops(d))
    got = graph.forward([c.to(d) for c in context] if isinstance(context, list) else context.to(d))
    check(got, ctx, 2e-5, "context embedding (training-mode SpecFormer)")
    g = graph.backward(dctx.to(d))
    total = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.requires_grad)))
    bad, worst = [], ("", 0.0)
    for k, v in sd.items():
        if not v.requires_grad:
            continue
        err = float((g[k].cpu().double() - v.grad.double()).abs().max())
        lim = 1e-4 * float(v.grad.abs().max()) + 1e-7 * total
        if err > lim:
            bad.append((k, err, lim))
        if err / (float(v.grad.abs().max()) + 1e-30) > worst[1] and float(v.grad.abs().max()) > 1e-6 * total:
            worst = (k, err / float(v.grad.abs().max()))
    print(f"[specformer train {version}] total grad norm {total:.4f}; worst relative deviation {worst}")
    assert not bad, bad[:10]
    for k, v in stats.items():
        check(bufs[k], v, 1e-5, k)
    assert int(bufs["cond_encoder.backbone.encoder.layers.0.norm_attn.1.num_batches_tracked"]) == int(sd["cond_encoder.backbone.encoder.layers.0.norm_attn.1.num_batches_tracked"]) + 1



@pytest.mark.parametrize("version", ["ir", "allspectra"])
def test_specformer_flash_attention_matches_materialised_scores(gpu_device, version):
    """The score-free attention of the bf16 mode (dst_spec_attn_flash_*: layer l's residual scores recomputed as scale * sum_j q_j k_j^T
    on the bf16 matrix pipe, probabilities re-created in the backward, the chained score gradient as direct contributions to the
    lower layers' dq / dk) against the fp32 kernels that materialise the [B,H,L,L] scores - same graph, same fp32 GEMMs, only the
    attention differs; tolerance = bf16 operand rounding (2^-9 relative per q / k / v / p element)."""
    from diffspectra_amd import train_engine as T
    from diffspectra_amd.spec_train import SpecTrainGraph
    d = gpu_device
    cfg, sd0 = procedural_state_dict(version)
    params = {k: v.clone().to(d).contiguous() for k, v in sd0.items()
              if (k.startswith("cond_encoder.") or k.startswith("cond_lin.")) and v.is_floating_point() and "running" not in k}
    B = 6
    context = cases.spectra_for(version, B, salt=5)
    context = [c.to(d) for c in context] if isinstance(context, list) else context.to(d)
    dctx = (torch.randn(B, 1024, generator=torch.Generator().manual_seed(4)) * 0.1).to(d)
    res = {}
    for flash in (False, True):
        bufs = {k: v.detach().clone().to(d) for k, v in sd0.items() if "running" in k or "num_batches" in k}
        graph = SpecTrainGraph(params, bufs, cfg, T.Ops(d))
        graph.flash = flash
        ctx = graph.forward(context)
        aos = [lt["ao"].clone() for lt in graph.t["layers"]]
        g = graph.backward(dctx)
        res[flash] = (ctx.clone(), aos, {k: v.clone() for k, v in g.items()})
    for l, (a, b) in enumerate(zip(res[True][1], res[False][1])):
        check(a, b, 2e-2, f"attention output of layer {l}")
    check(res[True][0], res[False][0], 2e-2, "context embedding")
    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in res[False][2].values())))
    dot = sum(float((res[True][2][k].double() * v.double()).sum()) for k, v in res[False][2].items())
    n1 = float(torch.sqrt(sum((v.double() ** 2).sum() for v in res[True][2].values())))
    worst = ("", 0.0)
    for k, v in res[False][2].items():
        # a V / to_out bias in front of a BatchNorm has a (nearly) ZERO true gradient - a sum of cancelling terms - so the bf16 rounding
        # noise is measured against the tensor's norm plus a small share of the total gradient norm
        err = float((res[True][2][k].double() - v.double()).norm()) / (float(v.double().norm()) + 3e-3 * total)
        if err > worst[1]:
            worst = (k, err)
    print(f"[flash attention {version}] gradient cosine {dot / (total * n1):.6f}; worst per-tensor relative L2 deviation {worst}")
    assert dot / (total * n1) > 0.9995 and worst[1] < 5e-2

# ------------------------------------------------------------------------------------------------ the reference's loss_fn surface vs G13
class _Replay:
    """Replays queued CPU tensors for torch.rand / torch.randn calls of matching shape, on the requested device."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __call__(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        t = self.queue.pop(0)
        assert tuple(t.shape) == tuple(size), (t.shape, size)
        return t.clone().to(kw.get("device", "cpu"))


def _train_model(version, d):
    from diffspectra_amd import filler
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.registry import create_model
    import diffspectra_amd.dmt  # noqa: F401
    cfg = qm9s_config(version, device=d)
    cfg.model.dropout = 0.0
    model = create_model(cfg)
    filler.fill_module_(model)
    return cfg, model


@pytest.mark.parametrize("version,coin_name", [("ir", "selfcond"), ("ir", "plain"), ("allspectra", "selfcond")])
def test_loss_fn_matches_reference_loss_and_gradients(gpu_device, monkeypatch, version, coin_name):
    """``get_sde_graph_loss_fn(...)(model, batch)`` + ``loss.backward()`` through the HIP training library against golden G13 (the
    reference's own loss_fn, train=True, dropout 0, every random draw injected): loss rtol 1e-5, every parameter's gradient norm and
    strided sample and the stored full gradients rtol 1e-4, Kabsch rotations, BatchNorm running statistics."""
    _loss_fn_against_golden(gpu_device, monkeypatch, version, coin_name, "g13_training.npz", 0.0)


@pytest.mark.parametrize("version,coin_name", [("ir", "selfcond"), ("allspectra", "plain")])
def test_loss_fn_matches_reference_with_dropout(gpu_device, monkeypatch, version, coin_name):
    """Config 5 AS SHIPPED (``config.model.dropout = 0.1``) against golden G17: the reference's loss_fn with its ``nn.Dropout`` masks
    injected - the masks this library's kernels generate (Philox, keyed on (seed, 4 * block + site, element)), pair-symmetric on the
    edge side.  Same gates as G13: loss 1e-5, all gradient norms / samples / full tensors rtol 1e-4."""
    _loss_fn_against_golden(gpu_device, monkeypatch, version, coin_name, "g17_training_dropout.npz", 0.1)


def test_dropout_kernel_masks_equal_the_oracle_masks(gpu_device):
    """``dst_dropout``'s mask against ``oracle.philox.dropout_keep`` (the generator golden G17's injected masks come from), bit for bit,
    incl. a length that is not a multiple of the 4-element Philox block."""
    from diffspectra_amd import train_engine as T
    from oracle import philox
    o = T.Ops(gpu_device)
    for n, p, seed, stream in ((1_000_003, 0.1, 1111, 0), (4099, 0.1, 2222, 31), (512 * 37, 0.25, (1 << 61) + 12345, 7)):
        y = torch.ones(n, device=gpu_device)
        o.dropout(y, p, seed, stream)
        keep = philox.dropout_keep(seed, stream, n, p)
        want = torch.from_numpy(keep.astype(np.float32) * philox.dropout_scale(p))
        assert torch.equal(y.cpu(), want), (n, p, seed, stream)


def _loss_fn_against_golden(gpu_device, monkeypatch, version, coin_name, fixture, dropout_p):
    from diffspectra_amd import losses as Lh
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    d = gpu_device
    cfg, model = _train_model(version, d)
    cfg.model.dropout = dropout_p
    g = cases.load_npz(fixture)
    tag = f"{version}_{coin_name}"
    batch, draws = cases.training_batch(version), cases.training_draws()
    batch = {k: v for k, v in batch.items() if k != "n_atoms"}
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, None, cfg)
    monkeypatch.setattr(torch, "rand", _Replay([draws["t_raw"]]))
    monkeypatch.setattr(torch, "randn", _Replay(draws["randn"]))
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.tensor(list(cases.TRAIN_DROPOUT_SEEDS)))
    monkeypatch.setattr(Lh, "random", lambda: 0.0 if coin_name == "selfcond" else 1.0)
    loss = loss_fn(model, batch)
    monkeypatch.undo()
    loss.backward()
    ref_loss = float(g[tag + "_loss"])
    assert abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss), (float(loss), ref_loss)
    last = loss_fn.last
    TL = last["layout"]
    check(TL.unpack_nodes(last["z"]), g[tag + "_z_t"], 1e-5, "z_t")
    check(TL.unpack_pairs(last["ez"]), g[tag + "_edge_z_t"], 1e-5, "edge_z_t")
    check(TL.unpack_nodes(last["aligned"]), g[tag + "_align_pos"], 2e-5, "Kabsch-aligned target")
    for b, n in enumerate(cases.TRAIN_ATOMS):
        if n >= 3:
            check(last["rot"][b].reshape(3, 3), g[tag + "_rotations"][b], 1e-4, f"rotation {b}")
    check(TL.unpack_nodes(torch.cat(last["pred"][:2], 1)), g[tag + "_pred"], 3e-5, "pred")
    _check_gradients_and_bn(model, g, tag, f"loss_fn {fixture[:3]}", float(loss), ref_loss)


def _check_gradients_and_bn(model, g, tag, label, loss, ref_loss):
    """Every parameter's gradient norm and strided sample, the stored full gradients (rtol 1e-4 + a floor of 1e-7 of the total norm), and
    the BatchNorm running statistics after the step, against a training golden (G13 / G17)."""
    names = json.loads(g[tag + "_grad_names"])
    norms = g[tag + "_grad_norms"].numpy()
    total = float(np.sqrt(np.sum(np.square(norms))))
    floor = 1e-7 * total
    grads = {n: p.grad for n, p in model.module.named_parameters()}
    bad = []
    for i, n in enumerate(names):
        ref_norm = float(norms[i])
        gr = grads[n]
        if gr is None:
            assert ref_norm == 0.0, n
            continue
        gr = gr.detach().cpu()
        if abs(float(gr.double().norm()) - ref_norm) > 1e-4 * ref_norm + floor:
            bad.append((n, "norm", float(gr.double().norm()), ref_norm))
        idx = torch.linspace(0, gr.numel() - 1, min(64, gr.numel())).round().long()
        want = g[tag + "_grad_samples"][i][:len(idx)]
        err = float((gr.reshape(-1)[idx] - want).abs().max())
        if err > 1e-4 * float(want.abs().max()) + 1e-3 * ref_norm / max(1.0, gr.numel() ** 0.5) + floor:
            bad.append((n, "sample", err, float(want.abs().max())))
        if n in cases.TRAIN_FULL_GRADS:
            full = g[f"{tag}_grad::{n}"]
            if not torch.allclose(gr, full, rtol=1e-4, atol=1e-4 * float(full.abs().max()) + floor):
                bad.append((n, "full", float((gr - full).abs().max()), float(full.abs().max())))
    print(f"[{label} {tag}] loss {loss:.6f} (reference {ref_loss:.6f}); total grad norm {total:.3f}; {len(names)} parameters checked")
    assert not bad, bad[:10]
    bn = "cond_encoder.backbone.encoder.layers.0.norm_attn.1."
    bufs = dict(model.module.named_buffers())
    check(bufs[bn + "running_mean"], g[tag + "_bn_running_mean"], 1e-5, "BatchNorm running_mean after the step")
    check(bufs[bn + "running_var"], g[tag + "_bn_running_var"], 1e-5, "BatchNorm running_var after the step")
    assert int(bufs[bn + "num_batches_tracked"]) == int(g[tag + "_bn_batches"])


@pytest.mark.parametrize("coin_name", ["selfcond", "plain"])
def test_model_forward_under_autograd_reproduces_reference_gradients(gpu_device, coin_name):
    """The score-function boundary (reference callers differentiate straight through ``model(...)``, losses.py:346-357): a
    reference-style loss written by hand around ``model(...)`` in training mode - inputs are golden G13's own z_t / noise level /
    aligned target, the loss is torch arithmetic on the returned tensors - must fill ``p.grad`` with the reference's gradients."""
    d = gpu_device
    cfg, model = _train_model("ir", d)
    model.train()
    g = cases.load_npz("g13_training.npz")
    tag = f"ir_{coin_name}"
    batch = cases.training_batch("ir")
    node_mask, edge_mask = batch["atom_mask"].unsqueeze(2).to(d), batch["edge_mask"].to(d)
    z_t, edge_z_t, nl = g[tag + "_z_t"].to(d), g[tag + "_edge_z_t"].to(d), g[tag + "_noise_level"].to(d)
    ctx = batch["context"].to(d)
    B = z_t.shape[0]
    kw = dict(context=ctx, edge_x=edge_z_t, noise_level=nl, alpha_t=g[tag + "_alpha_t"].to(d), sigma_t=g[tag + "_sigma_t"].to(d))
    cond_x = cond_e = None
    if coin_name == "selfcond":
        with torch.no_grad():                                            # losses.py:344-351: training mode, no gradient
            cond_x, cond_e = model(torch.zeros(B, device=d), z_t, node_mask, edge_mask, cond_x=None, cond_edge_x=None, **kw)
        assert not cond_x.requires_grad
        check(cond_x, g[tag + "_cond_x"], 3e-5, "self-conditioning prediction")
    pred, edge_pred = model(torch.zeros(B, device=d), z_t, node_mask, edge_mask, cond_x=cond_x, cond_edge_x=cond_e, **kw)
    assert pred.requires_grad and edge_pred.requires_grad and pred.grad_fn is not None
    check(pred, g[tag + "_pred"], 3e-5, "pred")
    check(edge_pred, g[tag + "_edge_pred"], 3e-5, "edge_pred")
    loss = otrain.loss_from_predictions(pred, edge_pred, g[tag + "_xh"].to(d), g[tag + "_edge_x"].to(d), g[tag + "_align_pos"].to(d),
                                        g[tag + "_alpha_t"].to(d), g[tag + "_sigma_t"].to(d))
    loss.backward()
    ref_loss = float(g[tag + "_loss"])
    assert abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    _check_gradients_and_bn(model, g, tag, "model(...) under autograd", float(loss), ref_loss)
    with pytest.raises(RuntimeError, match="twice"):
        pred.sum().backward()
    model.eval()
    out = model(torch.zeros(B, device=d), z_t, node_mask, edge_mask, cond_x=cond_x, cond_edge_x=cond_e, **kw)[0]
    assert not out.requires_grad                                         # eval mode: the sampling kernels, no graph


@pytest.mark.parametrize("max_grad", [10.0, 0.5])
def test_clip_update_replays_the_reference_queue(ops, gpu_device, max_grad):
    """dst_clip_update over 70 steps (the 50-entry history wraps) against the host restatement of gradient_clipping + Queue
    (losses.py:28-72): coefficient, allowed norm, history contents and order."""
    from diffspectra_amd import engine as E, losses as Lh
    T, o = ops
    d = gpu_device
    rng = np.random.default_rng(3)
    norms = np.concatenate([rng.uniform(5.0, 60.0, 30), rng.uniform(0.5, 8.0, 40)])          # clipped at first, below the bound later
    q = Lh.Queue()
    q.add(3000)
    st = torch.zeros(64, device=d)
    st[0], st[50] = 3000.0, 1.0
    world = 2.0
    for it, nrm in enumerate(norms):
        nsq = torch.tensor([(nrm * world) ** 2], dtype=torch.float32, device=d)               # the shards hold sums over ranks
        E._check(o.lib.dst_clip_update(E._ptr(nsq), C.c_float(1.0 / world), C.c_float(max_grad), E._ptr(st), E._stream()), "dst_clip_update")
        got = st.cpu().double()
        n32 = float(np.sqrt(np.float64(np.float32((nrm * world) ** 2)))) / world
        coef, allowed = Lh.clip_coefficient(n32, q, max_grad)
        assert abs(got[51] - coef) <= 2e-6 * max(coef, 1e-3), (it, float(got[51]), coef)
        assert abs(got[52] - n32) <= 1e-6 * n32 and abs(got[53] - float(allowed)) <= 2e-6 * float(allowed)
        if max_grad > 1.0:
            assert int(got[50]) == len(q)
            np.testing.assert_allclose(got[:len(q)].numpy(), np.asarray(q.items, dtype=np.float64), rtol=2e-6)
    if max_grad > 1.0:
        assert len(q) == 50


def test_step_fn_fused_optimizer_and_ema(gpu_device, monkeypatch):
    """``get_step_fn`` (losses.py:97-125): zero_grad -> loss -> backward -> warm-up lr + adaptive clip -> fused AdamW-amsgrad -> EMA,
    two steps, against torch.optim.AdamW(amsgrad=True, weight_decay=1e-12) + clip_grad_norm_ + the reference EMA formula applied to the
    same gradients on the CPU; checkpoint state in torch's format."""
    from diffspectra_amd import losses as Lh
    from diffspectra_amd.ema import ExponentialMovingAverage
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    d = gpu_device
    cfg, model = _train_model("ir", d)
    cfg.optim.warmup = 10
    cfg.optim.grad_clip = 10.0
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    opt = Lh.get_optimizer(cfg, model.parameters())
    optimize_fn = Lh.optimization_manager(cfg)
    step_fn = Lh.get_step_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, optimize_fn, None, cfg)
    state = dict(optimizer=opt, model=model, ema=ema, step=3)
    ref_p = [p.detach().cpu().clone().requires_grad_(p.requires_grad) for p in model.parameters()]
    ref_opt = torch.optim.AdamW([p for p in ref_p], lr=cfg.optim.lr, amsgrad=True, weight_decay=1e-12)
    ref_ema = [p.detach().clone() for p in ref_p if p.requires_grad]
    queue = Lh.Queue()
    queue.add(3000)
    batch = {k: v for k, v in cases.training_batch("ir").items() if k != "n_atoms"}
    torch.manual_seed(5)
    for it in range(2):
        loss = step_fn(state, batch)
        assert torch.isfinite(loss)
        grads = [p.grad.detach().cpu().clone() for p in model.parameters()]      # the step leaves the gradients in place
        for rp, gr in zip(ref_p, grads):
            rp.grad = gr.clone() if rp.requires_grad else None
        for gopt in ref_opt.param_groups:
            gopt["lr"] = cfg.optim.lr * min((3 + it) / 10, 1.0)
        max_norm = min(1.5 * queue.mean() + 2 * queue.std(), 10.0)
        # clip_grad_norm_ (losses.py:38) with the norm accumulated in fp64: torch's fp32 CPU reduction over the 11 M-element head weight
        # is itself 1e-4 off (measured: 18.1926 against 18.1944 in fp64, which the fused kernel reproduces to 1e-7)
        trainable = [p for p in ref_p if p.requires_grad]
        norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in trainable)))
        coef = min(1.0, max_norm / (norm + 1e-6))
        for p in trainable:
            p.grad.mul_(coef)
        queue.add(float(max_norm) if norm > max_norm else norm)
        assert abs(optimize_fn.last_grad_norm - norm) <= 1e-5 * norm
        ref_opt.step()
        n_upd = it + 1
        decay = min(0.999, (1 + n_upd) / (10 + n_upd))
        for s_, rp in zip(ref_ema, [p for p in ref_p if p.requires_grad]):
            s_.sub_((1.0 - decay) * (s_ - rp.detach()))
        worst = 0.0
        for p, rp in zip(model.parameters(), ref_p):
            delta = float((p.detach().cpu() - rp.detach()).abs().max())
            worst = max(worst, delta)
            assert delta <= 2e-7 + 2e-6 * float(rp.detach().abs().max()), delta
        for s_, rs in zip(ema.shadow_params, ref_ema):
            assert float((s_.cpu() - rs).abs().max()) <= 2e-7 + 2e-6 * float(rs.abs().max())
        print(f"[step_fn step {it}] loss {float(loss):.5f}, grad norm {norm:.3f}, clip at {max_norm:.1f}, worst parameter deviation {worst:.2e}")
    assert state["step"] == 5 and ema.num_updates == 2
    sd = opt.state_dict()
    man = json.load(open(cases.fixture_path("g14_checkpoint_manifest.json")))
    assert sorted(next(iter(sd["state"].values())).keys()) == man["optimizer_state_keys"]
    assert len(sd["state"]) == sum(p.requires_grad for p in model.parameters())          # as torch: no state for the frozen sdp_attn.scale
    assert list(sd.keys()) == man["optimizer_keys"]
    # the sampling path sees the trained weights (engine re-packed)
    a = cases.forward_inputs("ir", True)
    model.eval()
    out, _ = model(torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d), context=a["context"].to(d),
                   edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d), cond_x=None, cond_edge_x=None)
    assert torch.isfinite(out).all()


# ------------------------------------------------------------------------------------------------ data-parallel step (2 ranks on this GPU)
def _ddp_worker(rank, world, port, out_path):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffspectra_amd import losses as Lh
        from diffspectra_amd.ema import ExponentialMovingAverage
        from diffspectra_amd.noise_schedule import NoiseScheduleVP
        d = torch.device("cuda:0")
        cfg, model = _train_model("ir", d)
        cfg.optim.warmup = 0
        ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
        opt = Lh.get_optimizer(cfg, model.parameters())
        assert opt.world == world and opt.shard * world == opt.n_pad
        step_fn = Lh.get_step_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, Lh.optimization_manager(cfg), None, cfg)
        state = dict(optimizer=opt, model=model, ema=ema, step=0)
        batch = {k: v for k, v in cases.training_batch("ir", salt=rank).items() if k != "n_atoms"}       # a different batch per rank
        torch.manual_seed(100 + rank)
        import random
        random.seed(7)                                                                                      # the same self-conditioning coin on both ranks
        p_before = opt.P.detach().cpu().clone()
        loss = step_fn(state, batch)
        assert not torch.equal(p_before, opt.P.detach().cpu())
        p_step = opt.P.detach().cpu().clone()
        stale = opt.ema_flat.detach().cpu().clone()
        ema.copy_to(model.parameters())                                  # reading the EMA gathers the shards
        assert opt._ema_stale is False and not torch.equal(stale, opt.ema_flat.detach().cpu())
        assert torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), opt.unpadded(opt.ema_flat))
        flat_grad_local = opt.G.detach().cpu().clone()
        ema_gathered = opt.ema_flat.detach().cpu().clone()
        # checkpoint in a sharded job: state_dict() gathers the sharded moments - a collective every rank joins; rank 0 writes
        from diffspectra_amd.evaluate import save_checkpoint, restore_checkpoint
        opt.P.copy_(p_step.to(d))                                                  # undo the copy_to above: parameters of the step
        save_checkpoint(out_path + ".ckpt", state)
        dist.barrier()
        cfg2, model2 = _train_model("ir", d)
        cfg2.optim.warmup = 0
        ema2 = ExponentialMovingAverage(model2.parameters(), decay=0.999)
        opt2 = Lh.get_optimizer(cfg2, model2.parameters())
        step2 = Lh.get_step_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, Lh.optimization_manager(cfg2), None, cfg2)
        state2 = restore_checkpoint(out_path + ".ckpt", dict(optimizer=opt2, model=model2, ema=ema2, step=0), d)
        assert state2["step"] == 1 and torch.equal(opt2.P, opt.P) and torch.equal(opt2.M, opt.M) and torch.equal(opt2.Vmax, opt.Vmax)
        finals = []
        for st_, fn_ in ((state, step_fn), (state2, step2)):
            torch.manual_seed(500 + rank)
            random.seed(9)
            fn_(st_, batch)
            finals.append(st_["optimizer"].P.detach().cpu().clone())
        assert torch.equal(finals[0], finals[1]), "save -> restore -> step diverged from the uninterrupted run"
        up = lambda t: opt.unpadded(t.to(d)).cpu()                           # parameters() order without the alignment gaps of the flat buffers
        torch.save(dict(P=up(p_step), G=up(flat_grad_local), loss=float(loss.detach()), ema=up(ema_gathered), P2=up(finals[1])), out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_two_ranks(gpu_device, tmp_path):
    """Config 5's data-parallel step on 2 ranks (one process each, both on this GPU, gloo rehearsal backend): gradients are averaged over
    the ranks, every rank updates its half of the flat parameter buffer with the fused kernel, the halves are all-gathered - both ranks
    end with identical parameters, equal to one AdamW-amsgrad step on the mean gradient."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "ddp.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["P"], r1["P"]), "ranks ended with different parameters"
    assert torch.equal(r0["ema"], r1["ema"])
    assert torch.equal(r0["P2"], r1["P2"]) and not torch.equal(r0["P2"], r0["P"])      # the step after the checkpoint round trip
    assert abs(r0["loss"] - r1["loss"]) > 1e-3                                   # the ranks did see different batches
    assert torch.equal(r0["G"], r1["G"])                                         # gloo path: all_reduce in place, both hold the sum
    cfg, model = _train_model("ir", gpu_device)
    p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    n = p0.numel()
    # after the gloo all_reduce both flat buffers hold the SUM of the two ranks' gradients
    gsum = r0["G"][:n]
    ref = p0.clone().requires_grad_(True)
    ref.grad = gsum / 2
    norm = float(ref.grad.norm())
    coef = min(1.0, 10.0 / (norm + 1e-6))
    ref.grad.mul_(coef)
    torch.optim.AdamW([ref], lr=cfg.optim.lr, amsgrad=True, weight_decay=1e-12).step()
    assert float((r0["P"][:n] - ref.detach()).abs().max()) <= 2e-7 + 2e-6 * float(ref.detach().abs().max())


# ------------------------------------------------------------------------------------------------ dropout
def test_dropout_kernel_and_training_with_dropout(gpu_device, monkeypatch):
    """FF dropout of the training forward (dmt.py:114-120) with in-kernel Philox masks: keep rate and scaling, the backward re-creates
    the forward's mask, streams are independent; and the gradient of a p = 0.1 training loss agrees with a central finite difference
    along the gradient direction (same draws, same masks)."""
    from diffspectra_amd import losses as Lh, train_engine as T
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    d = gpu_device
    o = T.Ops(d)
    x = torch.ones(1_000_003, device=d)
    y = x.clone()
    o.dropout(y, 0.1, 12345, 7)
    keep = float((y != 0).float().mean())
    assert abs(keep - 0.9) < 2e-3 and torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1.0 / 0.9))
    y2 = x.clone()
    o.dropout(y2, 0.1, 12345, 7)
    assert torch.equal(y, y2)                                                    # the mask is a function of (seed, stream, index)
    y3 = x.clone()
    o.dropout(y3, 0.1, 12345, 8)
    agree = float(((y3 != 0) == (y != 0)).float().mean())
    assert abs(agree - (0.81 + 0.01)) < 5e-3                                      # independent masks agree on 0.9^2 + 0.1^2 of the elements
    cfg, model = _train_model("ir", d)
    cfg.model.dropout = 0.1
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, None, cfg)
    batch = {k: v for k, v in cases.training_batch("ir").items() if k != "n_atoms"}
    draws = cases.training_draws()
    real_randint = torch.randint

    def run():
        monkeypatch.setattr(torch, "rand", _Replay([draws["t_raw"]]))
        monkeypatch.setattr(torch, "randn", _Replay(draws["randn"]))
        monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.tensor([1111, 2222]))
        monkeypatch.setattr(Lh, "random", lambda: 0.0)
        try:
            return loss_fn(model, batch)
        finally:
            monkeypatch.undo()

    loss = run()
    loss.backward()
    # direction: the gradient over every feed-forward weight (the tensors whose gradient passes through the dropout masks)
    ps = [p for n, p in model.module.named_parameters() if ".ff_linear" in n and n.endswith("weight")]
    gs = [p.grad.detach().clone() for p in ps]
    an = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gs)))
    eps = 2e-2

    def shift(sign):
        with torch.no_grad():
            for p, g in zip(ps, gs):
                p.data.add_(sign * eps / an * g)

    shift(+1)
    lp = float(run())
    shift(-2)
    lm = float(run())
    shift(+1)
    fd = (lp - lm) / (2 * eps)
    print(f"[dropout] loss {float(loss):.5f}; directional derivative along the FF-weight gradient: finite difference {fd:.5f}, analytic {an:.5f}")
    assert abs(fd - an) <= 0.03 * an + 1e-4
    loss0 = float(loss)
    cfg.model.dropout = 0.0
    assert math.isfinite(loss0)


def test_bf16_training_precision(gpu_device, monkeypatch):
    """config.training.precision = 'bf16' (BASELINE config 5): GEMM operands rounded to bf16, fp32 accumulation.  Not the reference's
    arithmetic (it has no AMP), so it is held to the fp32 golden G13 only loosely: loss within 1 %, total gradient direction
    within cos > 0.995, every parameter finite; and the GEMM itself against a bf16-rounded fp64 product tightly."""
    from diffspectra_amd import losses as Lh, train_engine as T
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    d = gpu_device
    o = T.Ops(d)
    o.bf16 = True
    g = torch.Generator().manual_seed(1)
    for M, N, K in ((200, 130, 300), (64, 64, 48), (130, 40, 2000)):
        A, Bm = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        Cd = torch.zeros(M, N, device=d)
        o.gemm(T.mv(A.to(d)), T.mv(Bm.to(d)), T.mv(Cd), False, True)
        ref = A.bfloat16().double() @ Bm.bfloat16().double().t()
        assert float((Cd.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * max(1.0, K ** 0.5 / 8), (M, N, K)
    cfg, model = _train_model("ir", d)
    cfg.training.precision = "bf16"
    ref_g = cases.load_npz("g13_training.npz")
    tag = "ir_selfcond"
    batch, draws = cases.training_batch("ir"), cases.training_draws()
    batch = {k: v for k, v in batch.items() if k != "n_atoms"}
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, None, cfg)
    monkeypatch.setattr(torch, "rand", _Replay([draws["t_raw"]]))
    monkeypatch.setattr(torch, "randn", _Replay(draws["randn"]))
    monkeypatch.setattr(Lh, "random", lambda: 0.0)
    loss = loss_fn(model, batch)
    monkeypatch.undo()
    loss.backward()
    ref_loss = float(ref_g[tag + "_loss"])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-2 * ref_loss, (float(loss.detach()), ref_loss)
    names = json.loads(ref_g[tag + "_grad_names"])
    grads = {n: p.grad for n, p in model.module.named_parameters()}
    dot = n1 = n2 = 0.0
    for i, n in enumerate(names):
        if grads[n] is None:
            continue
        gr = grads[n].detach().cpu()
        assert torch.isfinite(gr).all(), n
        idx = torch.linspace(0, gr.numel() - 1, min(64, gr.numel())).round().long()
        a, b = gr.reshape(-1)[idx].double(), ref_g[tag + "_grad_samples"][i][:len(idx)].double()
        dot, n1, n2 = dot + float((a * b).sum()), n1 + float((a * a).sum()), n2 + float((b * b).sum())
    cos = dot / (n1 * n2) ** 0.5
    print(f"[bf16] loss {float(loss.detach()):.5f} (fp32 reference {ref_loss:.5f}); cosine of the sampled gradient against G13 {cos:.5f}")
    assert cos > 0.995


def test_training_from_processed_file_end_to_end(gpu_device, tmp_path):
    """The reference's training loop shape (run_lib.diffspectra_train: DataLoader -> train_step_fn) from a processed QM9S file in the PyG
    layout: ``qm9s_reader.ProcessedQM9S`` -> ``train_data.TrainBatches`` (transform + collate + augmentation, pinned by golden G16) ->
    ``losses.get_step_fn`` with the shipped dropout 0.1, two optimizer steps; then the trained weights sample through the HIP sampler."""
    from diffspectra_amd import losses as Lh, train_data as TD
    from diffspectra_amd.ema import ExponentialMovingAverage
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.qm9s_reader import ProcessedQM9S
    from tests.test_host_cpu import _write_processed_qm9s
    d = gpu_device
    mols = cases.raw_molecules(n_atoms=(3, 7, 2, 12, 9, 5, 8, 4))
    _write_processed_qm9s(str(tmp_path / "QM9S" / "processed"), mols, "pyg2")
    proc = ProcessedQM9S(str(tmp_path / "QM9S"))
    cfg, model = _train_model("allspectra", d)
    cfg.model.dropout = 0.1
    cfg.optim.warmup = 0
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
    opt = Lh.get_optimizer(cfg, model.parameters())
    step_fn = Lh.get_step_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, Lh.optimization_manager(cfg), None, cfg)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    before = opt.P.clone()
    torch.manual_seed(0)
    np.random.seed(0)
    losses = []
    for epoch in range(2):
        for batch in TD.TrainBatches(proc, "test", batch_size=2, spectra_version="allspectra", device=d):
            losses.append(float(step_fn(state, batch).detach()))
    assert len(losses) == 2 and all(math.isfinite(v) for v in losses) and state["step"] == 2
    assert float((opt.P - before).abs().max()) > 1e-5 and torch.isfinite(opt.P).all()
    eval_step = Lh.get_step_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), False, None, None, cfg)
    val = eval_step(state, next(iter(TD.TrainBatches(proc, "valid", batch_size=2, spectra_version="allspectra", shuffle=False, aug_rotation=False,
                                                       aug_translation=False, device=d))))
    assert math.isfinite(float(val)) and not val.requires_grad                      # EMA weights, eval mode, no gradient (losses.py:117-122)
    print(f"[train from file] losses {losses}, eval-mode EMA loss {float(val):.4f}")


def test_training_step_is_bit_reproducible_across_stream_modes(gpu_device, monkeypatch):
    """The three-stream step (main / node rows / weight gradients) against itself and against the single-stream order, on a batch large
    enough for the streams to overlap (96 molecules, all-spectra, dropout 0.1, bf16, self-conditioning forward taken): every kernel reduces
    in a fixed order, so loss and all gradients must agree BIT FOR BIT - a missing cross-stream dependency shows up here as a difference."""
    import random as _random
    from diffspectra_amd import filler, losses as Lh
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.registry import create_model
    d = gpu_device
    cfg = qm9s_config("allspectra", device=d)
    cfg.training.precision = "bf16"
    model = create_model(cfg)
    filler.fill_module_(model)
    Bt = 96
    n_atoms = filler.sample_n_atoms(Bt, seed=3).tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    g = torch.Generator().manual_seed(11)
    types = torch.randint(0, 5, (Bt, N), generator=g)
    order = torch.triu((torch.rand(Bt, N, N, generator=g) > 0.8).float() * torch.randint(1, 4, (Bt, N, N), generator=g), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(Bt, N, N)
    ctx = filler.synthetic_spectra(Bt, "allspectra", seed=5)
    batch = dict(positions=(torch.randn(Bt, N, 3, generator=g) * 1.3 * node_mask).to(d), atom_mask=node_mask.squeeze(-1).to(d),
                 edge_mask=edge_mask.to(d), atom_one_hot=(F.one_hot(types, 5).float() * node_mask).to(d),
                 edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1).to(d), formal_charges=torch.zeros(Bt, N, 1, device=d),
                 context=[c.to(d) for c in ctx])
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, None, cfg)
    params = [p for p in model.parameters() if p.requires_grad]
    bn = [b for n_, b in model.named_buffers() if "running" in n_]
    bn0 = [b.detach().clone() for b in bn]

    def run():
        for b, b0 in zip(bn, bn0):
            b.copy_(b0)
        for p in params:
            p.grad = None
        torch.manual_seed(123)
        _random.seed(7)
        monkeypatch.setattr(Lh, "random", lambda: 0.0)                       # the self-conditioning forward is taken
        try:
            loss = loss_fn(model, batch)
            loss.backward()
        finally:
            monkeypatch.undo()
        torch.cuda.synchronize()
        return float(loss.detach()), torch.cat([p.grad.reshape(-1) for p in params]).clone()

    names = [n_ for n_, p in model.named_parameters() if p.requires_grad]

    def differing(ga, gb):                                                   # the parameters whose gradients differ (for the failure message)
        out, o_ = [], 0
        for n_, p in zip(names, params):
            if not torch.equal(ga[o_:o_ + p.numel()], gb[o_:o_ + p.numel()]):
                out.append(n_)
            o_ += p.numel()
        return out[:4] + ["..."] + out[-14:] if len(out) > 18 else out

    l0, g0 = run()
    assert math.isfinite(l0) and bool(torch.isfinite(g0).all())
    for rep in range(3):
        l1, g1 = run()
        assert l1 == l0 and torch.equal(g1, g0), (f"three-stream step differs from itself in repetition {rep} (largest difference "
                                                  f"{float((g1 - g0).abs().max()):.3e} of {float(g0.abs().max()):.3e}): {differing(g1, g0)}")
    monkeypatch.setenv("DIFFSPECTRA_NODE_STREAM", "0")
    monkeypatch.setenv("DIFFSPECTRA_ASYNC_DW", "0")
    l2, g2 = run()
    assert l2 == l0 and torch.equal(g2, g0), f"single-stream order gives different bits: {differing(g2, g0)}"


@pytest.mark.gpu
def test_fused_chain_entry_points_reject_bad_arguments(gpu_device):
    """The C-ABI of the fused row chains validates what a raw caller could get wrong: NULL operands, pointers that are not 16-byte aligned (every
    access is a 16-byte vector), column offsets that are not multiples of 4, a dropout probability outside [0, 1) - DS_ERR_ARG (-1), nothing launched."""
    import ctypes as C
    import struct
    from diffspectra_amd import engine as E, filler, train_engine as T
    d = gpu_device
    lib = T.load_train_library()
    node_mask, _ = filler.masks_from_n_atoms([5, 3, 1])
    TL = T.TrainLayout(node_mask, d)
    Nn, Pp, D = TL.Nn, TL.Pp, 2 * TL.Pp
    keep = []                                                                    # (the kernels of the valid calls run: every operand stays alive)

    def f(*s):
        keep.append(torch.zeros(*s, device=d))
        return keep[-1]

    def hb(*s):
        keep.append(torch.zeros(*s, dtype=torch.bfloat16, device=d))
        return keep[-1]
    ada = f(TL.B, T.ADA)
    P = lambda t: t.data_ptr()
    # dst_node_chain_fwd
    good = [TL.node_mol_ptr, P(f(Nn, 256)), P(f(Nn, 256)), P(ada), T.ADA, 0, 256, 512, 768, P(hb(512, 256)), P(f(512)), P(hb(256, 512)), P(f(256)), P(hb(512, 256)),
            P(hb(64, 256)), P(f(64)), 0.1, 1, 2, 0, 7, 0, 0, 0, 0, 0, 0, P(f(Nn, 256)), P(f(Nn, 512)), P(f(Nn, 64))]
    call = lambda a: lib.dst_node_chain_fwd(C.byref(TL.c), T._NODE_PACK(*a), E._stream())
    assert call(good) == 0
    for idx, val in ((1, 0), (1, good[1] + 4), (5, 2), (9, 0), (16, 1.0), (27, 0)):     # NULL h_in, misaligned h_in, gate offset 2, NULL W1, p = 1, NULL h_out
        bad = list(good)
        bad[idx] = val
        assert call(bad) == -1, idx
    # dst_dir_chain_bwd / dst_pair_chain_bwd: the tile tables and scratch are required
    tt = TL.dir_tiles
    part = f(max(tt[4], 1) * 512)
    good = [tt[0], tt[1], tt[2], tt[3], tt[4], P(f(D, 3)), P(f(D, 256)), P(f(D, 256)), P(f(D, 2)), P(ada), P(f(TL.B, T.ADA)), T.ADA, 0, 256, P(f(3, 256)),
            P(hb(256, 256)), P(f(D, 256)), P(f(D, 256)), P(part)]
    call = lambda a: lib.dst_dir_chain_bwd(C.byref(TL.c), T._DIRB_PACK(*a), E._stream())
    assert call(good) == 0
    for idx, val in ((0, 0), (4, -1), (6, good[6] + 8), (12, 6), (15, 0), (18, 0)):      # NULL table, negative tile count, misaligned c0, offset 6, NULL W0T, NULL scratch
        bad = list(good)
        bad[idx] = val
        assert call(bad) == -1, idx
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ config 5 AS BENCHMARKED vs G17
def _group_of(name):
    if name.startswith("cond_encoder."):
        return "SpecFormer"
    if name.startswith("e_block_"):
        return "DMT blocks"
    if name.startswith(("node_pred_mlp.", "edge_exist_mlp.", "edge_type_mlp.", "node_", "edge_")) and not name.startswith(("node_emb", "edge_emb")):
        return "readouts"
    return "embeddings + time / adaLN tables"


def test_config5_as_benchmarked_against_the_reference_loss(gpu_device, monkeypatch):
    """BASELINE config 5 in the form ``bench.py --mode train`` times it - ALL-SPECTRA, bf16 GEMM operands, SpecFormer attention without
    score tensors (the flash kernels are what bf16 mode runs), FF dropout 0.1 with the kernels' own Philox masks, three streams - against
    golden G17 = the reference's ``loss_fn`` (losses.py:286-396) run with exactly those masks in fp32.  bf16 is not the reference's
    arithmetic, so the gates are the ones a mixed-precision run can be held to: loss within 1 %, gradient cosine > 0.995 for EVERY
    parameter group (DMT blocks / SpecFormer / readouts / embeddings + tables) on the golden's strided samples and on the stored full
    gradients, group gradient norms within 3 %, BatchNorm running statistics within 1e-3."""
    import os
    from diffspectra_amd import losses as Lh
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    assert os.environ.get("DIFFSPECTRA_NODE_STREAM", "1") == "1" and os.environ.get("DIFFSPECTRA_ASYNC_DW", "1") == "1"
    d = gpu_device
    version, coin_name = "allspectra", "plain"
    cfg, model = _train_model(version, d)
    cfg.model.dropout = 0.1
    cfg.training.precision = "bf16"
    g = cases.load_npz("g17_training_dropout.npz")
    tag = f"{version}_{coin_name}"
    batch, draws = cases.training_batch(version), cases.training_draws()
    batch = {k: v for k, v in batch.items() if k != "n_atoms"}
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, None, cfg)
    monkeypatch.setattr(torch, "rand", _Replay([draws["t_raw"]]))
    monkeypatch.setattr(torch, "randn", _Replay(draws["randn"]))
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.tensor(list(cases.TRAIN_DROPOUT_SEEDS)))
    monkeypatch.setattr(Lh, "random", lambda: 1.0)
    loss = loss_fn(model, batch)
    monkeypatch.undo()
    loss.backward()
    tr = model.module._hip_trainer
    assert tr.ops.bf16                                                               # bf16 products (and with them the flash attention kernels)
    ref_loss = float(g[tag + "_loss"])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-2 * abs(ref_loss), (float(loss.detach()), ref_loss)
    names = json.loads(g[tag + "_grad_names"])
    norms = g[tag + "_grad_norms"].numpy()
    grads = {n: p.grad for n, p in model.module.named_parameters()}
    acc = {}
    for i, n in enumerate(names):
        gr = grads[n]
        if gr is None:
            continue
        gr = gr.detach().cpu()
        assert torch.isfinite(gr).all(), n
        grp = acc.setdefault(_group_of(n), dict(dot=0.0, a=0.0, b=0.0, n_hip=0.0, n_ref=0.0, count=0))
        idx = torch.linspace(0, gr.numel() - 1, min(64, gr.numel())).round().long()
        a, b = gr.reshape(-1)[idx].double(), g[tag + "_grad_samples"][i][:len(idx)].double()
        if n in cases.TRAIN_FULL_GRADS:                                              # whole tensors where the golden stores them
            a, b = gr.reshape(-1).double(), g[f"{tag}_grad::{n}"].reshape(-1).double()
        grp["dot"] += float((a * b).sum()); grp["a"] += float((a * a).sum()); grp["b"] += float((b * b).sum())
        grp["n_hip"] += float(gr.double().pow(2).sum()); grp["n_ref"] += float(norms[i]) ** 2
        grp["count"] += 1
    assert set(acc) == {"SpecFormer", "DMT blocks", "readouts", "embeddings + time / adaLN tables"}, sorted(acc)
    for name, r in acc.items():
        cos = r["dot"] / max(1e-300, (r["a"] * r["b"]) ** 0.5)
        ratio = (r["n_hip"] / r["n_ref"]) ** 0.5
        print(f"[config 5 as benchmarked | {name}] {r['count']} tensors, gradient cosine {cos:.5f}, norm ratio {ratio:.4f}")
        assert cos > 0.995, (name, cos)
        assert abs(ratio - 1.0) < 0.03, (name, ratio)
    bn = "cond_encoder.backbone.encoder.layers.0.norm_attn.1."
    bufs = dict(model.module.named_buffers())
    check(bufs[bn + "running_mean"], g[tag + "_bn_running_mean"], 1e-3, "BatchNorm running_mean (bf16 products)")
    check(bufs[bn + "running_var"], g[tag + "_bn_running_var"], 1e-3, "BatchNorm running_var (bf16 products)")
    print(f"[config 5 as benchmarked] loss {float(loss.detach()):.5f} (reference, fp32: {ref_loss:.5f})")


def _synthetic_train_batch(Bt, version, seed, d):
    """A training batch in CollateSpectra's format (build_dataset.py:357-395) from the synthetic size histogram."""
    from diffspectra_amd import filler
    n_atoms = filler.sample_n_atoms(Bt, seed=seed).tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    g = torch.Generator().manual_seed(100 + seed)
    types = torch.randint(0, 5, (Bt, N), generator=g)
    order = torch.triu((torch.rand(Bt, N, N, generator=g) > 0.8).float() * torch.randint(1, 4, (Bt, N, N), generator=g), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(Bt, N, N)
    ctx = filler.synthetic_spectra(Bt, version, seed=5 + seed)
    pos = torch.randn(Bt, N, 3, generator=g) * 1.3 * node_mask
    pos = pos - pos.sum(1, keepdim=True) / node_mask.sum(1, keepdim=True) * node_mask            # centred, as the dataset's transform leaves it
    return dict(positions=pos.to(d), atom_mask=node_mask.squeeze(-1).to(d), edge_mask=edge_mask.to(d),
                atom_one_hot=(F.one_hot(types, 5).float() * node_mask).to(d),
                edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1).to(d), formal_charges=torch.zeros(Bt, N, 1, device=d),
                context=[c.to(d) for c in ctx] if isinstance(ctx, list) else ctx.to(d))


def _train_run(d, version, precision, steps, batches, lr=2e-4, seed=7, perturb=0.0):
    """``steps`` optimizer steps of the product's step_fn (fused AdamW + clip + EMA) from the procedural weights, every source of randomness
    seeded; returns (losses, model, ema, state)."""
    import random as _random
    from diffspectra_amd import losses as Lh
    from diffspectra_amd.ema import ExponentialMovingAverage
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    cfg, model = _train_model(version, d)
    cfg.model.dropout = 0.1
    cfg.training.precision = precision
    cfg.optim.lr, cfg.optim.warmup = lr, 0
    if perturb:
        with torch.no_grad():
            for prm in model.parameters():
                prm.mul_(1.0 + perturb)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
    opt = Lh.get_optimizer(cfg, model.parameters())
    step_fn = Lh.get_step_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, Lh.optimization_manager(cfg), None, cfg)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    torch.manual_seed(seed)
    rng = _random.Random(seed)
    old = Lh.random
    Lh.random = rng.random
    try:
        losses = [step_fn(state, batches[k % len(batches)]).detach() for k in range(steps)]
    finally:
        Lh.random = old
    return [float(v) for v in losses], cfg, model, ema, state


def test_bf16_and_fp32_training_curves_agree(gpu_device):
    """50 optimizer steps of config 5's step (all-spectra, dropout 0.1, the shipped peak lr 2e-4, no warm-up - the shipped 100 000-step
    warm-up would leave the weights untouched) in bf16 mode and in fp32 mode from the same weights, batches and seeds (noise, diffusion
    times, self-conditioning coins, dropout masks).  AdamW amplifies ANY rounding difference (an element whose gradient is rounding noise
    still moves by +-lr), so two runs of this length decorrelate whatever their arithmetic; measured here: an fp32 run whose initial
    weights are scaled by 1 + 1e-6 leaves the unperturbed fp32 curve by several per cent.  The test therefore holds bf16 to
      (a) the first optimizer steps, where the curves are still a property of the arithmetic: each of the first 3 losses within 0.5 %;
      (b) the chaos baseline over the whole run: the largest deviation of a 10-step mean between bf16 and fp32 is no more than twice the
          deviation between the two fp32 runs (or 2 %, whichever is larger), and every curve falls."""
    d = gpu_device
    batches = [_synthetic_train_batch(24, "allspectra", s, d) for s in range(4)]
    a = np.asarray(_train_run(d, "allspectra", "fp32", 50, batches)[0])
    a2 = np.asarray(_train_run(d, "allspectra", "fp32", 50, batches, perturb=1e-6)[0])
    b = np.asarray(_train_run(d, "allspectra", "bf16", 50, batches)[0])
    assert np.isfinite(a).all() and np.isfinite(a2).all() and np.isfinite(b).all()
    m = lambda v: v.reshape(5, 10).mean(1)
    dev = lambda x, y: float((np.abs(m(x) - m(y)) / np.abs(m(x))).max())
    chaos, gap = dev(a, a2), dev(a, b)
    print(f"[50 steps] fp32 10-step means            {np.round(m(a), 3).tolist()}\n[50 steps] fp32, weights x (1 + 1e-6)    {np.round(m(a2), 3).tolist()}\n"
          f"[50 steps] bf16 10-step means            {np.round(m(b), 3).tolist()}\n[50 steps] largest deviation of a 10-step mean: fp32 vs perturbed fp32 "
          f"{chaos:.4f}, fp32 vs bf16 {gap:.4f}; first three losses fp32 {np.round(a[:3], 4).tolist()} bf16 {np.round(b[:3], 4).tolist()}")
    assert np.all(np.abs(a[:3] - b[:3]) <= 5e-3 * np.abs(a[:3])), (a[:3], b[:3])
    assert gap <= max(0.02, 2.0 * chaos), (gap, chaos)
    for v in (a, a2, b):
        assert m(v)[-1] < 0.7 * m(v)[0]


def test_split_fp16_sampling_parity_on_weights_the_hip_trainer_produced(gpu_device):
    """The split-fp16 sampling kernels on weights AdamW actually produced (the published checkpoint is unreachable; this is the closest
    substitute the repo can make itself): 200 optimizer steps of the product's training step (ir, dropout 0.1, lr 2e-4) -> EMA weights
    -> (a) single forwards, first-step and general branch, HIP against the fp32 CPU oracle at the standing 2e-5 gate; (b) a 50-step
    injected-noise trajectory at the standing 5e-4 gate with the post-processed integer outputs compared decision by decision.
    Prints the largest |weight| the split packing saw and the largest |activation| of the oracle's forward."""
    import oracle
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.scalers import get_data_inverse_scaler
    d = gpu_device
    batches = [_synthetic_train_batch(32, "ir", s, d) for s in range(8)]
    losses, cfg, model, ema, state = _train_run(d, "ir", "fp32", 200, batches)
    assert all(math.isfinite(v) for v in losses) and state["step"] == 200
    first, last = float(np.mean(losses[:20])), float(np.mean(losses[-20:]))
    init = {k: v.detach().clone() for k, v in model.module.state_dict().items()}
    ema.copy_to(model.parameters())
    model.module.invalidate_engine()
    model.eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.module.state_dict().items()}
    cpu_cfg, sd0 = procedural_state_dict("ir")
    moved = max(float((sd[k].float() - sd0[k].float()).abs().max()) for k in sd0 if sd[k].is_floating_point())
    wmax = max(float(v.abs().max()) for k, v in sd.items() if v.is_floating_point() and v.dim() >= 2)
    print(f"[trained weights] loss {first:.3f} -> {last:.3f} over 200 steps; EMA weights moved by up to {moved:.4f} from the initial ones; "
          f"largest |weight| of a matrix {wmax:.4f}")
    assert last < first and moved > 1e-3
    amax = 0.0
    for first_step in (True, False):
        a = cases.forward_inputs("ir", first_step)
        ctx_cpu = oracle.context_embedding(sd, a["context"], cpu_cfg)
        ref_xh, ref_edge, dbg = oracle.dmt_forward(sd, cpu_cfg, a["xh"], a["node_mask"], a["edge_mask"], a["edge_x"], a["noise_level"],
                                                   a["cond_x"], a["cond_edge_x"], context_emb=ctx_cpu, return_debug=True)
        flat = []
        def walk(v):
            if torch.is_tensor(v):
                flat.append(v)
            elif isinstance(v, dict):
                [walk(x) for x in v.values()]
            elif isinstance(v, (list, tuple)):
                [walk(x) for x in v]
        walk(dbg)
        amax = max([amax] + [float(t.abs().max()) for t in flat if t.is_floating_point() and t.numel()])
        dev = lambda t: None if t is None else ([x.to(d) for x in t] if isinstance(t, (list, tuple)) else t.to(d))
        B = a["xh"].shape[0]
        xh, ef = model(torch.zeros(B, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d), context=dev(a["context"]),
                       edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d), cond_x=dev(a["cond_x"]), cond_edge_x=dev(a["cond_edge_x"]))
        ex, ee = float((xh.cpu() - ref_xh).abs().max()), float((ef.cpu() - ref_edge).abs().max())
        print(f"[trained weights] forward ({'first step' if first_step else 'general'}): max |HIP - oracle| nodes {ex:.2e}, edges {ee:.2e} (gate 2e-5)")
        assert ex <= 2e-5 + 1e-5 * float(ref_xh.abs().max()) and ee <= 2e-5 + 1e-5 * float(ref_edge.abs().max())
    print(f"[trained weights] largest |activation| in the oracle's forward {amax:.2f} (split-fp16 is exact below 65 520)")
    # 50-step trajectory, injected noise
    steps = 50
    tr = cases.trajectory_inputs("ir", steps)
    scfg = cfg.clone()
    scfg.sampling.steps = steps
    sampler = S._make_sampler(scfg, NoiseScheduleVP("cosine"), 1e-3, 1.0)
    sampler.noise_fn = lambda i: tr["raws"][i]
    z = oracle.combined_noise(*tr["raw0"][:2], tr["node_mask"])
    ez = oracle.symmetric_edge_noise(tr["raw0"][2], tr["edge_mask"])
    ctx_dev = [c.to(d) for c in tr["context"]] if isinstance(tr["context"], (list, tuple)) else tr["context"].to(d)
    x_mean, e_mean = sampler.sampling(model, z.to(d), tr["node_mask"].to(d), tr["edge_mask"].to(d), ez.to(d), ctx_dev)
    ctx_cpu = oracle.context_embedding(sd, tr["context"], cpu_cfg)
    model_fn = lambda x, ex_, nl, cx, cex: oracle.dmt_forward(sd, cpu_cfg, x, tr["node_mask"], tr["edge_mask"], ex_, nl, cx, cex, context_emb=ctx_cpu)
    rx, re_ = oracle.ancestral_sampling(model_fn, z, tr["node_mask"], tr["edge_mask"], ez, steps, lambda i: tr["raws"][i])
    drift = max(float((x_mean.cpu() - rx).abs().max()), float((e_mean.cpu() - re_).abs().max()))
    _, r_hot, r_fc, r_et = oracle.post_process(rx, tr["node_mask"], re_, tr["edge_mask"])
    eng = model.module.engine()
    _, h_hot, h_fc, h_et = S.post_process(x_mean, 5, True, tr["node_mask"].to(d), get_data_inverse_scaler(scfg), e_mean, tr["edge_mask"].to(d), True, engine=eng)
    nm = tr["node_mask"].squeeze(-1).bool()
    em = tr["edge_mask"].reshape(r_et.shape).bool()
    mis = (int((h_hot.argmax(-1).cpu() != r_hot.argmax(-1))[nm].sum()), int((h_fc.reshape(nm.shape).cpu().long() != r_fc.reshape(nm.shape).long())[nm].sum()),
           int((h_et.cpu() != r_et)[em].sum()))
    print(f"[trained weights] 50-step trajectory drift {drift:.2e} (gate 5e-4); decisions {int(nm.sum())} atom types / charges, {int(em.sum())} bond orders; "
          f"mismatches (type, charge, bond) {mis}")
    assert drift <= 5e-4 and mis == (0, 0, 0)
    del init


@pytest.mark.parametrize("L", [347, 69])
def test_specformer_flash_attention_vs_torch_fp64(gpu_device, L):
    """The score-free SpecFormer attention kernels of bf16 mode (``dst_spec_attn_flash_fwd / _bwd``) against an INDEPENDENT reference: torch
    autograd in fp64 of specformer.py:385-425's residual attention written out (scores_l = scale q_l k_l^T + scores_{l-1}, softmax,
    P V) over three chained layers.  The kernels round q, k, v and the probabilities to bf16 (fp32 accumulation), so the gate is the
    bf16 level: relative L2 error <= 1.5 % for every layer's output and every layer's dq / dk / dv, gradient cosine > 0.9998."""
    import ctypes as C
    from diffspectra_amd import engine as E, train_engine as T
    lib = T.load_train_library()
    d = gpu_device
    B, H, DK, DM = 3, 16, 8, 128
    gen = torch.Generator().manual_seed(L)
    qkv = [torch.randn(B * L, 3 * DM, generator=gen).to(d) for _ in range(3)]
    dao = [torch.randn(B * L, DM, generator=gen).to(d) for _ in range(3)]
    scale = DK ** -0.5
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=d)
    outs, stats = [], []
    for l in range(3):
        ast, out = f(B, H, L, 2), f(B * L, DM)
        qp = [E._ptr(q) for q in qkv[:l + 1]] + [None] * (2 - l)
        E._check(lib.dst_spec_attn_flash_fwd(qp[0], qp[1], qp[2], C.c_int32(l + 1), E._ptr(ast), E._ptr(out), C.c_int32(B), C.c_int32(L), C.c_int32(H),
                                             C.c_int32(DK), C.c_float(scale), E._stream()), "dst_spec_attn_flash_fwd")
        outs.append(out); stats.append(ast)
    dq = [torch.full((B * L, 3 * DM), float("nan"), device=d) for _ in range(3)]      # never zeroed: the first call (accumulate = 0) assigns
    for l in (2, 1, 0):
        qp = [E._ptr(q) for q in qkv[:l + 1]] + [None] * (2 - l)
        gp = [E._ptr(q) for q in dq[:l + 1]] + [None] * (2 - l)
        E._check(lib.dst_spec_attn_flash_bwd(qp[0], qp[1], qp[2], C.c_int32(l + 1), E._ptr(stats[l]), E._ptr(outs[l]), E._ptr(dao[l]), gp[0], gp[1], gp[2],
                                             C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), C.c_int32(0), C.c_int32(0 if l == 2 else 1), E._stream()), "dst_spec_attn_flash_bwd")
    torch.cuda.synchronize()
    qr = [q.double().cpu().clone().requires_grad_(True) for q in qkv]
    loss, s_prev, ref_out = 0.0, 0.0, []
    for l in range(3):
        x = qr[l].view(B, L, 3, H, DK)
        q_, k_, v_ = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        s_ = scale * q_ @ k_.transpose(-1, -2) + s_prev
        o = (torch.softmax(s_, -1) @ v_).permute(0, 2, 1, 3).reshape(B * L, DM)
        ref_out.append(o.detach())
        loss = loss + (o * dao[l].double().cpu()).sum()
        s_prev = s_
    loss.backward()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / (b.norm() + 1e-30))
    worst = 0.0
    for l in range(3):
        gr = qr[l].grad
        errs = dict(out=rel(outs[l], ref_out[l]), dq=rel(dq[l][:, :128], gr[:, :128]), dk=rel(dq[l][:, 128:256], gr[:, 128:256]), dv=rel(dq[l][:, 256:], gr[:, 256:]))
        cos = float((dq[l].double().cpu() * gr).sum() / (dq[l].double().cpu().norm() * gr.norm()))
        print(f"[flash attention vs torch fp64, L = {L}] layer {l}: " + ", ".join(f"{k} {v:.4f}" for k, v in errs.items()) + f"; gradient cosine {cos:.6f}")
        worst = max(worst, *errs.values())
        assert cos > 0.9998, (l, cos)
    assert worst <= 1.5e-2, worst


def test_fused_pair_chain_matches_the_unfused_kernels(gpu_device, monkeypatch):
    """``dst_pair_front_fwd`` / ``dst_pair_chain_fwd`` / ``dst_dir_chain_fwd`` (bf16 mode: the pair rows of a block in front of / behind the
    attention and its directed rows as one kernel each) against the per-operation kernels they replace, on a ragged batch incl. n = 29, 2 and 1, dropout 0.1: every tape tensor of every
    block - X1, x', d2, e1, en, te; he, xe1, the LayerNorm statistics, ye1, f3, s3, f4, e_out, X2, ed, the read-out slice - and the three
    outputs of the forward.  Same arithmetic, same Philox masks (tolerances: see below); then the backward over both tapes: gradient
    cosine > 0.9999, every gradient tensor within 3 % in L2."""
    from diffspectra_amd import filler, train_engine as T
    d = gpu_device
    cfg, sd0 = procedural_state_dict("ir")
    params = {k: v.detach().to(d).contiguous() for k, v in sd0.items() if not k.startswith("cond_encoder.") and v.is_floating_point()}
    n_atoms = [29, 2, 1, 18, 9, 23, 3, 12]
    node_mask, _ = filler.masks_from_n_atoms(n_atoms)
    TL = T.TrainLayout(node_mask, d)
    g = torch.Generator().manual_seed(5)
    xn, ex = torch.randn(TL.Nn, 9, generator=g).to(d), torch.randn(TL.Pp, 2, generator=g).to(d)
    cn, ce = torch.randn(TL.Nn, 9, generator=g).to(d), torch.randn(TL.Pp, 2, generator=g).to(d)
    nl = (torch.rand(TL.B, generator=g) * 8 - 4).to(d)
    ctx = (torch.randn(TL.B, 1024, generator=g) * 0.5).to(d)
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DIFFSPECTRA_FUSED_CHAIN", mode)
        graph = T.DmtTrainGraph(params, cfg, d)
        graph.ops.bf16 = True
        graph.dropout_p, graph.dropout_seed = 0.1, 987654321
        out = graph.forward(TL, xn, ex, nl, ctx, cn, ce)
        tape = [{k: bt[k].clone() for k in ("X1", "xs", "d2", "e1", "st_e1", "en", "te", "he", "xe1", "st_e2", "ye1", "f3", "s3", "f4", "e_out", "X2", "ed",
                                            "re_", "zz", "st_z", "zn", "c0", "sc0", "c2",
                                            "x1", "st_n2", "y1", "f1", "s1", "f2", "h_out", "ac", "rn")} for bt in graph.t["blocks"]]     # (+ the node chain)
        dpos, datom, dedge = (torch.randn(o.shape, generator=torch.Generator().manual_seed(9)).to(d) for o in out)
        grads = {k: v.clone() for k, v in graph.backward(dpos, datom, dedge).items()}
        runs[mode] = ([o.clone() for o in out], tape, grads)
    # What can be asked of two bf16 evaluations of the same chain: a product whose operands are bit-identical agrees to the fp32 accumulation
    # order (block 0: X1, x', d2 exactly; e1 to 2e-5); everything downstream sees operands that differ in the last fp32 bits, and rounding
    # those to bf16 moves an element by a whole bf16 step (2^-8) now and then - so the later tensors are held to the bf16 level: largest
    # deviation <= 2 % of the tensor's scale, mean deviation <= 1e-3 of it.  The dropout zero patterns must be identical throughout.
    worst, worst_mean = ("", 0.0), ("", 0.0)
    for i, (ta, tb) in enumerate(zip(runs["0"][1], runs["1"][1])):
        for k in ta:
            ref, got = ta[k], tb[k]
            scale = float(ref.abs().max()) + 1e-30
            err, mean_err = float((got - ref).abs().max()) / scale, float((got - ref).abs().mean()) / scale
            if err > worst[1]:
                worst = (f"block {i} {k}", err)
            if mean_err > worst_mean[1]:
                worst_mean = (f"block {i} {k}", mean_err)
            if i == 0 and k in ("X1", "xs", "d2"):
                assert err <= 1e-6, (i, k, err)
            elif i == 0 and k in ("e1", "st_e1", "en"):
                assert err <= 2e-5, (i, k, err)
            else:
                assert err <= 2e-2 and mean_err <= 1e-3, (i, k, err, mean_err)
            if k in ("s3", "f4", "s1", "f2"):
                # the masks are a function of (seed, stream, element) alone: both runs must zero exactly the elements the numpy restatement of
                # dst_dropout drops (an element it keeps may still be zero: SiLU underflows below -88)
                from oracle import philox
                keep = torch.from_numpy(philox.dropout_keep(987654321, 4 * i + dict(s1=0, f2=1, s3=2, f4=3)[k], ref.numel(), 0.1)).reshape(ref.shape)
                for name, t in (("unfused", ref), ("fused", got)):
                    tz = (t == 0).cpu()
                    dropped_but_alive = int((~keep & ~tz).sum())
                    assert dropped_but_alive == 0, (i, k, name, dropped_but_alive)
                    kept_zero = keep & tz
                    if k in ("s3", "s1") and bool(kept_zero.any()):
                        pre = (tb if name == "fused" else ta)["f3" if k == "s3" else "f1"].cpu()[kept_zero]
                        assert float(pre.max()) < -80.0 or float(pre.abs().max()) == 0.0, (i, k, name, "kept element is zero", pre[:4])
    for a, b in zip(runs["0"][0], runs["1"][0]):
        assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max())
    dot = na = nb = 0.0
    gbad = []
    for k, ga in runs["0"][2].items():
        gb = runs["1"][2][k]
        ga_, gb_ = ga.double(), gb.double()
        dot, na, nb = dot + float((ga_ * gb_).sum()), na + float((ga_ * ga_).sum()), nb + float((gb_ * gb_).sum())
        if float((ga_ - gb_).norm()) > 3e-2 * float(ga_.norm()) + 1e-9:
            gbad.append((k, float((ga_ - gb_).norm()), float(ga_.norm())))
    cos = dot / (na * nb) ** 0.5
    print(f"[fused pair chains] worst tape deviation {worst}, worst mean deviation {worst_mean}; {len(runs['0'][2])} gradients compared, cosine {cos:.7f}")
    assert cos > 0.9999, cos
    assert not gbad, gbad[:8]
