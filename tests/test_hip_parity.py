"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the committed golden vectors.

Tolerances (fp32): per-kernel 1e-5 abs (+1e-5 rel), single forward 2e-5, 50-step trajectory 5e-4;
integer outputs (atom type argmax, charges, bond orders) must be bit-exact.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from oracle import dmt as odmt
from tests.golden import cases
from tests.helpers import procedural_state_dict, max_abs_diff

pytestmark = pytest.mark.gpu

TOL_KERNEL = 1e-5
TOL_FORWARD = 2e-5
TOL_TRAJ = 5e-4


def close(a, b, atol, rtol=1e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = err > lim
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        return False, f"max err {float(err.max()):.3e} (limit {atol:g}); first bad index {idx}: {float(a[tuple(idx)])} vs {float(b[tuple(idx)])}; n_bad {int(bad.sum())}/{bad.numel()}"
    return True, f"max err {float(err.max()):.3e}"


def assert_close(a, b, atol, what="", rtol=1e-5):
    ok, msg = close(a, b, atol, rtol)
    assert ok, f"{what}: {msg}"


_MODELS = {}


def gpu_model(version, device):
    if version not in _MODELS:
        from diffspectra_amd import filler
        from diffspectra_amd.config import qm9s_config
        from diffspectra_amd.registry import create_model
        import diffspectra_amd.dmt  # noqa: F401
        cfg = qm9s_config(version, device=device)
        model = create_model(cfg)
        filler.fill_module_(model)
        model.eval()
        _MODELS[version] = (cfg, model)
    return _MODELS[version]


def to_dev(a, d):
    if a is None:
        return None
    if isinstance(a, (list, tuple)):
        return [x.to(d) for x in a]
    return a.to(d)


import contextlib


@contextlib.contextmanager
def swapped_weights(model, transform):
    """Run a block with ``transform(state_dict)`` loaded into the shared model, then restore the procedural weights."""
    original = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(transform(original), strict=True)      # the engine re-packs on the next call
    try:
        yield
    finally:
        model.load_state_dict(original, strict=True)


def integer_parity_report(tag, x_mean, e_mean, g, node_mask, edge_mask, got_atom, got_fc, got_et):
    """Margin statistics and mismatch counts of the integer outputs (SURVEY §7 'hard parts').

    A decision's margin is its distance (in model-output units) from the threshold that would flip it, measured on the
    REFERENCE trajectory; ``drift`` is the largest |HIP - reference| on the same tensors.  The committed fixtures must be
    reproduced EXACTLY (north_star: bit-exact argmax): zero mismatches, whatever the margin.  The margin histogram is printed
    as a diagnostic of how much head-room that exactness has."""
    rx, re_ = g[tag + "_x_mean"], g[tag + "_edge_mean"]
    nm = node_mask.squeeze(-1).bool().cpu()
    em = edge_mask.reshape(re_.shape[:3]).bool().cpu()
    drift = max(float((x_mean.cpu() - rx).abs().max()), float((e_mean.cpu() - re_).abs().max()))
    top2 = rx[:, :, 3:8].topk(2, -1).values
    m_atom = (top2[..., 0] - top2[..., 1])[nm]
    c4 = rx[:, :, 8] * 4.0
    m_fc = ((0.5 - (c4 - c4.round()).abs()) / 4.0)[nm]
    e0, e1 = re_[..., 0], re_[..., 1]
    m_exist = e0.abs()[em]
    m_type = torch.stack([(e1 + 2.0 / 3.0).abs(), e1.abs(), (e1 - 2.0 / 3.0).abs()]).min(0).values[em]
    mis_atom = (got_atom.cpu() != g[tag + "_atom_type"])[nm]
    mis_fc = (got_fc.squeeze(-1).cpu() != g[tag + "_fc"].squeeze(-1).long())[nm]
    mis_et = (got_et.cpu() != g[tag + "_edge_type"])[em]
    m_bond = torch.minimum(m_exist, m_type)
    edges = [0.0, 1e-4, 1e-3, 1e-2, 1e-1, float("inf")]
    print(f"[{tag}] drift {drift:.3e} (gate {TOL_TRAJ:g})")
    for name, marg, mis in (("atom type", m_atom, mis_atom), ("charge", m_fc, mis_fc), ("bond order", m_bond, mis_et)):
        hist = [int(((marg >= lo) & (marg < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        print(f"[{tag}] {name:10s} decisions {marg.numel():5d}  margin min {float(marg.min()):.2e} median {float(marg.median()):.2e}  "
              f"histogram [<1e-4,<1e-3,<1e-2,<1e-1,>=1e-1] {hist}  mismatches {int(mis.sum())}")
        assert int(mis.sum()) == 0, f"{tag}: {int(mis.sum())} {name} decisions differ from the reference (closest margin {float(marg.min()):.2e}, drift {drift:.2e})"
    cls = g[tag + "_atom_type"][nm].unique().tolist()
    orders = g[tag + "_edge_type"][em].unique().tolist()
    print(f"[{tag}] reference atom types {cls}, bond orders {orders}, non-zero charges {int((g[tag + '_fc'].squeeze(-1)[nm] != 0).sum())}")
    return drift


def run_injected_trajectory(model, cfg, version, steps, n_atoms, d):
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    tr = cases.trajectory_inputs(version, steps) if n_atoms is None else cases.trajectory_inputs(version, steps, n_atoms)
    sampler = S._make_sampler(cfg, NoiseScheduleVP("cosine"), 1e-3, 1.0)
    sampler.noise_fn = lambda i: tr["raws"][i]
    z = oracle.combined_noise(*tr["raw0"][:2], tr["node_mask"])
    ez = oracle.symmetric_edge_noise(tr["raw0"][2], tr["edge_mask"])
    x_mean, e_mean = sampler.sampling(model, z.to(d), tr["node_mask"].to(d), tr["edge_mask"].to(d), ez.to(d),
                                      to_dev(tr["context"], d))
    return tr, x_mean, e_mean


def check_trajectory_golden(gpu_device, fixture, version, steps, n_atoms=None, tag_suffix=""):
    """Injected-noise ancestral trajectory on the HIP path vs the reference's own run, with the de-trivialised readout
    weights of the case (tests/golden/calibrate_diverse.py): tolerance on coordinates/logits, integer outputs exact
    wherever the decision margin allows, margins and mismatch counts printed."""
    from diffspectra_amd import sampling as S
    from diffspectra_amd.scalers import get_data_inverse_scaler
    cfg, model = gpu_model(version, gpu_device)
    cfg = cfg.clone()
    cfg.sampling.steps = steps
    g = cases.load_npz(fixture)
    tag = f"{version}_S{steps}{tag_suffix}"
    d = gpu_device
    with swapped_weights(model, lambda sd: cases.readout_diverse(sd, tag)):
        tr, x_mean, e_mean = run_injected_trajectory(model, cfg, version, steps, n_atoms, d)
        eng = model.module.engine()
        pos, one_hot, fc, et = S.post_process(x_mean, 5, True, tr["node_mask"].to(d), get_data_inverse_scaler(cfg), e_mean,
                                              tr["edge_mask"].to(d), True, engine=eng)
    ok_x, msg_x = close(x_mean, g[tag + "_x_mean"], TOL_TRAJ)
    ok_e, msg_e = close(e_mean, g[tag + "_edge_mean"], TOL_TRAJ)
    print(f"[{tag}] x_mean: {msg_x} | edge_mean: {msg_e}")
    assert ok_x, msg_x
    assert ok_e, msg_e
    integer_parity_report(tag, x_mean, e_mean, g, tr["node_mask"], tr["edge_mask"], one_hot.argmax(-1), fc, et)
    mols = S.mol_process(one_hot, pos, fc, tr["n_atoms"], et)
    for m, (p, at, e, c) in enumerate(mols):
        assert_close(p, g[f"{tag}_mol{m}_pos"], TOL_TRAJ, f"mol {m} pos")
        assert at.dtype == torch.int64 and e.dtype == torch.float32 and c.dtype == torch.int64
        assert at.shape == g[f"{tag}_mol{m}_atom"].shape and e.shape == g[f"{tag}_mol{m}_edge"].shape


# ------------------------------------------------------------------------------------------------ GEMM

@pytest.mark.parametrize("M,K,N,act", [(64, 64, 32, 0), (70, 24, 1024, 2), (5, 1024, 1024, 0), (130, 256, 96, 1),
                                       (33, 56, 128, 3), (257, 128, 384, 0), (4, 1000, 40, 0),
                                       (700, 256, 200, 1), (1024, 1024, 160, 0)])   # the last two take the 128x128-tile kernel
def test_gemm_vs_torch(gpu_device, M, K, N, act):
    """MFMA fragment layout, packing, K/N padding and guards of ds_gemm (asymmetric random operands)."""
    from diffspectra_amd import engine as E
    lib = E.load_library()
    g = torch.Generator().manual_seed(M * 1000 + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    cs, csh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    ref = torch.nn.functional.linear(A.double(), W.double(), b.double())
    ref = [lambda x: x, torch.nn.functional.silu, torch.nn.functional.gelu, torch.tanh][act](ref)
    ref2 = (ref + R.double()) * cs.double() + csh.double()
    d = gpu_device
    Wp = E.pack_linear(W).to(d)
    out = torch.full((M, N), float("nan"), device=d)
    E.gemm(lib, A.to(d), K, Wp, E.pad_vec(b).to(d), out, N, M, K, N, act=act)
    assert_close(out, ref, 2e-5, f"gemm {M}x{K}x{N} act{act}", rtol=2e-5)
    out2 = torch.full((M, N), float("nan"), device=d)
    E.gemm(lib, A.to(d), K, Wp, E.pad_vec(b).to(d), out2, N, M, K, N, act=act, R=R.to(d), ldr=N,
           col_scale=cs.to(d), col_shift=csh.to(d))
    assert_close(out2, ref2, 4e-5, f"gemm+residual+affine {M}x{K}x{N}", rtol=4e-5)


def test_split_fp16_gemm_accuracy(gpu_device):
    """The arithmetic every block GEMM now uses (ds_device.h wave_mma_h): operands split in two fp16 planes, three f16 MFMAs
    per product, fp32 accumulate.  Against an fp64 reference its error must be at the level of the fp32-MFMA GEMM's own
    rounding error (both measured here), over magnitudes from 1e-4 to 50 and K up to 1024 - and exact on small integers."""
    from diffspectra_amd import engine as E
    lib = E.load_library()
    d = gpu_device
    g = torch.Generator().manual_seed(123)
    for M, K, N, scale in ((200, 256, 96, 1.0), (64, 1024, 160, 1.0), (130, 64, 256, 50.0), (77, 128, 64, 1e-4)):
        A = torch.randn(M, K, generator=g) * scale
        A[:, ::7] *= 1e-3                                               # mixed magnitudes inside a row
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        ref = A.double() @ W.double().t() + b.double()
        out = torch.full((M, N), float("nan"), device=d)
        E.gemm_split(lib, E.split_rows_f16(A).to(d), E.pack_linear_f16_split(W).to(d), b.to(d), out, M, K, N)
        out32 = torch.full((M, N), float("nan"), device=d)
        E.gemm(lib, A.to(d), K, E.pack_linear(W).to(d), E.pad_vec(b).to(d), out32, N, M, K, N)
        norm = (A.double().abs() @ W.double().abs().t() + b.double().abs())   # scale of the terms that were summed
        e_split = float(((out.cpu().double() - ref).abs() / norm).max())
        e_fp32 = float(((out32.cpu().double() - ref).abs() / norm).max())
        print(f"[split-gemm {M}x{K}x{N} scale {scale:g}] max error / sum|terms|: split-fp16 {e_split:.2e}, fp32 MFMA {e_fp32:.2e}")
        assert e_split < 4e-7, e_split                                  # ~2^-22 (representation + dropped a2 b2) .. 2^-21
        assert e_split < 8 * max(e_fp32, 2e-8)
    A = torch.randint(-8, 9, (64, 64), generator=g).float()             # exact: integers are representable in one fp16 plane
    W = torch.randint(-8, 9, (32, 64), generator=g).float()
    out = torch.zeros(64, 32, device=d)
    E.gemm_split(lib, E.split_rows_f16(A).to(d), E.pack_linear_f16_split(W).to(d), None, out, 64, 64, 32)
    assert torch.equal(out.cpu(), A @ W.t())


def test_split_fp16_range_saturates(gpu_device):
    """The split-fp16 arithmetic has fp16's exponent range.  Activations at or beyond it SATURATE at +-65536 (MODE.FP16_OVFL in
    every converting kernel) instead of turning into inf - inf = NaN; weights beyond it are refused at pack time."""
    from diffspectra_amd import engine as E
    lib = E.load_library()
    d = gpu_device
    with pytest.raises(ValueError):
        E.pack_linear_f16_split(torch.full((32, 64), 7e4))
    with pytest.raises(ValueError):
        E.pack_ff4_chain(torch.full((64, 128), -1e5))
    E.pack_linear_f16_split(torch.full((32, 64), 65503.0))                       # the largest finite fp16 magnitudes are fine
    # host converter (what ds_gemm_split's callers use) = the kernels' saturating conversion
    A = torch.zeros(64, 64)
    A[0, 0], A[1, 1], A[2, 2], A[3, 3], A[4, 4] = 65519.0, 65520.0, 7e4, -1e6, 3e38
    sp = E.split_rows_f16(A)
    rec = sp[:, 0].float() + sp[:, 1].float() / 2048.0
    assert torch.isfinite(rec).all()
    sat = 65504.0 + 65504.0 / 2048.0                                             # both planes at fp16's largest finite value
    assert rec[0, 0] == 65519.0 and rec[1, 1] == 65520.0 and rec[2, 2] == sat and rec[3, 3] == -sat and rec[4, 4] == sat
    W = torch.eye(64)
    out = torch.full((64, 64), float("nan"), device=d)
    E.gemm_split(lib, sp.to(d), E.pack_linear_f16_split(W).to(d), None, out, 64, 64, 64)
    o = out.cpu()
    assert torch.isfinite(o).all() and o[0, 0] == 65519.0 and o[2, 2] == sat and o[3, 3] == -sat


def _oracle_forward_f64(sd, cfg, a):
    """The oracle in fp64 (same code, default dtype switched): the truth both fp32 evaluations are measured against."""
    torch.set_default_dtype(torch.float64)
    try:
        dd = lambda t: None if t is None else ([x.double() for x in t] if isinstance(t, (list, tuple)) else t.double())
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        return oracle.dmt_forward(sd64, cfg, dd(a["xh"]), dd(a["node_mask"]), dd(a["edge_mask"]), dd(a["edge_x"]), dd(a["noise_level"]),
                                  dd(a["cond_x"]), dd(a["cond_edge_x"]), context=dd(a["context"]))
    finally:
        torch.set_default_dtype(torch.float32)


def _checkpoint_like(sd):
    """Weight statistics a trained checkpoint can have and the U(+-1/sqrt(fan_in)) filler never does: adaLN scale / shift / gate
    outputs 30x larger (every block's node / edge / equi / dist time_mlp), a few residual-stream channels at 1e4 magnitude,
    weights with 1e-6 entries.  (The top-level 17 -> 1024 time MLP is left alone: scaling it as well makes the forward
    ill-conditioned - the fp32 CPU oracle itself is then 10 % away from fp64.)"""
    import re
    out = dict(sd)
    for key, v in sd.items():
        k = key[7:] if key.startswith("module.") else key
        if re.match(r"e_block_\d+\.(node_time_mlp|edge_time_mlp|equi_update\.time_mlp|dist_layer\.time_mlp)\.1\.", k):
            out[key] = v * 30.0
        elif k == "node_emb.weight":
            w = v.clone()
            w[[3, 77, 200]] *= 1e4                                      # three residual-stream channels ~1e4
            out[key] = w
        elif re.match(r"e_block_\d+\.(ff_linear1|ff_linear3|equi_update\.coord_mlp\.0)\.weight", k):
            w = v.clone()
            w[::3] *= 1e-5                                              # rows of ~1e-6 entries next to ordinary ones
            out[key] = w
    return out


def test_forward_with_checkpoint_like_statistics(gpu_device):
    """Range-proofing of the split-fp16 path: a forward with checkpoint-like weight statistics stays finite and
    within 2e-5 (relative to the output scale) of the fp64 truth, or within 8x the fp32 CPU oracle's own distance from fp64 where
    that is larger (measured: positions 1.3e-6 vs 4.8e-7 for the CPU; edge logits 4.6e-5 vs 8.5e-6)."""
    cfg, model = gpu_model("ir", gpu_device)
    d = gpu_device
    a = cases.forward_inputs("ir", False)
    with swapped_weights(model, _checkpoint_like):
        xh, ef = model(torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d),
                       context=to_dev(a["context"], d), edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d),
                       cond_x=to_dev(a["cond_x"], d), cond_edge_x=to_dev(a["cond_edge_x"], d))
        sd = {k[7:]: v.detach().cpu() for k, v in model.state_dict().items()}
    cpu_cfg = cases.config_for("ir")
    ref32 = oracle.dmt_forward(sd, cpu_cfg, a["xh"], a["node_mask"], a["edge_mask"], a["edge_x"], a["noise_level"], a["cond_x"],
                               a["cond_edge_x"], context=a["context"])
    ref64 = _oracle_forward_f64(sd, cpu_cfg, a)
    assert torch.isfinite(xh).all() and torch.isfinite(ef).all()
    for name, got, r32, r64 in (("xh", xh, ref32[0], ref64[0]), ("edge", ef, ref32[1], ref64[1])):
        scale = max(1.0, float(r64.abs().max()))
        e_hip = float((got.cpu().double() - r64).abs().max()) / scale
        e_cpu = float((r32.double() - r64).abs().max()) / scale
        print(f"[checkpoint-like {name}] output scale {scale:.3g}: HIP vs fp64 {e_hip:.2e}, fp32 CPU oracle vs fp64 {e_cpu:.2e}")
        # 2e-5 of the output scale, or - where the amplified intermediates put the fp32 CPU evaluation itself above 2.5e-6 - no
        # more than 8x that evaluation's own distance from fp64
        assert e_hip <= max(2e-5, 8 * e_cpu), (name, e_hip, e_cpu)


def test_forward_saturates_instead_of_nan(gpu_device):
    """Activations driven far beyond 65504 (edge embedding x 1e7, node embedding x 1e6): the fp32 reference stays finite, and so
    must the split-fp16 path - saturated operands, no inf - inf = NaN, NaN guard not triggered."""
    cfg, model = gpu_model("ir", gpu_device)
    d = gpu_device
    a = cases.forward_inputs("ir", False)

    def huge(sd):
        out = dict(sd)
        for k, v in sd.items():
            if k.endswith("module.edge_emb.weight") or k == "edge_emb.weight":
                out[k] = v * 1e7
            elif k.endswith("module.node_emb.weight") or k == "node_emb.weight":
                out[k] = v * 1e6
        return out

    with swapped_weights(model, huge):
        xh, ef = model(torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d),
                       context=to_dev(a["context"], d), edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d),
                       cond_x=to_dev(a["cond_x"], d), cond_edge_x=to_dev(a["cond_edge_x"], d))
        eng = model.module.engine()
        L, ws = eng.layout_for(a["node_mask"].to(d), a["edge_mask"].to(d))
        flags = ws.t["flags"].cpu()
        sd = {k[7:]: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = oracle.dmt_forward(sd, cases.config_for("ir"), a["xh"], a["node_mask"], a["edge_mask"], a["edge_x"], a["noise_level"],
                             a["cond_x"], a["cond_edge_x"], context=a["context"])
    assert torch.isfinite(ref[0]).all() and torch.isfinite(ref[1]).all()          # the fp32 reference is finite here
    assert torch.isfinite(xh).all() and torch.isfinite(ef).all(), "split-fp16 path produced inf/NaN where fp32 is finite"
    assert int(flags[1]) == 0, "NaN guard (dmt.py:407-409) fired"
    assert float(xh[..., :3].abs().max()) > 0


def test_gemm_identity_asymmetric(gpu_device):
    """A = I with an asymmetric B: catches a transposed C/D fragment map (cdna guide §3)."""
    from diffspectra_amd import engine as E
    lib = E.load_library()
    d = gpu_device
    K = N = 64
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 0.01 + torch.arange(N).reshape(N, 1) * 3.0
    A = torch.eye(64)
    out = torch.zeros(64, N, device=d)
    E.gemm(lib, A.to(d), K, E.pack_linear(W).to(d), None, out, N, 64, K, N)
    assert torch.equal(out.cpu(), W.t().contiguous()), "C/D fragment layout or packing is transposed/permuted"


def test_gemm_row_groups(gpu_device):
    """Unfold-view A, grouped C and broadcast residual (the SpecFormer patch projection)."""
    from diffspectra_amd import engine as E
    lib = E.load_library()
    d = gpu_device
    B, Lspec, pl, st = 3, 701, 20, 10
    npatch = (Lspec - pl) // st + 1
    g = torch.Generator().manual_seed(7)
    spec = torch.randn(B, Lspec, generator=g)
    W, b = torch.randn(128, pl, generator=g) / pl ** 0.5, torch.randn(128, generator=g)
    pos = torch.randn(npatch, 128, generator=g)
    Ltot = npatch + 5
    z = torch.zeros(B, Ltot, 128, device=d)
    E.gemm(lib, spec.to(d), st, E.pack_linear(W).to(d), E.pad_vec(b).to(d), z.data_ptr() + 4 * 5 * 128, 128, B * npatch, pl, 128,
           R=pos.to(d), ldr=128, r_grp_rows=npatch, a_grp=(npatch, Lspec), c_grp=(npatch, Ltot * 128))
    ref = torch.nn.functional.linear(spec.unfold(-1, pl, st), W, b) + pos
    assert_close(z[:, 5:], ref, 1e-5, "grouped gemm")
    assert float(z[:, :5].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("L,has_prev", [(347, 0), (139, 1), (1100, 1)])
def test_spec_attention_vs_torch(gpu_device, L, has_prev):
    """ds_spec_attention (specformer.py:401-424: scores = q k^T * scale + prev, softmax over keys, @ v; the new pre-softmax
    scores are handed to the next layer) against a plain torch fp32 evaluation - including a sequence longer than one
    1024-thread workgroup.  The score tensor is kept key-major ([b][h][key][query])."""
    import ctypes as C
    from diffspectra_amd import engine as E
    lib = E.load_library()
    d = gpu_device
    B, H, DK = 2, 3, 8
    D = H * DK
    g = torch.Generator().manual_seed(5 + L)
    qkv = torch.randn(B, L, 3 * D, generator=g)
    prev = torch.randn(B, H, L, L, generator=g) * 0.5          # [b][h][key][query]
    scale = 0.37
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, L, H, DK).permute(0, 2, 1, 3) for i in range(3))
    s_ref = torch.einsum("bhid,bhjd->bhij", q, k) * scale     # [b][h][query][key]
    if has_prev:
        s_ref = s_ref + prev.transpose(-1, -2)
    o_ref = torch.einsum("bhij,bhjd->bhid", torch.softmax(s_ref, -1), v).permute(0, 2, 1, 3).reshape(B, L, D)
    scores = prev.clone().to(d) if has_prev else torch.zeros(B, H, L, L, device=d)
    out = torch.empty(B, L, D, device=d)
    qd = qkv.to(d)
    E._check(lib.ds_spec_attention(E._ptr(qd), E._ptr(scores), E._ptr(out), C.c_int(B), C.c_int(L), C.c_int(H), C.c_int(DK),
                                   C.c_float(scale), C.c_int(has_prev), E._stream()), "ds_spec_attention")
    assert_close(out, o_ref, 2e-5, "attention output")
    assert_close(scores.transpose(-1, -2), s_ref, 2e-5, "pre-softmax scores for the next layer")


# ------------------------------------------------------------------------------------------------ stages

def _oracle_edge_maps(L, N):
    """Index of every packed pair (a<b) in the oracle's directed edge list (row-major nonzero order)."""
    nd = L.t["node_dense"].cpu().long()
    pa, pb = nd[L.t["pair_a"].cpu().long()], nd[L.t["pair_b"].cpu().long()]
    valid = torch.from_numpy(L.valid)
    adj = (valid.unsqueeze(1) & valid.unsqueeze(2)) & ~torch.eye(N, dtype=torch.bool).unsqueeze(0)
    eid = torch.full(adj.shape, -1, dtype=torch.long)
    eid[adj] = torch.arange(int(adj.sum()))
    b = pa // N
    return eid[b, pa % N, pb % N], eid[b, pb % N, pa % N], nd


@pytest.mark.parametrize("first", [True, False])
def test_stages_vs_oracle(gpu_device, first):
    """time/adaLN table, init, each of the 8 blocks (h, e, pos) and the readout, localising any mismatch."""
    cfg, model = gpu_model("ir", gpu_device)
    cpu_cfg, sd = procedural_state_dict("ir")
    a = cases.forward_inputs("ir", first, n_atoms=[3, 9, 18, 29, 2, 1, 12])
    eng = model.module.engine()
    d = gpu_device
    L, ws = eng.layout_for(a["node_mask"], a["edge_mask"], validate=True)
    ctx_cpu = oracle.context_embedding(sd, a["context"], cpu_cfg)
    ref_out = oracle.dmt_forward(sd, cpu_cfg, a["xh"], a["node_mask"], a["edge_mask"], a["edge_x"], a["noise_level"],
                                 a["cond_x"], a["cond_edge_x"], context_emb=ctx_cpu, return_debug=True)
    ref_xh, ref_edge, dbg = ref_out
    N = L.N
    fwd, bwd, nd = _oracle_edge_maps(L, N)
    # --- time embedding + adaLN table
    xh, edge_x, nl = a["xh"].to(d), a["edge_x"].to(d), a["noise_level"].to(d)
    cx, ce = to_dev(a["cond_x"], d), to_dev(a["cond_edge_x"], d)
    ctx = ctx_cpu.to(d)
    eng.stage_time(L, ws, nl, ctx)
    temb = odmt.time_embedding(sd, a["noise_level"]) + ctx_cpu
    silu = torch.nn.functional.silu(temb)
    planes = ws.t["temb_silu"].view(torch.float16).reshape(-1, 2, 1024).float()      # split-fp16 layout: a = a1 + a2 / 2048
    assert_close(planes[:, 0] + planes[:, 1] / 2048.0, silu, TOL_KERNEL, "SiLU(time_emb)")
    from diffspectra_amd import engine as E
    ada = ws.t["ada"].cpu()
    for blk in (0, 7):
        base = blk * E.ADA_STRIDE
        for name, off, width in (("node_time_mlp", 0, 1536), ("edge_time_mlp", 1536, 384), ("equi_update.time_mlp", 1920, 512),
                                 ("dist_layer.time_mlp", 2432, 2)):
            want = odmt._lin(sd, f"e_block_{blk}.{name}.1", silu)
            assert_close(ada[:, base + off: base + off + width], want, TOL_KERNEL, f"adaLN table block {blk} {name}")
    assert_close(ada[:, 8 * E.ADA_STRIDE: 8 * E.ADA_STRIDE + 2], odmt._lin(sd, "dist_layer.time_mlp.1", silu), TOL_KERNEL, "adaLN top dist")
    # --- init
    eng.stage_init(L, ws, xh, edge_x, cx, ce)
    torch.cuda.synchronize()
    for blk in range(8):
        eng.stage_block(L, ws, blk, last=(blk == 7))
        torch.cuda.synchronize()
        h_ref = dbg[f"h_{blk}"][nd]
        pos_ref = dbg[f"pos_{blk}"][nd]
        e_ref = dbg[f"e_{blk}"]
        assert_close(ws.t["h"], h_ref, TOL_KERNEL * 2, f"block {blk} node features")
        assert_close(ws.t["e"], e_ref[fwd], TOL_KERNEL * 2, f"block {blk} edge features (a->b)")
        assert_close(ws.t["e"], e_ref[bwd], TOL_KERNEL * 2, f"block {blk} edge features (b->a)")
        assert_close(ws.t["pos"][:, :3], pos_ref, TOL_KERNEL * 2, f"block {blk} positions")
    out_xh = torch.empty(L.B, N, 9, device=d)
    out_edge = torch.empty(L.B, N, N, 2, device=d)
    eng.stage_readout(L, ws, out_xh, out_edge)
    assert_close(out_xh, ref_xh, TOL_FORWARD, "forward xh")
    assert_close(out_edge, ref_edge, TOL_FORWARD, "forward edges")
    assert float((out_xh.cpu() * (1 - a["node_mask"])).abs().max()) == 0.0
    assert torch.equal(out_edge, out_edge.transpose(1, 2))


# ------------------------------------------------------------------------------------------------ goldens

@pytest.mark.parametrize("version", ["ir", "allspectra"])
def test_g2_specformer_golden(gpu_device, version):
    cfg, model = gpu_model(version, gpu_device)
    g = cases.load_npz("g2_specformer.npz")
    ctx = model.module.engine().context_embedding(to_dev(cases.spectra_for(version, 4), gpu_device))
    assert_close(ctx, g[f"{version}_ctx"], TOL_KERNEL * 2, f"SpecFormer+cond_lin {version}")


@pytest.mark.parametrize("version", ["ir", "allspectra"])
@pytest.mark.parametrize("first", [True, False])
def test_g4_forward_golden(gpu_device, version, first):
    """model(...) through the reference call convention against the reference's own output."""
    cfg, model = gpu_model(version, gpu_device)
    g = cases.load_npz("g4_forward.npz")
    a = cases.forward_inputs(version, first)
    d = gpu_device
    xh, ef = model(torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d),
                   context=to_dev(a["context"], d), edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d),
                   cond_x=to_dev(a["cond_x"], d), cond_edge_x=to_dev(a["cond_edge_x"], d))
    tag = f"{version}_{'first' if first else 'general'}"
    assert_close(xh, g[tag + "_xh"], TOL_FORWARD, tag + " xh")
    assert_close(ef, g[tag + "_edge"], TOL_FORWARD, tag + " edge")


def test_forward_requires_kwargs(gpu_device):
    cfg, model = gpu_model("ir", gpu_device)
    a = cases.forward_inputs("ir", True)
    d = gpu_device
    with pytest.raises(KeyError):      # reference raises KeyError for a missing edge_x/cond_x (dmt.py:321)
        model(torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d), context=a["context"].to(d),
              noise_level=a["noise_level"].to(d))


@pytest.mark.parametrize("version,steps", [("allspectra", 5), ("ir", 50)])
def test_g5_trajectory_golden(gpu_device, version, steps):
    """Injected-noise ancestral trajectories + post-processing against the reference's run (diverse integer outputs)."""
    check_trajectory_golden(gpu_device, "g5_trajectory.npz", version, steps)


def test_g8_clamp_self_cond_golden(gpu_device):
    """self_cond_type='clamp' through the HIP sampler vs the reference trajectory (utils.py:137-148, sampling.py:590)."""
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.scalers import get_data_inverse_scaler
    version, steps = "ir", 8
    cfg, model = gpu_model(version, gpu_device)
    cfg = cfg.clone()
    cfg.sampling.steps = steps
    cfg.model.self_cond_type = "clamp"
    original = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(cases.readout_gain(original), strict=True)   # the engine repacks on the version bump
    try:
        g = cases.load_npz("g8_trajectory_clamp.npz")
        tr = cases.trajectory_inputs(version, steps)
        d = gpu_device
        z = oracle.combined_noise(*tr["raw0"][:2], tr["node_mask"])
        ez = oracle.symmetric_edge_noise(tr["raw0"][2], tr["edge_mask"])

        def run(c):
            sampler = S._make_sampler(c, NoiseScheduleVP("cosine"), 1e-3, 1.0)
            sampler.noise_fn = lambda i: tr["raws"][i]
            return sampler.sampling(model, z.to(d), tr["node_mask"].to(d), tr["edge_mask"].to(d), ez.to(d),
                                    to_dev(tr["context"], d))

        x_mean, e_mean = run(cfg)
        tag = f"{version}_S{steps}"
        assert_close(x_mean, g[tag + "_x_mean"], TOL_TRAJ, tag + " x_mean (clamp)")
        assert_close(e_mean, g[tag + "_edge_mean"], TOL_TRAJ, tag + " edge_mean (clamp)")
        ori = cfg.clone()
        ori.model.self_cond_type = "ori"
        x_ori, _ = run(ori)
        assert float((x_ori.cpu() - g[tag + "_x_mean"]).abs().max()) > 100 * TOL_TRAJ   # the clamp really acts here
        eng = model.module.engine()
        _, one_hot, fc, et = S.post_process(x_mean, 5, True, tr["node_mask"].to(d), get_data_inverse_scaler(cfg), e_mean,
                                            tr["edge_mask"].to(d), True, engine=eng)
        assert torch.equal(one_hot.argmax(-1).cpu(), g[tag + "_atom_type"])
        assert torch.equal(fc.squeeze(-1).cpu(), g[tag + "_fc"].squeeze(-1).long())
        assert torch.equal(et.cpu(), g[tag + "_edge_type"])
    finally:
        model.load_state_dict(original, strict=True)


def test_g6_post_process_golden(gpu_device):
    from diffspectra_amd import filler
    cfg, model = gpu_model("ir", gpu_device)
    g = cases.load_npz("g6_post_process.npz")
    node_mask, edge_mask = filler.masks_from_n_atoms([2, 5, 7])
    eng = model.module.engine()
    L, _ = eng.layout_for(node_mask, edge_mask, validate=True)
    pos, atom, fc, et = eng.post_process(L, g["xh"].to(gpu_device), g["edge_x"].to(gpu_device))
    assert torch.equal(pos.cpu(), g["pos"])
    valid = node_mask.squeeze(-1).bool()
    assert torch.equal(atom.cpu().long()[valid], g["atom_type"][valid])
    assert torch.equal(fc.cpu().long(), g["fc"].squeeze(-1).long())
    assert torch.equal(et.cpu(), g["edge_type"])


def test_sampler_step_vs_oracle(gpu_device):
    from diffspectra_amd import filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    n_atoms = [4, 29, 11, 2]
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "ss.x")
    p, pe, _, _ = filler.synthetic_state(n_atoms, "ss.p")
    B, N = 4, 29
    raw = (filler.normal("ss.rp", (B, N, 3)), filler.normal("ss.rf", (B, N, 6)), filler.normal("ss.re", (B, 2, N, N)))
    c_x, c_p, sig, temp = 0.8125, 0.31, 0.27, 0.9
    xm_ref = c_x * x + c_p * p
    x_ref = xm_ref + sig * oracle.combined_noise(raw[0], raw[1], node_mask) * temp
    em_ref = c_x * ex + c_p * pe
    e_ref = em_ref + sig * oracle.symmetric_edge_noise(raw[2], edge_mask) * temp
    d = gpu_device
    L, _ = eng.layout_for(node_mask, edge_mask)
    xd, exd = x.to(d).clone(), ex.to(d).clone()
    xm, em = torch.zeros(B, N, 9, device=d), torch.zeros(B, N, N, 2, device=d)
    eng.sampler_step(L, c_x, c_p, sig, temp, xd, exd, p.to(d), pe.to(d), raw[0].to(d), raw[1].to(d), raw[2].to(d), xm, em)
    for got, want, nm in ((xm, xm_ref, "x_mean"), (xd, x_ref, "x"), (em, em_ref, "edge_mean"), (exd, e_ref, "edge_x")):
        assert_close(got, want, 2e-6, nm)


def test_philox_noise_kernels_vs_oracle(gpu_device):
    """ds_initial_noise / ds_sampler_step_philox against the numpy restatement (oracle/philox.py, pinned by the Random123
    known answers): same per-molecule streams, CoM projection, symmetric edge noise; masked entries untouched."""
    from diffspectra_amd import filler
    from oracle import philox as P
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    node_mask = torch.zeros(5, 29, 1)
    for b, n in enumerate([4, 29, 1, 11]):
        node_mask[b, :n] = 1
    node_mask[4, [0, 2, 3, 7, 8]] = 1                                  # non-prefix mask: streams follow the valid-atom rank
    nm2 = node_mask.squeeze(-1)
    edge_mask = ((nm2.unsqueeze(1) * nm2.unsqueeze(2)) * (~torch.eye(29, dtype=torch.bool)).unsqueeze(0)).reshape(-1, 1)
    L, _ = eng.layout_for(node_mask, edge_mask, validate=True)
    mol = torch.tensor([7, (1 << 33) + 5, 0, 123456789, 99], dtype=torch.int64, device=d)
    seed = (3 << 32) + 42
    x, ex = eng.initial_noise_philox(L, seed, mol)
    B, N = 5, 29

    def oracle_noise(draw):
        wx, we = torch.zeros(B, N, 9), torch.zeros(B, N, N, 2)
        for b in range(B):
            idx = torch.nonzero(nm2[b]).reshape(-1)
            pos, feat, edge = P.molecule_noise(seed, draw, int(mol[b]), len(idx))
            wx[b, idx, :3], wx[b, idx, 3:] = torch.from_numpy(pos), torch.from_numpy(feat)
            we[b, idx.unsqueeze(1), idx.unsqueeze(0)] = torch.from_numpy(edge)
        return wx, we

    wx, we = oracle_noise(0)
    assert_close(x, wx, 2e-5, "initial node noise")
    assert_close(ex, we, 2e-5, "initial edge noise")
    assert float((x.cpu() * (1 - node_mask)).abs().max()) == 0.0 and torch.equal(ex, ex.transpose(1, 2))
    assert float(x[:, :, :3].sum(1).abs().max()) < 1e-5
    p, pe, _, _ = filler.synthetic_state([29] * 5, "ph.p")
    p, pe = p * node_mask, pe * edge_mask.reshape(B, N, N, 1)
    c_x, c_p, sig, temp, step = 0.8125, 0.31, 0.27, 0.9, 17
    x0, ex0 = x.clone(), ex.clone()
    xm, em = torch.zeros(B, N, 9, device=d), torch.zeros(B, N, N, 2, device=d)
    eng.sampler_step_philox(L, c_x, c_p, sig, temp, seed, step, mol, x, ex, p.to(d), pe.to(d), xm, em)
    nx, ne = oracle_noise(step + 1)
    xm_ref, em_ref = c_x * x0.cpu() + c_p * p, c_x * ex0.cpu() + c_p * pe
    for got, want, name in ((xm, xm_ref, "x_mean"), (x, xm_ref + sig * nx * temp, "x"), (em, em_ref, "edge_mean"),
                            (ex, em_ref + sig * ne * temp, "edge_x")):
        assert_close(got, want, 1e-5, name)
    # distribution at bench size: N(0,1) per channel, edges symmetric, molecules uncorrelated
    n_atoms = filler.sample_n_atoms(2048, seed=0).tolist()
    nmk, emk = filler.masks_from_n_atoms(n_atoms)
    Lb, _ = eng.layout_for(nmk, emk)
    xb, exb = eng.initial_noise_philox(Lb, 42, torch.arange(2048, dtype=torch.int64, device=d))
    feat = xb[:, :, 3:].cpu()[nmk.squeeze(-1).bool()]
    ed = exb.cpu()[emk.reshape(exb.shape[:3]).bool()]
    for t, name in ((feat, "feature"), (ed, "edge")):
        assert abs(float(t.mean())) < 0.01 and abs(float(t.var()) - 1.0) < 0.01, name
        assert abs(float((t ** 4).mean()) - 3.0) < 0.06, name


# ------------------------------------------------------------------------------------------------ properties at full size

def _random_rotation(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.float()


def test_full_size_properties(gpu_device):
    """BASELINE-size batch (all-spectra, QM9 size histogram): invariants, SE(3) equivariance, batch independence."""
    from diffspectra_amd import filler
    cfg, model = gpu_model("allspectra", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    n_atoms = filler.sample_n_atoms(256, seed=0).tolist()
    n_atoms[0], n_atoms[1] = 29, 3
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "fs.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "fs.c")
    B, N = len(n_atoms), 29
    nl = filler.uniform("fs.nl", (B,), -6, 6)
    ctx_emb = filler.normal("fs.ctx", (B, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask, validate=True)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx_emb.to(d))
    out, oute = out.cpu(), oute.cpu()
    assert torch.isfinite(out).all() and torch.isfinite(oute).all()
    assert float((out * (1 - node_mask)).abs().max()) == 0.0
    assert float((oute * (1 - edge_mask.reshape(B, N, N, 1))).abs().max()) == 0.0
    assert float(out[:, :, :3].sum(1).abs().max()) < 1e-5, "CoM"
    assert torch.equal(oute, oute.transpose(1, 2)), "edge symmetry"
    # SE(3): rotate + translate-free (CoM) inputs -> positions rotate, types/edges invariant
    Rm = _random_rotation(3)
    xr, cxr = x.clone(), cx.clone()
    xr[:, :, :3] = x[:, :, :3] @ Rm.T
    cxr[:, :, :3] = cx[:, :, :3] @ Rm.T
    outr, outer = eng.forward(L, ws, xr.to(d), ex.to(d), nl.to(d), cxr.to(d), cex.to(d), ctx_emb.to(d))
    assert_close(outr.cpu()[:, :, :3], out[:, :, :3] @ Rm.T, 5e-5, "rotation equivariance of positions")
    assert_close(outr.cpu()[:, :, 3:], out[:, :, 3:], 5e-5, "rotation invariance of type logits")
    assert_close(outer.cpu(), oute, 5e-5, "rotation invariance of edge logits")
    # batch independence + agreement with the oracle on a slice
    sel = [0, 1, 17, 101]
    cpu_cfg, sd = procedural_state_dict("allspectra")
    for b in sel:
        n = n_atoms[b]
        nm1, em1 = filler.masks_from_n_atoms([n])
        ref, refe = oracle.dmt_forward(sd, cpu_cfg, x[b:b + 1, :n], nm1, em1, ex[b:b + 1, :n, :n], nl[b:b + 1], cx[b:b + 1, :n],
                                       cex[b:b + 1, :n, :n], context_emb=ctx_emb[b:b + 1])
        assert_close(out[b:b + 1, :n], ref, TOL_FORWARD, f"molecule {b} (n={n}) vs single-molecule oracle")
        assert_close(oute[b:b + 1, :n, :n], refe, TOL_FORWARD, f"molecule {b} edges vs oracle")


def test_permutation_equivariance(gpu_device):
    from diffspectra_amd import filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    n_atoms = [13, 7]
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "pe.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "pe.c")
    nl = filler.uniform("pe.nl", (2,), -3, 3)
    ctx = filler.normal("pe.ctx", (2, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
    perm = torch.randperm(13, generator=torch.Generator().manual_seed(5))
    def permute(t, e):
        t2, e2 = t.clone(), e.clone()
        t2[0, :13] = t[0, perm]
        e2[0, :13, :13] = e[0, perm][:, perm]
        return t2, e2
    xp, exp_ = permute(x, ex)
    cxp, cexp = permute(cx, cex)
    outp, outep = eng.forward(L, ws, xp.to(d), exp_.to(d), nl.to(d), cxp.to(d), cexp.to(d), ctx.to(d))
    want, wante = permute(out.cpu(), oute.cpu())
    assert_close(outp.cpu(), want, 5e-5, "atom permutation equivariance")
    assert_close(outep.cpu(), wante, 5e-5, "atom permutation equivariance (edges)")


def test_permutation_equivariance_every_size(gpu_device):
    """Every molecule size 2 .. 29 in one batch, each molecule's atoms permuted: the attention kernel's difference-class row order,
    its per-target visit lists and its softmax address table all depend on n (odd / even n, the half class of an even n) and on which
    atom sits where - outputs must follow the permutation for every n, and match the CPU oracle for every n."""
    from diffspectra_amd import filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    n_atoms = list(range(2, 30))
    B = len(n_atoms)
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "pes.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "pes.c")
    nl = filler.uniform("pes.nl", (B,), -3, 3)
    ctx = filler.normal("pes.ctx", (B, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
    out, oute = out.cpu().clone(), oute.cpu().clone()
    gen = torch.Generator().manual_seed(11)
    perms = [torch.randperm(n, generator=gen) for n in n_atoms]
    def permute(t, e):
        t2, e2 = t.clone(), e.clone()
        for b, (n, pm) in enumerate(zip(n_atoms, perms)):
            t2[b, :n] = t[b, pm]
            e2[b, :n, :n] = e[b, pm][:, pm]
        return t2, e2
    xp, exp_ = permute(x, ex)
    cxp, cexp = permute(cx, cex)
    outp, outep = eng.forward(L, ws, xp.to(d), exp_.to(d), nl.to(d), cxp.to(d), cexp.to(d), ctx.to(d))
    want, wante = permute(out, oute)
    cpu_cfg, sd = procedural_state_dict("ir")
    for b, n in enumerate(n_atoms):
        assert_close(outp.cpu()[b], want[b], 5e-5, f"n = {n}: atom permutation equivariance")
        assert_close(outep.cpu()[b], wante[b], 5e-5, f"n = {n}: atom permutation equivariance (edges)")
        nm1, em1 = filler.masks_from_n_atoms([n])          # ... and every size against the oracle, one molecule at a time
        ref, refe = oracle.dmt_forward(sd, cpu_cfg, x[b:b + 1, :n], nm1, em1, ex[b:b + 1, :n, :n], nl[b:b + 1], cx[b:b + 1, :n],
                                       cex[b:b + 1, :n, :n], context_emb=ctx[b:b + 1])
        assert_close(out[b:b + 1, :n], ref, TOL_FORWARD, f"n = {n} vs single-molecule oracle")
        assert_close(oute[b:b + 1, :n, :n], refe, TOL_FORWARD, f"n = {n} edges vs single-molecule oracle")


def test_largest_molecules_batch(gpu_device):
    """A batch of only the largest molecules (n = 29 and the even n = 28: 13 chunks of pair rows, every LDS table at its maximum, the
    half class at its largest) against the oracle, and identical molecules inside one batch giving identical outputs (the per-molecule
    kernel has no cross-molecule state)."""
    from diffspectra_amd import filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    n_atoms = [29, 28] * 6
    B = len(n_atoms)
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "lm.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "lm.c")
    nl = filler.uniform("lm.nl", (B,), -3, 3)
    ctx = filler.normal("lm.ctx", (B, 1024)) * 0.5
    for t in (x, ex, cx, cex, nl, ctx):          # molecules 10 / 11 are copies of 0 / 1
        t[10], t[11] = t[0], t[1]
    L, ws = eng.layout_for(node_mask, edge_mask)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
    out, oute = out.cpu(), oute.cpu()
    assert torch.equal(out[10], out[0]) and torch.equal(oute[10], oute[0]) and torch.equal(out[11], out[1]) and torch.equal(oute[11], oute[1])
    cpu_cfg, sd = procedural_state_dict("ir")
    for b in (0, 1, 7):
        n = n_atoms[b]
        nm1, em1 = filler.masks_from_n_atoms([n])
        ref, refe = oracle.dmt_forward(sd, cpu_cfg, x[b:b + 1, :n], nm1, em1, ex[b:b + 1, :n, :n], nl[b:b + 1], cx[b:b + 1, :n],
                                       cex[b:b + 1, :n, :n], context_emb=ctx[b:b + 1])
        assert_close(out[b:b + 1, :n], ref, TOL_FORWARD, f"molecule {b} (n = {n}) vs single-molecule oracle")
        assert_close(oute[b:b + 1, :n, :n], refe, TOL_FORWARD, f"molecule {b} (n = {n}) edges vs oracle")


def test_molecule_launch_order_does_not_change_results(gpu_device):
    """ds_layout.mol_by_size (the per-molecule attention kernel takes its workgroups' molecules from size-sorted records) is a
    scheduling hint: with the field NULL (index order through node_off / pair_off) the outputs are bit-identical; ragged sizes incl.
    single atoms, pairs and the largest molecule."""
    from diffspectra_amd import engine as E, filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    n_atoms = [5, 29, 1, 2, 18, 18, 3, 24, 1, 11, 28, 9]
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "lo.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "lo.c")
    nl = filler.uniform("lo.nl", (len(n_atoms),), -3, 3)
    ctx = filler.normal("lo.ctx", (len(n_atoms), 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask)
    rec = L.t["mol_by_size"].cpu()
    assert rec.shape == (len(n_atoms), 4) and rec[:, 1].tolist() == sorted(n_atoms, reverse=True)
    assert torch.equal(rec[:, 3], rec[:, 1] * (rec[:, 1] - 1) // 2)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
    out, oute = out.clone(), oute.clone()
    saved = L.c.mol_by_size
    try:
        L.c.mol_by_size = None
        out2, oute2 = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
        assert torch.equal(out, out2) and torch.equal(oute, oute2)
    finally:
        L.c.mol_by_size = saved


def test_non_prefix_mask_and_errors(gpu_device):
    """Valid atoms need not be a prefix; bad structures raise instead of computing garbage."""
    from diffspectra_amd import filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    cpu_cfg, sd = procedural_state_dict("ir")
    node_mask = torch.zeros(2, 8, 1)
    node_mask[0, [0, 2, 3, 6]] = 1
    node_mask[1, 1:6] = 1
    em = (node_mask.squeeze(-1).unsqueeze(1) * node_mask.squeeze(-1).unsqueeze(2)) * (~torch.eye(8, dtype=torch.bool)).unsqueeze(0)
    edge_mask = em.reshape(-1, 1)
    x = filler.normal("np.x", (2, 8, 9)) * node_mask
    x[:, :, :3] = x[:, :, :3] - (x[:, :, :3].sum(1, keepdim=True) / node_mask.sum(1, keepdim=True)) * node_mask
    e = filler.normal("np.e", (2, 8, 8, 2))
    e = (e + e.transpose(1, 2)) * em.unsqueeze(-1)
    nl = torch.tensor([0.3, -2.0])
    ctx = filler.normal("np.ctx", (2, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask, validate=True)
    out, oute = eng.forward(L, ws, x.to(d), e.to(d), nl.to(d), None, None, ctx.to(d))
    ref, refe = oracle.dmt_forward(sd, cpu_cfg, x, node_mask, edge_mask, e, nl, None, None, context_emb=ctx)
    assert_close(out, ref, TOL_FORWARD, "non-prefix mask xh")
    assert_close(oute, refe, TOL_FORWARD, "non-prefix mask edges")
    bad = edge_mask.clone()
    bad[1] = 1 - bad[1]
    with pytest.raises(ValueError):
        eng.layout_for(node_mask, bad, validate=True)
    with pytest.raises(ValueError):
        eng.layout_for(torch.ones(1, 40, 1))          # more atoms than DS_MAX_ATOMS


# ------------------------------------------------------------------------------------------------ factory / driver surface

class _Item:
    def __init__(self, i, n, version_specs):
        self.num_atom = torch.tensor(n)
        self.pos = torch.zeros(n, 3)
        self.rdmol = f"mol{i}"
        self.uv, self.ir, self.raman = version_specs


def _tiny_dataset(count):
    from diffspectra_amd import filler
    n_atoms = filler.sample_n_atoms(count, seed=3).tolist()
    specs = cases.spectra_for("allspectra", count, salt=5)
    return [_Item(i, n_atoms[i], (specs[0][i], specs[1][i], specs[2][i])) for i in range(count)], n_atoms


class _ReplayRandn:
    """torch.randn stand-in that hands out queued tensors (moved to the requested device) in call order."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __call__(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        t = self.queue.pop(0)
        assert tuple(t.shape) == tuple(size), (t.shape, size)
        return t.clone().to(kw.get("device", "cpu"))


def test_g11_sampling_fn_golden(gpu_device, monkeypatch):
    """The OUTER loop against the reference: ``get_cond_sampling_eval_fn(...)(model)`` (sampling.py:353-468) on the same
    in-memory dataset with every randn draw replayed - seed-42 permutation, rounds, masks, initial noise, temperature,
    sampler, post-processing and mol_process; molecules compared with the reference's ``processed_mols``."""
    import json as _json
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.scalers import get_data_inverse_scaler
    c = cases.sampling_fn_case()
    g = cases.load_npz("g11_sampling_fn.npz")
    cfg, model = gpu_model("allspectra", gpu_device)
    cfg = cfg.clone()
    cfg.sampling.steps = c["steps"]
    cfg.sampling.noise_source = "torch"          # replayed randn draws in the reference's order
    cfg.eval.sampling_temperature = c["temperature"]
    ds = [_Item(i, c["n_atoms"][i], (c["spectra"][0][i], c["spectra"][1][i], c["spectra"][2][i])) for i in range(c["count"])]
    for i, it in enumerate(ds):
        it.pos = torch.full((c["n_atoms"][i], 3), float(i))
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    fn = S.get_cond_sampling_eval_fn(cfg, ns, c["batch_size"], c["n_samples"], get_data_inverse_scaler(cfg), ds)
    torch.manual_seed(42)
    perm = torch.randperm(c["count"])
    assert torch.equal(perm, g["perm"]), "seed-42 permutation differs from the reference's"
    replay = _ReplayRandn(cases.sampling_fn_noise_queue(c, perm.tolist()))
    with swapped_weights(model, lambda sd: cases.readout_diverse(sd, "allspectra_S5")):
        monkeypatch.setattr(torch, "randn", replay)
        try:
            mols, gt_pos, gt_mols = fn(model)
        finally:
            monkeypatch.undo()
    assert not replay.queue, "the HIP sampling function drew fewer randn tensors than the reference"
    assert gt_mols == _json.loads(g["gt_mols"])
    assert [float(p[0, 0]) for p in gt_pos] == g["gt_pos0"].tolist()
    assert len(mols) == c["n_samples"]
    n_int = n_mis = 0
    types_seen = set()
    for m, (pos, atom, edge, fc) in enumerate(mols):
        assert_close(pos, g[f"mol{m}_pos"], TOL_TRAJ, f"sampling_fn molecule {m} positions")
        assert atom.dtype == torch.int64 and fc.dtype == torch.int64 and edge.dtype == torch.float32
        for got, want in ((atom, g[f"mol{m}_atom"]), (edge, g[f"mol{m}_edge"]), (fc, g[f"mol{m}_fc"])):
            assert got.shape == want.shape
            n_int += want.numel()
            n_mis += int((got != want).sum())
        types_seen |= set(g[f"mol{m}_atom"].tolist())
    print(f"[g11] integer outputs compared: {n_int}, mismatches: {n_mis}; reference atom types {sorted(types_seen)}")
    assert n_mis == 0
    assert len(types_seen) >= 3


def test_cond_sampling_eval_fn_end_to_end(gpu_device):
    """get_cond_sampling_eval_fn(...)(model) with on-device noise: seed-42 permutation, rounds, tuple format, determinism,
    and the HBM-resident table as a drop-in dataset.  (Values are pinned to the reference by test_g11_sampling_fn_golden.)"""
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.scalers import get_data_inverse_scaler
    cfg, model = gpu_model("allspectra", gpu_device)
    cfg = cfg.clone()
    cfg.sampling.steps = 6
    cfg.sampling.noise_source = "torch"          # the reference's draw order (full rounds, torch.randn on the device)
    ds, n_atoms = _tiny_dataset(7)
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    fn = S.get_cond_sampling_eval_fn(cfg, ns, 3, 5, get_data_inverse_scaler(cfg), ds)
    mols, gt_pos, gt_mols = fn(model)
    assert len(mols) == 5 and len(gt_pos) == 5 and len(gt_mols) == 5
    torch.manual_seed(42)
    perm = torch.randperm(7).tolist()
    assert gt_mols == [f"mol{i}" for i in perm[:5]]                      # sampling.py:387-392 order
    for (pos, atom, edge, fc), i in zip(mols, perm[:5]):
        n = n_atoms[i]
        assert pos.shape == (n, 3) and pos.dtype == torch.float32 and pos.device.type == "cpu"
        assert atom.shape == (n,) and atom.dtype == torch.int64 and int(atom.max()) < 5
        assert edge.shape == (n, n) and edge.dtype == torch.float32 and torch.equal(edge, edge.T)
        assert set(edge.unique().tolist()) <= {0.0, 1.0, 2.0, 3.0} and float(edge.diagonal().abs().max()) == 0.0
        assert fc.shape == (n,) and fc.dtype == torch.int64
        assert float(pos.sum(0).abs().max()) < 1e-4                      # zero centre of mass
    mols2, _, _ = fn(model)                                               # same seeds -> same molecules
    for a, b in zip(mols, mols2):
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert float((a[0] - b[0]).abs().max()) < 1e-5
    # the HBM-resident conditioning table (N3) is a drop-in for the dataset: identical rounds, identical molecules
    from diffspectra_amd.dataset_pack import PackedSpectraTable
    tab = PackedSpectraTable.from_dataset(ds, "allspectra", device=gpu_device)
    fn3 = S.get_cond_sampling_eval_fn(cfg, ns, 3, 5, get_data_inverse_scaler(cfg), tab)
    mols3, gt_pos3, gt_mols3 = fn3(model)
    assert gt_mols3 == gt_mols and all(torch.equal(p, q) for p, q in zip(gt_pos3, gt_pos))
    for a, b in zip(mols, mols3):
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert float((a[0] - b[0]).abs().max()) < 1e-5


def _philox_sampling_setup(gpu_device, steps=4, count=11):
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.scalers import get_data_inverse_scaler
    cfg, model = gpu_model("allspectra", gpu_device)
    cfg = cfg.clone()
    cfg.sampling.steps = steps
    ds, n_atoms = _tiny_dataset(count)
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    return S, cfg, model, ds, n_atoms, ns, get_data_inverse_scaler(cfg)


def _same_molecules(a, b):
    return len(a) == len(b) and all(torch.equal(p[0], q[0]) and torch.equal(p[1], q[1]) and torch.equal(p[2], q[2])
                                    and torch.equal(p[3], q[3]) for p, q in zip(a, b))


def test_philox_sampling_is_batch_independent_and_topk(gpu_device):
    """Per-molecule noise streams: the sampled molecules do not depend on how the run is cut into micro-batches (bit for
    bit, positions included), and Top-K mode draws K different molecules per spectrum in the same batched run."""
    S, cfg, model, ds, n_atoms, ns, inv = _philox_sampling_setup(gpu_device)
    with swapped_weights(model, lambda sd: cases.readout_diverse(sd, "allspectra_S5")):
        a, gt_pos, gt_mols = S.get_cond_sampling_eval_fn(cfg, ns, 4, 9, inv, ds)(model)
        b, _, _ = S.get_cond_sampling_eval_fn(cfg, ns, 9, 9, inv, ds)(model)
        c, _, _ = S.get_cond_sampling_eval_fn(cfg, ns, 1, 9, inv, ds)(model)
        k3, gt3, mols3 = S.get_cond_sampling_eval_fn(cfg, ns, 5, 3, inv, ds, top_k=3)(model)
        other = cfg.clone()
        other.sampling.seed = 43
        d43, _, _ = S.get_cond_sampling_eval_fn(other, ns, 4, 9, inv, ds)(model)
    torch.manual_seed(42)
    perm = torch.randperm(len(ds)).tolist()
    assert gt_mols == [f"mol{i}" for i in perm[:9]]
    assert [m[0].shape[0] for m in a] == [n_atoms[i] for i in perm[:9]]
    assert _same_molecules(a, b) and _same_molecules(a, c), "molecules depend on the micro-batch size"
    assert not _same_molecules(a, d43)
    assert len(k3) == 9 and mols3 == [f"mol{i}" for i in perm[:3] for _ in range(3)]
    assert [m[0].shape[0] for m in k3] == [n_atoms[i] for i in perm[:3] for _ in range(3)]
    for i in range(3):                                                   # K candidates of one spectrum: distinct draws
        assert float((k3[3 * i][0] - k3[3 * i + 1][0]).abs().max()) > 1e-3
    assert torch.equal(k3[0][0], a[0][0])                                # slot 0 is slot 0 in both runs


def test_get_sampling_fn_validation_sampler(gpu_device):
    """``get_sampling_fn`` (sampling.py:148-248, the training-time validation sampler): UNSEEDED permutation of the validation set and
    temperature 1, against ``get_cond_sampling_eval_fn`` (seed-42 permutation, eval temperature) on the same model."""
    S, cfg, model, ds, n_atoms, ns, inv = _philox_sampling_setup(gpu_device)
    cfg.eval.sampling_temperature = 0.7
    with swapped_weights(model, lambda sd: cases.readout_diverse(sd, "allspectra_S5")):
        torch.manual_seed(3)
        a, gt_pos_a, gt_a = S.get_sampling_fn(cfg, ns, 4, 6, inv, ds)(model)
        torch.manual_seed(3)
        b, _, gt_b = S.get_sampling_fn(cfg, ns, 3, 6, inv, ds)(model)
        torch.manual_seed(4)
        c_, _, gt_c = S.get_sampling_fn(cfg, ns, 4, 6, inv, ds)(model)
        ev, _, gt_e = S.get_cond_sampling_eval_fn(cfg, ns, 4, 6, inv, ds)(model)
        # the validation sampler is the eval sampler at temperature 1 on whatever permutation the global generator yields
        cfg1 = cfg.clone()
        cfg1.eval.sampling_temperature = 1.0
        torch.manual_seed(42)
        perm42 = torch.randperm(len(ds)).tolist()
        torch.manual_seed(42)
        v42, _, gt_v42 = S.get_sampling_fn(cfg, ns, 4, 6, inv, ds)(model)
        e1, _, gt_e1 = S.get_cond_sampling_eval_fn(cfg1, ns, 4, 6, inv, ds)(model)
    torch.manual_seed(3)
    perm3 = torch.randperm(len(ds)).tolist()
    assert gt_a == [f"mol{i}" for i in perm3[:6]] == gt_b and len(a) == 6          # the permutation follows the global generator
    assert _same_molecules(a, b)                                                    # same permutation -> same molecules, any batch size
    assert gt_c != gt_a                                                             # unseeded: another generator state, another permutation
    assert [m[0].shape[0] for m in a] == [n_atoms[i] for i in perm3[:6]]
    assert gt_v42 == gt_e1 == [f"mol{i}" for i in perm42[:6]] == gt_e and _same_molecules(v42, e1)   # temperature 1 = the eval sampler at T = 1
    assert not _same_molecules(ev, e1)                                              # and the eval temperature (0.7) does act


def test_graph_replay_equals_eager(gpu_device):
    """hipGraph replay of the denoise iteration (device-side step index, in-place self-conditioning buffer) gives the
    eager launch sequence's result bit for bit, also when a pass is advanced in slices."""
    from diffspectra_amd import filler, sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    cfg, model = gpu_model("allspectra", gpu_device)
    cfg = cfg.clone()
    cfg.sampling.steps = 13
    d = gpu_device
    n_atoms = [5, 29, 2, 17, 9, 1, 12]
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    ctx = to_dev(cases.spectra_for("allspectra", len(n_atoms), salt=9), d)
    ids = torch.arange(100, 100 + len(n_atoms))
    outs = {}
    with swapped_weights(model, lambda sd: cases.readout_diverse(sd, "allspectra_S5")):
        for mode, slices in (("eager", [13]), ("graph", [13]), ("graph_sliced", [1, 1, 4, 7]), ("graph_late", [3, 10])):
            sampler = S._make_sampler(cfg, NoiseScheduleVP("cosine"), 1e-3, 0.9)
            sampler.use_graph = mode != "eager"
            st = sampler.begin(model, None, node_mask.to(d), edge_mask.to(d), None, ctx, mol_ids=ids, seed=7)
            if mode == "graph_late":
                sampler.use_graph = False                     # first slice eager (ping-pong buffers), then switch
            for k, n in enumerate(slices):
                if mode == "graph_late" and k == 1:
                    sampler.use_graph = True
                done = sampler.advance(st, n)
            assert done and st.i == 13
            assert (st.graph is not None) == (mode != "eager")
            outs[mode] = (st.x_mean.clone(), st.edge_mean.clone(), st.x.clone())
    for mode in ("graph", "graph_sliced", "graph_late"):
        for a, b in zip(outs["eager"], outs[mode]):
            assert torch.equal(a, b), f"{mode} differs from the eager launch sequence"
    assert float(outs["eager"][0].abs().max()) > 0.1


def _rank_worker(rank, world, port, out_path):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        S, cfg, model, ds, n_atoms, ns, inv = _philox_sampling_setup(dev)
        with swapped_weights(model, lambda sd: cases.readout_diverse(sd, "allspectra_S5")):
            mols, gt_pos, gt_mols = S.get_cond_sampling_eval_fn(cfg, ns, 3, 9, inv, ds)(model)
        if rank == 1:                                                    # every rank holds the full result; save rank 1's
            torch.save({"mols": mols.tolist(), "gt_mols": gt_mols}, out_path)      # MoleculeList -> plain list of tuples
    finally:
        dist.destroy_process_group()


def test_philox_sampling_two_ranks_equals_one(gpu_device, tmp_path):
    """The product's sharded sampling function on 2 ranks (one process each, both on this GPU, gloo for the final gather)
    returns, on every rank, exactly the molecules of the 1-rank run: slots are dealt by size, noise follows the slot."""
    import socket
    import torch.multiprocessing as mp
    S, cfg, model, ds, n_atoms, ns, inv = _philox_sampling_setup(gpu_device)
    with swapped_weights(model, lambda sd: cases.readout_diverse(sd, "allspectra_S5")):
        one, _, gt_one = S.get_cond_sampling_eval_fn(cfg, ns, 3, 9, inv, ds)(model)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "rank1.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    two = torch.load(out)
    assert two["gt_mols"] == gt_one
    assert _same_molecules(one, two["mols"]), "2-rank molecules differ from the 1-rank run"


def test_checkpoint_and_ema_roundtrip(gpu_device, tmp_path):
    """Reference checkpoint contract: {'model': module.-prefixed sd, 'ema': {'shadow_params': [...]}}; weights re-pack."""
    from diffspectra_amd import filler
    from diffspectra_amd.registry import create_model
    from diffspectra_amd.config import qm9s_config
    cfg = qm9s_config("ir", device=gpu_device)
    model = create_model(cfg)
    model.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    filler.fill_module_(model, salt=1)
    a = cases.forward_inputs("ir", True)
    d = gpu_device
    args = (torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d))
    kw = dict(context=a["context"].to(d), edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d), cond_x=None, cond_edge_x=None)
    out_salt1 = model(*args, **kw)[0].clone()
    # "checkpoint" with different weights: strict load of a module.-prefixed state dict (utils.py:17)
    ckpt = {"model": {k: v.cpu() for k, v in filler.fill_state_dict(model.state_dict(), salt=0).items()}}
    shadow = [filler.fill_tensor(n[len("module."):], p.shape, like=p, salt=0) for n, p in model.named_parameters() if p.requires_grad]
    ckpt["ema"] = {"decay": 0.999, "num_updates": 10, "shadow_params": shadow}
    path = tmp_path / "checkpoint_40.pth"
    torch.save(ckpt, path)
    loaded = torch.load(path, map_location=d)
    model.load_state_dict(loaded["model"], strict=True)
    params = [p for p in model.parameters() if p.requires_grad]
    assert len(params) == len(loaded["ema"]["shadow_params"])
    for s_param, p in zip(loaded["ema"]["shadow_params"], params):          # models/ema.py:44-55 copy_to
        p.data.copy_(s_param.data)
    out_salt0 = model(*args, **kw)[0]
    cfg0, ref_model = gpu_model("ir", gpu_device)                            # procedural salt-0 weights
    want = ref_model(*args, **kw)[0]
    assert_close(out_salt0, want, 1e-6, "output after checkpoint + EMA load")
    assert float((out_salt1 - out_salt0).abs().max()) > 1e-4                 # the engine really re-packed the weights


def test_evaluate_driver(gpu_device, tmp_path):
    """diffspectra_evaluate host plumbing: checkpoint_{k}.pth -> strict load -> EMA copy -> sampling_fn gives the molecules of
    the same sampling function on a model assembled by hand from those weights (HIP vs HIP: this checks WHICH weights are
    used, not the arithmetic - that is what the golden tests do)."""
    from diffspectra_amd import filler, sampling as S, evaluate as EV
    from diffspectra_amd.config import qm9s_config, Config
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.registry import create_model
    from diffspectra_amd.scalers import get_data_inverse_scaler
    cfg = qm9s_config("allspectra", device=gpu_device, steps=4, batch_size=3, num_samples=4)
    cfg.eval.begin_ckpt, cfg.eval.end_ckpt, cfg.eval.ckpts = 40, 40, ""
    ds, n_atoms = _tiny_dataset(5)
    # a "trained" checkpoint: model weights salt 2, EMA shadow salt 3 (eval must use the EMA ones)
    donor = create_model(cfg)
    donor.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    filler.fill_module_(donor, salt=2)
    ema = EV.ExponentialMovingAverage(donor.parameters(), decay=0.999)
    ema.shadow_params = [filler.fill_tensor(n[len("module."):], p.shape, like=p, salt=3).to(gpu_device)
                         for n, p in donor.named_parameters() if p.requires_grad]
    (tmp_path / "checkpoints").mkdir()
    EV.save_checkpoint(str(tmp_path / "checkpoints" / "checkpoint_40.pth"), dict(optimizer=None, model=donor, ema=ema, step=123))
    got = {}
    res = EV.diffspectra_evaluate(cfg, str(tmp_path), ds, metric_fns={"count": lambda m, p, r: len(m)})
    assert list(res) == [40] and res[40]["step"] == 123 and res[40]["metrics"]["count"] == 4
    # reference: same weights assembled by hand (EMA for parameters, checkpoint buffers for BatchNorm statistics)
    want_model = create_model(cfg)
    want_model.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    filler.fill_module_(want_model, salt=2)
    for s_p, p in zip(ema.shadow_params, [p for p in want_model.parameters() if p.requires_grad]):
        p.data.copy_(s_p)
    ns = NoiseScheduleVP("cosine")
    fn = S.get_cond_sampling_eval_fn(cfg, ns, 3, 4, get_data_inverse_scaler(cfg), ds)
    want, _, _ = fn(want_model)
    for a, b in zip(res[40]["processed_mols"], want):
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        assert float((a[0] - b[0]).abs().max()) < 1e-5
    # test_ds=None: the driver reads the reference's processed files under config.data.root (qm9s_reader) - same molecules as
    # handing it the 'test' split of the same data item by item
    from tests.test_host_cpu import _write_processed_qm9s
    big, _ = _tiny_dataset(9)
    mols = [{"atom_type": torch.zeros(int(it.num_atom), dtype=torch.long), "pos": it.pos, "edge_index": torch.zeros(2, 0, dtype=torch.long),
             "edge_type": torch.zeros(0, dtype=torch.long), "uv": it.uv, "ir": it.ir, "raman": it.raman} for it in big]
    perm = _write_processed_qm9s(str(tmp_path / "QM9S" / "processed"), mols, "pyg2")
    cfg.data.root, cfg.data.use_normalize = str(tmp_path / "QM9S"), False
    from_disk = EV.diffspectra_evaluate(cfg, str(tmp_path))
    by_hand = EV.diffspectra_evaluate(cfg, str(tmp_path), [big[j] for j in perm[6:].tolist()])
    assert len(from_disk[40]["processed_mols"]) == 3     # the split holds 3 molecules: one round of 3 (sampling.py:389-390)
    for a, b in zip(from_disk[40]["processed_mols"], by_hand[40]["processed_mols"]):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    with pytest.raises(FileNotFoundError):
        cfg.eval.begin_ckpt = cfg.eval.end_ckpt = 41
        EV.diffspectra_evaluate(cfg, str(tmp_path), ds)


def test_g7_full_length_trajectory_golden(gpu_device):
    """The metric's own length: 1000 injected-noise steps on the HIP path vs the reference's run (ir conditioning)."""
    check_trajectory_golden(gpu_device, "g7_trajectory_1000.npz", "ir", 1000, cases.FULL_LENGTH_ATOMS)


def test_g9_full_length_trajectory_allspectra_golden(gpu_device):
    """The headline configuration: all-spectra conditioning, 1000 injected-noise steps, vs the reference's run."""
    check_trajectory_golden(gpu_device, "g9_trajectory_1000_allspectra.npz", "allspectra", 1000, cases.ALLSPECTRA_FULL_ATOMS)


def test_g15_full_length_trajectory_max_size_golden(gpu_device):
    """1000 injected-noise steps with a maximum-size (n = 29) molecule in the batch, vs the reference's run."""
    check_trajectory_golden(gpu_device, "g15_trajectory_1000_n29.npz", "ir", 1000, cases.MAX_SIZE_FULL_ATOMS, tag_suffix="_n29")


def test_unconditional_config4_vs_oracle(gpu_device):
    """BASELINE config 4: ``ctx_emb = NULL`` (zero context embedding, SpecFormer skipped) on a 256-molecule QM9-histogram
    batch against ``oracle.dmt_forward(context_emb = 0)``, first-step and general branch."""
    from diffspectra_amd import filler
    cfg, model = gpu_model("allspectra", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    cpu_cfg, sd = procedural_state_dict("allspectra")
    n_atoms = filler.sample_n_atoms(256, seed=2).tolist()
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "c4.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "c4.c")
    B = len(n_atoms)
    nl = filler.uniform("c4.nl", (B,), -6, 6)
    L, ws = eng.layout_for(node_mask, edge_mask, validate=True)
    zero_ctx = torch.zeros(B, 1024)
    for first in (True, False):
        c1, c2 = (None, None) if first else (cx, cex)
        out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), to_dev(c1, d), to_dev(c2, d), None)
        ref, refe = oracle.dmt_forward(sd, cpu_cfg, x, node_mask, edge_mask, ex, nl, c1, c2, context_emb=zero_ctx)
        assert_close(out, ref, TOL_FORWARD, f"unconditional forward xh (first={first})")
        assert_close(oute, refe, TOL_FORWARD, f"unconditional forward edges (first={first})")
        # and a NULL context really is the zero embedding, not a stale one from an earlier call
        out0, oute0 = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), to_dev(c1, d), to_dev(c2, d), zero_ctx.to(d))
        assert torch.equal(out0, out) and torch.equal(oute0, oute)


def test_benchmark_batch_sizes(gpu_device):
    """The batch sizes the benchmark runs at.
    (a) BASELINE config 4's batch of 4096 molecules, zero context: slices of the batched forward against the single-call oracle
        on the same molecules (as test_full_size_properties does at 256).
    (b) 14 000 molecules resident at once - every [Pp, 256]-wide pair tensor beyond 2^31 bytes - against 512-molecule batches
        of the same molecules: catches 32-bit offset overflow in kernels and buffer descriptors."""
    from diffspectra_amd import filler
    cfg, model = gpu_model("allspectra", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    cpu_cfg, sd = procedural_state_dict("allspectra")
    # (a)
    M = 4096
    n_atoms = filler.sample_n_atoms(M, seed=4).tolist()
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "b4096.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "b4096.c")
    nl = filler.uniform("b4096.nl", (M,), -6, 6)
    L, ws = eng.layout_for(node_mask, edge_mask)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), None)
    out, oute = out.cpu(), oute.cpu()
    assert torch.isfinite(out).all() and torch.isfinite(oute).all()
    for lo in (0, 2040, M - 8):
        sl = slice(lo, lo + 8)
        nm, em = filler.masks_from_n_atoms(n_atoms[sl])
        N = nm.shape[1]
        ref, refe = oracle.dmt_forward(sd, cpu_cfg, x[sl, :N], nm, em, ex[sl, :N, :N], nl[sl], cx[sl, :N], cex[sl, :N, :N],
                                       context_emb=torch.zeros(8, 1024))
        assert_close(out[sl, :N], ref, TOL_FORWARD, f"4096-molecule batch, molecules {lo}..{lo + 8} xh")
        assert_close(oute[sl, :N, :N], refe, TOL_FORWARD, f"4096-molecule batch, molecules {lo}..{lo + 8} edges")
        assert float(out[sl, N:].abs().max() if N < out.shape[1] else 0.0) == 0.0
    del out, oute, L, ws
    # (b)
    M = 14000
    n_atoms = filler.sample_n_atoms(M, seed=0).tolist()
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "bb.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "bb.c")
    nl = torch.full((M,), 0.5)
    ctx = filler.normal("bb.ctx", (M, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask)
    assert L.Pp * 256 * 4 > 2 ** 31, "the batch must push the widest pair tensor beyond 2^31 bytes"
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
    out, oute = out.cpu(), oute.cpu()
    assert torch.isfinite(out).all() and torch.isfinite(oute).all()
    for lo in (0, M // 2 - 256, M - 512):
        sl = slice(lo, lo + 512)
        nm, em = filler.masks_from_n_atoms(n_atoms[sl])
        N = nm.shape[1]
        L2, ws2 = eng.layout_for(nm, em)
        o2, e2 = eng.forward(L2, ws2, x[sl, :N].contiguous().to(d), ex[sl, :N, :N].contiguous().to(d), nl[sl].to(d),
                             cx[sl, :N].contiguous().to(d), cex[sl, :N, :N].contiguous().to(d), ctx[sl].to(d))
        assert_close(o2.cpu(), out[sl, :N], TOL_FORWARD, f"14000-molecule batch vs 512-molecule batch at {lo}: xh")
        assert_close(e2.cpu(), oute[sl, :N, :N], TOL_FORWARD, f"14000-molecule batch vs 512-molecule batch at {lo}: edges")
    eng._layouts.clear()                                   # release the 14 000-molecule workspace


@pytest.mark.parametrize("variant", ["spec_model", "plain_model"])
def test_g10_pretrained_specformer_golden(gpu_device, variant, tmp_path):
    """BASELINE config 3: ``config.model.pretrained_specformer_path`` -> the reference's key mapping (dmt.py:268-303) ->
    SpecFormer + cond_lin on the GPU equals what the reference computed after ITS loader read the same checkpoint."""
    from diffspectra_amd import filler
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.registry import create_model
    g = cases.load_npz("g10_pretrained_specformer.npz")
    cfg = qm9s_config("allspectra", device=gpu_device)
    plain = create_model(cfg)
    plain.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    filler.fill_module_(plain)
    enc_sd = {k: v.cpu() for k, v in plain.module.cond_encoder.state_dict().items()}
    path = tmp_path / "pretrained_specformer.ckpt"
    torch.save(cases.pretrained_specformer_ckpt(enc_sd, variant), path)
    ctx_in = to_dev(cases.spectra_for("allspectra", 4), gpu_device)
    before = plain.module.engine().context_embedding(ctx_in).clone()
    plain.module.load_pretrained_specformer(str(path))               # in place, after a forward: the engine must re-pack
    got = plain.module.engine().context_embedding(ctx_in)
    assert_close(got, g[f"{variant}_ctx"], TOL_KERNEL * 2, f"pretrained SpecFormer ({variant}) conditioning embedding")
    assert float((got - before).abs().max()) > 1e-3


def test_engine_repacks_after_data_copy(gpu_device):
    """EMA ``copy_to``/``restore`` write with ``p.data.copy_`` (reference models/ema.py:55,77), which does not
    bump ``Tensor._version``; the packed weights must follow anyway."""
    from diffspectra_amd import filler
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.registry import create_model
    cfg = qm9s_config("ir", device=gpu_device)
    model = create_model(cfg)
    model.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    filler.fill_module_(model, salt=0)
    a = cases.forward_inputs("ir", False)
    d = gpu_device
    args = (torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d))
    kw = dict(context=a["context"].to(d), edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d),
              cond_x=a["cond_x"].to(d), cond_edge_x=a["cond_edge_x"].to(d))
    out0 = model(*args, **kw)[0].clone()
    versions = [int(p._version) for p in model.parameters()]
    other = filler.fill_state_dict(model.state_dict(), salt=4)
    for (name, p) in model.named_parameters():
        p.data.copy_(other[name].to(d))                                 # exactly what ema.copy_to does
    for (name, b) in model.named_buffers():
        b.data.copy_(other[name].to(d))                                 # BatchNorm statistics of the same "checkpoint"
    assert versions == [int(p._version) for p in model.parameters()]   # the hazard: no version bump
    out1 = model(*args, **kw)[0]
    fresh = create_model(cfg)
    fresh.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    fresh.load_state_dict({k: v.to(d) for k, v in other.items()}, strict=True)
    want = fresh(*args, **kw)[0]
    assert float((out1 - out0).abs().max()) > 1e-4, "engine kept the stale packed weights"
    assert_close(out1, want, 1e-6, "forward after p.data.copy_")


def test_dropin_forward_caches_loop_invariant_work(gpu_device):
    """A reference-style caller invokes ``model(...)`` 1000 times per round with the same context / mask tensors (sampling.py:588).
    The drop-in forward then reuses the spectra embedding, the layout and the mask checks of the first call - and notices when any
    of those tensors is modified in place."""
    cfg, model = gpu_model("ir", gpu_device)
    a = cases.forward_inputs("ir", False)
    d = gpu_device
    args = [torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d)]
    kw = dict(context=a["context"].to(d), edge_x=a["edge_x"].to(d), noise_level=a["noise_level"].to(d),
              cond_x=a["cond_x"].to(d), cond_edge_x=a["cond_edge_x"].to(d))
    out0 = model(*args, **kw)[0].clone()
    mod = model.module
    ctx0 = mod._ctx_cache[2]
    out1 = model(*args, **kw)[0]
    assert mod._ctx_cache[2] is ctx0 and torch.equal(out0, out1)                 # second call: cached embedding, identical result
    kw["context"].mul_(1.05)                                                      # in-place edit bumps the version: re-encoded
    out2 = model(*args, **kw)[0]
    assert mod._ctx_cache[2] is not ctx0 and float((out2 - out0).abs().max()) > 1e-5
    fresh = model(*args, **dict(kw, context=kw["context"].clone()))[0]           # a new tensor with the same values: same result
    assert torch.equal(fresh, out2)
    bad = args[3].clone()
    bad[5] = 1.0 - bad[5]
    with pytest.raises(ValueError):                                               # a different edge_mask tensor is validated again
        model(args[0], args[1], args[2], bad, **kw)


def test_caches_do_not_confuse_a_reallocated_tensor_with_the_freed_one(gpu_device):
    """The caching allocator hands a freed tensor's address to the next tensor of the same size, with version counter 0 again
    (ADVICE r3): a mask / context freed and re-created with OTHER contents between two calls must be seen as new.  The loop frees
    and re-allocates until the address really repeats (it does on the first try with torch's allocator), then compares with a
    model that has never seen the first tensors."""
    cfg, model = gpu_model("ir", gpu_device)
    d = gpu_device
    from diffspectra_amd import filler
    from diffspectra_amd.sampling import build_masks

    def inputs(n_atoms, salt):
        x, ex, _, _ = filler.synthetic_state(n_atoms, f"realloc.{salt}", n_max=9)
        nm, em = build_masks(n_atoms, len(n_atoms), d, max_n=9)
        c = torch.log10(1.0 + filler.uniform(f"realloc.ir.{salt}", (len(n_atoms), 1, 3501)))
        return x.to(d), ex.to(d), nm, em, c

    def call(m, x, ex, nm, em, c):
        return m(torch.zeros(x.shape[0], device=d), x, nm, em, context=c, edge_x=ex, noise_level=torch.zeros(x.shape[0], device=d),
                 cond_x=None, cond_edge_x=None)

    x1, ex1, nm1, em1, c1 = inputs([9, 4, 7], 1)
    c1 = c1.to(d)
    call(model, x1, ex1, nm1, em1, c1)
    addr = (nm1.data_ptr(), em1.data_ptr(), c1.data_ptr())
    del nm1, em1, c1
    x2, ex2, nm2, em2, c2h = inputs([5, 9, 3], 2)
    c2 = c2h.to(d)
    reused = (nm2.data_ptr() == addr[0], em2.data_ptr() == addr[1], c2.data_ptr() == addr[2])
    print(f"addresses re-issued by the allocator (node_mask, edge_mask, context): {reused}")
    out = call(model, x2, ex2, nm2, em2, c2)
    from diffspectra_amd.registry import create_model
    fresh = create_model(cfg)                                       # a model that has never seen the first tensors
    fresh.eval()          # the sampling path: a model left in training mode runs the training forward (dropout, BatchNorm batch statistics)
    filler.fill_module_(fresh)
    fresh.eval()
    want = call(fresh, x2, ex2, nm2.clone(), em2.clone(), c2.clone())
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])


def test_forward_rejects_asymmetric_edges(gpu_device):
    """The pair layout stores one value per unordered pair; a directed edge input must raise, not be silently symmetrised."""
    cfg, model = gpu_model("ir", gpu_device)
    a = cases.forward_inputs("ir", False)
    d = gpu_device
    bad = a["edge_x"].clone()
    bad[3, 0, 1, 0] += 0.5
    with pytest.raises(ValueError, match="not symmetric"):
        model(torch.zeros(4, device=d), a["xh"].to(d), a["node_mask"].to(d), a["edge_mask"].to(d), context=a["context"].to(d),
              edge_x=bad.to(d), noise_level=a["noise_level"].to(d), cond_x=a["cond_x"].to(d), cond_edge_x=a["cond_edge_x"].to(d))


def test_c_abi_rejects_oversized_layout(gpu_device):
    """ds_sampler_step / ds_post_process validate the layout themselves (a raw C-ABI caller has no Python Layout class)."""
    import copy
    from diffspectra_amd import engine as E, filler
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    node_mask, edge_mask = filler.masks_from_n_atoms([3, 4])
    L, _ = eng.layout_for(node_mask, edge_mask)
    d = gpu_device
    z9, z2 = torch.zeros(2, 4, 9, device=d), torch.zeros(2, 4, 4, 2, device=d)
    r3, r6, r22 = torch.zeros(2, 4, 3, device=d), torch.zeros(2, 4, 6, device=d), torch.zeros(2, 2, 4, 4, device=d)
    for field, value in (("max_n", 33), ("B", 0)):
        bad = E.DsLayout.from_buffer_copy(L.c)
        setattr(bad, field, value)
        st = eng.lib.ds_sampler_step(C.byref(bad), C.c_float(1), C.c_float(0), C.c_float(0), C.c_float(1), E._ptr(z9), E._ptr(z2),
                                     E._ptr(z9), E._ptr(z2), E._ptr(r3), E._ptr(r6), E._ptr(r22), E._ptr(z9), E._ptr(z2), E._stream())
        assert st == -1
        pos, at, fc, et = torch.zeros(2, 4, 3, device=d), torch.zeros(2, 4, dtype=torch.int32, device=d), torch.zeros(2, 4, dtype=torch.int32, device=d), torch.zeros(2, 4, 4, device=d)
        st = eng.lib.ds_post_process(C.byref(bad), E._ptr(z9), E._ptr(z2), E._ptr(pos), E._ptr(at), E._ptr(fc), E._ptr(et), E._stream())
        assert st == -1


def test_stability_kernel_vs_reference_decisions(gpu_device):
    """ds_check_stability against the scalar restatement of the reference's check (oracle/stability.py, pinned to the
    reference's own get_bond_order by G12) on ragged molecules, and on the G12 threshold sweep itself."""
    from diffspectra_amd import filler
    from diffspectra_amd.stability import check_stability_batch
    from oracle import stability as ost
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    d = gpu_device
    gen = torch.Generator().manual_seed(11)
    n_atoms = [1, 2, 5, 9, 18, 29, 13, 7]
    B, N = len(n_atoms), max(n_atoms)
    pos, mask = torch.zeros(B, N, 3), torch.zeros(B, N)
    types = torch.randint(0, 5, (B, N), generator=gen)
    grid = torch.stack(torch.meshgrid(torch.arange(4.0), torch.arange(4.0), torch.arange(2.0), indexing="ij"), -1).reshape(-1, 3)
    for b, n in enumerate(n_atoms):
        pos[b, :n] = grid[:n] * (1.0 + 0.08 * b) + 0.08 * torch.randn(n, 3, generator=gen)
        mask[b, :n] = 1
    types[3, 0], types[3, 1] = 1, 2
    pos[3, 1] = pos[3, 0] + torch.tensor([1.15, 0.0, 0.0])                     # a C#N pair at triple-bond distance
    stable, nr, cnt, order = check_stability_batch(pos.to(d), types.to(d), mask.to(d), engine=eng)
    seen = set()
    for b, n in enumerate(n_atoms):
        want = ost.check_stability(pos[b, :n].tolist(), types[b, :n].tolist())
        assert (bool(stable[b]), int(nr[b]), int(cnt[b])) == want[:3], b
        assert order[b, :n, :n].cpu().tolist() == want[3], b
        assert int(order[b, n:].abs().sum()) == 0 and int(order[b, :, n:].abs().sum()) == 0
        seen |= {o for row in want[3] for o in row}
    assert seen == {0, 1, 2, 3}
    g = cases.load_npz("g12_bond_orders.npz")                                    # the reference's own decisions
    dist = cases.bond_distance_sweep()
    off = np.abs(dist * 100 - np.round(dist * 100)) > 1e-6
    d32 = torch.from_numpy(dist[off]).float()
    M = len(d32)
    for i in range(5):
        for j in range(5):
            p2 = torch.zeros(M, 2, 3)
            p2[:, 1, 0] = d32
            t2 = torch.tensor([[i, j]]).expand(M, 2).contiguous()
            _, _, _, o = check_stability_batch(p2.to(d), t2.to(d), torch.ones(M, 2, device=d), engine=eng)
            assert o[:, 0, 1].cpu().tolist() == g["orders"][i, j][torch.from_numpy(off)].tolist(), (i, j)
            assert torch.equal(o[:, 0, 1], o[:, 1, 0])


def test_batched_stability_on_device(gpu_device):
    """N4: ``check_stability_batch`` (the HIP kernel behind the package API) equals the reference's pair-by-pair decision
    (``oracle/stability.py``, pinned to ``get_bond_order`` by golden G12) on a fixture that exercises every outcome."""
    from diffspectra_amd.stability import check_stability_batch
    from oracle import stability as ost
    cfg, model = gpu_model("ir", gpu_device)
    eng = model.module.engine()
    gen = torch.Generator().manual_seed(11)
    n_atoms = [1, 2, 5, 9, 18, 29]
    B, N = len(n_atoms), max(n_atoms)
    # positions on a jittered grid with ~1.1-1.6 A spacing so that all four outcomes (none/single/double/triple) occur
    pos = torch.zeros(B, N, 3)
    types = torch.randint(0, 5, (B, N), generator=gen)
    mask = torch.zeros(B, N)
    for b, n in enumerate(n_atoms):
        grid = torch.stack(torch.meshgrid(torch.arange(4.0), torch.arange(4.0), torch.arange(2.0), indexing="ij"), -1).reshape(-1, 3)
        pos[b, :n] = grid[:n] * (1.05 + 0.1 * b) + 0.08 * torch.randn(n, 3, generator=gen)
        mask[b, :n] = 1
    types[3, 0], types[3, 1] = 1, 2                   # a C#N pair at triple-bond distance
    pos[3, 1] = pos[3, 0] + torch.tensor([1.15, 0.0, 0.0])
    d = gpu_device
    stable, nr_stable, cnt, order = (t.cpu() for t in check_stability_batch(pos.to(d), types.to(d), mask.to(d), engine=eng))
    seen = set()
    for b, n in enumerate(n_atoms):
        want = ost.check_stability(pos[b, :n].tolist(), types[b, :n].tolist())
        assert (bool(stable[b]), int(nr_stable[b]), int(cnt[b])) == want[:3]
        assert order[b, :n, :n].tolist() == want[3]
        assert int(order[b, n:].abs().sum()) == 0 and int(order[b, :, n:].abs().sum()) == 0
        seen |= {o for row in want[3] for o in row}
    assert seen == {0, 1, 2, 3}                       # the fixture exercises every branch of get_bond_order
    with pytest.raises(RuntimeError):
        check_stability_batch(pos.to(d), types.to(d), mask.to(d))


def test_two_stream_forward_equals_one_stream_bit_for_bit(gpu_device):
    """``ds_forward`` runs a block's node rows (k_node_update without its node2edge part, then the next block's q|k|v) on a side stream beside
    the pair rows' k_edge_update; ``ds_set_two_stream(0)`` gives the single-stream order of ``ds_stage_block``.  Same kernels, same
    arithmetic: outputs must agree BIT FOR BIT (a missing cross-stream dependency would show up as a difference), on a batch large
    enough for the two sides to overlap, for both branches of the forward, repeatedly."""
    from diffspectra_amd import engine as E, filler
    cfg, model = gpu_model("allspectra", gpu_device)
    eng = model.module.engine()
    lib = E.load_library()
    d = gpu_device
    n_atoms = filler.sample_n_atoms(1500, seed=2).tolist()
    n_atoms[0], n_atoms[1], n_atoms[2] = 29, 1, 2
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "ts.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "ts.c")
    B = len(n_atoms)
    nl = filler.uniform("ts.nl", (B,), -6, 6)
    ctx_emb = filler.normal("ts.ctx", (B, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask, validate=True)
    xd, exd, nld, cxd, cexd, ctxd = (t.to(d) for t in (x, ex, nl, cx, cex, ctx_emb))
    prev = lib.ds_set_two_stream(0)
    try:
        ref = {}
        for first in (True, False):
            o, oe = eng.forward(L, ws, xd, exd, nld, None if first else cxd, None if first else cexd, ctxd)
            ref[first] = (o.clone(), oe.clone())
        lib.ds_set_two_stream(1)
        for rep in range(3):
            for first in (True, False):
                o, oe = eng.forward(L, ws, xd, exd, nld, None if first else cxd, None if first else cexd, ctxd)
                assert torch.equal(o, ref[first][0]) and torch.equal(oe, ref[first][1]), (rep, first)
    finally:
        lib.ds_set_two_stream(prev)
