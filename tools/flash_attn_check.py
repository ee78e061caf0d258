"""Kernel-level comparison: dst_spec_attn_flash_{fwd,bwd} against the materialised-score kernels, per layer and per q / k / v part."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from diffspectra_amd import engine as E, train_engine as T
lib = T.load_train_library()
d = torch.device("cuda:0")
B, H, DK, DM = 3, 16, 8, 128
L = int(os.environ.get("L", "347"))
SC = float(os.environ.get("SC", "1.0"))
gen = torch.Generator().manual_seed(0)
qkv = [(torch.randn(B * L, 3 * DM, generator=gen) * SC).to(d) for _ in range(3)]
dao = [(torch.randn(B * L, DM, generator=gen)).to(d) for _ in range(3)]
scale = DK ** -0.5
Lp = (L + 31) // 32 * 32
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=d)
s = E._stream
# materialised chain
prev, sc, st, ao = None, [], [], []
for l in range(3):
    scores, ast, out = f(B, H, L, Lp), f(B, H, L, 2), f(B * L, DM)
    E._check(lib.dst_spec_attn_fwd(E._ptr(qkv[l]), E._ptr(prev), E._ptr(scores), E._ptr(ast), E._ptr(out), C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), s()), "fwd")
    sc.append(scores); st.append(ast); ao.append(out); prev = scores
dq_ref, dsin = [None] * 3, None
for l in (2, 1, 0):
    dqkv, dscores = f(B * L, 3 * DM), f(B, H, L, Lp)
    E._check(lib.dst_spec_attn_bwd(E._ptr(qkv[l]), E._ptr(sc[l]), E._ptr(st[l]), E._ptr(dao[l]), E._ptr(dsin), E._ptr(dqkv), E._ptr(dscores), C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), s()), "bwd")
    dq_ref[l] = dqkv
    dsin = dscores if l > 0 else None
# flash
ao2, st2 = [], []
for l in range(3):
    ast, out = f(B, H, L, 2), f(B * L, DM)
    qp = [E._ptr(q) for q in qkv[:l + 1]] + [None] * (2 - l)
    E._check(lib.dst_spec_attn_flash_fwd(qp[0], qp[1], qp[2], C.c_int32(l + 1), E._ptr(ast), E._ptr(out), C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), s()), "ffwd")
    ao2.append(out); st2.append(ast)
dq = [torch.zeros(B * L, 3 * DM, device=d) for _ in range(3)]
for l in (2, 1, 0):
    qp = [E._ptr(q) for q in qkv[:l + 1]] + [None] * (2 - l)
    gp = [E._ptr(q) for q in dq[:l + 1]] + [None] * (2 - l)
    E._check(lib.dst_spec_attn_flash_bwd(qp[0], qp[1], qp[2], C.c_int32(l + 1), E._ptr(st2[l]), E._ptr(ao2[l]), E._ptr(dao[l]), gp[0], gp[1], gp[2], C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), C.c_int32(0), C.c_int32(0 if l == 2 else 1), s()), "fbwd")
torch.cuda.synchronize()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
for l in range(3):
    print(f"layer {l}: out rel {rel(ao2[l], ao[l]):.4f}   dq rel {rel(dq[l][:, :128], dq_ref[l][:, :128]):.4f}  dk rel {rel(dq[l][:, 128:256], dq_ref[l][:, 128:256]):.4f}  "
          f"dv rel {rel(dq[l][:, 256:], dq_ref[l][:, 256:]):.4f}   norms {float(dq_ref[l][:, :128].norm()):.2f} {float(dq_ref[l][:, 128:256].norm()):.2f} {float(dq_ref[l][:, 256:].norm()):.2f}")
# independent reference: torch autograd in fp64 on the GPU
qr = [q.double().clone().requires_grad_(True) for q in qkv]
loss = 0.0
Sprev = 0.0
for l in range(3):
    x = qr[l].view(B, L, 3, H, DK)
    q_, k_, v_ = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    S = scale * q_ @ k_.transpose(-1, -2) + Sprev
    O = torch.softmax(S, -1) @ v_
    loss = loss + (O.permute(0, 2, 1, 3).reshape(B * L, DM) * dao[l].double()).sum()
    Sprev = S
loss.backward()
for l in range(3):
    gr = qr[l].grad.float()
    print(f"vs torch layer {l}: flash dq {rel(dq[l][:, :128], gr[:, :128]):.4f} dk {rel(dq[l][:, 128:256], gr[:, 128:256]):.4f} dv {rel(dq[l][:, 256:], gr[:, 256:]):.4f} | "
          f"materialised dq {rel(dq_ref[l][:, :128], gr[:, :128]):.4f} dk {rel(dq_ref[l][:, 128:256], gr[:, 128:256]):.4f} dv {rel(dq_ref[l][:, 256:], gr[:, 256:]):.4f}")
cosr = lambda a, b: (float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30)))
for l in range(3):
    gr = qr[l].grad.float()
    a, b_ = dq[l][:, 256:].view(B, L, H, DK), gr[:, 256:].view(B, L, H, DK)
    print(f"dv layer {l}: cos/ratio all {cosr(a, b_)}; head0 {cosr(a[:, :, 0], b_[:, :, 0])}; first 4 dims {cosr(a[..., :4], b_[..., :4])} last 4 dims {cosr(a[..., 4:], b_[..., 4:])}; keys<32 {cosr(a[:, :32], b_[:, :32])}")
    # is it a permutation of the dims?
    best = [max(range(DK), key=lambda e2: float((a[..., e1] * b_[..., e2]).sum())) for e1 in range(DK)]
    print("   best-matching reference dim for each computed dim:", best)
