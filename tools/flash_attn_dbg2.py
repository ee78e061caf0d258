import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from diffspectra_amd import engine as E, train_engine as T
lib = T.load_train_library()
d = torch.device("cuda:0")
B, H, DK, DM, L = 1, 16, 8, 128, 64
gen = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * L, 384, generator=gen) * 0.5).to(d)
dao = torch.randn(B * L, 128, generator=gen).to(d)
scale = DK ** -0.5
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=d)
ast, out = f(B, H, L, 2), f(B * L, DM)
E._check(lib.dst_spec_attn_flash_fwd(E._ptr(qkv), None, None, C.c_int32(1), E._ptr(ast), E._ptr(out), C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), E._stream()), "f")
dq = torch.zeros(B * L, 384, device=d)
E._check(lib.dst_spec_attn_flash_bwd(E._ptr(qkv), None, None, C.c_int32(1), E._ptr(ast), E._ptr(out), E._ptr(dao), E._ptr(dq), None, None, C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), C.c_int32(0), E._stream()), "b")
torch.cuda.synchronize()
x = qkv.double().clone().requires_grad_(True)
v = x.view(B, L, 3, H, DK)
q_, k_, v_ = v[:, :, 0].permute(0, 2, 1, 3), v[:, :, 1].permute(0, 2, 1, 3), v[:, :, 2].permute(0, 2, 1, 3)
P = torch.softmax(scale * q_ @ k_.transpose(-1, -2), -1)
O = P @ v_
(O.permute(0, 2, 1, 3).reshape(B * L, DM) * dao.double()).sum().backward()
gr = x.grad.float()
torch.set_printoptions(precision=4, linewidth=220, sci_mode=False)
print("out rel", float((out - O.permute(0, 2, 1, 3).reshape(B * L, DM).float()).norm() / O.norm()))
for name, c0 in (("dq", 0), ("dk", 128), ("dv", 256)):
    a, b = dq[:, c0:c0 + 8].cpu(), gr[:, c0:c0 + 8].cpu()
    print(name, "head 0, rows 0..5 computed:\n", a[:6], "\n expected:\n", b[:6])
    # which expected row does each computed row match best?
    sim = (a @ b.t())
    print(name, "argmax match of computed rows 0..15 among expected rows:", sim.argmax(1)[:16].tolist(), " row-norm ratio", (a.norm(dim=1) / b.norm(dim=1))[:8].tolist())
