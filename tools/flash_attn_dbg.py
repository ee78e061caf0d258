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
qkv = torch.zeros(B * L, 384)
qkv[:, 256:] = torch.randn(B * L, 128, generator=gen)
qkv = qkv.to(d)
dao = torch.zeros(B * L, 128)
mode = os.environ.get("MODE", "a")
if mode == "a":      # dO[q][d] = 1 for q == 5 only, head 0, d = 2
    dao[5, 2] = 1.0
elif mode == "b":
    dao[37, 6] = 1.0
dao = dao.to(d)
f = lambda *s: torch.empty(*s, dtype=torch.float32, device=d)
ast, out = f(B, H, L, 2), f(B * L, DM)
E._check(lib.dst_spec_attn_flash_fwd(E._ptr(qkv), None, None, C.c_int32(1), E._ptr(ast), E._ptr(out), C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(DK ** -0.5), E._stream()), "f")
dq = torch.zeros(B * L, 384, device=d)
E._check(lib.dst_spec_attn_flash_bwd(E._ptr(qkv), None, None, C.c_int32(1), E._ptr(ast), E._ptr(out), E._ptr(dao), E._ptr(dq), None, None, C.c_int32(B), C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(DK ** -0.5), C.c_int32(0), E._stream()), "b")
torch.cuda.synchronize()
print("stats m,l row0:", ast[0, 0, 0].tolist(), " expected (0, L)")
dv = dq[:, 256:264].cpu()      # head 0
print("expected dv[k][d] = 1/L =", 1.0 / L, "at the one (d) column for every k")
torch.set_printoptions(precision=4, linewidth=200)
print(dv[:8])
print("nonzero columns:", (dv.abs().sum(0) > 1e-9).nonzero().flatten().tolist(), " column sums:", dv.sum(0).tolist())
print("norms dq dk dv (all heads):", float(dq[:, :128].norm()), float(dq[:, 128:256].norm()), float(dq[:, 256:].norm()))
nz = dq[:, 256:].nonzero()
print("nonzero dv entries:", nz.shape[0], nz[:6].tolist(), dq[:, 256:][dq[:, 256:] != 0][:6].tolist())
