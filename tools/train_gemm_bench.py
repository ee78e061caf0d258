"""Time dst_gemm on the shapes of one config-5 training step (256 molecules: ~4.6 k node rows, ~40 k pair rows, ~81 k directed rows).
    python tools/train_gemm_bench.py            (new kernel)
    DST_GEMM_OLD=1 python tools/train_gemm_bench.py   (round 3's kernel, same epilogue)
    DST_GEMM_BN=64 ...                           (force the tile width)
Prints per shape: microseconds per call, algorithmic TFLOP/s, and the total over the listed calls."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from diffspectra_amd import train_engine as T

d = torch.device("cuda:0")
o = T.Ops(d)
o.bf16 = os.environ.get("PREC", "bf16") == "bf16"
Nn, Pp, D, B, Ls = 4600, 40400, 80800, 256, 256 * 347
# (name, M, N, K, ta, tb, calls per step)
SHAPES = [
    ("fwd node 256->768 (qkv as 3)", Nn, 252, 256, False, True, 24), ("fwd node ff1 256->512", Nn, 512, 256, False, True, 8),
    ("fwd node ff2 512->256", Nn, 256, 512, False, True, 8), ("fwd pair 64->256 (te0/te1)", Pp, 256, 64, False, True, 16),
    ("fwd pair 128->64 edge_emb", Pp, 64, 128, False, True, 8), ("fwd pair ff3 64->128", Pp, 128, 64, False, True, 8),
    ("fwd pair ff4 128->64", Pp, 64, 128, False, True, 8), ("fwd pair ed 128->256", Pp, 256, 128, False, True, 8),
    ("fwd directed coord0 256->256", D, 256, 256, False, True, 8), ("fwd directed coord2 256->3", D, 3, 256, False, True, 8),
    ("dX directed coord0", D, 256, 256, False, False, 8), ("dW directed coord0", 256, 256, D, True, False, 8),
    ("dX pair ed 256->128", Pp, 64, 256, False, False, 16), ("dW pair ed", 256, 128, Pp, True, False, 8),
    ("dX pair te 256->64", Pp, 64, 256, False, False, 16), ("dW pair te", 256, 64, Pp, True, False, 16),
    ("dW pair ff3", 128, 64, Pp, True, False, 8), ("dW pair ff4", 64, 128, Pp, True, False, 8),
    ("dX node ff1", Nn, 256, 512, False, False, 8), ("dW node ff1", 512, 256, Nn, True, False, 8),
    ("dW node qkv", 252, 256, Nn, True, False, 24), ("adaLN table fwd", B, 19744, 1024, False, True, 1),
    ("adaLN table dW", 19744, 1024, B, True, False, 1), ("adaLN table dX", B, 1024, 19744, False, False, 1),
    ("spec tokens 128->128", Ls, 128, 128, False, True, 12), ("spec tokens ff 128->256", Ls, 256, 128, False, True, 3),
    ("spec dW tokens 128x128", 128, 128, Ls, True, False, 12), ("spec dX tokens", Ls, 128, 128, False, False, 12),
    ("spec head fwd", B, 256, 44416, False, True, 1), ("spec head dW", 256, 44416, B, True, False, 1), ("spec head dX", B, 44416, 256, False, False, 1),
]
tot_us, tot_flop = 0.0, 0.0
rows = []
for name, M, N, K, ta, tb, calls in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=d)
    Bm = torch.randn((N, K) if tb else (K, N), device=d)
    Cd = torch.empty(M, N, device=d)
    for _ in range(3):
        o.gemm(T.mv(A), T.mv(Bm), T.mv(Cd), ta, tb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        o.gemm(T.mv(A), T.mv(Bm), T.mv(Cd), ta, tb)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flop = 2.0 * M * N * K
    rows.append((name, M, N, K, us, flop / us / 1e6, calls))
    tot_us += us * calls
    tot_flop += flop * calls
    del A, Bm, Cd
for r in rows:
    print(f"{r[0]:34s} M {r[1]:6d} N {r[2]:6d} K {r[3]:6d}  {r[4]:9.1f} us  {r[5]:7.1f} TFLOP/s  x{r[6]}")
print(json.dumps({"mode": "old" if os.environ.get("DST_GEMM_OLD") == "1" else "new", "bn": os.environ.get("DST_GEMM_BN", "auto"),
                  "total_ms_listed_calls": tot_us / 1e3, "tflops": tot_flop / tot_us / 1e6}))
