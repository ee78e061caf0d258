"""Time dst_gemm on shapes given as M,N,K,ta,tb[,acc] arguments (bf16 operands unless PREC=fp32): microseconds per call, warm operands.
    python tools/gemm_probe.py 4600,256,512,0,1 4600,256,512,0,0"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from diffspectra_amd import train_engine as T

d = torch.device("cuda:0")
o = T.Ops(d)
o.bf16 = os.environ.get("PREC", "bf16") == "bf16"
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    M, N, K, ta, tb = v[:5]
    acc = bool(v[5]) if len(v) > 5 else False
    A = torch.randn((K, M) if ta else (M, K), device=d)
    Bm = torch.randn((N, K) if tb else (K, N), device=d)
    Cd = torch.zeros(M, N, device=d)
    for _ in range(5):
        o.gemm(T.mv(A), T.mv(Bm), T.mv(Cd), bool(ta), bool(tb), acc=acc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        o.gemm(T.mv(A), T.mv(Bm), T.mv(Cd), bool(ta), bool(tb), acc=acc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"M {M:6d} N {N:6d} K {K:6d} ta {ta} tb {tb} acc {int(acc)}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
