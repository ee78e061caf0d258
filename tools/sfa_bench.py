"""Times the score-free SpecFormer attention kernels (csrc/ds_train_attn.hip) at BASELINE config 5's shape - 256 molecules, 347 patches,
16 heads of 8 - one launch kind at a time (development tool; DIFFSPECTRA_HIP_LIB picks a variant build, tools/variant_build.py).

    python tools/sfa_bench.py [--mols 256] [--patches 347] [--iters 10]
Prints, per layer count 1..3, the average microseconds of the forward, the query-side and the key-side backward, and a checksum of every
output so that variants can be compared bit for bit."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffspectra_amd import engine as E, train_engine as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=256)
    ap.add_argument("--patches", type=int, default=347)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    lib = T.load_train_library()
    d = torch.device("cuda:0")
    B, L, H, DK, DM = a.mols, a.patches, 16, 8, 128
    gen = torch.Generator().manual_seed(1)
    qkv = [torch.randn(B * L, 3 * DM, generator=gen).to(d) for _ in range(3)]
    dao = torch.randn(B * L, DM, generator=gen).to(d)
    scale = DK ** -0.5
    st, out = torch.empty(B, H, L, 2, device=d), torch.empty(B * L, DM, device=d)
    dq = [torch.zeros(B * L, 3 * DM, device=d) for _ in range(3)]

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    for nl in (1, 2, 3):
        qp = [E._ptr(q) for q in qkv[:nl]] + [None] * (3 - nl)
        gp = [E._ptr(q) for q in dq[:nl]] + [None] * (3 - nl)

        def fwd():
            E._check(lib.dst_spec_attn_flash_fwd(qp[0], qp[1], qp[2], C.c_int32(nl), E._ptr(st), E._ptr(out), C.c_int32(B), C.c_int32(L), C.c_int32(H),
                                                 C.c_int32(DK), C.c_float(scale), E._stream()), "fwd")

        def bwd(part, acc=1):
            E._check(lib.dst_spec_attn_flash_bwd(qp[0], qp[1], qp[2], C.c_int32(nl), E._ptr(st), E._ptr(out), E._ptr(dao), gp[0], gp[1], gp[2], C.c_int32(B),
                                                 C.c_int32(L), C.c_int32(H), C.c_int32(DK), C.c_float(scale), C.c_int32(part), C.c_int32(acc), E._stream()), "bwd")

        tf = timed(fwd)
        tq = timed(lambda: bwd(1))
        tk = timed(lambda: bwd(2))
        tq0 = timed(lambda: bwd(1, 0))                      # the assigning flavour (the last layer's call)
        tk0 = timed(lambda: bwd(2, 0))
        for g in dq:
            g.zero_()
        fwd()
        bwd(0)
        torch.cuda.synchronize()
        cs = [float(out.double().sum()), float(st.double().sum())] + [float(g.double().abs().sum()) for g in dq[:nl]]
        print(f"layers {nl}: fwd {tf:7.1f} us  bwd_q {tq:7.1f} us  bwd_kv {tk:7.1f} us  assigning: {tq0:7.1f} {tk0:7.1f} us   checksums " + " ".join(f"{c:.10e}" for c in cs), flush=True)


if __name__ == "__main__":
    main()
