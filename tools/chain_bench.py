"""Stand-alone timing of the fused training chains (dst_pair_front_fwd / dst_pair_chain_fwd / dst_dir_chain_fwd) through the C-ABI, with and
without the tape stores (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffspectra_amd import filler, train_engine as T  # noqa: E402
from diffspectra_amd.train_engine import ADA, ADA_STRIDE, DIST_OFF, EDGE_OFF, EQUI_OFF  # noqa: E402


def main():
    d = torch.device("cuda:0")
    B = int(os.environ.get("B", "256"))
    n_atoms = filler.sample_n_atoms(B, seed=3).tolist()
    node_mask, _ = filler.masks_from_n_atoms(n_atoms)
    TL = T.TrainLayout(node_mask, d)
    o = T.Ops(d)
    Nn, Pp, D = TL.Nn, TL.Pp, 2 * TL.Pp
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(d)
    ada, u, e, X1, pos, ac, ed = r(B, ADA), r(Nn, 64), r(Pp, 64), r(Pp, 128), r(Nn, 3), r(Nn, 512), r(Pp, 256)
    W3, b3, W4, b4, Win, bed, Wro, bro, n2eb = r(128, 64), r(128), r(64, 128), r(64), r(256, 640), r(256), r(16, 64), r(16), r(64)
    means, stds, Wee, bee, Wte = r(63), r(63).abs() + 0.5, r(64, 128), r(64), r(512, 64)
    W0, b0, W2 = r(256, 256), r(256), r(3, 256)
    f = lambda *s: torch.empty(*s, device=d)
    hb = lambda t: t.contiguous().to(torch.bfloat16)                  # the chains take their weights as bf16 (dst_pack_bf16_pieces in the product)
    W3, W4, Wedb, Wro, Wee, Wte, W0, W2 = hb(W3), hb(W4), hb(Win[:, 512:640]), hb(Wro), hb(Wee), hb(Wte), hb(W0), hb(W2)

    def run(name, fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(f"{name:34s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us")

    print(f"B {B} Nn {Nn} Pp {Pp}")
    for save in (True, False):
        outs = dict(e_out=f(Pp, 64), ed=f(Pp, 256), ro=f(Pp, 16))
        if save:
            outs.update(he=f(Pp, 64), xe1=f(Pp, 64), st=f(Pp, 2), ye1=f(Pp, 64), f3=f(Pp, 128), s3=f(Pp, 128), f4=f(Pp, 64), X2=f(Pp, 128))
        run(f"pair_chain_fwd save={save}", lambda: o.pair_chain_fwd(TL, u, n2eb, e, X1, 128, ada, EDGE_OFF + 128, EDGE_OFF + 192, EDGE_OFF + 256, EDGE_OFF + 320,
                                                                   W3, b3, W4, b4, Wedb, 128, bed, Wro, bro, (0.1, 1234, 2, 3), outs))
        outs = dict(X1=f(Pp, 128), te=f(Pp, 512))
        if save:
            outs.update(xs=f(Pp), d2=f(Pp), e1=f(Pp, 64), st=f(Pp, 2), en=f(Pp, 64))
        run(f"pair_front_fwd save={save}", lambda: o.pair_front_fwd(TL, pos, ada, DIST_OFF, EDGE_OFF, EDGE_OFF + 64, means, stds, e, Wee, bee, Wte, outs))
        outs = dict(c2=f(D, 3))
        if save:
            outs.update(zz=f(D, 256), st=f(D, 2), zn=f(D, 256), c0=f(D, 256), sc0=f(D, 256))
        run(f"dir_chain_fwd save={save}", lambda: o.dir_chain_fwd(TL, ac, ed, ada, EQUI_OFF, EQUI_OFF + 256, W0, b0, W2, outs))


if __name__ == "__main__":
    main()
