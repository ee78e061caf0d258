"""Does running two independent molecule batches on two HIP streams beat running them back to back?  (development tool)"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffspectra_amd import filler  # noqa: E402
from diffspectra_amd.config import qm9s_config  # noqa: E402
from diffspectra_amd.registry import create_model  # noqa: E402
from diffspectra_amd.engine import Layout, Workspace  # noqa: E402
import diffspectra_amd.dmt  # noqa: F401,E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=2048, help="molecules per batch (two batches are run)")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    d = torch.device("cuda:0")
    cfg = qm9s_config("allspectra", device=d)
    model = create_model(cfg)
    filler.fill_module_(model)
    eng = model.module.engine()
    batches = []
    for k in range(2):
        n_atoms = filler.sample_n_atoms(args.mols, seed=k).tolist()
        x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "tf.x")
        cx, cex, _, _ = filler.synthetic_state(n_atoms, "tf.c")
        B = len(n_atoms)
        L = Layout(node_mask, d)
        ws = Workspace(L, d)
        t = [v.to(d) for v in (x, ex, torch.full((B,), 0.5), cx, cex, filler.normal("tf.ctx", (B, 1024)) * 0.5)]
        out = torch.empty(B, L.N, 9, device=d)
        oute = torch.empty(B, L.N, L.N, 2, device=d)
        batches.append((L, ws, t, out, oute))

    def fwd(b):
        L, ws, (x, ex, nl, cx, cex, ctx), out, oute = b
        eng.forward(L, ws, x, ex, nl, cx, cex, ctx, out, oute)

    for b in batches:
        fwd(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        for b in batches:
            fwd(b)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / args.iters
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        for s, b in zip(streams, batches):
            with torch.cuda.stream(s):
                fwd(b)
    torch.cuda.synchronize()
    par = (time.perf_counter() - t0) / args.iters
    print(f"2 x {args.mols} molecules: back to back {seq * 1e3:.2f} ms, two streams {par * 1e3:.2f} ms ({seq / par:.3f}x)")


if __name__ == "__main__":
    main()
