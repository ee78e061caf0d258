"""One-off check that a batch whose pair tensors exceed 2^31 bytes gives the same per-molecule outputs as small batches
(catches 32-bit offset overflow in the kernels / buffer descriptors).  Development tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffspectra_amd import filler  # noqa: E402
from diffspectra_amd.config import qm9s_config  # noqa: E402
from diffspectra_amd.registry import create_model  # noqa: E402
import diffspectra_amd.dmt  # noqa: F401,E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 14000
    d = torch.device("cuda:0")
    cfg = qm9s_config("allspectra", device=d)
    model = create_model(cfg)
    filler.fill_module_(model)
    eng = model.module.engine()
    n_atoms = filler.sample_n_atoms(M, seed=0).tolist()
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "bb.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "bb.c")
    nl = torch.full((M,), 0.5)
    ctx = filler.normal("bb.ctx", (M, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask)
    print(f"molecules {M}: Nn {L.Nn} Pp {L.Pp}; ed bytes {L.Pp * 1024 / 2**30:.2f} GiB", flush=True)
    out, oute = eng.forward(L, ws, x.to(d), ex.to(d), nl.to(d), cx.to(d), cex.to(d), ctx.to(d))
    out, oute = out.cpu(), oute.cpu()
    worst = 0.0
    for lo in (0, M // 2 - 256, M - 512):
        sl = slice(lo, lo + 512)
        nm, em = filler.masks_from_n_atoms(n_atoms[sl])
        N = nm.shape[1]
        L2, ws2 = eng.layout_for(nm, em)
        o2, e2 = eng.forward(L2, ws2, x[sl, :N].contiguous().to(d), ex[sl, :N, :N].contiguous().to(d), nl[sl].to(d),
                             cx[sl, :N].contiguous().to(d), cex[sl, :N, :N].contiguous().to(d), ctx[sl].to(d))
        dx = float((o2.cpu() - out[sl, :N]).abs().max())
        de = float((e2.cpu() - oute[sl, :N, :N]).abs().max())
        print(f"  molecules {lo}..{lo + 512}: max |diff| nodes {dx:.2e} edges {de:.2e}", flush=True)
        worst = max(worst, dx, de)
    assert torch.isfinite(out).all() and worst < 2e-5, worst
    print("OK")


if __name__ == "__main__":
    main()
