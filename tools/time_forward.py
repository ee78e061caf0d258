"""Time ds_forward + ds_sampler_step on synthetic QM9-sized batches (development tool; bench.py is the contract)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffspectra_amd import filler  # noqa: E402
from diffspectra_amd.config import qm9s_config  # noqa: E402
from diffspectra_amd.registry import create_model  # noqa: E402
import diffspectra_amd.dmt  # noqa: F401,E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mols", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--version", default="allspectra")
    ap.add_argument("--spec", action="store_true", help="also time the SpecFormer conditioning encoder")
    ap.add_argument("--dropin", action="store_true",
                    help="also time the reference-style loop `model(t, xh, ...)` (weight fingerprint, layout lookup, symmetry checks, SpecFormer "
                         "per call, as dmt.py:348-350 does) against the bare engine.forward")
    args = ap.parse_args()
    d = torch.device("cuda:0")
    cfg = qm9s_config(args.version, device=d)
    model = create_model(cfg)
    filler.fill_module_(model)
    eng = model.module.engine()
    n_atoms = filler.sample_n_atoms(args.mols, seed=0).tolist()
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "tf.x")
    cx, cex, _, _ = filler.synthetic_state(n_atoms, "tf.c")
    B = len(n_atoms)
    nl = torch.full((B,), 0.5)
    ctx = filler.normal("tf.ctx", (B, 1024)) * 0.5
    L, ws = eng.layout_for(node_mask, edge_mask)
    x, ex, cx, cex, nl, ctx = (t.to(d) for t in (x, ex, cx, cex, nl, ctx))
    out = torch.empty(B, L.N, 9, device=d)
    oute = torch.empty(B, L.N, L.N, 2, device=d)
    for _ in range(2):
        eng.forward(L, ws, x, ex, nl, cx, cex, ctx, out, oute)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        eng.forward(L, ws, x, ex, nl, cx, cex, ctx, out, oute)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    E = sum(n * (n - 1) for n in n_atoms)
    macs = 8 * (620544 * sum(n_atoms) + 157184 * E + 2492416 * B) + (233216 * sum(n_atoms) + 33088 * E + 1330176 * B)
    if args.spec:
        spectra = filler.synthetic_spectra(B, args.version, seed=1)
        spectra = [t.to(d) for t in spectra] if isinstance(spectra, list) else spectra.to(d)
        eng.context_embedding(spectra)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.context_embedding(spectra)
        torch.cuda.synchronize()
        print(f"  SpecFormer + cond_lin for {B} molecules: {(time.perf_counter() - t1) * 1e3:.1f} ms (once per 1000 steps)")
    if args.dropin:
        spectra = filler.synthetic_spectra(B, args.version, seed=1)
        spectra = [t.to(d) for t in spectra] if isinstance(spectra, list) else spectra.to(d)
        nm, em = node_mask.to(d), edge_mask.to(d)
        tt = torch.zeros(B, device=d)
        call = lambda: model(tt, x, nm, em, context=spectra, edge_x=ex, noise_level=nl, cond_x=cx, cond_edge_x=cex)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.iters):
            call()
        torch.cuda.synchronize()
        dd = (time.perf_counter() - t1) / args.iters
        eng.context_embedding(spectra)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.iters):
            eng.context_embedding(spectra)
        torch.cuda.synchronize()
        ds_ = (time.perf_counter() - t1) / args.iters
        print(f"  drop-in model(...) call with the same context / mask tensors: {dd * 1e3:.3f} ms = engine.forward {dt * 1e3:.3f} + weight "
              f"fingerprint and host checks {(dd - dt) * 1e3:.3f} ms (the spectra embedding is cached: re-encoding it per call, as the "
              f"reference does, would add {ds_ * 1e3:.3f} ms)")
    import ctypes as C
    names = ["edge_geom", "node_qkv", "attn_fused", "node_update", "edge_update", "equi_pairs"]
    per = []
    for kid in range(6):
        eng.lib.ds_profile_config(C.c_int(kid), C.c_int(1), C.c_int(256))
        for _ in range(3):
            eng.forward(L, ws, x, ex, nl, cx, cex, ctx, out, oute)
        tot, cnt = C.c_double(0), C.c_int64(0)
        eng.lib.ds_profile_read(C.byref(tot), C.byref(cnt))
        per.append(tot.value / max(1, cnt.value))
    eng.lib.ds_profile_config(C.c_int(-1), C.c_int(1), C.c_int(0))
    print("  per-launch ms: " + ", ".join(f"{n} {t:.3f}" for n, t in zip(names, per)) + f" | block sum {sum(per):.3f}")
    if os.environ.get("DIFFSPECTRA_HIP_LIB"):
        ws.t["flags"].zero_()
        eng.forward(L, ws, x, ex, nl, cx, cex, ctx, out, oute)
        torch.cuda.synchronize()
        st = ws.t["flags"][16:64].cpu().view(torch.int64)
        tot = float(st.sum())
        print("  stamp shares (wave 0 cycles per phase): " + ", ".join(f"P{i} {float(v) / max(tot, 1):.3f}" for i, v in enumerate(st.tolist()) if v) + f" | total cycles/WG-launch {tot:.3e}")
    print(f"mols {B} Nn {L.Nn} Pp {L.Pp}: {dt * 1e3:.3f} ms/forward, {dt / B * 1e6:.2f} us/mol-step, "
          f"{2 * macs / dt / 1e12:.2f} TFLOP/s algorithmic, {B / dt / 1000:.1f} mol/s @1000 steps")


if __name__ == "__main__":
    main()
