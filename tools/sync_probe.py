"""Lists the host-device synchronisations of one training step (development tool): torch's sync debug mode warns at every synchronising
torch call (.item(), .cpu(), .tolist(), nonzero ...) with the Python stack that made it."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import argparse
    a = argparse.Namespace(spectra="allspectra", precision="bf16", force_collectives=False, train_batch=256, backend="nccl")
    from diffspectra_amd import filler, losses as Lh
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.ema import ExponentialMovingAverage
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.registry import create_model
    import diffspectra_amd.dmt  # noqa: F401
    device = torch.device("cuda:0")
    cfg = qm9s_config(a.spectra, device=device)
    cfg.training.precision = a.precision
    model = create_model(cfg)
    filler.fill_module_(model)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
    opt = Lh.get_optimizer(cfg, model.parameters())
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    step_fn = Lh.get_step_fn(ns, True, Lh.optimization_manager(cfg), None, cfg)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    Bt = a.train_batch
    n_atoms = filler.sample_n_atoms(Bt, seed=3).tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    g = torch.Generator().manual_seed(11)
    types = torch.randint(0, 5, (Bt, N), generator=g)
    order = torch.triu((torch.rand(Bt, N, N, generator=g) > 0.8).float() * torch.randint(1, 4, (Bt, N, N), generator=g), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(Bt, N, N)
    ctx = filler.synthetic_spectra(Bt, a.spectra, seed=5)
    batch = dict(positions=(torch.randn(Bt, N, 3, generator=g) * 1.3 * node_mask).to(device), atom_mask=node_mask.squeeze(-1).to(device),
                 edge_mask=edge_mask.to(device), atom_one_hot=(torch.nn.functional.one_hot(types, 5).float() * node_mask).to(device),
                 edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1).to(device), formal_charges=torch.zeros(Bt, N, 1, device=device),
                 context=[c.to(device) for c in ctx] if isinstance(ctx, list) else ctx.to(device))
    for _ in range(3):
        step_fn(state, batch)
    torch.cuda.synchronize()
    if "--profile" in sys.argv:                       # where the HOST spends a step (cumulative time per function, ten steps)
        import cProfile
        import pstats
        import random
        random.seed(1234)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            step_fn(state, batch)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(28)
        return
    import traceback
    def show(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack() if "diffspectra_amd" in f.filename]
        print("SYNC:", str(message)[:80], "<-", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:]), flush=True)
    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    step_fn(state, batch)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
