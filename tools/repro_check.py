"""Runs the same training step several times under given stream modes and reports whether the gradients repeat bit for bit (development tool)."""
import math
import os
import random as _random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from diffspectra_amd import filler, losses as Lh
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.registry import create_model
    import diffspectra_amd.dmt  # noqa: F401
    d = torch.device("cuda:0")
    cfg = qm9s_config("allspectra", device=d)
    cfg.training.precision = "bf16"
    model = create_model(cfg)
    filler.fill_module_(model)
    Bt = 96
    n_atoms = filler.sample_n_atoms(Bt, seed=3).tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    g = torch.Generator().manual_seed(11)
    types = torch.randint(0, 5, (Bt, N), generator=g)
    order = torch.triu((torch.rand(Bt, N, N, generator=g) > 0.8).float() * torch.randint(1, 4, (Bt, N, N), generator=g), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(Bt, N, N)
    ctx = filler.synthetic_spectra(Bt, "allspectra", seed=5)
    batch = dict(positions=(torch.randn(Bt, N, 3, generator=g) * 1.3 * node_mask).to(d), atom_mask=node_mask.squeeze(-1).to(d),
                 edge_mask=edge_mask.to(d), atom_one_hot=(F.one_hot(types, 5).float() * node_mask).to(d),
                 edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1).to(d), formal_charges=torch.zeros(Bt, N, 1, device=d),
                 context=[c.to(d) for c in ctx])
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0), True, None, cfg)
    params = [p for p in model.parameters() if p.requires_grad]
    names = [n_ for n_, p in model.named_parameters() if p.requires_grad]
    bn = [b for n_, b in model.named_buffers() if "running" in n_]
    bn0 = [b.detach().clone() for b in bn]
    Lh.random = lambda: 0.0

    def run():
        for b, b0 in zip(bn, bn0):
            b.copy_(b0)
        for p in params:
            p.grad = None
        torch.manual_seed(123)
        _random.seed(7)
        loss = loss_fn(model, batch)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), [p.grad.detach().clone() for p in params]

    l0, g0 = run()
    bad_runs = 0
    prev, vs_prev = g0, 0
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        l1, g1 = run()
        diff = [n_ for n_, a, b in zip(names, g1, g0) if not torch.equal(a, b)]
        vs_prev += int(any(not torch.equal(a, b) for a, b in zip(g1, prev)))
        prev = g1
        if diff:
            bad_runs += 1
            blocks = sorted({n_.split(".")[1] for n_ in diff if n_.startswith("module.e_block_")})
            print(f"rep {rep}: {len(diff)} of {len(names)} gradients differ; blocks {blocks}; first {diff[:3]}", flush=True)
    print("runs that differ:", bad_runs, "from the first run;", vs_prev, "from the run before", flush=True)


if __name__ == "__main__":
    main()
