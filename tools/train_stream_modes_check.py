"""Race detector for the three-stream training step: N optimizer steps from the same initial state and seeds in the three-stream order and in
the single-stream order must give the same loss BITS at every step (every kernel reduces in a fixed order; a missing cross-stream dependency
or a buffer handed out too early shows up as a difference that the following steps amplify).
    python tools/train_stream_modes_check.py [steps] [molecules]"""
import os, random, struct, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from diffspectra_amd import filler, losses as Lh
from diffspectra_amd.config import qm9s_config
from diffspectra_amd.ema import ExponentialMovingAverage
from diffspectra_amd.noise_schedule import NoiseScheduleVP
from diffspectra_amd.registry import create_model
import diffspectra_amd.dmt  # noqa: F401

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Bt = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = torch.device("cuda:0")


def run(mode):
    os.environ["DIFFSPECTRA_NODE_STREAM"], os.environ["DIFFSPECTRA_ASYNC_DW"] = mode
    cfg = qm9s_config("allspectra", device=d)
    cfg.training.precision = "bf16"
    model = create_model(cfg)
    filler.fill_module_(model)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
    opt = Lh.get_optimizer(cfg, model.parameters())
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    step_fn = Lh.get_step_fn(ns, True, Lh.optimization_manager(cfg), None, cfg)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    n_atoms = filler.sample_n_atoms(Bt, seed=3).tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    gen = torch.Generator().manual_seed(11)
    types = torch.randint(0, 5, (Bt, N), generator=gen)
    order = torch.triu((torch.rand(Bt, N, N, generator=gen) > 0.8).float() * torch.randint(1, 4, (Bt, N, N), generator=gen), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(Bt, N, N)
    ctx = filler.synthetic_spectra(Bt, "allspectra", seed=5)
    batch = dict(positions=(torch.randn(Bt, N, 3, generator=gen) * 1.3 * node_mask).to(d), atom_mask=node_mask.squeeze(-1).to(d),
                 edge_mask=edge_mask.to(d), atom_one_hot=(F.one_hot(types, 5).float() * node_mask).to(d),
                 edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1).to(d), formal_charges=torch.zeros(Bt, N, 1, device=d),
                 context=[c.to(d) for c in ctx])
    torch.manual_seed(0)
    random.seed(1234)
    out = []
    for _ in range(steps):
        out.append(step_fn(state, batch).detach())
    torch.cuda.synchronize()
    return [struct.unpack("<I", struct.pack("<f", float(x)))[0] for x in out], [float(x) for x in out]


a_bits, a = run(("1", "1"))
b_bits, b = run(("0", "0"))
c_bits, c = run(("1", "1"))
first = next((i for i, (x, y, z) in enumerate(zip(a_bits, b_bits, c_bits)) if not (x == y == z)), None)
print(f"{steps} steps, {Bt} molecules: losses {a[0]:.6f} -> {a[-1]:.6f}; three-stream vs single-stream vs three-stream again: "
      + ("identical bits at every step" if first is None else f"FIRST DIFFERENCE at step {first}: {a[first]!r} {b[first]!r} {c[first]!r}"))
sys.exit(0 if first is None else 1)
