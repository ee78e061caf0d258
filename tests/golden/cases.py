"""Shared definitions of the golden cases: inputs are regenerated procedurally on both sides
(the generator that imports the reference, and the tests that never see it), so the committed
fixtures hold expected OUTPUTS only."""
from __future__ import annotations

import os

import numpy as np
import torch

from diffspectra_amd import filler
from diffspectra_amd.config import qm9s_config, SPECTRUM_LENGTHS

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
RAGGED = [3, 9, 18, 29]          # n_atoms of the B=4 parity batch
FULL_LENGTH_ATOMS = [4, 7, 11]   # molecules of the 1000-step golden trajectory (G7)


def fixture_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name)


def spectra_for(version: str, batch: int, salt: int = 0):
    """Non-negative log10(1+u)-style spectra, names keyed so every case is reproducible."""
    specs = [torch.log10(1.0 + filler.uniform(f"spectra.{n}", (batch, 1, L), salt=salt))
             for n, L in zip(("uv", "ir", "raman"), SPECTRUM_LENGTHS)]
    if version == "allspectra":
        return specs
    return specs[{"uv": 0, "ir": 1, "raman": 2}[version]]


def forward_inputs(version: str, first_step: bool, n_atoms=RAGGED, salt: int = 0):
    """Inputs of one ``model(...)`` call (dmt.py:306-321) for the G4 cases."""
    xh, edge_x, node_mask, edge_mask = filler.synthetic_state(n_atoms, "g4.x", salt=salt)
    B = len(n_atoms)
    noise_level = filler.uniform("g4.noise_level", (B,), -6.0, 6.0, salt=salt)
    if first_step:
        cond_x = cond_edge_x = None
    else:
        cond_x, cond_edge_x, _, _ = filler.synthetic_state(n_atoms, "g4.cond", salt=salt)
        cond_x = cond_x * 0.7
        cond_edge_x = cond_edge_x * 0.5
    return dict(xh=xh, node_mask=node_mask, edge_mask=edge_mask, edge_x=edge_x, noise_level=noise_level,
                cond_x=cond_x, cond_edge_x=cond_edge_x, context=spectra_for(version, B, salt))


def trajectory_inputs(version: str, steps: int, n_atoms=RAGGED, salt: int = 0):
    """Initial state + per-step raw randn draws (order pos, feat, edge — sampling.py:611-612,623-624)."""
    B, N = len(n_atoms), max(n_atoms)
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    raw0 = (filler.normal("g5.init.pos", (B, N, 3), salt), filler.normal("g5.init.feat", (B, N, 6), salt),
            filler.normal("g5.init.edge", (B, 2, N, N), salt))
    raws = [(filler.normal(f"g5.step{i}.pos", (B, N, 3), salt), filler.normal(f"g5.step{i}.feat", (B, N, 6), salt),
             filler.normal(f"g5.step{i}.edge", (B, 2, N, N), salt)) for i in range(steps)]
    return dict(node_mask=node_mask, edge_mask=edge_mask, raw0=raw0, raws=raws,
                context=spectra_for(version, B, salt), n_atoms=list(n_atoms))


def block_inputs(n_atoms=RAGGED, salt: int = 0):
    """Inputs of one EquivariantMixBlock call (dmt.py:122) for the G3 component cases."""
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    B, N, _ = node_mask.shape
    nm = node_mask.reshape(-1, 1)
    pos = (filler.normal("g3.pos", (B * N, 3), salt) * 1.5) * nm
    h = filler.normal("g3.h", (B * N, 256), salt) * nm
    adj = edge_mask.reshape(B, N, N)
    b, i, j = adj.nonzero(as_tuple=True)
    edge_index = torch.stack([b * N + i, b * N + j])
    # symmetric edge features / adjacency heads, as every caller on the path provides
    e_dense = filler.normal("g3.e", (B, N, N, 64), salt)
    e_dense = 0.5 * (e_dense + e_dense.transpose(1, 2))
    a_dense = (filler.uniform("g3.adj", (B, N, N, 2), salt=salt) > 0.5).float()
    a_dense = torch.maximum(a_dense, a_dense.transpose(1, 2))
    temb = filler.normal("g3.temb", (B, 1024), salt)
    return dict(pos=pos, h=h, edge_attr=e_dense[b, i, j], edge_index=edge_index, node_mask=nm,
                extra_heads=a_dense[b, i, j], node_time_emb=temb.repeat_interleave(N, 0), edge_time_emb=temb[b],
                temb=temb, n_atoms=list(n_atoms), N=N, B=B, dense_index=(b, i, j))


READOUT_GAIN_KEYS = ("node_pred_mlp.4.", "edge_type_mlp.4.", "edge_exist_mlp.4.")


def readout_gain(state_dict, gain: float = 8.0):
    """G8 only: multiply the last readout layers by ``gain`` so that predictions leave the self-conditioning clamp
    range (random-init readouts stay inside it and the clamp would never act).  Works with or without ``module.``."""
    out = dict(state_dict)
    for k, v in state_dict.items():
        if any(key in k for key in READOUT_GAIN_KEYS):
            out[k] = v * gain
    return out


# ---- 'diverse readout' perturbation of the trajectory goldens (see calibrate_diverse.py) ----
NODE_GAIN = (8.0, 8.0, 8.0, 8.0, 8.0, 24.0)     # rows of node_pred_mlp.4.weight: 5 atom-type logits, formal charge
EDGE_GAIN = (16.0, 64.0)                        # edge_exist_mlp.4.weight, edge_type_mlp.4.weight
NODE_BIAS0 = (0.06, -0.09, 0.02, 0.10, 0.28, -0.37)   # starting point of the per-case calibration
EDGE_BIAS0 = (-0.19, -0.71)
ALLSPECTRA_FULL_ATOMS = [5, 9, 12]              # molecules of the all-spectra 1000-step golden trajectory (G9)
MAX_SIZE_FULL_ATOMS = [29, 6]                   # G15: a 1000-step trajectory that contains a maximum-size (n = 29) molecule
DIVERSE_CASES = {                               # tag -> (spectra version, denoise steps, n_atoms)
    "allspectra_S5": ("allspectra", 5, RAGGED),
    "ir_S50": ("ir", 50, RAGGED),
    "ir_S1000": ("ir", 1000, FULL_LENGTH_ATOMS),
    "allspectra_S1000": ("allspectra", 1000, ALLSPECTRA_FULL_ATOMS),
}
# G15 re-uses the calibrated readout of ir_S1000 (diverse_readout.json holds a copy under its tag): with 812 directed edges and
# 29 atoms the outputs spread over their range without a calibration of their own (30+ minutes of CPU oracle per iteration).


def apply_readout(state_dict, node_bias, edge_bias):
    """Scale the last layer of the three readout MLPs by NODE_GAIN / EDGE_GAIN and replace its bias (works with or
    without the ``module.`` prefix; returns a new dict)."""
    out = dict(state_dict)
    for k, v in state_dict.items():
        if k.endswith("node_pred_mlp.4.weight"):
            out[k] = v * torch.tensor(NODE_GAIN, dtype=v.dtype, device=v.device).reshape(-1, 1)
        elif k.endswith("node_pred_mlp.4.bias"):
            out[k] = torch.tensor(node_bias, dtype=v.dtype, device=v.device)
        elif k.endswith("edge_exist_mlp.4.weight"):
            out[k] = v * EDGE_GAIN[0]
        elif k.endswith("edge_type_mlp.4.weight"):
            out[k] = v * EDGE_GAIN[1]
        elif k.endswith("edge_exist_mlp.4.bias"):
            out[k] = torch.tensor([edge_bias[0]], dtype=v.dtype, device=v.device)
        elif k.endswith("edge_type_mlp.4.bias"):
            out[k] = torch.tensor([edge_bias[1]], dtype=v.dtype, device=v.device)
    return out


def readout_diverse(state_dict, tag: str):
    """The calibrated perturbation of golden case ``tag`` (``diverse_readout.json``): atom types, charges and bond
    orders of the sampled molecules then cover their whole range, with decisions close to the thresholds."""
    import json
    with open(fixture_path("diverse_readout.json")) as f:
        rec = json.load(f)[tag]
    return apply_readout(state_dict, rec["node_bias"], rec["edge_bias"])


def pretrained_specformer_ckpt(encoder_state, variant: str):
    """A Lightning-style SpecFormer pre-training checkpoint for the key mapping of reference dmt.py:268-303.

    ``variant='spec_model'``: encoder under ``model.representation_spec_model.*`` (the first prefix the loader tries) while
    ``out_norm`` lives under ``model.representation_model.out_norm.*`` (the loader's special case); one entry has a
    mismatched shape and one is absent (both must keep their current values); unrelated keys are ignored.
    ``variant='plain_model'``: everything under ``model.representation_model.*``.
    Values: the procedural filler with salt 11, so they differ from the model's own salt-0 weights."""
    prefix = {"spec_model": "model.representation_spec_model", "plain_model": "model.representation_model"}[variant]
    sd = {}
    for k, v in encoder_state.items():
        val = filler.fill_tensor("cond_encoder." + k, v.shape, like=v, salt=11)
        if k in ("out_norm.weight", "out_norm.bias"):
            sd[f"model.representation_model.{k}"] = val
            if variant == "spec_model":     # a decoy under the tried prefix that the special case must NOT read
                sd[f"{prefix}.{k}"] = torch.full_like(val, 123.0)
            continue
        if variant == "spec_model" and k == "backbone.W_P.1.bias":
            continue                                                        # absent from the checkpoint
        if variant == "spec_model" and k == "backbone.W_pos_ir":
            val = torch.zeros(v.shape[0] + 1, v.shape[1])                   # wrong shape: skipped by the loader
        sd[f"{prefix}.{k}"] = val
    sd["model.some_other_head.weight"] = torch.ones(3, 3)
    sd["optimizer_like_entry"] = torch.zeros(1)
    return {"state_dict": sd, "epoch": 3}


def sampling_fn_case():
    """Inputs of the outer-loop golden (G11): 7-item dataset, rounds of 3, 6 molecules kept, 5 denoise steps."""
    count, batch_size, n_samples, steps = 7, 3, 6, 5
    n_atoms = [5, 9, 3, 12, 7, 4, 10]
    return dict(count=count, batch_size=batch_size, n_samples=n_samples, steps=steps, temperature=0.9, n_atoms=n_atoms,
                spectra=spectra_for("allspectra", count, salt=5))


def sampling_fn_noise_queue(case, perm):
    """Every randn draw of ``sampling_fn`` in call order (sampling.py:442-447 then :611-612,:623-624 per step), for the
    rounds the seed-42 permutation ``perm`` produces.  The generator replays it into the reference; the GPU test replays
    it into the HIP sampling function."""
    queue = []
    bs, rounds = case["batch_size"], -(-case["n_samples"] // case["batch_size"])
    for r in range(rounds):
        ids = perm[r * bs:(r + 1) * bs]
        N = max(case["n_atoms"][i] for i in ids)
        B = bs
        queue += [filler.normal(f"g11.r{r}.init.pos", (B, N, 3)), filler.normal(f"g11.r{r}.init.feat", (B, N, 6)),
                  filler.normal(f"g11.r{r}.init.edge", (B, 2, N, N))]
        for i in range(case["steps"]):
            queue += [filler.normal(f"g11.r{r}.s{i}.pos", (B, N, 3)), filler.normal(f"g11.r{r}.s{i}.feat", (B, N, 6)),
                      filler.normal(f"g11.r{r}.s{i}.edge", (B, 2, N, N))]
    return queue


TRAIN_ATOMS = [3, 7, 1, 12, 29, 9]             # G13 training batch: ragged, with a single-atom and a maximum-size molecule


def training_batch(version: str, n_atoms=TRAIN_ATOMS, salt: int = 0):
    """A collated training batch in the format of ``CollateSpectra.__call__`` (datasets/build_dataset.py:357-395): un-centred
    positions, one-hot atom types, formal charges, dense edge features [exist, bond order / 3] (``EdgeComSpectraTransform``
    :106-138, compress_edge), masks and spectra.  Procedural, reproducible on both sides."""
    B, N = len(n_atoms), max(n_atoms)
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    nm = node_mask.squeeze(-1)
    pos = (filler.normal("g13.pos", (B, N, 3), salt) * 1.3 + 0.4) * node_mask          # not zero-CoM (translation augmentation)
    types = (filler.uniform("g13.type", (B, N), salt=salt) * 5).long().clamp(0, 4)
    atom_one_hot = torch.nn.functional.one_hot(types, 5).float() * node_mask
    fc = ((filler.uniform("g13.fc", (B, N, 1), salt=salt) * 3).floor() - 1.0) * node_mask
    u = filler.uniform("g13.bond", (B, N, N), salt=salt)
    u = torch.triu(u, 1)
    u = u + u.transpose(1, 2)
    order = torch.zeros(B, N, N)
    order[u > 0.6], order[u > 0.8], order[u > 0.93] = 1.0, 2.0, 3.0
    em = edge_mask.reshape(B, N, N)
    order = order * em
    edge_one_hot = torch.stack([(order > 0).float(), order / 3.0], dim=-1)
    return dict(atom_one_hot=atom_one_hot, edge_one_hot=edge_one_hot, positions=pos, formal_charges=fc, atom_mask=nm,
                edge_mask=edge_mask, context=spectra_for(version, B, salt + 3), n_atoms=list(n_atoms))


def training_draws(n_atoms=TRAIN_ATOMS, salt: int = 0):
    """The random draws of one ``loss_fn`` call (losses.py:314-317), injected on both sides: ``torch.rand(B)`` for t, then the
    three ``randn`` tensors of the noise samplers (pos [B,N,3], feat [B,N,6], edge [B,2,N,N])."""
    B, N = len(n_atoms), max(n_atoms)
    return dict(t_raw=filler.uniform("g13.t", (B,), 0.02, 0.98, salt=salt),
                randn=[filler.normal("g13.n.pos", (B, N, 3), salt), filler.normal("g13.n.feat", (B, N, 6), salt),
                       filler.normal("g13.n.edge", (B, 2, N, N), salt)])


# parameters whose FULL gradient the training golden stores (one per kernel family); every other parameter is pinned by its
# gradient norm and by a 64-entry strided sample
TRAIN_DROPOUT_SEEDS = (1111, 2222)            # G17: Philox keys of the FF-dropout masks (self-conditioning forward, main forward)

TRAIN_FULL_GRADS = ("e_block_0.attn_mpnn.lin_edge0.weight", "e_block_7.equi_update.coord_mlp.0.weight", "node_emb.weight",
                    "e_block_3.node_time_mlp.1.bias", "e_block_0.dist_layer.means.weight", "e_block_5.dist_layer.stds.weight",
                    "e_block_2.equi_update.coord_norm.scale", "time_mlp.0.weights", "e_block_4.attn_mpnn.lin_query.weight",
                    "e_block_6.ff_linear4.weight", "edge_type_mlp.4.weight", "cond_lin.weight",
                    "cond_encoder.backbone.encoder.layers.1.norm_attn.1.weight", "cond_encoder.out_norm.bias")


def raw_molecules(n_atoms=(3, 7, 1, 12, 9), salt: int = 0):
    """Procedural QM9S-style raw items (what ``QM9SDataset.process`` stores per molecule, qm9s_dataset.py:245-268): atom types 0..4,
    directed edge list (both directions, sorted as PyG does) with bond types 1..4 (4 = aromatic), formal charges, positions, raw spectra."""
    mols = []
    for m, n in enumerate(n_atoms):
        at = (filler.uniform(f"g16.at{m}", (n,), salt=salt) * 5).long().clamp(0, 4)
        u = filler.uniform(f"g16.b{m}", (n, n), salt=salt)
        rows, cols, types = [], [], []
        for i in range(n):
            for j in range(i + 1, n):
                if float(u[i, j]) > 0.55:
                    t = 1 + int(float(u[j, i]) * 4)                   # bond type 1..4
                    rows += [i, j]; cols += [j, i]; types += [t, t]
        ei = torch.tensor([rows, cols], dtype=torch.long).reshape(2, -1)
        et = torch.tensor(types, dtype=torch.long)
        if et.numel():
            perm = (ei[0] * n + ei[1]).argsort()                      # qm9s_dataset.py:259-260
            ei, et = ei[:, perm], et[perm]
        fc = ((filler.uniform(f"g16.fc{m}", (n,), salt=salt) * 3).floor() - 1).long()
        spec = {nm: filler.uniform(f"g16.{nm}{m}", (1, L), 0.0, 5.0, salt=salt) for nm, L in zip(("uv", "ir", "raman"), SPECTRUM_LENGTHS)}
        mols.append(dict(atom_type=at, edge_index=ei, edge_type=et, fc=fc, pos=filler.normal(f"g16.pos{m}", (n, 3), salt) * 1.4, num_atom=n, **spec))
    return mols


def bond_distance_sweep():
    """Distances (Angstrom) for the bond-order golden (G12): a 1 pm grid over 0.5-2.0 A plus points 1e-4 A either side of
    every integer-picometre threshold in that range, so each ``<`` comparison is exercised on both sides."""
    grid = np.arange(50, 201, dtype=np.float64) / 100.0
    near = np.concatenate([grid - 1e-4, grid + 1e-4])
    return np.sort(np.concatenate([grid, near]))


def config_for(version: str, steps: int = 1000):
    return qm9s_config(spectra_version=version, steps=steps)


def save_npz(name: str, **arrays):
    np.savez_compressed(fixture_path(name), **{k: np.asarray(v) for k, v in arrays.items()})


def load_npz(name: str):
    with np.load(fixture_path(name)) as z:
        return {k: (str(z[k]) if z[k].dtype.kind in "US" else torch.from_numpy(z[k])) for k in z.files}
