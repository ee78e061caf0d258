"""CPU: the oracle restatement against the golden vectors produced by the reference's own source
(tests/golden/generate_golden.py).  These pin the oracle (prompt ③ / SURVEY §8c G0-G6)."""
import json

import pytest
import torch

import oracle
from oracle import dmt as odmt
from tests.golden import cases
from tests.helpers import procedural_state_dict, max_abs_diff

# Tolerances (fp32): the reference's own batch-composition noise floor is 5e-7 per forward and
# <= 3.2e-5 over a 50-step trajectory.
TOL_KERNEL = 1e-5
TOL_FORWARD = 2e-5
TOL_TRAJ = 5e-4


@pytest.mark.parametrize("version", ["ir", "allspectra"])
def test_g0_state_dict_manifest(version):
    cfg, sd = procedural_state_dict(version)
    with open(cases.fixture_path(f"state_dict_manifest_{version}.json")) as f:
        man = json.load(f)
    ref = [(k[len("module."):], tuple(s), d) for k, s, d in man["entries"]]
    mine = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    assert mine == ref
    assert len(mine) == (435 if version == "allspectra" else 429)


def test_g1_schedule_coefficients():
    g = cases.load_npz("g1_schedule.npz")
    for S in (50, 1000):
        co = oracle.ancestral_coefficients(S)
        for k in ("t", "s", "alpha_t", "sigma_t", "alpha_s", "sigma_s", "c_x", "c_pred", "noise_level"):
            assert torch.equal(co[k], g[f"S{S}_{k}"]), (S, k)      # same torch fp32 ops → bit-exact


@pytest.mark.parametrize("version", ["ir", "allspectra"])
def test_g2_specformer(version):
    cfg, sd = procedural_state_dict(version)
    g = cases.load_npz("g2_specformer.npz")
    ctx = cases.spectra_for(version, 4)
    z = oracle.specformer_forward(sd, ctx, version, cfg.model.patch_len, cfg.model.stride)
    assert max_abs_diff(z, g[f"{version}_z"]) <= TOL_KERNEL
    assert max_abs_diff(oracle.context_embedding(sd, ctx, cfg), g[f"{version}_ctx"]) <= TOL_KERNEL


def test_g3_components():
    cfg, sd = procedural_state_dict("ir")
    g = cases.load_npz("g3_components.npz")
    inp = cases.block_inputs()
    row, col = inp["edge_index"]
    d2 = ((inp["pos"][row] - inp["pos"][col]) ** 2).sum(1, keepdim=True)
    assert max_abs_diff(d2, g["d2"]) == 0.0
    dist = odmt._cond_gaussian(sd, "e_block_0.dist_layer", d2, inp["edge_time_emb"])
    assert max_abs_diff(dist, g["cond_gaussian"]) <= TOL_KERNEL
    tm = odmt._trans_mix(sd, "e_block_0.attn_mpnn", inp["h"], inp["edge_index"], inp["edge_attr"], inp["extra_heads"])
    assert max_abs_diff(tm, g["trans_mix"]) <= TOL_KERNEL
    eq = odmt._equi_update(sd, "e_block_0.equi_update", inp["h"], inp["pos"], inp["edge_index"], inp["edge_attr"],
                           dist, inp["edge_time_emb"], inp["extra_heads"])
    assert max_abs_diff(eq, g["equi_update"]) <= TOL_KERNEL
    h, e, pos = odmt._mix_block(sd, "e_block_0", inp["pos"], inp["h"], inp["edge_attr"], inp["edge_index"],
                                inp["node_mask"], inp["extra_heads"], inp["node_time_emb"], inp["edge_time_emb"])
    assert max_abs_diff(h, g["block_h"]) <= TOL_KERNEL
    assert max_abs_diff(e, g["block_e"]) <= TOL_KERNEL
    assert max_abs_diff(pos, g["block_pos"]) <= TOL_KERNEL
    nl = torch.tensor([-7.5, -1.0, 0.0, 0.3, 9.0])
    assert max_abs_diff(odmt.time_embedding(sd, nl), g["time_mlp"]) <= TOL_KERNEL


@pytest.mark.parametrize("version", ["ir", "allspectra"])
@pytest.mark.parametrize("first", [True, False])
def test_g4_forward(version, first):
    cfg, sd = procedural_state_dict(version)
    g = cases.load_npz("g4_forward.npz")
    a = cases.forward_inputs(version, first)
    xh, ef = oracle.dmt_forward(sd, cfg, a["xh"], a["node_mask"], a["edge_mask"], a["edge_x"], a["noise_level"],
                                a["cond_x"], a["cond_edge_x"], context=a["context"])
    tag = f"{version}_{'first' if first else 'general'}"
    assert max_abs_diff(xh, g[tag + "_xh"]) <= TOL_FORWARD
    assert max_abs_diff(ef, g[tag + "_edge"]) <= TOL_FORWARD
    # invariants the reference guarantees: masked rows zero, zero CoM, symmetric edges
    assert float((xh * (1 - a["node_mask"])).abs().max()) == 0.0
    assert float(xh[:, :, :3].sum(1).abs().max()) < 1e-5
    assert torch.equal(ef, ef.transpose(1, 2))


@pytest.mark.parametrize("version,steps", [("allspectra", 5), ("ir", 50)])
def test_g5_trajectory(version, steps):
    cfg, sd = procedural_state_dict(version)
    g = cases.load_npz("g5_trajectory.npz")
    tr = cases.trajectory_inputs(version, steps)
    ctx = oracle.context_embedding(sd, tr["context"], cfg)       # loop-invariant

    def model_fn(x, edge_x, noise_level, cond_x, cond_edge_x):
        return oracle.dmt_forward(sd, cfg, x, tr["node_mask"], tr["edge_mask"], edge_x, noise_level, cond_x,
                                  cond_edge_x, context_emb=ctx)

    z = oracle.combined_noise(tr["raw0"][0], tr["raw0"][1], tr["node_mask"])
    ez = oracle.symmetric_edge_noise(tr["raw0"][2], tr["edge_mask"])
    x_mean, e_mean = oracle.ancestral_sampling(model_fn, z, tr["node_mask"], tr["edge_mask"], ez, steps,
                                               lambda i: tr["raws"][i])
    tag = f"{version}_S{steps}"
    assert max_abs_diff(x_mean, g[tag + "_x_mean"]) <= TOL_TRAJ
    assert max_abs_diff(e_mean, g[tag + "_edge_mean"]) <= TOL_TRAJ
    pos, one_hot, fc, et = oracle.post_process(x_mean, tr["node_mask"], e_mean, tr["edge_mask"])
    assert torch.equal(one_hot.argmax(-1), g[tag + "_atom_type"])            # bit-exact integer outputs
    assert torch.equal(fc.to(g[tag + "_fc"].dtype), g[tag + "_fc"])
    assert torch.equal(et, g[tag + "_edge_type"])
    mols = oracle.mol_process(one_hot, pos, fc, tr["n_atoms"], et)
    for m, (p, at, e, c) in enumerate(mols):
        assert max_abs_diff(p, g[f"{tag}_mol{m}_pos"]) <= TOL_TRAJ
        assert torch.equal(at, g[f"{tag}_mol{m}_atom"]) and torch.equal(e, g[f"{tag}_mol{m}_edge"])
        assert torch.equal(c, g[f"{tag}_mol{m}_fc"])


def test_g8_clamp_self_cond():
    """self_cond_type='clamp' (utils.py:137-148): oracle vs the reference trajectory, and the clamp must matter."""
    version, steps = "ir", 8
    cfg, sd = procedural_state_dict(version)
    sd = cases.readout_gain(sd)
    g = cases.load_npz("g8_trajectory_clamp.npz")
    tr = cases.trajectory_inputs(version, steps)
    ctx = oracle.context_embedding(sd, tr["context"], cfg)

    def model_fn(x, edge_x, noise_level, cond_x, cond_edge_x):
        return oracle.dmt_forward(sd, cfg, x, tr["node_mask"], tr["edge_mask"], edge_x, noise_level, cond_x,
                                  cond_edge_x, context_emb=ctx)

    z = oracle.combined_noise(tr["raw0"][0], tr["raw0"][1], tr["node_mask"])
    ez = oracle.symmetric_edge_noise(tr["raw0"][2], tr["edge_mask"])
    run = lambda fn: oracle.ancestral_sampling(model_fn, z, tr["node_mask"], tr["edge_mask"], ez, steps,
                                               lambda i: tr["raws"][i], cond_process_fn=fn)
    x_mean, e_mean = run(oracle.self_cond_clamp)
    tag = f"{version}_S{steps}"
    assert max_abs_diff(x_mean, g[tag + "_x_mean"]) <= TOL_TRAJ
    assert max_abs_diff(e_mean, g[tag + "_edge_mean"]) <= TOL_TRAJ
    x_ori, _ = run(oracle.self_cond_ori)
    assert max_abs_diff(x_ori, g[tag + "_x_mean"]) > 100 * TOL_TRAJ      # the fixture really exercises the clamp
    _, one_hot, fc, et = oracle.post_process(x_mean, tr["node_mask"], e_mean, tr["edge_mask"])
    assert torch.equal(one_hot.argmax(-1), g[tag + "_atom_type"])
    assert torch.equal(fc.to(g[tag + "_fc"].dtype), g[tag + "_fc"])
    assert torch.equal(et, g[tag + "_edge_type"])


def test_g6_post_process_thresholds():
    g = cases.load_npz("g6_post_process.npz")
    from diffspectra_amd import filler
    node_mask, edge_mask = filler.masks_from_n_atoms([2, 5, 7])
    pos, one_hot, fc, et = oracle.post_process(g["xh"], node_mask, g["edge_x"], edge_mask)
    assert torch.equal(pos, g["pos"])
    assert torch.equal(one_hot.to(g["one_hot"].dtype), g["one_hot"])
    assert torch.equal(fc.to(g["fc"].dtype), g["fc"])
    assert torch.equal(et, g["edge_type"])


def test_g7_full_length_trajectory():
    """1000 ancestral steps (the metric's own length) with injected noise: oracle vs the reference's own run."""
    cfg, sd = procedural_state_dict("ir")
    g = cases.load_npz("g7_trajectory_1000.npz")
    tr = cases.trajectory_inputs("ir", 1000, cases.FULL_LENGTH_ATOMS)
    ctx = oracle.context_embedding(sd, tr["context"], cfg)

    def model_fn(x, edge_x, noise_level, cond_x, cond_edge_x):
        return oracle.dmt_forward(sd, cfg, x, tr["node_mask"], tr["edge_mask"], edge_x, noise_level, cond_x, cond_edge_x,
                                  context_emb=ctx)

    z = oracle.combined_noise(tr["raw0"][0], tr["raw0"][1], tr["node_mask"])
    ez = oracle.symmetric_edge_noise(tr["raw0"][2], tr["edge_mask"])
    x_mean, e_mean = oracle.ancestral_sampling(model_fn, z, tr["node_mask"], tr["edge_mask"], ez, 1000, lambda i: tr["raws"][i])
    tag = "ir_S1000"
    assert max_abs_diff(x_mean, g[tag + "_x_mean"]) <= TOL_TRAJ
    assert max_abs_diff(e_mean, g[tag + "_edge_mean"]) <= TOL_TRAJ
    pos, one_hot, fc, et = oracle.post_process(x_mean, tr["node_mask"], e_mean, tr["edge_mask"])
    assert torch.equal(one_hot.argmax(-1), g[tag + "_atom_type"])
    assert torch.equal(fc.to(g[tag + "_fc"].dtype), g[tag + "_fc"])
    assert torch.equal(et, g[tag + "_edge_type"])
