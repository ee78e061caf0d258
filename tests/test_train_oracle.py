"""Row N1 on the CPU: the training-loss oracle (oracle/train.py, torch autograd over the functional forward) against golden G13 -
the REFERENCE's own ``get_sde_graph_loss_fn`` loss and gradients with every random draw injected (losses.py:286-396)."""
import json

import numpy as np
import pytest
import torch

from oracle import train as otrain
from tests.golden import cases
from tests.helpers import procedural_state_dict

CASES = [("ir", "selfcond", "g13"), ("ir", "plain", "g13"), ("allspectra", "selfcond", "g13"),
         ("ir", "selfcond", "g17"), ("allspectra", "plain", "g17")]      # G17: dropout 0.1 with injected masks (config 5 as shipped)
FIXTURES = {"g13": "g13_training.npz", "g17": "g17_training_dropout.npz"}


def grad_sample_index(numel: int, count: int = 64):
    return torch.linspace(0, numel - 1, min(count, numel)).round().long()


def oracle_loss_and_grads(version, coin_name, dropout=None):
    cfg, sd0 = procedural_state_dict(version)
    cfg = cfg.clone()
    sd = {}
    for k, v in sd0.items():
        v = v.clone()
        if v.is_floating_point() and not any(s in k for s in ("running_mean", "running_var", "sdp_attn.scale")):
            v.requires_grad_(True)
        sd[k] = v
    batch = cases.training_batch(version)
    draws = cases.training_draws()
    loss, info = otrain.training_loss(sd, cfg, batch, draws["t_raw"], draws["randn"], coin_name == "selfcond", dropout=dropout)
    loss.backward()
    return sd, loss.detach(), info


@pytest.mark.parametrize("version,coin_name,fixture", CASES)
def test_training_oracle_matches_reference_loss_and_grads(version, coin_name, fixture):
    g = cases.load_npz(FIXTURES[fixture])
    tag = f"{version}_{coin_name}"
    sd, loss, info = oracle_loss_and_grads(version, coin_name, (0.1, cases.TRAIN_DROPOUT_SEEDS) if fixture == "g17" else None)
    for k in ("xh", "edge_x", "z_t", "edge_z_t", "alpha_t", "sigma_t", "noise_level", "align_pos", "pred", "edge_pred"):
        assert torch.allclose(info[k].detach(), g[f"{tag}_{k}"], rtol=1e-5, atol=2e-6), k
    if coin_name == "selfcond":
        assert torch.allclose(info["cond_x"], g[tag + "_cond_x"], rtol=1e-5, atol=2e-6)
    assert abs(float(loss) - float(g[tag + "_loss"])) <= 1e-5 * abs(float(g[tag + "_loss"]))
    names = json.loads(g[tag + "_grad_names"])
    total = float(np.sqrt(np.sum(np.square(g[tag + "_grad_norms"].numpy()))))
    floor = 1e-7 * total           # gradients that are sums of cancelling terms (e.g. a bias in front of a BatchNorm) carry fp32 noise of this size
    worst = 0.0
    for i, n in enumerate(names):
        ref_norm = float(g[tag + "_grad_norms"][i])
        grad = sd[n].grad
        if grad is None:
            assert ref_norm == 0.0, n
            continue
        assert abs(float(grad.double().norm()) - ref_norm) <= 1e-4 * ref_norm + floor, (n, float(grad.double().norm()), ref_norm)
        idx = grad_sample_index(grad.numel())
        want = g[tag + "_grad_samples"][i][:len(idx)]
        scale = ref_norm / max(1.0, grad.numel() ** 0.5)
        err = float((grad.reshape(-1)[idx] - want).abs().max())
        assert err <= 1e-4 * float(want.abs().max()) + 1e-3 * scale + floor, (n, err)
        worst = max(worst, err / (float(want.abs().max()) + 1e-12))
        if n in cases.TRAIN_FULL_GRADS:
            full = g[f"{tag}_grad::{n}"]
            assert torch.allclose(grad, full, rtol=1e-4, atol=1e-4 * float(full.abs().max()) + floor), n
    bn = "cond_encoder.backbone.encoder.layers.0.norm_attn.1."
    assert torch.allclose(info["bn"][bn + "running_mean"], g[tag + "_bn_running_mean"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(info["bn"][bn + "running_var"], g[tag + "_bn_running_var"], rtol=1e-5, atol=1e-6)
    print(f"[{fixture.upper()} {tag}] loss {float(loss):.6f}; worst sampled gradient deviation {worst:.2e}")


def test_kabsch_alignment_golden():
    """get_align_position / kabsch_batch (losses.py:414-452) on the ragged G13 batch, incl. a single-atom molecule."""
    g = cases.load_npz("g13_training.npz")
    tag = "ir_selfcond"
    rot = otrain.kabsch_batch(g[tag + "_z_t"][:, :, :3], g[tag + "_xh"][:, :, :3])
    n_atoms = cases.TRAIN_ATOMS
    for b, n in enumerate(n_atoms):
        if n >= 3:                                     # with fewer atoms the rotation is not unique (rank-deficient covariance)
            assert torch.allclose(rot[b], g[tag + "_rotations"][b], atol=1e-4), b
    assert torch.allclose(otrain.align_position(g[tag + "_z_t"], g[tag + "_xh"]), g[tag + "_align_pos"], atol=1e-5)


def test_dropout_masks_are_pair_symmetric_and_hit_the_keep_rate():
    """The injected FF-dropout masks (oracle.train.dropout_masks): both directions of a pair carry the pair's mask, padded atoms are
    kept, the keep rate is 1 - p, kept elements are scaled by 1 / (1 - p)."""
    n_atoms = [3, 5, 1, 4]
    fn = otrain.dropout_masks(n_atoms, 5, 0.1, 1234)
    E = sum(n * (n - 1) for n in n_atoms)
    m = fn(2, 3, torch.ones(E, 64))
    off = 0
    for n in n_atoms:
        blk = m[off:off + n * (n - 1)]
        ii, jj = np.nonzero(~np.eye(n, dtype=bool))
        dense = torch.zeros(n, n, 64)
        dense[ii, jj] = blk
        assert torch.equal(dense, dense.transpose(0, 1))
        off += n * (n - 1)
    node = fn(0, 0, torch.ones(4 * 5, 512)).reshape(4, 5, 512)
    assert float(node[2, 1:].min()) > 1.0                               # padded atoms of the single-atom molecule: kept
    big = otrain.dropout_masks([29] * 8, 29, 0.1, 99)(1, 2, torch.ones(8 * 29 * 28, 128))
    keep = float((big != 0).float().mean())
    assert abs(keep - 0.9) < 5e-3 and abs(float(big.max()) - 1.0 / 0.9) < 1e-6
