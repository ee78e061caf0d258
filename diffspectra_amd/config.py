"""Config object for the DMT + SpecFormer sampling path.

The reference keeps its hyper-parameters in an ``ml_collections.ConfigDict``
(reference ``configs/diffspectra_qm9s.py:9-153``).  ``ml_collections`` is not
part of this image and the loader itself is outside the hot path, so this
module provides a plain attribute tree with the *same field names* the model
factory, the sampler factory and the scalers read
(``models/dmt.py:185-207,260-265``, ``sampling.py:353-376``, ``utils.py:71-150``).
Any object exposing these attributes (including a real ConfigDict) works.
"""
from __future__ import annotations

import copy
from types import SimpleNamespace


class Config(SimpleNamespace):
    """Attribute tree; ``cfg.a.b`` access and ``hasattr`` behave like ConfigDict."""

    def clone(self) -> "Config":
        return copy.deepcopy(self)


def qm9s_config(spectra_version: str = "allspectra", device="cpu", steps: int = 1000,
                batch_size: int = 128, num_samples: int = 10000) -> Config:
    """Field values of reference ``configs/diffspectra_qm9s.py`` (QM9S, DMT)."""
    data = Config(
        name="QM9S", info_name="qm9_second_half", compress_edge=True, centered=True,
        include_aromatic=False, atom_types=5, bond_types=4, fc_scale=[-1.0, 1.0],
        max_node=29, spectra_version=spectra_version,
        root="/path/to/dataset/QM9S", use_normalize=True,   # configs/diffspectra_qm9s.py:17,35
    )
    model = Config(
        name="DMT", pred_data=True, include_fc_charge=True, normalize_factors="1, 4, 4, 1",
        ema_decay=0.999, edge_ch=2, nf=256, n_layers=8, n_heads=16, dropout=0.1,
        cond_time=True, dist_gbf=True, gbf_name="CondGaussianLayer", self_cond=True,
        self_cond_type="ori", edge_quan_th=0.0, n_extra_heads=2, CoM=True, mlp_ratio=2,
        spatial_cut_off=2.0, softmax_inf=True, trans_name="TransMixLayer", cond_ch=1,
        pretrained_specformer_path="", patch_len=[20, 50, 50], stride=[10, 25, 25],
        loss_weights="1., 0.25, 0.1", noise_align=True,                        # configs/diffspectra_qm9s.py:79-80
    )
    sde = Config(schedule="cosine", continuous_beta_0=0.1, continuous_beta_1=20.0)
    sampling = Config(method="ancestral", steps=steps, noise_source="philox", seed=42)
    evaluate = Config(batch_size=batch_size, num_samples=num_samples, sampling_temperature=1.0, enable_sampling=True,
                      begin_ckpt=40, end_ckpt=40, ckpts="")
    # training side (configs/diffspectra_qm9s.py:85-127): batch 128 per GPU, AdamW-amsgrad lr 2e-4, warm-up 100 000 steps,
    # adaptive gradient clipping capped at 10
    # precision: 'fp32' (the reference's arithmetic, what golden G13 pins) or 'bf16' (BASELINE config 5: GEMM operands rounded to bf16,
    # fp32 accumulation and fp32 master weights - a build-side option, the reference has no AMP)
    training = Config(batch_size=128, reduce_mean=False, n_iters=2000000, snapshot_freq=50000, num_gpus=1, precision="fp32")
    optim = Config(weight_decay=0, optimizer="AdamW", lr=2e-4, beta1=0.9, eps=1e-8, warmup=100000, grad_clip=10.0,
                   disable_grad_log=True)
    return Config(
        exp_type="diffspectra", pred_edge=True, only_2D=False, data=data, model=model, sde=sde,
        sampling=sampling, eval=evaluate, training=training, optim=optim, seed=42, device=device,
    )


# n_atoms histogram of the QM9S split the reference samples from
# (reference ``datasets/datasets_config.py:23-25``, ``qm9_second_half['train_n_nodes']``).
QM9_SECOND_HALF_N_NODES = {
    3: 1, 4: 3, 5: 3, 6: 5, 7: 7, 8: 25, 9: 62, 10: 178, 11: 412, 12: 845, 13: 1541, 14: 2587,
    15: 3865, 16: 5344, 17: 6461, 18: 6695, 19: 6944, 20: 4794, 21: 4962, 22: 1701, 23: 2380,
    24: 267, 25: 754, 26: 17, 27: 132, 29: 15,
}

SPECTRUM_LENGTHS = (701, 3501, 3501)  # uv, ir, raman (reference models/specformer.py:33)


def used_spectra(spectra_version: str):
    """Indices into (uv, ir, raman) used by a spectra_version (reference models/specformer.py:35-46)."""
    table = {"uv": [0], "ir": [1], "raman": [2], "allspectra": [0, 1, 2]}
    if spectra_version not in table:
        raise ValueError("spectra_version should be uv, ir, raman or allspectra")
    return table[spectra_version]
