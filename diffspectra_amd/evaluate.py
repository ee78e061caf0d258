"""Evaluation driver + checkpoint contract ("next" row N2 of SURVEY §8f).

Mirror of the sampling half of reference ``run_lib.diffspectra_evaluate`` (``run_lib.py:297-370``) and of the
checkpoint helpers (``utils.py:7-30``, ``models/ema.py``): build the model through the factory, restore
``checkpoints/checkpoint_{k}.pth`` (``{'optimizer', 'model', 'ema', 'step'}``, ``model`` keys ``module.``-prefixed,
``ema['shadow_params']`` a list in ``parameters()`` order), copy the EMA weights into the model and run the
conditional sampling function.  The reference's RDKit / MOSES / FCD metric stack (``run_lib.py:371-441``) is host-side
analytics outside this path; callers plug metrics in as callbacks that receive ``processed_mols`` in the reference's
tuple format.  (The reference's own ``run_lib`` cannot be imported as shipped: ``import visualize`` has no module.)
"""
from __future__ import annotations

import logging
import os
from typing import Callable, Dict, Optional

import torch

from .ema import ExponentialMovingAverage  # noqa: F401  (re-exported: the checkpoint contract's EMA holder)
from .noise_schedule import NoiseScheduleVP
from .registry import create_model
from .sampling import get_cond_sampling_eval_fn
from .scalers import get_data_inverse_scaler


def restore_checkpoint(ckpt_path: str, state: Dict, device) -> Dict:
    """utils.py:7-20: strict model load, EMA + step restored; the optimizer entry is optional on the inference path."""
    if not os.path.exists(ckpt_path):
        os.makedirs(os.path.dirname(ckpt_path) or ".", exist_ok=True)
        logging.warning("No checkpoint found at %s. Returned the same state as input", ckpt_path)
        return state
    loaded = torch.load(ckpt_path, map_location=device)
    if state.get("optimizer") is not None and "optimizer" in loaded:
        state["optimizer"].load_state_dict(loaded["optimizer"])
    state["model"].load_state_dict(loaded["model"], strict=True)
    state["ema"].load_state_dict(loaded["ema"])
    state["step"] = loaded["step"]
    return state


def save_checkpoint(ckpt_path: str, state: Dict) -> None:
    """utils.py:23-30.  In a multi-rank job EVERY rank calls this (the sharded optimizer / EMA state is gathered by collectives
    inside ``state_dict()``); rank 0 writes the file."""
    saved = {}
    if state.get("optimizer") is not None:            # same key order as the reference's file
        saved["optimizer"] = state["optimizer"].state_dict()
    saved.update(model=state["model"].state_dict(), ema=state["ema"].state_dict(), step=state["step"])
    from .shard import world_info
    if world_info()[0] == 0:
        torch.save(saved, ckpt_path)


def checkpoint_ids(config):
    """run_lib.py:326-331: explicit comma list or the inclusive begin..end range."""
    ckpts = getattr(config.eval, "ckpts", "")
    if ckpts != "":
        return [int(c) for c in str(ckpts).split(",")]
    return list(range(config.eval.begin_ckpt, config.eval.end_ckpt + 1))


def diffspectra_evaluate(config, workdir: str, test_ds=None, eval_folder: str = "eval",
                         metric_fns: Optional[Dict[str, Callable]] = None):
    """Sampling evaluation over the configured checkpoints; returns ``{ckpt: {'processed_mols', 'gt_pos', 'gt_rdmols',
    'metrics'}}``.  ``metric_fns[name](processed_mols, gt_pos, gt_rdmols)`` are optional host-side callbacks.

    ``test_ds=None`` reads the reference's processed files under ``config.data.root`` (``run_lib.py:313`` ->
    ``build_dataset.py:31-42``: the 'test' entry of ``split_dict_diffspectra_qm9.pt``) into the device-resident table of
    ``qm9s_reader.ProcessedQM9S.packed_table`` - no PyG, no per-molecule Python."""
    os.makedirs(os.path.join(workdir, eval_folder), exist_ok=True)
    if test_ds is None:
        from .qm9s_reader import ProcessedQM9S
        test_ds = ProcessedQM9S(config.data.root).packed_table(
            config.data.spectra_version, split="test", device=config.device, normalize=getattr(config.data, "use_normalize", True))
    model = create_model(config)
    ema = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_decay)
    state = dict(optimizer=None, model=model, ema=ema, step=0)
    logging.info("model size: %.1fMB", sum(p.numel() for p in model.parameters()) * 4 / 2 ** 20)
    noise_scheduler = NoiseScheduleVP(config.sde.schedule, continuous_beta_0=config.sde.continuous_beta_0,
                                      continuous_beta_1=config.sde.continuous_beta_1)
    inverse_scaler = get_data_inverse_scaler(config)
    sampling_fn = get_cond_sampling_eval_fn(config, noise_scheduler, config.eval.batch_size, config.eval.num_samples,
                                            inverse_scaler, test_ds)
    results = {}
    for ckpt in checkpoint_ids(config):
        ckpt_path = os.path.join(workdir, "checkpoints", "checkpoint_{}.pth".format(ckpt))
        if not os.path.exists(ckpt_path):
            raise FileNotFoundError("Checkpoint path error: " + ckpt_path)
        logging.info("load checkpoint: %s", ckpt_path)
        state = restore_checkpoint(ckpt_path, state, device=config.device)
        ema.copy_to(model.parameters())          # eval uses EMA weights; BatchNorm buffers stay the model's (run_lib.py:361-362)
        processed_mols, gt_pos, gt_rdmols = sampling_fn(model)
        metrics = {name: fn(processed_mols, gt_pos, gt_rdmols) for name, fn in (metric_fns or {}).items()}
        results[ckpt] = dict(processed_mols=processed_mols, gt_pos=gt_pos, gt_rdmols=gt_rdmols, metrics=metrics, step=state["step"])
    return results
