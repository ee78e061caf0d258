"""The RCCL call sites of the path, executed on ONE GPU.

Every collective of this code base sits behind ``world > 1`` - and a GPU box has one card, so until the driver's 8-GPU run none of
``reduce_scatter_tensor`` / ``all_gather_into_tensor`` / device ``broadcast`` had ever executed.  Here a child process initialises a
world-size-1 ``nccl`` (= RCCL) process group and forces the sharded branches (``shard.force_collectives``, ``FusedAdamW(force_sharded=True)``):
with one rank every collective is an identity, so results must equal the unsharded ones BIT FOR BIT, while dtype / contiguity / aliasing
mistakes in the calls surface as RCCL errors.  A child process, so that the process group does not outlive the test."""
from __future__ import annotations

import os
import socket

import pytest
import torch

from tests.golden import cases

pytestmark = pytest.mark.gpu


def _worker(port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    d = torch.device("cuda:0")
    torch.cuda.set_device(d)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    res = {}
    try:
        from diffspectra_amd import shard, losses as Lh, filler, sampling as S
        from diffspectra_amd.ema import ExponentialMovingAverage
        from diffspectra_amd.evaluate import save_checkpoint, restore_checkpoint
        from diffspectra_amd.noise_schedule import NoiseScheduleVP
        from diffspectra_amd.scalers import get_data_inverse_scaler
        from diffspectra_amd.dataset_pack import PackedSpectraTable
        from tests.test_train_hip import _train_model
        assert dist.get_backend() == "nccl"
        # ---- shard.py: broadcast / all_gather_counts / gather_records with and without the shortcut
        perm = torch.randperm(37)
        rec = torch.randint(0, 255, (11, shard.RECORD_BYTES), dtype=torch.uint8, device=d)
        plain = (shard.broadcast_from_rank0(perm, d), shard.all_gather_counts(torch.tensor([11]), d), shard.gather_records(rec))
        assert not shard.collectives_on()
        shard.force_collectives(True)
        assert shard.collectives_on()
        forced = (shard.broadcast_from_rank0(perm, d), shard.all_gather_counts(torch.tensor([11]), d), shard.gather_records(rec, [11]))
        res["shard_equal"] = bool(torch.equal(plain[0], forced[0]) and plain[1] == forced[1] and torch.equal(plain[2], forced[2])
                                  and forced[2].data_ptr() != rec.data_ptr())
        # ---- the sharded sampling function (final all-gather of records through RCCL)
        cfg, model = _train_model("ir", d)
        cfg.sampling.steps = 3
        cfg.sampling.seed = 42
        model.eval()
        n_atoms = [5, 9, 3, 7, 4, 6]
        spec = torch.log10(1.0 + filler.uniform("rccl.ir", (len(n_atoms), 1, 3501))).to(d)
        table = PackedSpectraTable([None, spec, None], torch.tensor(n_atoms), device=d)
        ns = NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0)
        fn = S.get_cond_sampling_eval_fn(cfg, ns, 4, len(n_atoms), get_data_inverse_scaler(cfg), table)
        mols_forced = fn(model)[0]
        shard.force_collectives(False)
        mols_plain = fn(model)[0]
        res["sampling_equal"] = all(all(torch.equal(a, b) for a, b in zip(x, y)) for x, y in zip(mols_forced, mols_plain))
        # ---- FusedAdamW: reduce_scatter_tensor -> shard update -> all_gather_into_tensor (incl. the Ps.clone() aliasing case) vs the plain step
        model.train()
        runs = {}
        for mode in ("plain", "sharded"):
            cfg_m, m = _train_model("ir", d)
            cfg_m.optim.warmup = 0
            ema = ExponentialMovingAverage(m.parameters(), decay=0.999)
            opt = Lh.FusedAdamW(m.parameters(), lr=cfg_m.optim.lr, force_sharded=(mode == "sharded"))
            assert opt.sharded == (mode == "sharded") and opt.shard == opt.n_pad
            step_fn = Lh.get_step_fn(ns, True, Lh.optimization_manager(cfg_m), None, cfg_m)
            state = dict(optimizer=opt, model=m, ema=ema, step=0)
            batch = {k: v for k, v in cases.training_batch("ir").items() if k != "n_atoms"}
            import random
            losses = []
            for it in range(2):
                torch.manual_seed(100 + it)
                random.seed(7 + it)
                losses.append(float(step_fn(state, batch).detach()))
            # checkpoint round trip in a sharded job: every rank calls save; then restore into a fresh state and take one more step
            ck = out_path + f".{mode}.ckpt"
            save_checkpoint(ck, state)
            cfg_r, m_r = _train_model("ir", d)
            cfg_r.optim.warmup = 0
            ema_r = ExponentialMovingAverage(m_r.parameters(), decay=0.999)
            opt_r = Lh.FusedAdamW(m_r.parameters(), lr=cfg_r.optim.lr, force_sharded=(mode == "sharded"))
            step_r = Lh.get_step_fn(ns, True, Lh.optimization_manager(cfg_r), None, cfg_r)
            state_r = dict(optimizer=opt_r, model=m_r, ema=ema_r, step=0)
            # one step BEFORE the restore attaches the EMA to the flat buffer: load_state_dict must then write through the views
            torch.manual_seed(5)
            random.seed(5)
            step_r(state_r, batch)
            state_r = restore_checkpoint(ck, state_r, d)
            for st_ in (state, state_r):
                torch.manual_seed(300)
                random.seed(11)
                st_["loss3"] = float(step_fn(st_, batch).detach()) if st_ is state else float(step_r(st_, batch).detach())
            res[f"{mode}_restore_equal"] = bool(torch.equal(opt.P, opt_r.P) and torch.equal(opt.M, opt_r.M) and torch.equal(opt.Vmax, opt_r.Vmax)
                                                and all(torch.equal(a, b) for a, b in zip(ema.shadow_params, ema_r.shadow_params))
                                                and ema_r.shadow_params[0].data_ptr() == opt_r.ema_flat.data_ptr())
            p_after = opt.P.detach().cpu().clone()
            ema.copy_to(m.parameters())                       # sharded: gather_ema's all_gather_into_tensor
            assert torch.equal(opt.unpadded(opt.P), opt.unpadded(opt.ema_flat))
            runs[mode] = dict(P=p_after, ema=opt.ema_flat.detach().cpu().clone(), M=opt.M.detach().cpu().clone(),
                              Vmax=opt.Vmax.detach().cpu().clone(), losses=losses + [state["loss3"]], norm=float(opt.grad_norm()))
        res["adamw_equal"] = all(torch.equal(runs["plain"][k], runs["sharded"][k]) for k in ("P", "ema", "M", "Vmax"))
        res["losses"] = (runs["plain"]["losses"], runs["sharded"]["losses"])
        res["norms"] = (runs["plain"]["norm"], runs["sharded"]["norm"])
        torch.save(res, out_path)
    finally:
        dist.destroy_process_group()


def test_rccl_call_sites_on_a_one_rank_group(gpu_device, tmp_path):
    import torch.multiprocessing as mp
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    out = str(tmp_path / "rccl.pt")
    p = mp.get_context("spawn").Process(target=_worker, args=(port, out))
    p.start()
    p.join(900)
    assert p.exitcode == 0, "the world-size-1 nccl worker failed (see its traceback above)"
    res = torch.load(out)
    print(f"[rccl world 1] {res}")
    assert res["shard_equal"], "broadcast / all_gather_counts / gather_records changed their data"
    assert res["sampling_equal"], "sampling through the RCCL gather differs from the single-rank shortcut"
    assert res["adamw_equal"], "reduce-scatter / all-gather step differs from the unsharded step"
    assert res["plain_restore_equal"] and res["sharded_restore_equal"], "save -> restore -> step diverged from the uninterrupted run"
    assert res["losses"][0] == res["losses"][1] and res["norms"][0] == res["norms"][1]
