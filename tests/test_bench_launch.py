"""bench.py drives all GPUs from one command (as the reference's nn.DataParallel does, models/utils.py:27): with --gpus N > 1
and no WORLD_SIZE it spawns torch.distributed.run itself and relays rank 0's single JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                     # exactly one line on stdout: the JSON record
    return json.loads(lines[0]), p.stderr


def test_self_launch_two_ranks_dry_run():
    rec, err = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--dry-run", "--steps", "3", "--warmup", "1"], 300)
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["steps"] == 3 and rec["warmup"] == 1
    assert "torch.distributed.run" in err                 # went through the spawn path


@pytest.mark.parametrize("scaling,total,per_rank", [("weak", 80, [40, 40]), ("strong", 40, [20, 20])])
def test_scaling_modes_shard_the_evaluation(scaling, total, per_rank):
    """--scaling weak: --samples per GPU (what the driver's --gpus N runs); strong: --samples in all, sharded over the ranks (BASELINE
    config 3).  The dry run deals the sample slots with the product's shard.assign_slots and gathers the per-rank counts over gloo."""
    rec, _ = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--samples", "40", "--scaling", scaling], 300)
    assert rec["scaling"] == scaling and rec["config"]["samples_total"] == total and rec["config"]["molecules_per_gpu"] == per_rank


@pytest.mark.parametrize("scaling,samples,total,per_rank", [("weak", 1250, 10000, [1250] * 8), ("strong", 10001, 10001, [1251] + [1250] * 7),
                                                            ("strong", 5, 5, [1, 1, 1, 1, 1, 0, 0, 0])])
def test_eight_rank_dry_run(scaling, samples, total, per_rank):
    """The SCALE driver's widest shape (`--gpus 8`) without GPUs: eight gloo ranks through the spawn path, the evaluation dealt by the
    product's shard.assign_slots - an even weak split, an uneven strong split (10 001 samples) and ranks that own nothing - and exactly
    ONE JSON line on stdout (from rank 0)."""
    rec, err = _run(["--gpus", "8", "--backend", "gloo", "--dry-run", "--samples", str(samples), "--scaling", scaling], 600)
    assert rec["n_gpus"] == 8 and rec["scaling"] == scaling and rec["dry_run"] is True
    assert rec["config"]["samples_total"] == total and rec["config"]["molecules_per_gpu"] == per_rank
    assert "torch.distributed.run" in err


def test_single_rank_does_not_spawn():
    rec, err = _run(["--gpus", "1", "--dry-run"], 120)
    assert rec["n_gpus"] == 1 and "self-launch" not in err


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu():
    """The whole multi-rank bench (product sampling function, slot sharding, final gather) through the spawn path: two gloo
    ranks sharing cuda:0."""
    rec, _ = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--samples", "40", "--batch", "24", "--denoise-steps", "20",
                   "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-live-traffic"], 900)
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["passes_completed"] == 1
    assert rec["config"]["molecules_per_gpu"] == 40 and "partial" not in rec["metric"]


@pytest.mark.gpu
def test_single_gpu_bench_prints_exactly_one_line():
    """The driver's N = 1 invocation shape on a tiny evaluation: stdout carries the JSON line and nothing else (the product's
    progress prints go to stderr), the run completes one evaluation and the line carries roofline + config."""
    rec, err = _run(["--gpus", "1", "--samples", "40", "--batch", "24", "--denoise-steps", "20", "--steps", "20", "--warmup", "2",
                     "--no-cpu-baseline", "--no-live-traffic"], 900)
    assert rec["n_gpus"] == 1 and rec["config"]["passes_completed"] == 1 and rec["value"] > 0
    assert "Generate 40, Total 40." in err and "roofline" in rec and rec["config"]["mode"] == "eval"


@pytest.mark.gpu
def test_strong_scaling_two_ranks_on_one_gpu():
    """BASELINE config 3's split on one card: 40 samples in all, 20 per rank, one final gather."""
    rec, _ = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--scaling", "strong", "--samples", "40", "--batch", "24",
                   "--denoise-steps", "20", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-live-traffic"], 900)
    assert rec["scaling"] == "strong" and rec["config"]["samples_total"] == 40 and rec["config"]["molecules_per_gpu"] == 20
    assert rec["value"] > 0 and rec["config"]["passes_completed"] == 1


@pytest.mark.gpu
def test_force_collectives_rehearsal_on_one_gpu():
    """`bench.py --gpus 1 --backend nccl --force-collectives`: a world-size-1 RCCL group, the record gather really executed."""
    rec, _ = _run(["--gpus", "1", "--backend", "nccl", "--force-collectives", "--samples", "40", "--batch", "24", "--denoise-steps", "20",
                   "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-live-traffic"], 900)
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and "forced" in rec["config"]["collectives"]


@pytest.mark.gpu
@pytest.mark.parametrize("samples,per_rank", [(41, [14, 14, 13]), (2, [1, 1, 0])])
def test_uneven_strong_split_on_one_gpu(samples, per_rank):
    """Shares that differ, and a rank that owns no molecule at all: every rank still meets the others in the one closing gather of the
    same bench step (three gloo ranks sharing cuda:0; rank 0 reports its own share)."""
    rec, _ = _run(["--gpus", "3", "--backend", "gloo", "--same-device", "--scaling", "strong", "--samples", str(samples), "--batch", "24",
                   "--denoise-steps", "20", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-live-traffic"], 900)
    assert rec["config"]["samples_total"] == samples and rec["config"]["molecules_per_gpu"] == per_rank[0]
    assert rec["value"] > 0 and rec["config"]["passes_completed"] == 1
