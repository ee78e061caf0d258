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
