"""Headline benchmark: molecules/sec for 1000-step QM9S all-spectra conditional sampling (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W
One command drives all GPUs, as the reference's ``nn.DataParallel`` does (models/utils.py:27): with ``--gpus N > 1`` and no
``WORLD_SIZE`` in the environment the process only spawns ``python -m torch.distributed.run --nproc-per-node N bench.py ...``
(it never touches the GPU itself) and relays rank 0's JSON line; launched under ``torch.distributed.run`` directly (RANK /
WORLD_SIZE set) it is one rank of the job.

Default workload (``--mode eval``) = BASELINE config 2 literally: **10 000 samples per GPU** drawn through the product's
``get_cond_sampling_eval_fn(...)`` on a ``PackedSpectraTable``-backed synthetic test set - seed-42 permutation, size-sorted slot
assignment, micro-batches of ``--batch`` molecules, per micro-batch SpecFormer conditioning (once per molecule) + in-kernel
initial noise + 1000 DMT evaluations with the fused ancestral update + post-processing + 1 248-byte records, then the final
gather (the only collective) and the one device->host copy of the result tensors (the reference's per-molecule tuples are built
from them on access, ``sampling.MoleculeList``).  A bench *step* is one twentieth of
that whole evaluation, so the driver's ``--steps 20 --warmup 5`` times exactly one complete 10 000-sample run; warm-up steps
are slices of a throw-away run.  molecules/sec = molecules x (denoise iterations timed / iterations of the evaluation) / elapsed.
Molecules are independent, so ranks own disjoint sample slots (weak scaling: 10 000 per GPU).
``--mode resident`` keeps round 2's kernel-level workload (``--mols`` molecules resident, back-to-back passes; also config 4
with ``--unconditional``).

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     live HIP-event timing of the dominant kernel (k_equi_pairs) vs the ceiling of its arithmetic (f16 MFMA peak / 3),
  "cpu_baseline": the CPU oracle (faithful restatement of the reference path) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (155 measured)
PEAK_F16_MFMA_TFLOPS = 2516.6      # dense f16/bf16 MFMA peak (16x the fp32 MFMA rate; MI355X_MICROARCH.md "~2.5 PF dense")
PEAK_HBM_GBPS = 8000.0            # HBM3E (MI355X_MICROARCH.md: ~8 TB/s)
SPLIT_MFMAS_PER_PRODUCT = 3        # split-fp16 arithmetic: a*b = a1*b1 + (a1*b2 + a2*b1)/2048 -> three f16 MFMAs per fp32-accurate product
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / SPLIT_MFMAS_PER_PRODUCT   # ceiling of an fp32-accurate GEMM on the f16 matrix pipe
EQUI_MACS_PER_DIRECTED_EDGE = 256 * 256 + 256 * 3   # coord_mlp.0 + coord_mlp.2 (SURVEY §8d constants)


_T0 = time.perf_counter()


_OUT = None


def claim_stdout():
    """The process's stdout carries ONE JSON line.  Native libraries write there too (RCCL prints a five-line version banner on
    stdout when its first communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the JSON line
    goes to a private duplicate of the original descriptor."""
    global _OUT
    if _OUT is None:
        sys.stdout.flush()
        _OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _OUT


def emit(line: dict) -> None:
    out = claim_stdout()
    out.write(json.dumps(line) + "\n")
    out.flush()


def log(msg: str) -> None:
    """Progress on stderr (stdout carries only the JSON line)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """CPU cores this process can actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def algorithmic_macs(n_atoms) -> int:
    """de-duplicated GEMM MACs of one DMT evaluation over a batch."""
    n = np.asarray(n_atoms, dtype=np.int64)
    N, E, B = int(n.sum()), int((n * (n - 1)).sum()), len(n)
    return 8 * (620544 * N + 157184 * E + 2492416 * B) + (233216 * N + 33088 * E + 1330176 * B)


def pmc_traffic(kernel: str, mols: float):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r0N_pmc_traffic.json).

    bench.py cannot run the profiler on itself; the counters are collected with `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` in separate passes of this same workload (tools/pmc_report.py) and scaled per molecule."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            rec = json.load(open(path))[kernel]
            return rec["bytes_per_launch_per_molecule"] * mols, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def live_pmc_traffic(kernel: str, mols: int, spectra: str, budget_s: float = 120.0):
    """HBM bytes per launch of `kernel`, measured NOW: two child runs of this script under `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE,
    then WRITE_SIZE - separate passes, as MI355X_MICROARCH.md prescribes) on a resident batch of `mols` molecules drawn from the same
    size histogram, 4 denoise iterations each.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in kB; gfx950 reports
    half of wide fetch streams).  Children, not exec: this process has initialised the GPU.  Returns (bytes, E_dir of the profiled
    batch, note) or (None, None, reason) - the caller then falls back to the committed profile."""
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, None, "rocprofv3 not found"
    # Never start a profiler from a process that is itself being profiled: rocprofv3 is a `#!/usr/bin/env python3` script, and with
    # the outer profiler's LD_PRELOAD / ROCP* environment the preloaded tool would initialise the GPU inside `env` before it execs
    # python3 - the exec of a GPU-initialised process that this pool's machines do not survive.
    profiled = [k for k in os.environ if k.startswith(("ROCP", "ROCPROF", "ROCTRACER"))]
    if profiled or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None, None, "this process runs under a profiler (LD_PRELOAD / ROCP* set): no nested rocprofv3"
    t_begin = time.perf_counter()
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="ds_pmc_", dir="/tmp")
        cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--mode", "resident", "--mols", str(mols), "--spectra", spectra, "--steps", "1", "--warmup", "0", "--denoise-steps", "4",
               "--steps-per-pass", "1", "--no-cpu-baseline", "--no-live-traffic", "--profile-kernel", "-1"]
        try:
            left = budget_s - (time.perf_counter() - t_begin)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                                     "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "LD_PRELOAD")
                   and not k.startswith(("TORCHELASTIC", "ROCP", "ROCPROF", "ROCTRACER"))}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=max(30.0, left))
            if r.returncode != 0:
                return None, None, f"rocprofv3 --pmc {counter} exited with {r.returncode}"
            tot, cnt = 0.0, 0
            for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                import csv
                for row in csv.DictReader(open(path)):
                    if row.get("Counter_Name") == counter and kernel in row.get("Kernel_Name", ""):
                        tot += float(row["Counter_Value"])
                        cnt += 1
            if cnt == 0:
                return None, None, f"no {counter} samples of {kernel} in the rocprofv3 output"
            vals[counter] = tot / cnt
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
            return None, None, f"{type(exc).__name__}: {exc}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    from diffspectra_amd import filler
    n = filler.sample_n_atoms(mols, seed=0).astype(np.int64)
    e_dir = float((n * (n - 1)).sum())
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, e_dir, (
        f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child runs of this script, resident batch of {mols} molecules, "
        f"4 denoise iterations, {time.perf_counter() - t_begin:.0f} s), scaled by directed edges to this run's mean launch")


def live_pmc_family_bytes(kernel_substr: str, child_args, budget_s: float = 150.0):
    """HBM bytes moved by every launch of the kernels whose name contains ``kernel_substr`` over one child run of this script
    (``child_args``) under ``rocprofv3 --kernel-trace --pmc`` - FETCH_SIZE and WRITE_SIZE in separate passes, bytes = (2 * FETCH_SIZE +
    WRITE_SIZE) * 1024 as in ``live_pmc_traffic``.  Returns (total bytes, launches, note) or (None, 0, reason).  Same guards: never
    nested under a profiler, children not exec."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, 0, "rocprofv3 not found"
    if [k for k in os.environ if k.startswith(("ROCP", "ROCPROF", "ROCTRACER"))] or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None, 0, "this process runs under a profiler (LD_PRELOAD / ROCP* set): no nested rocprofv3"
    t_begin = time.perf_counter()
    tot, cnt = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="ds_pmc_", dir="/tmp")
        cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), *child_args]
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                                     "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "LD_PRELOAD")
                   and not k.startswith(("TORCHELASTIC", "ROCP", "ROCPROF", "ROCTRACER"))}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=max(30.0, budget_s - (time.perf_counter() - t_begin)))
            if r.returncode != 0:
                return None, 0, f"rocprofv3 --pmc {counter} exited with {r.returncode}"
            tot[counter], cnt[counter] = 0.0, 0
            for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(path)):
                    if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                        tot[counter] += float(row["Counter_Value"])
                        cnt[counter] += 1
            if cnt[counter] == 0:
                return None, 0, f"no {counter} samples of {kernel_substr} in the rocprofv3 output"
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
            return None, 0, f"{type(exc).__name__}: {exc}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0, cnt["FETCH_SIZE"], (
        f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two child runs of this script ({time.perf_counter() - t_begin:.0f} s), "
        f"summed over the {cnt['FETCH_SIZE']} launches of *{kernel_substr}* kernels")


def executed_macs(n_atoms) -> int:
    """MACs the kernels actually issue: the edge-side GEMMs whose operands are symmetric in (a, b) run once per unordered
    pair, only MultiCondEquiUpdate's coord_mlp (66 304 MACs) runs per directed edge."""
    n = np.asarray(n_atoms, dtype=np.int64)
    N, E, B = int(n.sum()), int((n * (n - 1)).sum()), len(n)
    sym, directed = 157184 - EQUI_MACS_PER_DIRECTED_EDGE, EQUI_MACS_PER_DIRECTED_EDGE
    return 8 * (620544 * N + (sym // 2 + directed) * E + 2492416 * B) + (233216 * N + (33088 // 2) * E + 1330176 * B)


def cpu_baseline(version: str, denoise_steps: int, sample_mols: int = 64, sample_steps: int = 6, budget_s: float = 25.0):
    """Time the CPU oracle (reference algorithm, SpecFormer re-encoded every step as the reference does) on the host.

    Bounded: at most ``sample_steps`` timed denoise steps and ``budget_s`` seconds, so the JSON line is always emitted."""
    import oracle
    from diffspectra_amd import filler
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.params import build_dmt_tree, Holder
    cfg = qm9s_config(version)
    tree = Holder()
    build_dmt_tree(tree, cfg)
    sd = filler.fill_state_dict(tree.state_dict())
    n_atoms = filler.sample_n_atoms(sample_mols, seed=0).tolist()
    x, ex, node_mask, edge_mask = filler.synthetic_state(n_atoms, "cpu.x")
    ctx = filler.synthetic_spectra(sample_mols, version, seed=1)
    nl = torch.zeros(sample_mols)
    cores = usable_cores()
    torch.set_num_threads(cores)                   # every core this process may run on (cgroup quota respected)
    cond = (None, None)
    times = []
    t_begin = time.perf_counter()
    for i in range(sample_steps + 1):              # first iteration = warm-up (and first-step branch)
        t0 = time.perf_counter()
        out = oracle.dmt_forward(sd, cfg, x, node_mask, edge_mask, ex, nl, cond[0], cond[1], context=ctx)
        times.append(time.perf_counter() - t0)
        cond = out
        if len(times) >= 3 and time.perf_counter() - t_begin > budget_s:
            break
    per_step = float(np.mean(times[1:]))
    return {"value": sample_mols / (per_step * denoise_steps), "unit": "molecules/sec", "cores": int(cores), "kind": "port",
            "sample": f"{sample_mols} molecules (QM9 size histogram, seed 0), {len(times) - 1} timed denoise steps of the "
                      f"faithful CPU oracle ({per_step:.3f} s/step), extrapolated to {denoise_steps} steps"}


def cpu_baseline_train(version: str, sample_mols: int = 16):
    """Time the CPU oracle's training step (oracle/train.py: the reference's loss_fn restated, torch autograd for the backward) on
    a bounded sample: one step with the self-conditioning forward and one without (the reference flips a fair coin per step)."""
    from oracle import train as otrain
    from diffspectra_amd import filler
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.params import build_dmt_tree, Holder
    cfg = qm9s_config(version)
    tree = Holder()
    build_dmt_tree(tree, cfg)
    sd = {}
    for k, v in filler.fill_state_dict(tree.state_dict()).items():
        v = v.clone()
        if v.is_floating_point() and not any(s in k for s in ("running_mean", "running_var", "sdp_attn.scale")):
            v.requires_grad_(True)
        sd[k] = v
    B = sample_mols
    n_atoms = filler.sample_n_atoms(B, seed=3).tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    g = torch.Generator().manual_seed(11)
    types = torch.randint(0, 5, (B, N), generator=g)
    order = torch.triu((torch.rand(B, N, N, generator=g) > 0.8).float() * torch.randint(1, 4, (B, N, N), generator=g), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(B, N, N)
    batch = dict(positions=torch.randn(B, N, 3, generator=g) * 1.3 * node_mask, atom_mask=node_mask.squeeze(-1), edge_mask=edge_mask,
                 atom_one_hot=torch.nn.functional.one_hot(types, 5).float() * node_mask,
                 edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1), formal_charges=torch.zeros(B, N, 1),
                 context=filler.synthetic_spectra(B, version, seed=5), n_atoms=n_atoms)
    draws = [torch.randn(B, N, 3, generator=g), torch.randn(B, N, 6, generator=g), torch.randn(B, 2, N, N, generator=g)]
    t_raw = torch.rand(B, generator=g) * 0.96 + 0.02
    cores = usable_cores()
    torch.set_num_threads(cores)
    times = []
    for coin in (True, False):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        loss, _ = otrain.training_loss(sd, cfg, batch, t_raw, draws, coin)
        loss.backward()
        times.append(time.perf_counter() - t0)
    per_step = float(np.mean(times))
    return {"value": B / per_step, "unit": "molecules/sec", "cores": int(cores), "kind": "port",
            "sample": f"{B} molecules (QM9 size histogram, seed 3), two training steps of the CPU oracle (forward of oracle/train.py, "
                      f"torch autograd backward; one with the self-conditioning forward, one without: {times[0]:.2f} s / {times[1]:.2f} s)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="eval", choices=["eval", "resident", "train"],
                    help="eval: BASELINE config 2 through the product sampling function (10 000 samples per GPU); "
                         "resident: --mols molecules resident, back-to-back passes (kernel-level A/B, config 4); "
                         "train: BASELINE config 5, one optimizer step per bench step on --train-batch molecules per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                    help="train mode: GEMM arithmetic (bf16 = config 5: bf16 products, fp32 accumulation and master weights)")
    ap.add_argument("--train-batch", type=int, default=256, help="train mode: molecules per GPU and step (config 5: 2048 over 8 GPUs)")
    ap.add_argument("--samples", type=int, default=10000, help="eval mode: samples per GPU")
    ap.add_argument("--batch", type=int, default=10000,
                    help="eval mode: micro-batch (molecules resident at a time).  10 000 = the whole per-GPU evaluation in one batch (~6 GB of workspace; "
                         "same-box round 5: 235.6 molecules/s against 233.4 with two micro-batches of 5 000)")
    ap.add_argument("--mols", type=int, default=4096, help="resident mode: molecules resident per GPU")
    ap.add_argument("--denoise-steps", type=int, default=1000)
    ap.add_argument("--steps-per-pass", type=int, default=20,
                    help="bench steps one complete evaluation (eval) / sampling pass (resident) is cut into")
    ap.add_argument("--budget-s", type=float, default=300.0,
                    help="wall-clock cap of the timed region: --steps is lowered (and reported) if it would not fit")
    ap.add_argument("--spectra", default="allspectra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true",
                    help="eval mode: do not append the short BASELINE config 5 measurement (8 bf16 training steps, ~10 s) to the line")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not collect roofline.traffic with two rocprofv3 --pmc child runs; use the committed profile instead")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal on one GPU: initialise a world-size-1 process group of --backend and run every collective of the path "
                         "(record all-gather, count gather; train mode: gradient reduce-scatter + parameter all-gather) instead of skipping them")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="eval mode with N GPUs: weak = --samples per GPU (N x 10 000 in all; what the driver's --gpus N runs); strong = "
                         "--samples in all, sharded --samples / N per GPU (BASELINE config 3: the 10 000-sample evaluation over 8 GPUs)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0 (1-GPU box, gloo backend)")
    ap.add_argument("--unconditional", action="store_true",
                    help="BASELINE config 4 (resident mode): zero context embedding, SpecFormer skipped (build extension)")
    ap.add_argument("--profile-kernel", type=int, default=5, help="block-stage kernel timed with HIP events (5 = k_equi_pairs)")
    ap.add_argument("--graph", default="off", choices=["auto", "on", "off"],
                    help="hipGraph replay of the denoise iteration (measured: no gain at any batch size, so off by default)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: ranks rendezvous (gloo), rank 0 prints a line with value null")
    args = ap.parse_args(argv)
    if args.unconditional:
        args.mode = "resident"
    return args


def self_launch(argv) -> int:
    """``python bench.py --gpus N`` without a launcher: spawn ``torch.distributed.run`` with N fresh ranks and relay rank 0's JSON
    line.  This parent never imports torch.cuda state or calls a GPU API (it must stay exec/fork-safe on the GPU box); the
    children are ordinary child processes, their stderr passes through, their stdout is filtered down to the one JSON line."""
    import socket
    import subprocess
    args = parse_args(argv)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    log(f"self-launch: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True)
    printed = 0
    for line in proc.stdout:
        t = line.strip()
        if t.startswith("{") and '"metric"' in t and printed == 0:
            print(t, flush=True)
            printed += 1
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and printed != 1:
        log("the ranks exited cleanly but printed no JSON line")
        return 1
    return rc


def config5_child(args, budget_s: float = 240.0) -> dict:
    """The config-5 object of the default line: `bench.py --mode train` (bf16, 256 molecules, 8 steps after 3 warm-up steps) as a child
    process; its one JSON line is reduced to the fields the driver line carries.  Never raises."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "train", "--gpus", "1", "--steps", "8", "--warmup", "3", "--precision", "bf16",
           "--train-batch", "256", "--spectra", args.spectra, "--no-cpu-baseline"] + (["--no-live-traffic"] if args.no_live_traffic else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                             "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "LD_PRELOAD")
           and not k.startswith(("TORCHELASTIC", "ROCP", "ROCPROF", "ROCTRACER"))}
    try:
        r = subprocess.run(cmd, cwd=os.path.dirname(os.path.abspath(__file__)), env=env, capture_output=True, text=True, timeout=budget_s)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not rows:
            return {"value": None, "error": f"child exited with {r.returncode}: {r.stderr.strip().splitlines()[-1:] or ''}"}
        t_line = json.loads(rows[-1])
        out = {"metric": t_line["metric"], "value": t_line["value"], "unit": t_line["unit"], "ms_per_step": t_line["ms_per_step"],
               "steps": t_line["steps"], "warmup": t_line["warmup"], "dtype": t_line["dtype"], "workload": t_line["config"]["workload"],
               "roofline": t_line["roofline"], "whole_path": t_line.get("whole_path"), "process": "child (bench.py --mode train)"}
        log(f"config 5: {t_line['value']:.0f} molecules/sec ({t_line['ms_per_step']:.1f} ms per step)")
        return out
    except Exception as exc:  # noqa: BLE001 - the headline line must not be lost to the secondary measurement
        return {"value": None, "error": f"{type(exc).__name__}: {exc}"}


def train_bench(args, world, rank, device):
    """``--mode train``: the config-5 line of ``train_measure`` as the one JSON line."""
    grouped = world > 1 or args.force_collectives
    line = train_measure(args, world, rank, device, args.steps, args.warmup, not args.no_cpu_baseline)
    if rank == 0:
        emit(line)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def train_measure(args, world, rank, device, steps, warmup, with_cpu_baseline):
    """BASELINE config 5 (secondary line, not the headline metric): the DMT training step on QM9S all-spectra through the product's
    ``losses.get_step_fn`` - batch preparation, forward diffusion, Kabsch alignment, p = 0.5 self-conditioning forward, training-mode
    SpecFormer, DMT forward + hand-written backward, gradient reduce-scatter, fused AdamW-amsgrad + clip + EMA, parameter all-gather.
    --precision bf16 (default, config 5) rounds every GEMM operand to bf16 with fp32 accumulation; storage, master weights and the
    optimizer stay fp32.  A bench step = one optimizer step on --train-batch molecules per GPU."""
    from diffspectra_amd import filler, losses as Lh
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.ema import ExponentialMovingAverage
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.registry import create_model
    import diffspectra_amd.dmt  # noqa: F401
    cfg = qm9s_config(args.spectra, device=device)
    cfg.training.precision = args.precision
    cfg.optim.force_sharded = bool(args.force_collectives)        # one-rank rehearsal of the reduce-scatter / all-gather step
    grouped = world > 1 or args.force_collectives
    model = create_model(cfg)
    filler.fill_module_(model)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
    opt = Lh.get_optimizer(cfg, model.parameters())
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    step_fn = Lh.get_step_fn(ns, True, Lh.optimization_manager(cfg), None, cfg)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    Bt = args.train_batch
    n_atoms = filler.sample_n_atoms(world * Bt, seed=3)[rank * Bt:(rank + 1) * Bt].tolist()
    node_mask, edge_mask = filler.masks_from_n_atoms(n_atoms)
    N = node_mask.shape[1]
    g = torch.Generator().manual_seed(11 + rank)
    types = torch.randint(0, 5, (Bt, N), generator=g)
    order = torch.triu((torch.rand(Bt, N, N, generator=g) > 0.8).float() * torch.randint(1, 4, (Bt, N, N), generator=g), 1)
    order = (order + order.transpose(1, 2)) * edge_mask.reshape(Bt, N, N)
    ctx = filler.synthetic_spectra(Bt, args.spectra, seed=5 + rank)
    batch = dict(positions=(torch.randn(Bt, N, 3, generator=g) * 1.3 * node_mask).to(device), atom_mask=node_mask.squeeze(-1).to(device),
                 edge_mask=edge_mask.to(device), atom_one_hot=(torch.nn.functional.one_hot(types, 5).float() * node_mask).to(device),
                 edge_one_hot=torch.stack([(order > 0).float(), order / 3.0], -1).to(device), formal_charges=torch.zeros(Bt, N, 1, device=device),
                 context=[c.to(device) for c in ctx] if isinstance(ctx, list) else ctx.to(device))

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(rank)
    import random as _random
    _random.seed(1234)          # the self-conditioning coin of every step (losses.py:344: random() < 0.5): a fixed sequence, so that two runs time
                                # the same mix of one- and two-forward steps (the extra forward is ~25 % of a step; unseeded, 10-step runs spread +-4 ms)
    for w in range(warmup):
        loss = step_fn(state, batch)
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        loss = step_fn(state, batch)
    host_issue = time.perf_counter() - t0            # the host's share: every launch of the K steps enqueued (no device wait inside a step)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(loss.detach()).all()
    # roofline of the dominant kernel family (the GEMMs: k_tr_gemm_big + its split-K reduction): one more step - on every rank, the step
    # holds collectives - with a HIP event pair around every dst_gemm call on its stream, outside the timed region; algorithmic
    # FLOPs 2 M N K per call
    from diffspectra_amd import train_engine as TE
    orig_gemm, recs = TE.Ops.gemm, []

    def timed_gemm(self, A, Bm, Cm, ta, tb, **kw):
        M, K = (A.cols, A.rows) if ta else (A.rows, A.cols)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_gemm(self, A, Bm, Cm, ta, tb, **kw)
        e1.record()
        recs.append((e0, e1, 2.0 * M * Cm.cols * K, (M, Cm.cols, K, int(ta), int(tb)), 4.0 * (M * K + K * Cm.cols + M * Cm.cols)))
    TE.Ops.gemm = timed_gemm
    prev = {k: os.environ.get(k) for k in ("DIFFSPECTRA_ASYNC_DW", "DIFFSPECTRA_NODE_STREAM")}
    for k in prev:
        os.environ[k] = "0"                          # one stream for this instrumented step: an event pair (recorded on torch's current
    try:                                             # stream) then brackets exactly one product
        step_fn(state, batch)
        sync()
    finally:
        TE.Ops.gemm = orig_gemm
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if rank == 0:
        n = np.asarray(n_atoms, dtype=np.int64)
        flop = 3.0 * 2.0 * algorithmic_macs(n)                        # forward + two backward GEMMs per forward GEMM (self-cond forward not counted)
        line = {"metric": "molecules/sec, DMT training step on QM9S all-spectra (BASELINE config 5; secondary line)", "value": world * Bt * steps / elapsed,
                "unit": "molecules/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": ("bf16 (GEMM operands rounded to bf16, fp32 accumulation, fp32 master weights / activations / optimizer state)"
                          if args.precision == "bf16" else "f32 (fp32 MFMA GEMMs)"),
                "data": "synthetic",
                "config": {"workload": f"DMT training step, QM9S {args.spectra}, {Bt} molecules per GPU and step (global batch {world * Bt}), dropout "
                                       f"{cfg.model.dropout}, AdamW-amsgrad + adaptive clip + EMA fused, gradient reduce-scatter + parameter all-gather; self-conditioning coin from random.seed(1234)",
                           "mode": "train", "molecules_per_gpu": Bt, "parallelism": f"dp{world}", "last_loss": float(loss.detach()),
                           "host_issue_ms_per_step": host_issue / steps * 1e3},
                "whole_path": {"algorithmic_tflops_per_gpu": flop * steps / elapsed / 1e12}}
        gemm_ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        gemm_flop = sum(r[2] for r in recs)
        if os.environ.get("DIFFSPECTRA_GEMM_TABLE") == "1":             # per-shape totals of the step's GEMM calls, on stderr
            table = {}
            for e0, e1, f, shape, _ in recs:
                t = table.setdefault(shape, [0, 0.0, 0.0])
                t[0] += 1
                t[1] += e0.elapsed_time(e1)
                t[2] += f
            for shape, (cnt, ms, f) in sorted(table.items(), key=lambda kv: -kv[1][1]):
                log(f"gemm M {shape[0]:6d} N {shape[1]:6d} K {shape[2]:6d} ta {shape[3]} tb {shape[4]}: {cnt:3d} calls {ms:7.3f} ms  {ms / cnt * 1e3:7.1f} us/call  {f / ms / 1e9:6.1f} TFLOP/s")
        peak = PEAK_F16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_FP32_MFMA_TFLOPS
        ach = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
        line["roofline"] = {"bound": "mfma", "kernel": "k_tr_gemm_bf16 (+ k_tr_gemm_reduce, its split-K reduction): every dst_gemm call of one step",
                            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": None if ach is None else ach / peak,
                            "peak_note": ("dense bf16 MFMA peak" if args.precision == "bf16" else "dense fp32 MFMA peak") +
                                         " (MI355X_MICROARCH.md); the step's GEMMs are small (M = nodes / pairs of 256 molecules, N, K <= 1024) "
                                         "and bound by launch latency and their operand streams, not by the matrix pipe",
                            "traffic": None, "launches_timed": len(recs), "avg_launch_ms": gemm_ms / max(1, len(recs)),
                            "share_of_step": gemm_ms / (elapsed / steps * 1e3), "algorithmic_flop_per_step": gemm_flop,
                            # the bound these products actually have: their fp32 operands read once and their output written once
                            "hbm_view": {"algorithmic_bytes_per_step": sum(r[4] for r in recs),
                                         "achieved_GBps": sum(r[4] for r in recs) / (gemm_ms * 1e-3) / 1e9 if gemm_ms > 0 else None,
                                         "peak_GBps": PEAK_HBM_GBPS,
                                         "frac": sum(r[4] for r in recs) / (gemm_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS if gemm_ms > 0 else None},
                            "note": "timed in one extra single-stream step (in the timed steps the weight-gradient products and the node-row chain run "
                                    "concurrently on their own streams, so this share is an upper bound of their part of the step)"}
        # HBM bytes of the same kernel family per step, from the counters: two child runs of this mode (1 warm-up + 2 timed + the one
        # instrumented step = 4 optimizer steps each) under rocprofv3 --pmc, summed over every k_tr_gemm* launch (split-K reductions included)
        if world == 1 and not args.no_live_traffic:
            child_steps = 2
            child = ["--mode", "train", "--gpus", "1", "--steps", str(child_steps), "--warmup", "1", "--precision", args.precision, "--train-batch", str(Bt),
                     "--spectra", args.spectra, "--no-cpu-baseline", "--no-live-traffic"]
            tot_b, launches, note = live_pmc_family_bytes("k_tr_gemm", child)
            if tot_b is not None:
                n_steps = child_steps + 1 + 1
                line["roofline"]["traffic"] = tot_b / n_steps
                line["roofline"]["traffic_unit"] = "HBM bytes per optimizer step, all k_tr_gemm* launches (compare hbm_view.algorithmic_bytes_per_step)"
                line["roofline"]["traffic_source"] = note + f"; {n_steps} optimizer steps in the profiled run, {launches / n_steps:.0f} launches per step"
            else:
                line["roofline"]["traffic_source"] = "not measured: " + note
        if with_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_train(args.spectra)
            except Exception as exc:  # noqa: BLE001 - the GPU line is still valid
                line["cpu_baseline"] = {"value": None, "unit": "molecules/sec", "cores": usable_cores(), "kind": "port", "sample": f"failed: {exc}"}
        return line
    return None




def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(argv))

    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert int(t.item()) == world - 1
            dist.barrier()
        # the slot sharding of the evaluation, as the product does it (shard.assign_slots over the synthetic size histogram), without a GPU
        from diffspectra_amd import filler, shard
        total = args.samples if args.scaling == "strong" else world * args.samples      # strong: any total, shares differ by at most one
        mine = shard.assign_slots(filler.sample_n_atoms(total, seed=0), rank, world)
        counts = shard.all_gather_counts(torch.tensor([mine.numel()], dtype=torch.int64), "cpu")
        if rank == 0:
            emit({"metric": "molecules/sec, 1000-step QM9S all-spectra sampling", "value": None, "unit": "molecules/sec",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": args.scaling,
                  "config": {"samples_total": total, "molecules_per_gpu": counts}})
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_collectives:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                                   # one-rank rehearsal group: no launcher has set the rendezvous up
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                sk.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    grouped = world > 1 or args.force_collectives

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if grouped:
        dist.barrier()
    if args.force_collectives:
        from diffspectra_amd import shard as _shard
        _shard.force_collectives(True)
    if args.mode == "train":
        return train_bench(args, world, rank, device)
    from diffspectra_amd import filler, sampling as S, engine as E, shard
    from diffspectra_amd.config import qm9s_config
    from diffspectra_amd.dataset_pack import PackedSpectraTable
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    from diffspectra_amd.registry import create_model
    from diffspectra_amd.scalers import get_data_inverse_scaler
    import diffspectra_amd.dmt  # noqa: F401

    if rank == 0:
        log("library built; creating model")
    cfg = qm9s_config(args.spectra, device=device, steps=args.denoise_steps)
    model = create_model(cfg)
    filler.fill_module_(model)
    model.eval()
    eng = model.module.engine()
    lib = eng.lib
    if rank == 0:
        log("weights packed on device")
    noise_sched = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                  continuous_beta_1=cfg.sde.continuous_beta_1)
    inv = get_data_inverse_scaler(cfg)
    spp = max(1, args.steps_per_pass)

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v: float) -> float:
        if not grouped:
            return v
        t = torch.tensor([v], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.mode == "eval":
        # ---- BASELINE config 2: the product's sampling function on a synthetic test set (QM9S second-half size histogram,
        # SURVEY §8d), 10 000 sample slots per GPU.  The spectra table lives in HBM (PackedSpectraTable, row N3).
        total = args.samples if args.scaling == "strong" else world * args.samples      # strong: any total (10 001 over 8 ranks: 1 251 / 1 250)
        base = min(total, 10000)                       # distinct synthetic spectra; larger test sets repeat them
        spec = filler.synthetic_spectra(base, args.spectra, seed=1)
        spec = spec if isinstance(spec, list) else [spec]
        from diffspectra_amd.config import used_spectra
        cols = [None, None, None]
        for k, t in zip(used_spectra(args.spectra), spec):
            t = t.to(device)
            cols[k] = t.repeat((total + base - 1) // base, 1, 1)[:total].contiguous()
        all_atoms = filler.sample_n_atoms(total, seed=0)
        table = PackedSpectraTable(cols, torch.from_numpy(all_atoms), device=device)
        cfg.sampling.seed = 42
        fn = S.get_cond_sampling_eval_fn(cfg, noise_sched, args.batch, total, inv, table)
        unit_desc = "evaluation"

        class Stream:
            """The product evaluation advanced one bench step at a time.  Every rank closes an evaluation (finish(): the one gather) at
            bench step spp, 2 spp, ... whatever its own share is - ranks with fewer molecules (an uneven strong split, a rank with none)
            run shorter slices but meet the others in the same collective of the same step."""
            def __init__(self):
                self.run = fn.start(model)
                self.total = self.run.total_iters
                self.slice = -(-self.total // spp)
                self.iters, self.passes, self.result, self.k = 0, 0, None, 0

            def step(self):
                if self.k and self.k % spp == 0:                       # more steps than one evaluation: back-to-back evaluations
                    self.run = fn.start(model)
                before = self.run.iters_done
                done = self.run.advance(self.slice) if self.slice else self.run.done
                self.iters += self.run.iters_done - before
                self.k += 1
                if self.k % spp == 0:
                    assert done, "an evaluation must be complete after steps_per_pass bench steps"
                    self.result = self.run.finish()
                    self.passes += 1

            def close(self):
                self.run.finish(partial=True)

        probe = Stream()
        slice_len, iters_per_unit = probe.slice, probe.total
        n_atoms_mine = np.asarray(probe.run.n_atoms)[probe.run.mine.numpy()]
        launches_e_dir = float((n_atoms_mine * (n_atoms_mine - 1)).sum()) / max(1, len(probe.run.batches))
        mols_per_gpu = int(probe.run.mine.numel())     # this rank's share (weak: --samples; strong: total / world, +1 on the first total % world ranks)
        mols_resident = min(args.batch, mols_per_gpu)
        samples_total = total
        del probe
    else:
        M = args.mols
        all_atoms = filler.sample_n_atoms(world * M, seed=0)
        n_atoms = all_atoms[rank * M:(rank + 1) * M].tolist()
        context = filler.synthetic_spectra(world * M, args.spectra, seed=1)
        context = [c[rank * M:(rank + 1) * M].to(device) for c in context] if isinstance(context, list) else context[rank * M:(rank + 1) * M].to(device)
        node_mask, edge_mask = S.build_masks(n_atoms, M, device, max_n=int(all_atoms.max()))
        sampler = S._make_sampler(cfg, noise_sched, 1e-3, cfg.eval.sampling_temperature)
        mol_ids = torch.arange(rank * M, (rank + 1) * M, dtype=torch.int64, device=device)
        sampler.use_graph = {"auto": "auto", "on": True, "off": False}[args.graph]
        graphed = sampler.use_graph is True or (sampler.use_graph == "auto" and eng.layout_for(node_mask, edge_mask)[0].Pp <= sampler.graph_max_pairs)
        if graphed:
            args.profile_kernel = -1          # HIP-event sampling brackets eager launches; a replayed graph has none
        spp = max(1, min(spp, args.denoise_steps))
        slice_len = -(-args.denoise_steps // spp)          # denoise iterations per bench step
        iters_per_unit = args.denoise_steps
        unit_desc = "sampling pass"
        n_atoms_mine = np.asarray(n_atoms, dtype=np.int64)
        launches_e_dir = float((n_atoms_mine * (n_atoms_mine - 1)).sum())
        mols_per_gpu = mols_resident = M
        samples_total = world * M

        class Stream:
            """Back-to-back sampling passes over the resident micro-batch, advanced one bench step at a time."""
            def __init__(self):
                self.st, self.result, self.passes, self.iters = None, None, 0, 0

            def step(self):
                if self.st is None:
                    self.st = sampler.begin(model, None, node_mask, edge_mask, None, None if args.unconditional else context,
                                            mol_ids=mol_ids, seed=42)
                before = self.st.i
                done = sampler.advance(self.st, slice_len)
                self.iters += self.st.i - before
                if done:
                    self.close()
                    self.passes += 1

            def close(self):
                if self.st is None:
                    return
                pos, one_hot, fc, edge_types = S.post_process(self.st.x_mean, 5, True, node_mask, inv, self.st.edge_mean,
                                                              edge_mask, True, engine=eng)
                rec = shard.gather_records(shard.pack_records_u8(pos, one_hot.argmax(-1), fc, edge_types))
                self.result = rec[rank * M:(rank + 1) * M]
                self.st = None

    # ---- warm-up: W steps of a throwaway run (code paths, allocator, layout cache, clocks), closed once so that the closing
    # torch kernels and the collective are loaded; the last warm-up step sizes the timed region against the wall-clock budget
    warm = Stream()
    t_step = 0.0
    for w in range(args.warmup):
        sync()
        t0 = time.perf_counter()
        with __import__("contextlib").redirect_stdout(sys.stderr):
            warm.step()
        sync()
        t_step = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            log(f"warmup step {w + 1}/{args.warmup} done ({t_step * 1e3:.0f} ms, {slice_len} denoise iterations x {mols_resident} molecules)")
    if args.warmup > 0:
        warm.close()
        sync()
    del warm
    steps = args.steps
    if t_step > 0 and steps * t_step > args.budget_s:
        steps = max(1, int(args.budget_s / t_step))
        if rank == 0:
            log(f"--steps {args.steps} would take ~{args.steps * t_step:.0f} s; running {steps} steps to stay inside {args.budget_s:.0f} s")

    every = max(1, (steps * slice_len * 8) // 2000)
    if args.profile_kernel >= 0:
        E._check(lib.ds_profile_config(C.c_int(args.profile_kernel), C.c_int(every), C.c_int(4096)), "ds_profile_config")
    import contextlib
    run = Stream()
    sync()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):          # the product prints the reference's progress lines; stdout carries only the JSON line
        for k in range(steps):
            run.step()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    tot_ms, samples = C.c_double(0.0), C.c_int64(0)
    E._check(lib.ds_profile_read(C.byref(tot_ms), C.byref(samples)), "ds_profile_read")
    lib.ds_profile_config(C.c_int(-1), C.c_int(1), C.c_int(0))
    # the timed work must be the real computation: check invariants the reference guarantees on its outputs
    if args.mode == "eval" and run.result is not None:
        processed = run.result[0]
        assert len(processed) == samples_total
        for pos_o, atom_o, et_o, fc_o in processed[::97]:
            assert torch.isfinite(pos_o).all() and float(pos_o.sum(0).abs().max()) < 1e-3, "generated positions are not zero-CoM"
            assert int(atom_o.min()) >= 0 and int(atom_o.max()) < 5, "atom types out of range"
            assert torch.equal(et_o, et_o.t()) and float(et_o.max()) <= 3.0, "bond orders not symmetric in {0..3}"
    elif args.mode == "resident" and run.result is not None:
        pos_o, atom_o, _, et_o = shard.unpack_records_u8(run.result)
        max_n = node_mask.shape[1]
        pos_o, atom_o, et_o = pos_o[:, :max_n], atom_o[:, :max_n], et_o[:, :max_n, :max_n]
        assert torch.isfinite(pos_o).all()
        nm = node_mask.squeeze(-1)
        assert float((pos_o * nm.unsqueeze(-1)).sum(1).abs().max()) < 1e-3, "generated positions are not zero-CoM"
        assert float((pos_o * (1 - nm).unsqueeze(-1)).abs().max()) == 0.0, "padded atoms carry positions"
        assert int(atom_o.min()) >= 0 and int(atom_o.max()) < 5, "atom types out of range"
        assert torch.equal(et_o, et_o.transpose(1, 2)) and float(et_o.max()) <= 3.0, "bond orders not symmetric in {0..3}"

    if rank == 0:
        frac_done = run.iters / iters_per_unit                          # fraction of the evaluation / pass that was timed
        value = samples_total * frac_done / elapsed
        n = n_atoms_mine.astype(np.int64)
        kern_ms = tot_ms.value / max(1, samples.value)
        kernel_names = ["k_edge_geom", "k_node_qkv", "k_attn_fused", "k_node_update", "k_edge_update", "k_equi_pairs", "unused"]
        roofline = None
        if samples.value > 0 and args.profile_kernel == 5:
            flop = 2.0 * EQUI_MACS_PER_DIRECTED_EDGE * launches_e_dir
            ach = flop / (kern_ms * 1e-3) / 1e12
            traffic = traffic_src = None
            if not args.no_live_traffic and world == 1:      # N > 1: the other ranks would sit in the closing barrier meanwhile
                live, e_prof, note = live_pmc_traffic("k_equi_pairs", int(mols_resident), args.spectra)
                if live is not None:
                    traffic, traffic_src = live * launches_e_dir / e_prof, note
                else:
                    log(f"live PMC traffic unavailable ({note}); using the committed profile")
            if traffic is None:
                traffic, traffic_src = pmc_traffic("k_equi_pairs", launches_e_dir / float((n * (n - 1)).mean()))
                traffic_src = f"{traffic_src} (committed profile scaled to this run's mean molecules per launch; not collected by this run)"
            roofline = {"bound": "mfma", "kernel": "k_equi_pairs", "achieved": ach, "peak": PEAK_SPLIT_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach / PEAK_SPLIT_TFLOPS,
                        "peak_note": "dense f16 MFMA peak (2516.6 TFLOP/s) / 3: the kernel evaluates its fp32-accurate 256x256 GEMM as three "
                                     "f16 MFMAs per product (split operands, fp32 accumulate); algorithmic FLOPs counted once",
                        "vs_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                        "traffic_unit": "bytes of HBM traffic per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 PMC counters in separate passes",
                        "traffic_source": traffic_src, "avg_launch_ms": kern_ms, "launches_timed": int(samples.value),
                        "algorithmic_flop_per_launch": flop}
        elif samples.value > 0:
            roofline = {"bound": "mfma", "kernel": kernel_names[args.profile_kernel], "avg_launch_ms": kern_ms,
                        "launches_timed": int(samples.value), "achieved": None, "peak": PEAK_FP32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": None, "traffic": None}
        # whole-path FLOPs: one DMT evaluation per molecule per denoise iteration
        fwd_flop_per_mol = 2.0 * algorithmic_macs(n) / len(n)
        exe_flop_per_mol = 2.0 * executed_macs(n) / len(n)
        mol_steps_per_s = value * args.denoise_steps / world
        whole = fwd_flop_per_mol * mol_steps_per_s / 1e12
        whole_exe = exe_flop_per_mol * mol_steps_per_s / 1e12
        complete = run.passes > 0 and steps == args.steps
        if args.mode == "eval":
            workload = (f"BASELINE config 2: QM9S {args.spectra}, DMT + SpecFormer (no pretrain), random-init procedural weights, "
                        f"{args.denoise_steps} denoise steps, {mols_per_gpu} samples per GPU ({samples_total} in all, {args.scaling} scaling) through the product "
                        f"get_cond_sampling_eval_fn (synthetic PackedSpectraTable test set, n_atoms ~ qm9_second_half histogram, "
                        f"mean {float(n.mean()):.2f}; seed-42 permutation, size-sorted slots, micro-batches of {args.batch}); one bench "
                        f"step = 1/{spp} of the evaluation ({slice_len} denoise iterations), {spp} steps = the complete "
                        f"{samples_total}-sample run incl. SpecFormer, initial noise, post-processing, the final gather and "
                        "the single device->host copy of the result tensors (per-molecule tuples are views built on access)")
        else:
            workload = (("QM9S unconditional (zero context embedding), DMT only" if args.unconditional else
                         f"QM9S {args.spectra}, DMT + SpecFormer (no pretrain)") + ", random-init procedural weights, "
                        f"{args.denoise_steps} denoise steps per molecule, {mols_per_gpu} molecules resident per GPU "
                        f"(n_atoms ~ qm9_second_half histogram, mean {float(n.mean()):.2f}); one bench step = "
                        f"{slice_len} denoise iterations over the resident batch, {spp} steps = one complete "
                        f"{args.denoise_steps}-step sampling pass incl. SpecFormer, initial noise, post-processing "
                        "and the final gather")
        if not complete:
            workload += (f" -- PARTIAL: {run.iters} of {iters_per_unit} denoise iterations of one {unit_desc} were timed "
                         "(opening work charged in full, closing work not reached); not comparable with a complete run")
        line = {
            "metric": "molecules/sec, 1000-step QM9S all-spectra sampling" + ("" if complete else " (partial run)"),
            "value": value, "unit": "molecules/sec",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling if args.mode == "eval" else "weak", "vs_baseline": None,
            "dtype": "f32 (GEMMs as split-fp16 x3 MFMA with fp32 accumulate; fp32-level accuracy, parity gates unchanged)", "data": "synthetic",
            "library": entry.LIBRARY_STATE,
            "config": {"workload": workload, "mode": args.mode, "samples_total": samples_total,
                       "collectives": ("forced on a one-rank group (rehearsal)" if args.force_collectives and world == 1 else
                                       f"{args.backend} over {world} ranks" if world > 1 else "none (one rank)"),
                       "molecules_per_gpu": mols_per_gpu, "molecules_resident_per_gpu": mols_resident,
                       "denoise_steps": args.denoise_steps, "denoise_iterations_per_step": slice_len,
                       "steps_per_pass": spp, "passes_completed": run.passes, "denoise_iterations_timed": run.iters,
                       "steps_requested": args.steps, "parallelism": f"dp{world} (molecule shards)",
                       "launch_mode": "eager launches" if args.mode == "eval" or not graphed else "hipGraph replay per denoise iteration"},
            "roofline": roofline,
            "whole_path": {"algorithmic_tflops_per_gpu": whole, "vs_fp32_mfma_peak": whole / PEAK_FP32_MFMA_TFLOPS,
                           "frac_of_split_f16_ceiling": whole / PEAK_SPLIT_TFLOPS,
                           "executed_tflops_per_gpu": whole_exe, "executed_vs_fp32_mfma_peak": whole_exe / PEAK_FP32_MFMA_TFLOPS,
                           "executed_frac_of_split_f16_ceiling": whole_exe / PEAK_SPLIT_TFLOPS,
                           "algorithmic_gflop_per_molecule_step": fwd_flop_per_mol / 1e9,
                           "executed_gflop_per_molecule_step": exe_flop_per_mol / 1e9},
        }
        log(f"GPU timing done: {value:.2f} molecules/sec ({elapsed:.1f} s for {steps} steps)")
        if args.mode == "eval" and world == 1 and not args.no_config5:
            # BASELINE config 5 where the driver sees it: a short run of the training step (`--mode train`) after the timed region, in a
            # CHILD process with a timeout - a hang, a fault or the training library's side effects (its env switches, its patched GEMM
            # timer) cannot touch the headline number measured above, which is already complete at this point
            del run
            torch.cuda.empty_cache()
            line["config5"] = config5_child(args)
        log("timing CPU baseline")
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.spectra, args.denoise_steps)
            except Exception as exc:   # the GPU line must not be lost to a host-side problem
                line["cpu_baseline"] = {"value": None, "unit": "molecules/sec", "cores": usable_cores(), "kind": "port",
                                        "sample": f"failed: {type(exc).__name__}: {exc}"}
        emit(line)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
