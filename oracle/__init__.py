"""CPU oracle for the DiffSpectra denoising hot path — TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU fp32 restatements of the reference algorithm, every function
citing the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package, and only as
the checker / the timed CPU baseline — never as the shipped compute path.  The
product (``diffspectra_amd``) raises if its HIP library is missing; it never
falls back to this code.

Parity pinning: the reference ships no tests, so the oracle is pinned
by golden vectors generated in the build container by running the reference's own
source files (``tests/golden/generate_golden.py``); the torch-only reference files
(noise schedule, SpecFormer, model utils, scalers) run unmodified, the PyG-dependent
ones (``models/layers.py``, ``models/dmt.py``, ``sampling.py``) run unmodified on top
of small stand-ins that restate the published semantics of the five un-vendored
third-party symbols (``torch_geometric==2.4.0``: ``MessagePassing.propagate``,
``utils.softmax``, ``utils.dense_to_sparse``; ``torch_scatter.scatter``;
``torch_sparse.sample`` (unused)).  Parity *at that third-party boundary* is
therefore pinned by restated semantics, not by the third-party binaries.
"""
from .schedule import cosine_log_alpha, marginal_prob, ancestral_coefficients  # noqa: F401
from .specformer import specformer_forward  # noqa: F401
from .dmt import dmt_forward, context_embedding  # noqa: F401
from .sampler import (ancestral_sampling, combined_noise, symmetric_edge_noise, post_process,  # noqa: F401
                      mol_process, inverse_scale, self_cond_ori, self_cond_clamp)
from . import philox  # noqa: F401,E402
