"""Oracle (TEST INFRASTRUCTURE ONLY): the per-molecule counter-based noise streams of ``ds_initial_noise`` /
``ds_sampler_step_philox`` restated in numpy.

Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
generator behind ``torch.randn`` on GPUs and cuRAND/rocRAND): multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments
0x9E3779B9 / 0xBB67AE85, ten rounds.  Known-answer vectors of the Random123 distribution pin it
(``tests/test_oracle_golden.py::test_philox_known_answers``).  The noise transforms are the reference's
(``models/utils.py:67-106``): masked N(0,1), CoM-projected position noise, one edge draw per unordered pair and channel.
Counter layout (include/diffspectra_hip.h): (element, draw, mol_id low, kind | mol_id high << 1), key = 64-bit seed.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over equal-shaped uint32 arrays (scalars broadcast); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _box_muller(a, b):
    f = np.float32
    # fmaf(float(a), 2^-32, 2^-33): float(a) rounds to 24 bits, the product by a power of two is exact, one rounding at the end
    u0 = (a.astype(f).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(f)
    u1 = (b.astype(f).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(f)
    r = np.sqrt(f(-2.0) * np.log(u0)).astype(f)
    th = (f(6.283185307179586) * u1).astype(f)
    return (r * np.cos(th)).astype(f), (r * np.sin(th)).astype(f)


def normal4(elem, draw, mol_id, kind, seed):
    """[len(elem), 4] float32 normals of counter (elem, draw, mol_id, kind) under key ``seed``."""
    elem = np.asarray(elem, dtype=np.uint64)
    mol_id, seed = int(mol_id), int(seed)
    c3 = (kind | ((mol_id >> 32) << 1)) & 0xFFFFFFFF
    x, y, z, w = philox4x32_10(elem, np.uint64(draw), np.uint64(mol_id & 0xFFFFFFFF), np.uint64(c3), seed & 0xFFFFFFFF, seed >> 32)
    a0, a1 = _box_muller(x, y)
    b0, b1 = _box_muller(z, w)
    return np.stack([a0, a1, b0, b1], axis=1)


def molecule_noise(seed, draw, mol_id, n):
    """Noise of one molecule with ``n`` atoms for draw ``draw`` (0 = initial noise, 1 + step for denoise step ``step``):
    ``(pos [n,3] CoM-projected, feat [n,6], edge [n,n,2] symmetric with zero diagonal)``, float32."""
    f = np.float32
    nz = normal4(np.arange(3 * n), draw, mol_id, 0, seed).reshape(n, 12)
    pos = nz[:, :3].copy()
    mean = np.zeros(3, dtype=f)
    for a in range(n):                       # ascending-atom fp32 sum, as the kernel
        mean = (mean + pos[a]).astype(f)
    pos = (pos - (mean / f(n)).astype(f)).astype(f)
    feat = nz[:, 3:9].copy()
    edge = np.zeros((n, n, 2), dtype=f)
    if n > 1:
        hi, lo = np.tril_indices(n, -1)      # p = hi(hi-1)/2 + lo enumerates exactly this order
        p = hi * (hi - 1) // 2 + lo
        e = normal4(p, draw, mol_id, 1, seed)[:, :2]
        edge[hi, lo] = e
        edge[lo, hi] = e
    return pos, feat, edge


def dropout_keep(seed, stream_id, n, p):
    """The FF-dropout masks of the training kernels (``dst_dropout`` / the fused GEMM epilogues, csrc/ds_train.hip): element i of a
    tensor is kept iff word ``i & 3`` of Philox block (i >> 2, stream_id, 'DROP') under key ``seed`` is >= p * 2^32 (evaluated in
    fp32, as the kernel does).  Returns a bool array of ``n`` elements; kept elements are scaled by ``dropout_scale(p)``."""
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    w = philox4x32_10(q & MASK, q >> np.uint64(32), np.uint64(stream_id), np.uint64(0x44524F50), int(seed) & 0xFFFFFFFF, int(seed) >> 32)
    thr = np.uint32(min(np.float32(p) * np.float32(4294967296.0), np.float32(4294967040.0)))
    return (np.stack(w, axis=1).reshape(-1)[:n] >= thr)


def dropout_scale(p):
    """1 / (1 - p) in fp32, as the host side of ``dst_dropout`` computes it."""
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p))
