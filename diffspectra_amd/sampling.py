"""Ancestral reverse-diffusion sampler with the reference's surface, driven on the HIP library.

Mirrors reference ``sampling.py``: ``AncestralSampler(noise_scheduler, time_steps, model_pred_data, pred_edge,
self_cond, cond_process_fn, sampling_temperature).sampling(model, z_T, node_mask, edge_mask, edge_z_T, context)``
(``:553-631``), ``get_cond_sampling_eval_fn`` / ``get_sampling_fn`` (``:148,353``) returning ``sampling_fn(model)``
→ ``(processed_mols, gt_pos, gt_rdmols)``, ``post_process`` (``:53-97``) and ``mol_process`` (``:12-32``).

What differs is how it runs: the loop-invariant spectra embedding is computed once (the reference re-encodes it
every step, dmt.py:348-350), each step is one ``ds_forward`` + one fused ``ds_sampler_step``, the per-step
scalars come from a table computed up front, and molecules leave the GPU in one copy.
"""
from __future__ import annotations

from collections.abc import Sequence
from typing import Callable, Optional

import numpy as np
import torch

from .scalers import get_self_cond_fn, hip_post_process_supported


def _unwrap(model):
    return getattr(model, "module", model)


def _hip_model(model):
    m = _unwrap(model)
    if not hasattr(m, "engine"):
        raise TypeError("diffspectra_amd samplers drive the HIP DMT (diffspectra_amd.dmt.DMT); "
                        f"got {type(m).__name__}. There is no generic PyTorch fallback.")
    return m


class AncestralSampler:
    """Ancestral sampling for 2D & 3D joint generation (data-prediction + self-conditioning path)."""

    def __init__(self, noise_scheduler, time_steps, model_pred_data, pred_edge=False, self_cond=False,
                 cond_process_fn=None, sampling_temperature=1.0):
        if not (model_pred_data and pred_edge and self_cond):
            raise ValueError("the MI355X sampler implements the shipped mode: pred_data, pred_edge and self_cond all True")
        self.noise_scheduler = noise_scheduler
        self.t_array = time_steps
        self.s_array = torch.cat([time_steps[1:], torch.zeros(1, device=time_steps.device)])
        self.model_pred_data, self.pred_edge, self.self_cond = model_pred_data, pred_edge, self_cond
        self.cond_process_fn = cond_process_fn
        self.sampling_temperature = sampling_temperature
        self.noise_fn: Optional[Callable] = None     # noise_fn(i) -> (raw_pos, raw_feat, raw_edge): injected randn draws
        self.progress_fn: Optional[Callable] = None  # progress_fn(i, n_steps), called every 100 steps (long runs)
        self.philox_seed = 42                        # key of the per-molecule noise streams (begin(..., mol_ids=...))
        # hipGraph replay of the denoise iteration.  Off by default: measured on the MI355X
        # (profiles/r02_throughput_vs_batch.jsonl) a replayed iteration takes the same time as ~90 eager launches at every
        # batch size from 64 to 2048 molecules (2.66 ms vs 2.66 ms at 64) - small batches are bound by the latency of the
        # dependent kernel chain on a mostly empty chip, not by host launch work.  True forces it, 'auto' applies it below
        # graph_max_pairs packed pair rows; results are bit-identical either way (tests/test_hip_parity.py).
        self.use_graph = False
        self.graph_max_pairs = 120_000
        self._table = None

    def coefficient_table(self):
        """Per-step (c_x, c_pred, sigma, noise_level), the scalar algebra of sampling.py:572-584,604-606 in torch fp32."""
        if self._table is None:
            ns = self.noise_scheduler
            rows = []
            for i in range(len(self.t_array)):
                t, s = self.t_array[i].detach().cpu(), self.s_array[i].detach().cpu()
                alpha_t, sigma_t = ns.marginal_prob(t)
                alpha_s, sigma_s = ns.marginal_prob(s)
                alpha_t_given_s = alpha_t / alpha_s
                sigma2_t_given_s = sigma_t ** 2 - alpha_t_given_s ** 2 * sigma_s ** 2
                sigma = torch.sqrt(sigma2_t_given_s) * sigma_s / sigma_t
                rows.append(torch.stack([alpha_t_given_s * sigma_s ** 2 / sigma_t ** 2,
                                         alpha_s * sigma2_t_given_s / sigma_t ** 2, sigma,
                                         torch.log(alpha_t ** 2 / sigma_t ** 2)]))
            self._table = torch.stack(rows).to(torch.float32)
        return self._table

    @torch.no_grad()
    def begin(self, model, z_T, node_mask, edge_mask, edge_z_T=None, context=None, mol_ids=None, seed=None):
        """State of one sampling pass before its first denoise step: noisy tensors cloned onto the GPU, the (hoisted,
        loop-invariant) spectra embedding, the per-step coefficient table.  ``advance`` runs denoise steps on it.

        ``mol_ids`` (int64 [B]) + ``seed`` select the in-kernel noise: every molecule then has its own Philox stream keyed
        on (seed, mol_id), so its trajectory does not depend on the batch or rank it is sampled in.  ``z_T=None`` draws
        the initial noise from the same streams."""
        m = _hip_model(model)
        eng = m.engine()
        dev = eng.device
        L, ws = eng.layout_for(node_mask, edge_mask, validate=True)
        B, N = L.B, L.N
        st = _Pass()
        st.eng, st.L, st.ws, st.i = eng, L, ws, 0
        st.mol_ids, st.seed = None, 0
        if mol_ids is not None:
            st.mol_ids = torch.as_tensor(mol_ids, dtype=torch.int64).to(dev).contiguous()
            st.seed = int(self.philox_seed if seed is None else seed)
            if st.mol_ids.numel() != B:
                raise ValueError("mol_ids must hold one id per molecule of the batch")
        if z_T is None:
            if st.mol_ids is None:
                raise ValueError("z_T=None needs mol_ids (in-kernel initial noise)")
            st.x, st.edge_x = eng.initial_noise_philox(L, st.seed, st.mol_ids)
        else:
            st.x = z_T.detach().to(dev, torch.float32).contiguous().clone()
            st.edge_x = edge_z_T.detach().to(dev, torch.float32).contiguous().clone()
        st.ctx = eng.context_embedding(context)                      # hoisted: loop-invariant
        tab = self.coefficient_table()
        st.coef = tab.tolist()
        st.nl_rows = tab[:, 3].to(dev).reshape(-1, 1).expand(-1, B).contiguous()   # [S, B]: row i = noise level of step i
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        st.pred, st.edge_pred = [f(B, N, 9), f(B, N, 9)], [f(B, N, N, 2), f(B, N, N, 2)]
        st.x_mean, st.edge_mean = torch.zeros(B, N, 9, device=dev), torch.zeros(B, N, N, 2, device=dev)
        st.cond_x = st.cond_edge_x = None
        st.graph = None
        return st

    # ------------------------------------------------------------------ hipGraph replay of one denoise iteration
    def _graph_wanted(self, st):
        if st.mol_ids is None or self.noise_fn is not None or self.progress_fn is not None:
            return False                     # needs the in-kernel noise; injected noise / progress callbacks stay eager
        if self.cond_process_fn is not None and getattr(self.cond_process_fn, "in_place_clamp", False):
            return False                     # the 'clamp' hook allocates per call
        if self.use_graph == "auto":
            return st.L.Pp <= self.graph_max_pairs
        return bool(self.use_graph)

    def _graph_body(self, st):
        """One iteration with constant launch arguments: the step index lives in device memory (ds_step_begin), the
        prediction is written in place over the self-conditioning input (ds_stage_init has consumed it by then)."""
        eng, L, ws, g = st.eng, st.L, st.ws, st.graph
        eng.step_begin(g["table"], len(st.coef), g["step"], L.B, g["nl"])
        eng.forward(L, ws, st.x, st.edge_x, g["nl"], g["pred"], g["edge_pred"], st.ctx, g["pred"], g["edge_pred"])
        eng.sampler_step_philox_dev(L, g["table"], g["step"], float(self.sampling_temperature), st.seed, st.mol_ids, st.x,
                                    st.edge_x, g["pred"], g["edge_pred"], st.x_mean, st.edge_mean)

    def _graph_advance(self, st, end):
        """Iterations st.i .. end-1 by graph replay (st.i >= 1: the first iteration takes the no-conditioning branch)."""
        eng, dev = st.eng, st.eng.device
        if st.graph is None:
            tab = self.coefficient_table().to(dev).contiguous()
            g = dict(table=tab, step=torch.zeros(1, dtype=torch.int32, device=dev), nl=torch.empty(st.L.B, device=dev),
                     pred=st.cond_x, edge_pred=st.cond_edge_x, graph=None)
            st.graph = g
            g["step"].fill_(st.i - 1)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._graph_body(st)                              # a real iteration, eagerly, on the capture stream
            torch.cuda.current_stream(dev).wait_stream(side)
            st.i += 1
            if st.i >= end:
                return
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                self._graph_body(st)
            g["graph"] = graph
        g = st.graph
        g["step"].fill_(st.i - 1)
        if g["graph"] is None:                                    # captured lazily when only one iteration was asked for
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._graph_body(st)
            g["graph"] = graph
        for _ in range(st.i, end):
            g["graph"].replay()
        st.i = end

    @torch.no_grad()
    def advance(self, st, n_steps=None):
        """Run the next ``n_steps`` denoise steps of the pass (all remaining ones by default); True when it is complete."""
        eng, L, ws = st.eng, st.L, st.ws
        dev = eng.device
        B, N = L.B, L.N
        temp = float(self.sampling_temperature)
        end = len(st.coef) if n_steps is None else min(len(st.coef), st.i + int(n_steps))
        if st.i >= 1 and st.i < end and self._graph_wanted(st):
            self._graph_advance(st, end)
            return end >= len(st.coef)
        stop_eager = end
        if st.i == 0 and end > 1 and self._graph_wanted(st):
            stop_eager = 1                                         # first iteration eagerly, the rest by graph replay
        for i in range(st.i, stop_eager):
            c_x, c_pred, sigma, _ = st.coef[i]
            cur = i & 1
            eng.forward(L, ws, st.x, st.edge_x, st.nl_rows[i], st.cond_x, st.cond_edge_x, st.ctx, st.pred[cur], st.edge_pred[cur])
            st.cond_x, st.cond_edge_x = st.pred[cur], st.edge_pred[cur]
            if self.cond_process_fn is not None:                   # sampling.py:590 ('ori' identity, 'clamp' in place)
                st.cond_x, st.cond_edge_x = self.cond_process_fn(st.cond_x, st.cond_edge_x)
            if st.mol_ids is not None and self.noise_fn is None:   # per-molecule Philox streams, generated in the kernel
                eng.sampler_step_philox(L, c_x, c_pred, sigma, temp, st.seed, i, st.mol_ids, st.x, st.edge_x, st.pred[cur],
                                        st.edge_pred[cur], st.x_mean, st.edge_mean)
            else:
                if self.noise_fn is not None:
                    raw = [r.to(dev, torch.float32).contiguous() for r in self.noise_fn(i)]
                else:                                              # reference draw order/shapes (models/utils.py:69,78,102)
                    raw = [torch.randn((B, N, 3), device=dev), torch.randn((B, N, 6), device=dev),
                           torch.randn((B, 2, N, N), device=dev)]
                eng.sampler_step(L, c_x, c_pred, sigma, temp, st.x, st.edge_x, st.pred[cur], st.edge_pred[cur], raw[0], raw[1],
                                 raw[2], st.x_mean, st.edge_mean)
            if self.progress_fn is not None and (i + 1) % 100 == 0:
                self.progress_fn(i + 1, len(st.coef))
        st.i = stop_eager
        if stop_eager < end:
            self._graph_advance(st, end)
        return end >= len(st.coef)

    @torch.no_grad()
    def sampling(self, model, z_T, node_mask, edge_mask, edge_z_T=None, context=None, mol_ids=None, seed=None):
        st = self.begin(model, z_T, node_mask, edge_mask, edge_z_T, context, mol_ids=mol_ids, seed=seed)
        self.advance(st)
        return st.x_mean, st.edge_mean


class _Pass:
    """Mutable state of one in-flight sampling pass (``AncestralSampler.begin`` / ``advance``)."""
    __slots__ = ("eng", "L", "ws", "i", "x", "edge_x", "ctx", "coef", "nl_rows", "pred", "edge_pred", "x_mean", "edge_mean",
                 "cond_x", "cond_edge_x", "mol_ids", "seed", "graph")


def post_process(xh, atom_types, include_charge, node_mask, inverse_scaler, edge_x=None, edge_mask=None,
                 compress_edge=False, engine=None):
    """sampling.py:53-97 on the GPU (``ds_post_process``).  Returns (pos, one_hot, fc, edge_types) like the reference."""
    if engine is None:
        raise RuntimeError("post_process runs in the HIP library: pass engine=model.engine()")
    if not (compress_edge and include_charge and atom_types == 5 and edge_x is not None
            and getattr(inverse_scaler, "factors", (1, 4, 4, 1)) == (1, 4, 4, 1)):
        raise ValueError("ds_post_process implements the shipped configuration (compress_edge, charges, 5 atom types)")
    L, _ = engine.layout_for(node_mask, edge_mask)
    pos, atom, fc, et = engine.post_process(L, xh, edge_x)
    one_hot = torch.nn.functional.one_hot(atom.long(), atom_types) * node_mask.to(pos.device)
    return pos, one_hot, fc.long().unsqueeze(-1), et


def mol_process(one_hot, x, formal_charges, n_nodes, edge_types=None):
    """sampling.py:12-32 with ONE device→host copy per tensor instead of four per molecule."""
    atom_type = one_hot.argmax(-1).cpu()
    pos, fc = x.cpu(), formal_charges.reshape(formal_charges.shape[0], -1).long().cpu()
    et = None if edge_types is None else edge_types.cpu()
    mols = []
    for i in range(atom_type.shape[0]):
        n = int(n_nodes[i])
        if et is None:
            mols.append((pos[i, :n].clone(), atom_type[i, :n].clone()))
        else:
            mols.append((pos[i, :n].clone(), atom_type[i, :n].clone(), et[i, :n, :n].clone(), fc[i, :n].clone()))
    return mols


def build_masks(n_nodes, batch_size, device, max_n=None):
    """node_mask [B,N,1] / edge_mask [B*N*N,1] as sampling.py:432-439 (``max_n`` pads to a fixed width, e.g. so that
    every rank of a sharded run produces records of the same size)."""
    max_n = max(n_nodes) if max_n is None else max(int(max_n), max(n_nodes))
    node_mask = torch.zeros(batch_size, max_n)
    for i in range(batch_size):
        node_mask[i, 0:n_nodes[i]] = 1
    edge_mask = node_mask.unsqueeze(1) * node_mask.unsqueeze(2)
    edge_mask *= (~torch.eye(max_n, dtype=torch.bool)).unsqueeze(0)
    return node_mask.unsqueeze(2).to(device), edge_mask.view(batch_size * max_n * max_n, 1).to(device)


def initial_noise(batch_size, max_n, node_nf, edge_nf, node_mask, edge_mask):
    """z, edge_z of sampling.py:442-447 (same randn draw order and shapes on the model device)."""
    dev = node_mask.device
    zx = torch.randn((batch_size, max_n, 3), device=dev) * node_mask
    zx = zx - (zx.sum(1, keepdim=True) / node_mask.sum(1, keepdim=True)) * node_mask
    zh = torch.randn((batch_size, max_n, node_nf), device=dev) * node_mask
    ze = torch.tril(torch.randn((batch_size, edge_nf, max_n, max_n), device=dev), -1)
    ze = (ze + ze.transpose(-1, -2)).permute(0, 2, 3, 1) * edge_mask.reshape(batch_size, max_n, max_n, 1)
    return torch.cat([zx, zh], dim=2), ze.contiguous()


class MoleculeList(Sequence):
    """The ``processed_mols`` of a sharded run: behaves like the reference's list of per-molecule tuples ``(pos [n,3] f32,
    atom_type [n] i64, edge_type [n,n] f32, fc [n] i64)`` (sampling.py:17-28) - ``len``, indexing, slicing, iteration - but builds a
    tuple when it is asked for, from the four host tensors the one device->host copy produced.  Building 80 000 tuples eagerly costs
    seconds of Python on every rank of an 8-GPU run; consumers (``check_stability``, the RDKit metrics) walk the list once anyway.
    ``tolist()`` materialises a plain list."""

    def __init__(self, pos, atom, edge_type, fc, n_atoms):
        self.pos, self.atom, self.edge_type, self.fc, self.n_atoms = pos, atom, edge_type, fc, list(n_atoms)

    def __len__(self):
        return len(self.n_atoms)

    def _one(self, k):
        n = self.n_atoms[k]
        return (self.pos[k, :n].clone(), self.atom[k, :n].clone(), self.edge_type[k, :n, :n].clone(), self.fc[k, :n].clone())

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self._one(i) for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError("molecule index out of range")
        return self._one(k)

    def tolist(self):
        return self[:]


def _make_sampler(config, noise_scheduler, eps, temperature):
    if config.sampling.method != "ancestral":
        raise ValueError("Invalid sampling method!")
    if config.only_2D:
        raise ValueError("only_2D sampling is outside the MI355X hot path (every shipped config has only_2D=False)")
    time_steps = torch.linspace(noise_scheduler.T, eps, config.sampling.steps, device=config.device)
    return AncestralSampler(noise_scheduler, time_steps, config.model.pred_data, config.pred_edge, config.model.self_cond,
                            get_self_cond_fn(config), sampling_temperature=temperature)


def _assemble(ds, ids, version):
    """Context, n_nodes, ground-truth positions / molecules of the dataset items ``ids`` (sampling.py:391-420)."""
    if hasattr(ds, "batch"):                                   # PackedSpectraTable: resident in HBM, no per-item Python
        return ds.batch(ids, version)
    items = [ds[int(i)] for i in ids]
    n_nodes = [int(it.num_atom.item()) if hasattr(it.num_atom, "item") else int(it.num_atom) for it in items]
    stack = lambda name: torch.stack([getattr(it, name) for it in items])
    context = [stack("uv"), stack("ir"), stack("raman")] if version == "allspectra" else stack(version)
    return context, n_nodes, [it.pos for it in items], [getattr(it, "rdmol", None) for it in items]


def _ground_truth(ds, ids):
    """(gt_pos, gt_rdmols) of the dataset items ``ids`` without touching their spectra."""
    idl = [int(i) for i in ids]
    if hasattr(ds, "pos") and hasattr(ds, "rdmol") and isinstance(getattr(ds, "pos"), list):
        return [ds.pos[i] for i in idl], [ds.rdmol[i] for i in idl]
    items = [ds[i] for i in idl]
    return [it.pos for it in items], [getattr(it, "rdmol", None) for it in items]


def _num_atoms(ds, ids):
    if hasattr(ds, "num_atom") and torch.is_tensor(getattr(ds, "num_atom")):
        return ds.num_atom[torch.as_tensor(ids, dtype=torch.int64)].tolist()
    out = []
    for i in ids:
        na = ds[int(i)].num_atom
        out.append(int(na.item()) if hasattr(na, "item") else int(na))
    return out


def _sampling_fn_factory(config, sampler, batch_size, n_samples, inverse_scaler, ds, fixed_seed, top_k=1):
    """``sampling_fn(model) -> (processed_mols, gt_pos, gt_rdmols)`` of sampling.py:378-468 in two noise modes
    (``config.sampling.noise_source``):

    ``'philox'`` (default): sample slot k (= k-th entry of the seed-42 permutation; with ``top_k`` = K every spectrum owns
    K consecutive slots) has its own noise stream keyed on ``(config.sampling.seed, k)``.  Slots are dealt to the ranks of
    the initialised process group by size (``shard.assign_slots``), sampled in n-bucketed micro-batches of ``batch_size``
    with no collective in the loop, and the 1 248-byte result records are all-gathered once at the end (RCCL over xGMI),
    so every rank returns the full lists, in slot order, and the molecules are the same for any number of ranks and any
    batch size.
    ``'torch'``: the reference's own draw order - ``torch.randn`` of the padded batch shapes on the model device, full
    rounds of ``batch_size`` (sampling.py:442-447,611-612,623-624); single process; what the G11 golden replays."""
    from . import shard
    atom_types = config.data.atom_types
    include_fc = config.model.include_fc_charge
    node_nf = atom_types + int(include_fc)
    edge_nf = config.model.edge_ch
    version = config.data.spectra_version
    noise_source = getattr(config.sampling, "noise_source", "philox")
    if noise_source not in ("philox", "torch"):
        raise ValueError("config.sampling.noise_source must be 'philox' or 'torch'")
    if top_k < 1:
        raise ValueError("top_k must be >= 1")
    if not hip_post_process_supported(config):
        raise ValueError("ds_post_process implements the shipped scaling configuration only")

    def reference_order(model, eng, perm):
        device = eng.device
        processed, gt_pos, gt_mols = [], [], []
        for r in range(int(np.ceil(n_samples * top_k / batch_size))):
            ids = perm[r * batch_size:(r + 1) * batch_size]
            context, n_nodes, pos_r, mols_r = _assemble(ds, ids, version)
            gt_pos += pos_r
            gt_mols += mols_r
            bs = len(n_nodes)
            node_mask, edge_mask = build_masks(n_nodes, bs, device)
            max_n = node_mask.shape[1]
            z, edge_z = initial_noise(bs, max_n, node_nf, edge_nf, node_mask, edge_mask)
            x_node, x_edge = sampler.sampling(model, z, node_mask, edge_mask, edge_z, context)
            pos, one_hot, fc, edge_types = post_process(x_node, atom_types, include_fc, node_mask, inverse_scaler,
                                                        x_edge, edge_mask, config.data.compress_edge, engine=eng)
            processed += mol_process(one_hot, pos, fc, n_nodes, edge_types)
            print("Generate {}, Total {}.".format(len(processed), n_samples))
        return processed[:n_samples * top_k], gt_pos[:n_samples * top_k], gt_mols[:n_samples * top_k]

    class ShardedRun:
        """One sharded evaluation in resumable form: ``advance(n)`` runs the next n denoise iterations of this rank's
        micro-batches (opening a micro-batch with SpecFormer + initial noise and closing it with post-processing + record
        packing as it goes), ``finish()`` does the only collective of the path and returns the three lists.
        ``sampling_fn(model)`` is ``start(model)`` advanced to the end; ``bench.py`` times the same object in slices."""

        def __init__(self, model, eng, perm):
            self.model, self.eng, self.device = model, eng, eng.device
            self.rank, self.world = shard.world_info()
            self.slot_ds = perm[:n_samples].repeat_interleave(top_k)         # dataset item of every sample slot
            self.n_atoms = _num_atoms(ds, self.slot_ds.tolist())
            self.mine = shard.assign_slots(self.n_atoms, self.rank, self.world)
            self.seed = int(getattr(config.sampling, "seed", 42))
            self.batches = [self.mine[lo:lo + batch_size] for lo in range(0, self.mine.numel(), batch_size)]
            self.steps_per_batch = len(sampler.t_array)
            self.total_iters = len(self.batches) * self.steps_per_batch
            self.iters_done, self.next_batch, self.cur, self.recs = 0, 0, None, []

        def _open(self):
            slots = self.batches[self.next_batch]
            context, n_nodes, _, _ = _assemble(ds, self.slot_ds[slots], version)
            node_mask, edge_mask = build_masks(n_nodes, len(n_nodes), self.device)
            st = sampler.begin(self.model, None, node_mask, edge_mask, None, context, mol_ids=slots, seed=self.seed)
            self.cur = (st, node_mask, edge_mask)
            self.next_batch += 1

        def _close(self):
            st, node_mask, edge_mask = self.cur
            pos, one_hot, fc, edge_types = post_process(st.x_mean, atom_types, include_fc, node_mask, inverse_scaler,
                                                        st.edge_mean, edge_mask, config.data.compress_edge, engine=self.eng)
            self.recs.append(shard.pack_records_u8(pos, one_hot.argmax(-1), fc, edge_types))
            self.cur = None

        @property
        def done(self):
            return self.cur is None and self.next_batch >= len(self.batches)

        def advance(self, n_iters=None):
            """Run up to ``n_iters`` more denoise iterations (all remaining ones by default); True when every micro-batch of
            this rank is sampled and packed."""
            left = self.total_iters - self.iters_done if n_iters is None else int(n_iters)
            while left > 0 and not self.done:
                if self.cur is None:
                    self._open()
                st = self.cur[0]
                before = st.i
                finished = sampler.advance(st, left)
                left -= st.i - before
                self.iters_done += st.i - before
                if finished:
                    self._close()
            return self.done

        def finish(self, partial=False):
            """Gather the records of all ranks (the only collective) and unpack them, in slot order, on every rank.
            ``partial=True`` closes an in-flight micro-batch from its current state (bench warm-up of the closing code path)."""
            if partial:
                if self.cur is not None:
                    self._close()
                done_slots = [torch.cat(self.batches[:len(self.recs)])] if self.recs else []
                rec = torch.cat(self.recs) if self.recs else torch.zeros(0, shard.RECORD_BYTES, dtype=torch.uint8, device=self.device)
                cnt = torch.tensor([rec.shape[0]], dtype=torch.int64)
                counts = [int(c) for c in shard.all_gather_counts(cnt, self.device)]
                return shard.gather_records(rec, counts), done_slots
            if not self.done:
                raise RuntimeError("finish() before every micro-batch was sampled; call advance() to the end first")
            n_slots = self.slot_ds.numel()
            rec = torch.cat(self.recs) if self.recs else torch.zeros(0, shard.RECORD_BYTES, dtype=torch.uint8, device=self.device)
            by_slot = shard.gather_by_slot(rec, self.n_atoms)                        # the only collective of the path
            self.records_by_slot = by_slot
            pos, atom, fc, et = (t.cpu() for t in shard.unpack_records_u8(by_slot))     # ONE device->host copy per tensor
            processed = MoleculeList(pos, atom, et, fc, self.n_atoms[:n_slots])         # per-molecule tuples are built on access
            gt_pos, gt_mols = _ground_truth(ds, self.slot_ds.tolist())
            if self.rank == 0:
                print("Generate {}, Total {}.".format(len(processed), n_samples * top_k))
            return processed, gt_pos, gt_mols

    def _permutation(eng):
        if fixed_seed:
            torch.manual_seed(42)                              # sampling.py:387
        perm = torch.randperm(len(ds))
        if not fixed_seed:
            perm = shard.broadcast_from_rank0(perm, eng.device)      # an unseeded permutation must still be ONE permutation
        return perm

    def start(model):
        """The resumable form of ``sampling_fn(model)`` (philox noise source only): a ``ShardedRun``."""
        if noise_source != "philox":
            raise RuntimeError("start() drives the sharded (philox) path; noise_source='torch' is the monolithic reference order")
        model.eval()
        eng = _hip_model(model).engine()
        with torch.no_grad():
            return ShardedRun(model, eng, _permutation(eng))

    def sampling_fn(model):
        model.eval()
        eng = _hip_model(model).engine()
        with torch.no_grad():
            perm = _permutation(eng)
            if noise_source == "torch":
                if shard.world_info()[1] != 1:
                    raise RuntimeError("noise_source='torch' reproduces the reference's single-process draw order; "
                                       "multi-rank sampling needs noise_source='philox'")
                return reference_order(model, eng, perm)
            run = ShardedRun(model, eng, perm)
            run.advance()
            return run.finish()

    sampling_fn.start = start
    return sampling_fn


def get_cond_sampling_eval_fn(config, noise_scheduler, batch_size, n_samples, inverse_scaler, test_ds, eps=1e-3, top_k=1):
    """sampling.py:353-468 (fixed seed-42 permutation of the test set, eval temperature).  ``top_k`` = K draws K molecules
    per test spectrum (the reference's Top-K accuracy protocol, assets/3_performance.png panel c) in the same batched run:
    ``processed_mols[i*K:(i+1)*K]`` are the K candidates of the i-th spectrum."""
    sampler = _make_sampler(config, noise_scheduler, eps, config.eval.sampling_temperature)
    return _sampling_fn_factory(config, sampler, batch_size, n_samples, inverse_scaler, test_ds, fixed_seed=True, top_k=top_k)


def get_sampling_fn(config, noise_scheduler, batch_size, n_samples, inverse_scaler, val_ds, eps=1e-3):
    """sampling.py:148-248 (unseeded permutation of the validation set, temperature 1)."""
    sampler = _make_sampler(config, noise_scheduler, eps, 1.0)
    return _sampling_fn_factory(config, sampler, batch_size, n_samples, inverse_scaler, val_ds, fixed_seed=False)
