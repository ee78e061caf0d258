"""CPU: host logic and the C-ABI surface (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from diffspectra_amd import engine as E, filler
from diffspectra_amd.config import qm9s_config
from tests.helpers import procedural_state_dict


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = E.load_library()                      # includes the struct-size self check
    hdr = open(E.HEADER_PATH).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = re.findall(r"^\s*(?:int|void)\s+(ds_\w+)\s*\(", hdr, flags=re.M)
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), name
    # the training C-ABI (include/diffspectra_train.h) lives in the same library
    from diffspectra_amd import train_engine as T
    train = T.train_exports()
    assert len(train) >= 30 and "dst_gemm" in train and "dst_adamw_ema" in train
    tl = T.load_train_library()
    for name in train:
        assert hasattr(tl, name), name


def test_training_surface_refuses_cpu():
    """Row N1 has no CPU path either: optimizer, loss function and graphs raise off the GPU."""
    from diffspectra_amd import losses as Lh, train_engine as T
    from diffspectra_amd.registry import create_model
    import diffspectra_amd.dmt  # noqa: F401
    cfg = qm9s_config("ir")
    model = create_model(cfg)
    with pytest.raises(RuntimeError):
        Lh.get_optimizer(cfg, model.parameters())
    with pytest.raises(RuntimeError):
        T.DmtTrainGraph({}, cfg, "cpu")
    cfg.model.dropout = 0.0
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    loss_fn = Lh.get_sde_graph_loss_fn(NoiseScheduleVP("cosine"), True, None, cfg)
    with pytest.raises(RuntimeError):
        loss_fn(model, {})
    q = Lh.Queue()
    q.add(3000)
    coef, allowed = Lh.clip_coefficient(18.0, q, 10.0)              # losses.py:33-44: allowed = min(1.5 mean + 2 std, max_grad)
    assert allowed == 10.0 and abs(coef - 10.0 / (18.0 + 1e-6)) < 1e-12 and q.items[0] == 10.0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(E, "_lib", None)
    monkeypatch.setattr(E, "LIB_PATH", "/nonexistent/libdiffspectra_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        E.load_library()


def test_model_refuses_cpu():
    import diffspectra_amd.dmt  # noqa: F401
    from diffspectra_amd.registry import create_model
    cfg = qm9s_config("ir", device="cpu")
    model = create_model(cfg)
    a = filler.synthetic_state([3, 4], "cpu.x")
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.zeros(2), a[0], a[2], a[3], context=torch.zeros(2, 1, 3501), edge_x=a[1], noise_level=torch.zeros(2),
              cond_x=None, cond_edge_x=None)


def test_pack_linear_matches_kernel_indexing():
    """pack_linear vs the index formula the kernels use (wave_mma float4 fetch / wp_at)."""
    g = torch.Generator().manual_seed(0)
    for N, K in ((6, 128), (252, 64), (64, 68), (1024, 17)):
        W = torch.randn(N, K, generator=g)
        p = E.pack_linear(W)
        Kp, Np = (K + 7) // 8 * 8, (N + 31) // 32 * 32
        assert p.numel() == Kp * Np
        k = torch.arange(Kp).view(-1, 1)
        n = torch.arange(Np).view(1, -1)
        idx = (((k >> 3) * 2 + ((k >> 2) & 1)) * Np + n) * 4 + (k & 3)
        B = p[idx]
        want = torch.zeros(Kp, Np)
        want[:K, :N] = W.t()
        assert torch.equal(B, want)


def test_packed_weights_cover_every_parameter():
    cfg, sd = procedural_state_dict("ir")
    flat, off = E.pack_dmt_weights(sd)
    assert len(off) == E.W_NUM_SLOTS and len(set(off)) == len(off)
    assert all(o % 64 == 0 for o in off)
    assert torch.isfinite(flat).all()
    # every DMT (non-SpecFormer) parameter value must be recoverable from the packed buffer: check the totals
    n_dmt = sum(v.numel() for k, v in sd.items() if not k.startswith("cond_encoder.") and k != "cond_lin.weight" and k != "cond_lin.bias")
    assert int((flat != 0).sum()) >= int(0.999 * n_dmt)


def test_layout_tables():
    node_mask, edge_mask = filler.masks_from_n_atoms([3, 1, 5, 2])
    L = E.Layout(node_mask, "cpu")
    assert (L.B, L.N, L.Nn, L.Pp, L.max_n) == (4, 5, 11, 3 + 0 + 10 + 1, 5)
    t = {k: v.numpy() for k, v in L.t.items()}
    assert t["node_off"].tolist() == [0, 3, 4, 9, 11] and t["pair_off"].tolist() == [0, 3, 3, 13, 14]
    for m in range(4):
        n = t["node_off"][m + 1] - t["node_off"][m]
        for p in range(t["pair_off"][m], t["pair_off"][m + 1]):
            a, b = t["pair_a"][p] - t["node_off"][m], t["pair_b"][p] - t["node_off"][m]
            assert 0 <= a < b < n and t["pair_mol"][p] == m
            assert p - t["pair_off"][m] == a * (2 * n - a - 1) // 2 + (b - a - 1)   # the formula the kernels use
    L.check_edge_mask(edge_mask)
    bad = edge_mask.clone(); bad[0] = 1   # a diagonal entry
    with pytest.raises(ValueError):
        L.check_edge_mask(bad)


def test_factory_and_checkpoint_contract():
    """module.-prefixed state dict, strict load, EMA-style copy into parameters() order."""
    import diffspectra_amd.dmt  # noqa: F401
    from diffspectra_amd.registry import create_model, register_model
    cfg = qm9s_config("allspectra", device="cpu")
    model = create_model(cfg)
    sd = model.state_dict()
    assert len(sd) == 435 and all(k.startswith("module.") for k in sd)
    trainable = [p for p in model.parameters() if p.requires_grad]
    assert len(trainable) == 414 and sum(p.numel() for p in model.parameters()) == 39630619
    model.load_state_dict(filler.fill_state_dict(sd), strict=True)
    with pytest.raises(ValueError):
        register_model(name="DMT")(type("X", (), {}))


def test_pretrained_specformer_key_mapping():
    """dmt.py:268-303: Lightning-style keys → cond_encoder.*"""
    import diffspectra_amd.dmt as D
    cfg = qm9s_config("ir", device="cpu")
    m = D.DMT(cfg)
    src = {}
    for k, v in m.cond_encoder.state_dict().items():
        key = f"model.representation_model.{k}"
        src[key] = torch.full_like(v, 0.25) if v.dtype.is_floating_point else v
    n = m.load_pretrained_specformer_state(src)
    assert n == len(m.cond_encoder.state_dict())
    assert float(m.cond_encoder.head.linear.weight[0, 0]) == 0.25


def test_sampler_surface():
    from diffspectra_amd import sampling as S
    from diffspectra_amd.noise_schedule import NoiseScheduleVP
    cfg = qm9s_config("ir", steps=7)
    ns = NoiseScheduleVP("cosine", continuous_beta_0=0.1, continuous_beta_1=20.0)
    assert ns.T == 0.9946
    smp = S._make_sampler(cfg, ns, 1e-3, 1.0)
    assert smp.coefficient_table().shape == (7, 4)
    with pytest.raises(TypeError):
        smp.sampling(lambda *a, **k: None, torch.zeros(1, 2, 9), torch.ones(1, 2, 1), torch.ones(4, 1), torch.zeros(1, 2, 2, 2), None)
    bad = cfg.clone(); bad.sampling.method = "ode"
    with pytest.raises(ValueError, match="Invalid sampling method"):
        S.get_cond_sampling_eval_fn(bad, ns, 4, 4, None, [])
    nm, em = S.build_masks([2, 3], 2, "cpu")
    want_nm, want_em = filler.masks_from_n_atoms([2, 3])
    assert torch.equal(nm, want_nm) and torch.equal(em, want_em)


def test_packed_spectra_table():
    """N3 (conditioning side): the HBM-resident table hands a round exactly what per-item assembly would."""
    from types import SimpleNamespace
    from diffspectra_amd.dataset_pack import PackedSpectraTable
    M = 9
    raw = [filler.uniform(f"pack.{n}", (M, 1, L)) * 50.0 for n, L in zip(("uv", "ir", "raman"), (701, 3501, 3501))]
    n_atoms = filler.sample_n_atoms(M, seed=4).tolist()
    ds = [SimpleNamespace(uv=raw[0][i], ir=raw[1][i], raman=raw[2][i], num_atom=torch.tensor(n_atoms[i]),
                          pos=torch.full((n_atoms[i], 3), float(i)), rdmol=f"m{i}") for i in range(M)]
    tab = PackedSpectraTable.from_dataset(ds, "allspectra", normalize=True)
    assert len(tab) == M
    ids = torch.tensor([7, 2, 2, 5])
    ctx, n_nodes, pos, mols = tab.batch(ids, "allspectra")
    assert n_nodes == [n_atoms[i] for i in ids.tolist()] and mols == ["m7", "m2", "m2", "m5"]
    assert all(torch.equal(p, ds[i].pos) for p, i in zip(pos, ids.tolist()))
    for k in range(3):      # log10(x + 1) of the reference transform (build_dataset.py:141-148), stacked like sampling.py:404-420
        want = torch.stack([torch.log10(raw[k][i] + 1) for i in ids.tolist()])
        assert ctx[k].shape == want.shape and torch.equal(ctx[k], want)
    it = tab[5]             # dataset surface kept for code written against the reference's ds[i]
    assert int(it.num_atom) == n_atoms[5] and it.rdmol == "m5" and torch.equal(it.ir, torch.log10(raw[1][5] + 1))
    ir_only = PackedSpectraTable.from_dataset(ds, "ir")
    c1, n1, _, _ = ir_only.batch([0, 8], "ir")
    assert torch.equal(c1, torch.stack([raw[1][0], raw[1][8]])) and n1 == [n_atoms[0], n_atoms[8]]
    with pytest.raises(ValueError):
        ir_only.batch([0], "allspectra")
    with pytest.raises(ValueError):
        PackedSpectraTable([raw[0][:, :, :10], None, None], torch.tensor(n_atoms))


def _write_processed_qm9s(proc_dir, mols, layout):
    """Write data_qm9_allspectra.pt + split file the way InMemoryDataset.collate + torch.save do, WITHOUT PyG / RDKit: classes
    named like theirs live in throw-away modules while pickling (PyG 2.4.0: Data.__dict__ = {_edge_attr_cls, _tensor_attr_cls,
    _store}, GlobalStorage.__getstate__ = its __dict__ with _parent resolved; PyG 1.x: the tensors directly in Data.__dict__)."""
    import sys
    import types
    names = ["torch_geometric", "torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage",
             "rdkit", "rdkit.Chem", "rdkit.Chem.rdchem"]
    mods = {n: types.ModuleType(n) for n in names}

    def cls(module, name):
        c = type(name, (), {"__module__": module})
        setattr(mods[module], name, c)
        return c
    Data, Storage, Mol = cls("torch_geometric.data.data", "Data"), cls("torch_geometric.data.storage", "GlobalStorage"), cls("rdkit.Chem.rdchem", "Mol")
    EdgeAttr, TensorAttr = cls("torch_geometric.data.data", "DataEdgeAttr"), cls("torch_geometric.data.data", "DataTensorAttr")
    cat = lambda k, dim=0: torch.cat([m[k] for m in mols], dim)
    cum = lambda sizes: torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)
    M = len(mols)
    rdmols = []
    for i in range(M):
        r = Mol()
        r.tag = f"mol{i}"
        rdmols.append(r)
    mapping = {"atom_type": cat("atom_type"), "pos": cat("pos"), "edge_index": cat("edge_index", 1), "edge_type": cat("edge_type"),
               "uv": cat("uv"), "ir": cat("ir"), "raman": cat("raman"),
               "num_atom": torch.tensor([m["pos"].shape[0] for m in mols]), "idx": torch.arange(M), "rdmol": rdmols}
    n = [m["pos"].shape[0] for m in mols]
    ne = [m["edge_type"].shape[0] for m in mols]
    one = torch.arange(M + 1)
    slices = {"atom_type": cum(n), "pos": cum(n), "edge_index": cum(ne), "edge_type": cum(ne), "uv": one, "ir": one, "raman": one,
              "num_atom": one, "idx": one, "rdmol": one}
    if "fc" in mols[0]:                               # per-atom formal charges (qm9s_dataset.py:267)
        mapping["fc"], slices["fc"] = cat("fc"), cum(n)
    data = Data()
    if layout == "pyg2":
        st = Storage()
        st.__dict__.update({"_key": None, "_mapping": mapping, "_parent": data})
        data.__dict__.update({"_edge_attr_cls": EdgeAttr, "_tensor_attr_cls": TensorAttr, "_store": st})
    else:
        data.__dict__.update(mapping)
    os.makedirs(proc_dir, exist_ok=True)
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update(mods)
    try:
        torch.save((data, slices), os.path.join(proc_dir, "data_qm9_allspectra.pt"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    perm = np.random.RandomState(0).permutation(M)
    torch.save({"first_train": perm[:2], "second_train": perm[2:4], "valid": perm[4:6], "test": perm[6:]},
               os.path.join(proc_dir, "split_dict_diffspectra_qm9.pt"))
    return perm


@pytest.mark.parametrize("layout", ["pyg2", "pyg1"])
def test_processed_qm9s_reader(tmp_path, layout):
    """N3 (on-disk side): data_qm9_allspectra.pt + split_dict_diffspectra_qm9.pt -> the sampler's table, item for item what
    ``dataset.index_select(split['test'])[i]`` + EdgeComSpectraTransform give (qm9s_dataset.py:153,306-312,357-361;
    build_dataset.py:36-42,141-148), read without PyG or RDKit."""
    from diffspectra_amd.qm9s_reader import ProcessedQM9S
    M = 11
    n_atoms = filler.sample_n_atoms(M, seed=9).tolist()
    mols = []
    for i, n in enumerate(n_atoms):
        ne = 2 * max(n - 1, 0)
        mols.append({"atom_type": torch.randint(0, 5, (n,)), "pos": filler.uniform(f"rd.pos{i}", (n, 3)),
                     "edge_index": torch.randint(0, max(n, 1), (2, ne)), "edge_type": torch.randint(1, 4, (ne,)),
                     "uv": filler.uniform(f"rd.uv{i}", (1, 701)).abs() * 30, "ir": filler.uniform(f"rd.ir{i}", (1, 3501)).abs() * 30,
                     "raman": filler.uniform(f"rd.ra{i}", (1, 3501)).abs() * 30})
    perm = _write_processed_qm9s(str(tmp_path / "processed"), mols, layout)
    ds = ProcessedQM9S(str(tmp_path))                       # the dataset root, as config.data.root
    assert len(ds) == M and sorted(ds.splits) == ["first_train", "second_train", "test", "valid"]
    test_ids = perm[6:].tolist()
    tab = ds.packed_table("allspectra", split="test")
    assert len(tab) == len(test_ids)
    for i, j in enumerate(test_ids):
        it = tab[i]
        assert int(it.num_atom) == n_atoms[j] and torch.equal(it.pos, mols[j]["pos"])
        for name in ("uv", "ir", "raman"):
            assert torch.equal(getattr(it, name), torch.log10(mols[j][name] + 1))
        assert type(it.rdmol).__name__ == "Mol" and it.rdmol._state["tag"] == f"mol{j}"      # opaque, but the right molecule
        assert torch.equal(ds.item_field("edge_index", j), mols[j]["edge_index"])           # concatenated along dim 1
        assert torch.equal(ds.item_field("atom_type", j), mols[j]["atom_type"])
    ctx, n_nodes, pos, _ = tab.batch([2, 0], "allspectra")
    assert n_nodes == [n_atoms[test_ids[2]], n_atoms[test_ids[0]]] and ctx[1].shape == (2, 1, 3501)
    assert torch.equal(ctx[0][1, 0], torch.log10(mols[test_ids[0]]["uv"][0] + 1))
    raw = ds.packed_table("ir", split="valid", normalize=False)
    assert torch.equal(raw[1].ir, mols[perm[5]]["ir"]) and raw.spectra[0] is None
    assert len(ds.packed_table("ir", split=None)) == M
    with pytest.raises(KeyError):
        ds.split("train")
    with pytest.raises(FileNotFoundError):
        ProcessedQM9S(str(tmp_path / "nowhere"))


def test_batched_stability_needs_the_hip_engine():
    """N4 has no PyTorch implementation: without an engine the batched stability check refuses."""
    from diffspectra_amd.stability import check_stability_batch
    with pytest.raises(RuntimeError):
        check_stability_batch(torch.zeros(1, 2, 3), torch.zeros(1, 2, dtype=torch.long), torch.ones(1, 2))


def test_build_reports_on_stderr_only(tmp_path, capsys):
    """bench.py calls build() and its stdout must stay ONE JSON line: a rebuild on the benchmark box (sources newer than the
    library) may only talk on stderr."""
    import __graft_entry__ as g
    obj = str(tmp_path / "x.o")
    src = str(tmp_path / "x.hip")
    open(src, "w").write("// empty\n")
    g._compile("/bin/true", src, obj, ["-O3"])
    cap = capsys.readouterr()
    assert cap.out == "" and "[build]" in cap.err
    assert open(obj + ".flags").read() == "-O3"
    with pytest.raises(FileNotFoundError):                   # a dependency that does not exist is an error, not an empty digest
        g._digest([str(tmp_path / "missing.h")])
