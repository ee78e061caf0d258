"""Row N3, training half: batch assembly for the training step (diffspectra_amd/train_data.py) against golden G16 - the reference's own
``EdgeComSpectraTransform`` + ``CollateSpectra`` (datasets/build_dataset.py:94-149,306-395) run on the same procedural raw molecules - and
end to end from a processed file in the PyG layout through ``qm9s_reader``."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from diffspectra_amd import train_data as TD
from tests.golden import cases


def _items():
    items = []
    for m in cases.raw_molecules():
        one_hot, edge = TD.edge_com_transform(m["atom_type"], m["edge_index"], m["edge_type"])
        items.append(SimpleNamespace(atom_one_hot=one_hot, edge_one_hot=edge, fc=m["fc"], pos=m["pos"], num_atom=m["num_atom"],
                                     uv=torch.log10(m["uv"] + 1), ir=torch.log10(m["ir"] + 1), raman=torch.log10(m["raman"] + 1)))
    return items


@pytest.mark.parametrize("tag,rot,tr", [("plain", False, False), ("aug", True, True)])
@pytest.mark.parametrize("version", ["allspectra", "ir"])
def test_transform_and_collate_match_the_reference(tag, rot, tr, version):
    g = cases.load_npz("g16_training_collate.npz")
    np.random.seed(123)
    torch.manual_seed(321)
    b = TD.collate_spectra(_items(), version, aug_rotation=rot, aug_translation=tr, aug_translation_scale=0.01)
    for k, v in b.items():
        if k == "context":
            for i, c in enumerate(v if isinstance(v, list) else [v]):
                assert torch.equal(c, g[f"{tag}_{version}_context{i}"]), (k, i)
        else:
            want = g[f"{tag}_{version}_{k}"]
            assert v.shape == want.shape and v.dtype == want.dtype, (k, v.shape, want.shape, v.dtype, want.dtype)
            assert torch.equal(v, want), k
    assert b["edge_one_hot"].shape[-1] == 2 and float(b["edge_one_hot"][..., 1].max()) <= 1.0
    if tag == "aug":
        assert not torch.equal(b["positions"], g[f"plain_{version}_positions"])          # the augmentation did act


def test_train_batches_from_processed_file(tmp_path):
    """ProcessedQM9S (PyG 2.x layout written without PyG) -> TrainBatches -> the collated dict; equal to collating the same raw
    molecules by hand, every molecule of the split exactly once per epoch."""
    from diffspectra_amd.qm9s_reader import ProcessedQM9S
    from tests.test_host_cpu import _write_processed_qm9s
    mols = cases.raw_molecules(n_atoms=(3, 7, 1, 12, 9, 5, 8, 4))
    perm = _write_processed_qm9s(str(tmp_path / "QM9S" / "processed"), mols, "pyg2")
    proc = ProcessedQM9S(str(tmp_path / "QM9S"))
    tb = TD.TrainBatches(proc, "first_train", batch_size=2, spectra_version="allspectra", shuffle=False, aug_rotation=False,
                         aug_translation=False)
    batches = list(tb)
    assert len(tb) == 1 and len(batches) == 1
    ids = perm[:2].tolist()
    items = []
    for j in ids:
        m = mols[j]
        one_hot, edge = TD.edge_com_transform(m["atom_type"], m["edge_index"], m["edge_type"])
        items.append(SimpleNamespace(atom_one_hot=one_hot, edge_one_hot=edge, fc=m["fc"], pos=m["pos"], num_atom=m["num_atom"],
                                     uv=torch.log10(m["uv"] + 1), ir=torch.log10(m["ir"] + 1), raman=torch.log10(m["raman"] + 1)))
    want = TD.collate_spectra(items, "allspectra")
    for k, v in want.items():
        if k == "context":
            assert all(torch.equal(a, b) for a, b in zip(v, batches[0][k]))
        else:
            assert torch.equal(v, batches[0][k]), k
    tb2 = TD.TrainBatches(proc, "test", batch_size=3, spectra_version="ir", shuffle=True)
    seen = sum(b["atom_mask"].shape[0] for b in tb2)
    assert seen == proc.split("test").numel() and isinstance(next(iter(tb2))["context"], torch.Tensor)
