"""Reader of the reference's processed QM9S files, straight into the device-resident table.

The reference reads ``<root>/processed/data_qm9_allspectra.pt`` with ``self.data, self.slices = torch.load(...)``
(``datasets/qm9s_dataset.py:153,166-174``): a PyG ``InMemoryDataset.collate`` pair - one ``Data`` object whose tensors
are the concatenation of every molecule's, and ``slices[name]``, the cumulative offsets of the molecules along each
tensor's first dimension.  ``split_dict_diffspectra_qm9.pt`` holds the index arrays ``first_train / second_train /
valid / test`` (``qm9s_dataset.py:306-312``, ``build_dataset.py:36-42``).  Item ``i`` of a split is then
``data[name][slices[name][j] : slices[name][j + 1]]`` with ``j = split[i]`` (``qm9s_dataset.py:357-361``), passed through
``EdgeComSpectraTransform`` whose only effect on what the SAMPLER reads is ``log10(x + 1)`` on the spectra
(``build_dataset.py:141-148``).

Unpickling that file normally imports ``torch_geometric`` (the ``Data`` / ``GlobalStorage`` classes) and ``rdkit``
(the per-molecule ``rdmol`` objects).  Neither is needed to get at the tensors, so the unpickler below substitutes a
plain attribute bag for every class of a module that is not importable; PyG 2.x state layout (``Data.__dict__ ->
_store -> _mapping``) and the 1.x layout (tensors directly in ``Data.__dict__``) are both understood.  When RDKit IS
installed the ``rdmol`` entries unpickle as real molecules and the evaluation metrics can use them; otherwise they stay
opaque placeholders and only the spectra-conditioned sampling (which never looks inside them) is available.

Format restated from PyG 2.4.0 (``torch_geometric/data/{data,storage,collate}.py``); there is no PyG in this image, so the
reader is exercised on files written in that layout by ``tests/test_host_cpu.py`` - unpinned against a real download.
"""
from __future__ import annotations

import importlib
import os
import pickle
from typing import Dict, List, Optional

import torch

from .config import SPECTRUM_LENGTHS, used_spectra
from .dataset_pack import PackedSpectraTable

DATA_FILE = "data_qm9_allspectra.pt"            # qm9s_dataset.py:170-172 (every spectra_version shares it)
SPLIT_FILE = "split_dict_diffspectra_qm9.pt"    # qm9s_dataset.py:311
_NAMES = ("uv", "ir", "raman")


class Opaque:
    """Stand-in for an instance of a class whose module is not importable (PyG containers, RDKit molecules)."""

    def __init__(self, *args, **kwargs):
        self._args, self._kwargs = args, kwargs

    def __setstate__(self, state):
        self._state = state

    def __call__(self, *args, **kwargs):     # objects rebuilt through a factory function of a missing module
        return Opaque(*args, **kwargs)


def _opaque_class(module: str, name: str):
    return type(name, (Opaque,), {"__module__": module, "_opaque_origin": f"{module}.{name}"})


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            importlib.import_module(module.split(".")[0])
        except ImportError:
            return _opaque_class(module, name)
        return super().find_class(module, name)


class _tolerant_pickle:     # the ``pickle_module`` protocol torch.load expects
    Unpickler = _TolerantUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def _load(path: str):
    return torch.load(path, map_location="cpu", pickle_module=_tolerant_pickle, weights_only=False)


def _mapping_of(data) -> Dict[str, object]:
    """The ``name -> value`` dict of a collated ``Data`` object, whichever PyG generation wrote it."""
    if isinstance(data, dict):
        return data
    state = getattr(data, "_state", None)
    if state is None:
        state = getattr(data, "__dict__", {})
    if isinstance(state, tuple):                      # (dict, slots) form of __getstate__
        state = next((s for s in state if isinstance(s, dict)), {})
    store = state.get("_store")
    if store is not None:                             # PyG 2.x: Data.__dict__['_store'] is a GlobalStorage
        sstate = getattr(store, "_state", None) or getattr(store, "__dict__", {})
        mapping = sstate.get("_mapping")
        if mapping is None:
            raise ValueError("unrecognised PyG storage layout: no '_mapping' in the GlobalStorage state")
        return dict(mapping)
    return {k: v for k, v in state.items() if not k.startswith("_")}   # PyG 1.x: attributes directly


class ProcessedQM9S:
    """``data`` + ``slices`` of the processed file, and the conditional-generation split."""

    def __init__(self, root: str):
        proc = root if os.path.exists(os.path.join(root, DATA_FILE)) else os.path.join(root, "processed")
        path = os.path.join(proc, DATA_FILE)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{DATA_FILE} not found under {root} (expected <root>/processed/, qm9s_dataset.py:153)")
        loaded = _load(path)
        if not (isinstance(loaded, (tuple, list)) and len(loaded) >= 2):
            raise ValueError(f"{path}: expected the (data, slices) pair of InMemoryDataset.collate")
        self.fields = _mapping_of(loaded[0])
        self.slices = {k: torch.as_tensor(v, dtype=torch.int64) for k, v in _mapping_of(loaded[1]).items()}
        if "num_atom" not in self.fields or "num_atom" not in self.slices:
            raise ValueError(f"{path}: no 'num_atom' field (qm9s_dataset.py:263)")
        self.num_molecules = self.slices["num_atom"].numel() - 1
        split_path = os.path.join(proc, SPLIT_FILE)
        self.splits: Optional[Dict[str, torch.Tensor]] = None
        if os.path.exists(split_path):
            self.splits = {k: torch.as_tensor(v, dtype=torch.int64).reshape(-1) for k, v in _load(split_path).items()}

    def __len__(self) -> int:
        return self.num_molecules

    def split(self, name: str) -> torch.Tensor:
        if self.splits is None:
            raise FileNotFoundError(f"{SPLIT_FILE} not found next to {DATA_FILE} (qm9s_dataset.py:306-312)")
        if name not in self.splits:
            raise KeyError(f"split '{name}' not in {sorted(self.splits)}")
        return self.splits[name]

    def item_field(self, name: str, j: int):
        """Molecule ``j``'s slice of one field: what ``InMemoryDataset.get(j)`` puts on the item."""
        v, s = self.fields[name], self.slices[name]
        a, b = int(s[j]), int(s[j + 1])
        if torch.is_tensor(v):
            if name == "edge_index":                  # concatenated along the LAST dimension (PyG's __cat_dim__)
                return v[:, a:b]
            return v[a:b]
        return v[a] if b - a == 1 else v[a:b]         # python lists (rdmol)

    def packed_table(self, spectra_version: str, split: Optional[str] = "test", device="cpu",
                     normalize: bool = True) -> PackedSpectraTable:
        """The sampler's conditioning table for one split, in split order (``test_ds[i]`` == table[i]).

        Every molecule contributes ONE row of each spectrum (``[1, L]`` items, ``sampling.py:399-411``), so the rows of a
        split are one ``index_select`` of the collated tensor - no per-molecule Python.  ``normalize`` applies the
        reference transform's ``log10(x + 1)`` (``build_dataset.py:141-148``; ``config.data.use_normalize``).
        """
        ids = self.split(split) if split is not None else torch.arange(self.num_molecules)
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.num_molecules):
            raise IndexError("split indices outside the dataset")
        spectra: List[Optional[torch.Tensor]] = [None, None, None]
        for k in used_spectra(spectra_version):
            name, L = _NAMES[k], SPECTRUM_LENGTHS[k]
            if name not in self.fields:
                raise KeyError(f"the processed file holds no '{name}' spectra")
            t, s = self.fields[name], self.slices[name]
            if not torch.equal(s, torch.arange(self.num_molecules + 1)) or t.shape[-1] != L:
                raise ValueError(f"'{name}': expected one [1, {L}] row per molecule, got tensor {tuple(t.shape)}")
            rows = t.reshape(self.num_molecules, L).index_select(0, ids).to(torch.float32)
            spectra[k] = torch.log10(rows + 1) if normalize else rows
        na = self.fields["num_atom"]
        na = na if torch.is_tensor(na) else torch.tensor(list(na), dtype=torch.int64)
        idl = ids.tolist()
        pos = [self.item_field("pos", j) for j in idl] if "pos" in self.fields else None
        rdmol = [self.item_field("rdmol", j) for j in idl] if "rdmol" in self.fields else None
        return PackedSpectraTable(spectra, na.reshape(-1)[ids], pos, rdmol, device)
