"""Training-side batch assembly from the reference's processed QM9S files (training half).

The reference feeds ``loss_fn`` through PyG: ``QM9SDataset[j]`` -> ``EdgeComSpectraTransform`` (``datasets/build_dataset.py:94-149``: one-hot
atom types, dense ``[n, n, 2]`` edge features = [exists, bond order / 3] with aromatic (type 4) folded to 0, ``log10(x + 1)`` spectra) ->
``DataLoader(collate_fn=CollateSpectra(...))`` (``:306-395``: padding to the batch's largest molecule, node / edge masks, optional random
rotation + translation of the positions).  Here the same three steps read the collated tensors of ``qm9s_reader.ProcessedQM9S`` directly - no
PyG, no per-item transform objects - and yield the dict ``losses.get_step_fn`` / ``loss_fn`` consume (the format of ``CollateSpectra.__call__``).
Pinned against the reference's own transform + collate classes by golden G16 (tests/golden/generate_golden.py); the on-disk file layout
itself stays restated from PyG 2.4.0 (qm9s_reader.py).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Iterator, List, Optional, Sequence

import torch

from .config import used_spectra
from .qm9s_reader import ProcessedQM9S

_NAMES = ("uv", "ir", "raman")
QM9_ATOM_TYPES = (0, 1, 2, 3, 4)        # dataset_info['atom_encoder'].values() (datasets_config.py:3): H, C, N, O, F


def edge_com_transform(atom_type: torch.Tensor, edge_index: torch.Tensor, edge_type: torch.Tensor, atom_type_list: Sequence[int] = QM9_ATOM_TYPES,
                       include_aromatic: bool = False):
    """``EdgeComSpectraTransform.__call__`` without the spectra part (build_dataset.py:108-138): (atom_one_hot [n, T], edge_one_hot [n, n, 2 (+1)])."""
    n = atom_type.numel()
    one_hot = (atom_type.unsqueeze(-1) == torch.tensor(list(atom_type_list)).unsqueeze(0)).float()
    bond = edge_type.clone()
    bond[bond == 4] = 0
    feat = [bond / 3.0]
    if include_aromatic:
        feat.append((edge_type == 4).float())
    feat = torch.stack(feat, dim=-1)
    dense = torch.zeros((n * n, feat.size(-1)))
    idx = (edge_index[0] * n + edge_index[1]).unsqueeze(-1).expand(feat.size())
    dense.scatter_add_(0, idx, feat)
    dense = dense.reshape(n, n, feat.size(-1))
    exist = (dense.sum(dim=-1, keepdim=True) != 0).float()
    return one_hot, torch.cat([exist, dense], dim=-1)


def collate_spectra(items: List[SimpleNamespace], spectra_version: str = "allspectra", aug_rotation: bool = False, aug_translation: bool = False,
                    aug_translation_scale: float = 0.01):
    """``CollateSpectra.__call__`` (build_dataset.py:357-395): items with ``atom_one_hot, edge_one_hot, fc, pos, num_atom, uv, ir, raman``."""
    num_atoms = [int(it.num_atom) for it in items]
    N, B = max(num_atoms), len(items)
    T, E = items[0].atom_one_hot.shape[1], items[0].edge_one_hot.shape[2]
    atom_one_hot, positions = torch.zeros(B, N, T), torch.zeros(B, N, 3, dtype=items[0].pos.dtype)
    fc, edge_one_hot = torch.zeros(B, N, 1, dtype=items[0].fc.dtype), torch.zeros(B, N, N, E)
    node_mask = torch.zeros(B, N, dtype=atom_one_hot.dtype)
    for b, (it, n) in enumerate(zip(items, num_atoms)):
        atom_one_hot[b, :n], positions[b, :n], fc[b, :n, 0] = it.atom_one_hot, it.pos, it.fc
        edge_one_hot[b, :n, :n] = it.edge_one_hot
        node_mask[b, :n] = 1.0
    edge_mask = node_mask.unsqueeze(1) * node_mask.unsqueeze(2)
    edge_mask = (edge_mask * (~torch.eye(N, dtype=torch.bool)).unsqueeze(0)).reshape(-1, 1)
    stack = lambda name: torch.stack([getattr(it, name) for it in items], dim=0)
    context = [stack("uv"), stack("ir"), stack("raman")] if spectra_version == "allspectra" else stack(spectra_version)
    mask3 = node_mask.unsqueeze(-1)
    if aug_rotation:                                         # build_dataset.py:323-331: scipy's uniform random rotations, per molecule
        from scipy.spatial.transform import Rotation
        rot = Rotation.random(B)
        p = positions.numpy()
        for b in range(B):
            p[b] = rot[b].apply(p[b])
        positions = torch.from_numpy(p).to(positions.dtype) * mask3
    if aug_translation:                                      # :333-338
        positions = (positions + aug_translation_scale * torch.randn(B, 1, 3, dtype=positions.dtype).repeat(1, N, 1)) * mask3
    return dict(atom_one_hot=atom_one_hot, edge_one_hot=edge_one_hot, positions=positions, formal_charges=fc, atom_mask=node_mask,
                edge_mask=edge_mask, context=context)


class TrainBatches:
    """Iterable over collated training batches of one split of the processed file (``run_lib.diffspectra_train`` feeds ``train_step_fn`` from
    ``DataLoader(train_ds, batch_size, shuffle=True, collate_fn=CollateSpectra(...))``, build_dataset.py:84-85).  ``device``: where the batch
    tensors are placed (the loss function accepts host or device tensors)."""

    def __init__(self, proc: ProcessedQM9S, split: str, batch_size: int, spectra_version: str = "allspectra", shuffle: bool = True,
                 drop_last: bool = False, normalize: bool = True, aug_rotation: bool = True, aug_translation: bool = True,
                 aug_translation_scale: float = 0.01, include_aromatic: bool = False, device="cpu"):
        self.proc, self.ids = proc, proc.split(split)
        self.batch_size, self.version, self.shuffle, self.drop_last = batch_size, spectra_version, shuffle, drop_last
        self.normalize, self.aug = normalize, (aug_rotation, aug_translation, aug_translation_scale)
        self.include_aromatic, self.device = include_aromatic, torch.device(device)
        for f in ("atom_type", "edge_index", "edge_type", "fc", "pos"):
            if f not in proc.fields:
                raise KeyError(f"the processed file holds no '{f}' field (qm9s_dataset.py:267-268)")

    def __len__(self) -> int:
        n = self.ids.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def item(self, j: int) -> SimpleNamespace:
        """What ``QM9SDataset[j]`` + ``EdgeComSpectraTransform`` hand to the collate function."""
        p = self.proc
        at, ei, et = p.item_field("atom_type", j), p.item_field("edge_index", j), p.item_field("edge_type", j)
        one_hot, edge_one_hot = edge_com_transform(at, ei, et, include_aromatic=self.include_aromatic)
        na = p.item_field("num_atom", j)
        it = SimpleNamespace(atom_one_hot=one_hot, edge_one_hot=edge_one_hot, fc=p.item_field("fc", j), pos=p.item_field("pos", j),
                             num_atom=int(na.reshape(-1)[0]) if torch.is_tensor(na) else int(na))
        for k in used_spectra(self.version):
            s = p.item_field(_NAMES[k], j).to(torch.float32)
            setattr(it, _NAMES[k], torch.log10(s + 1) if self.normalize else s)       # build_dataset.py:141-148
        for name in _NAMES:
            if not hasattr(it, name):
                setattr(it, name, torch.zeros(1, 1))
        return it

    def __iter__(self) -> Iterator[dict]:
        order = self.ids[torch.randperm(self.ids.numel())] if self.shuffle else self.ids
        for lo in range(0, order.numel(), self.batch_size):
            chunk = order[lo:lo + self.batch_size].tolist()
            if self.drop_last and len(chunk) < self.batch_size:
                return
            batch = collate_spectra([self.item(j) for j in chunk], self.version, *self.aug)
            yield {k: ([t.to(self.device) for t in v] if isinstance(v, list) else v.to(self.device)) for k, v in batch.items()}
