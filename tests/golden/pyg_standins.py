"""Stand-ins for the five un-vendored third-party symbols the reference's hot path imports.

Used ONLY by ``generate_golden.py`` in the build container so the reference's own
``models/layers.py``, ``models/dmt.py`` and ``sampling.py`` can be imported unmodified
(SURVEY §8c).  They restate the documented behaviour of the pinned versions
(``env.sh:4-6``: torch_geometric==2.4.0, torch_scatter for torch-2.3.0):

* ``MessagePassing.propagate`` (flow ``source_to_target``, ``aggr='add'``, ``node_dim=0``):
  ``j = edge_index[0]`` (source), ``i = edge_index[1]`` (target); message args named ``*_i`` /
  ``*_j`` are ``index_select`` gathers of the same-named kwarg; ``index = i``, ``ptr = None``,
  ``size_i`` = number of nodes; the result is the scatter-sum of messages over ``i``.
* ``utils.softmax(src, index, ptr, num_nodes)``: ``exp(src - segmax[index]) / (segsum[index] + 1e-16)``.
* ``utils.dense_to_sparse(adj[B,N,N])``: nonzeros in row-major order → ``[b*N+i ; b*N+j]``.
* ``torch_scatter.scatter(src, index, dim, reduce='add', dim_size)``: ``index_add_``.
* ``torch_sparse.sample``: imported by ``sampling.py:6`` and never called.
"""
from __future__ import annotations

import inspect
import sys
import types
from typing import Optional, Tuple, Union

import torch
from torch import Tensor


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr: str = "add", flow: str = "source_to_target", node_dim: int = -2, **kwargs):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target" and node_dim == 0
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        src, tgt = edge_index[0], edge_index[1]
        num_nodes = None
        for v in kwargs.values():
            if isinstance(v, Tensor) and v.dim() >= 1:
                num_nodes = v.size(0)
                break
        args = {}
        for name in inspect.signature(self.message).parameters:
            if name == "index":
                args[name] = tgt
            elif name == "ptr":
                args[name] = None
            elif name == "size_i":
                args[name] = num_nodes
            elif name.endswith("_i"):
                args[name] = kwargs[name[:-2]].index_select(0, tgt)
            elif name.endswith("_j"):
                args[name] = kwargs[name[:-2]].index_select(0, src)
            else:
                args[name] = kwargs[name]
        # node count must come from a node-level tensor (the *_i/_j sources), not an edge tensor
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_i") or name.endswith("_j"):
                num_nodes = kwargs[name[:-2]].size(0)
                break
        args["size_i"] = num_nodes
        msg = self.message(**args)
        out = torch.zeros((num_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        return out.index_add_(0, tgt, msg)


def softmax(src: Tensor, index: Tensor, ptr=None, num_nodes: Optional[int] = None, dim: int = 0) -> Tensor:
    assert dim == 0 and ptr is None
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    mx = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
    mx = mx.scatter_reduce(0, idx, src.detach(), "amax", include_self=True)
    out = (src - mx.index_select(0, index)).exp()
    den = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, out) + 1e-16
    return out / den.index_select(0, index)


def dense_to_sparse(adj: Tensor):
    assert adj.dim() == 3
    b, i, j = adj.nonzero(as_tuple=True)
    n = adj.size(1)
    return torch.stack([b * n + i, b * n + j], dim=0), adj[b, i, j]


def scatter(src: Tensor, index: Tensor, dim: int = -1, out=None, dim_size: Optional[int] = None,
            reduce: str = "sum") -> Tensor:
    assert dim == 0 and reduce in ("add", "sum")
    n = int(index.max()) + 1 if dim_size is None else dim_size
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add_(0, index, src)


def install() -> None:
    """Register the stand-in modules under the third-party import names."""
    tg = types.ModuleType("torch_geometric")
    tg_typing = types.ModuleType("torch_geometric.typing")
    tg_typing.OptTensor = Optional[Tensor]
    tg_typing.PairTensor = Tuple[Tensor, Tensor]
    tg_typing.Adj = Union[Tensor]
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_conv = types.ModuleType("torch_geometric.nn.conv")
    tg_conv.MessagePassing = MessagePassing
    tg_nn.conv = tg_conv
    tg_utils = types.ModuleType("torch_geometric.utils")
    tg_utils.softmax = softmax
    tg_utils.dense_to_sparse = dense_to_sparse
    tg.typing, tg.nn, tg.utils = tg_typing, tg_nn, tg_utils
    ts = types.ModuleType("torch_scatter")
    ts.scatter = scatter
    tsp = types.ModuleType("torch_sparse")
    tsp.sample = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torch_sparse.sample stand-in: never called"))
    for name, mod in {"torch_geometric": tg, "torch_geometric.typing": tg_typing, "torch_geometric.nn": tg_nn,
                      "torch_geometric.nn.conv": tg_conv, "torch_geometric.utils": tg_utils,
                      "torch_scatter": ts, "torch_sparse": tsp}.items():
        sys.modules[name] = mod
