"""Model factory with the reference's surface (``models/utils.py:5-28``): ``register_model`` / ``create_model``.

``create_model`` returns the model wrapped so that ``state_dict()`` keys carry the ``module.`` prefix of the
reference's ``nn.DataParallel`` wrapper — reference checkpoints load with ``strict=True`` (``utils.py:17``) —
but there is no scatter/replicate/gather: scale-out is one process per GPU.
"""
from __future__ import annotations

import torch
from torch import nn

_MODELS = {}


def register_model(cls=None, *, name=None):
    """Decorator registering a model class under ``name`` (same contract as models/utils.py:5-21)."""

    def _register(c):
        local_name = c.__name__ if name is None else name
        if local_name in _MODELS:
            raise ValueError("Already registerd model")
        _MODELS[local_name] = c
        return c

    return _register if cls is None else _register(cls)


class SingleDeviceParallel(nn.Module):
    """``module.``-prefixed wrapper standing in for nn.DataParallel on a one-process-per-GPU deployment."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def get_model(name: str):
    if name not in _MODELS:
        from . import dmt  # noqa: F401  (the shipped model registers itself on import, as models/__init__.py does upstream)
    return _MODELS[name]


def create_model(config):
    model = get_model(config.model.name)(config)
    model = model.to(config.device)
    return SingleDeviceParallel(model)
