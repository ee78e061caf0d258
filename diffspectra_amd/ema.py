"""Exponential moving average of the trainable parameters - the surface of reference ``models/ema.py``.

Same constructor (``parameters, decay, use_num_updates=True``), same state layout (``decay``, ``num_updates``,
``shadow_params`` = list of the trainable tensors in ``parameters()`` order, ``ema.py:79-85``) and the same five operations
(``update :24-42``, ``copy_to :44-55``, ``store :57-64``, ``restore :66-77``, ``state_dict / load_state_dict``), so that
``run_lib``-style callers (``store -> copy_to -> sample -> restore``; ``losses.py:115-122``) and checkpoints written by the
reference work unchanged.  The arithmetic runs as multi-tensor ops over the whole parameter list (one launch per operation
instead of one per tensor); ``copy_to`` / ``restore`` notify a HIP ``DMT`` that its packed weights are stale.
"""
from __future__ import annotations

from typing import Iterable, List

import torch


def _trainable(parameters) -> List[torch.Tensor]:
    return [p for p in parameters if p.requires_grad]


class ExponentialMovingAverage:
    def __init__(self, parameters: Iterable[torch.nn.Parameter], decay: float, use_num_updates: bool = True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in parameters if p.requires_grad]
        self.collected_params: List[torch.Tensor] = []

    def effective_decay(self) -> float:
        """The decay ``update`` will apply next: min(decay, (1 + n) / (10 + n)) with n counted after the increment (ema.py:34-37)."""
        if self.num_updates is None:
            return self.decay
        n = self.num_updates + 1
        return min(self.decay, (1 + n) / (10 + n))

    @torch.no_grad()
    def update(self, parameters) -> None:
        """shadow -= (1 - decay) * (shadow - param) for every trainable parameter (ema.py:24-42)."""
        decay = self.effective_decay()
        if self.num_updates is not None:
            self.num_updates += 1
        params = [p.detach() for p in _trainable(parameters)]
        if len(params) != len(self.shadow_params):
            raise ValueError(f"EMA holds {len(self.shadow_params)} tensors, got {len(params)} trainable parameters")
        if not params:
            return
        one_minus_decay = 1.0 - decay
        diff = torch._foreach_sub(self.shadow_params, params)          # (shadow - param)
        torch._foreach_mul_(diff, one_minus_decay)
        torch._foreach_sub_(self.shadow_params, diff)

    def _sync(self) -> None:
        """Hook of a sharded optimizer (losses.FusedAdamW): every rank updates only its shard of the shadow parameters in the step
        kernel; the shards are all-gathered when the averages are READ (copy_to, state_dict), not on every step."""
        fn = getattr(self, "_before_read", None)
        if fn is not None:
            fn()

    @torch.no_grad()
    def copy_to(self, parameters) -> None:
        self._sync()
        params = _trainable(parameters)
        if len(params) != len(self.shadow_params):
            raise ValueError(f"EMA holds {len(self.shadow_params)} tensors, the model has {len(params)} trainable ones")
        for shadow, p in zip(self.shadow_params, params):
            p.data.copy_(shadow.data.to(p.device))

    def store(self, parameters) -> None:
        self.collected_params = [p.detach().clone() for p in parameters]

    @torch.no_grad()
    def restore(self, parameters) -> None:
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)

    def state_dict(self):
        """ema.py:79-81.  Under a sharded optimizer this gathers the shards first: a collective, every rank must call it."""
        self._sync()
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict) -> None:
        """ema.py:83-85.  When a fused optimizer has made the shadow parameters views of its flat buffer (``attach_ema``), the loaded
        values are copied INTO those views - rebinding the list would leave the step kernel averaging the old buffer."""
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        loaded = state_dict["shadow_params"]
        if getattr(self, "_before_read", None) is not None:
            if len(loaded) != len(self.shadow_params):
                raise ValueError(f"EMA holds {len(self.shadow_params)} tensors, the checkpoint has {len(loaded)}")
            with torch.no_grad():
                for mine, new in zip(self.shadow_params, loaded):
                    mine.copy_(new.to(mine.device))
            done = getattr(self, "_after_load", None)
            if done is not None:
                done()
        else:
            self.shadow_params = loaded
