"""EMA of the trainable parameters (reference: inference/exponential_moving_average.py:13-47).
The update is one fused ``torch._foreach_lerp_`` over all parameters instead of a Python loop."""
from __future__ import annotations

from contextlib import contextmanager
from typing import Iterator

import torch
from torch import Tensor, nn

from .constants import DEFAULT_EMA_DECAY


class ExponentialMovingAverage:
    def __init__(self, model: nn.Module, decay: float = DEFAULT_EMA_DECAY) -> None:
        self.model, self.decay = model, decay
        self.shadow: dict[str, Tensor] = {}
        self.version = 0               # bumped whenever the shadow tensors are replaced (the fused optimizer step caches their addresses)
        self.fused_step_done = False   # set by inference/fused_optimizer.py when its kernel already applied this step's update
        self._init_shadow()

    def _init_shadow(self) -> None:
        self.shadow = {name: p.detach().clone() for name, p in self.model.named_parameters()}
        self.version += 1

    @torch.no_grad()
    def update(self) -> None:
        if self.fused_step_done:       # the optimizer kernel updated parameters and shadow in one pass
            self.fused_step_done = False
            return
        names, params = zip(*self.model.named_parameters())
        torch._foreach_lerp_([self.shadow[n] for n in names], [p.detach() for p in params], 1.0 - self.decay)

    @contextmanager
    def apply(self) -> Iterator[None]:
        """Temporarily swap the averaged weights into the model."""
        named = list(self.model.named_parameters())
        params = [p.detach() for _, p in named]
        backup = [torch.empty_like(p) for p in params]
        # p.copy_() per parameter, as the reference does it, is ~200 launches each way: one multi-tensor copy instead (the detached
        # aliases share the parameters' version counters: caches keyed on Tensor._version see the swap)
        with torch.no_grad():
            torch._foreach_copy_(backup, params)
            torch._foreach_copy_(params, [self.shadow[name] for name, _ in named])
        try:
            yield
        finally:
            with torch.no_grad():
                torch._foreach_copy_(params, backup)

    def state_dict(self) -> dict[str, Tensor]:
        return {k: v.clone() for k, v in self.shadow.items()}

    def load_state_dict(self, state: dict[str, Tensor]) -> None:
        self.shadow = {k.replace("encoder.sit._orig_mod.", "encoder.sit."): v.clone() for k, v in state.items()}
        self.version += 1
