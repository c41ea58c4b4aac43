"""``Accelerator`` shim (reference: accelerate.py:47-62 wraps ``encoder.sit`` in ``torch.compile``).
This build does not use a tracing compiler for its hot path (hand-written HIP kernels + eager
hipBLASLt GEMMs), so ``optimize`` returns the module unchanged; the dataclass exists so that
``InferenceConfig(accelerator=Accelerator(...))`` written for the reference keeps working."""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass
from enum import Enum
from typing import Iterator

from torch import nn


class CompileMode(Enum):
    DEFAULT = "default"
    REDUCE_OVERHEAD = "reduce-overhead"
    MAX_AUTOTUNE = "max-autotune"


@dataclass(frozen=True)
class Accelerator:
    compile: bool = False
    compile_mode: CompileMode = CompileMode.DEFAULT

    def optimize(self, module: nn.Module) -> nn.Module:
        return module


@contextmanager
def suppress_torch_compile_output() -> Iterator[None]:
    yield
