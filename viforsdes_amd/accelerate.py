"""``Accelerator`` shim (reference: accelerate.py:47-62 wraps ``encoder.sit`` in ``torch.compile``).
This build does not use a tracing compiler for its hot path (hand-written HIP kernels + eager
hipBLASLt GEMMs), so ``optimize`` returns the module unchanged; the dataclass exists so that
``InferenceConfig(accelerator=Accelerator(...))`` written for the reference keeps working."""
from __future__ import annotations

import os
import sys
from contextlib import contextmanager
from dataclasses import dataclass
from enum import Enum
from typing import Iterator, Optional

import torch
from torch import nn

TUNED_GEMMS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemms_gfx950.csv")
_tuned_state = {"enabled": False}


def enable_tuned_gemms() -> bool:
    """Let the encoder's torch GEMMs use the hipBLASLt / rocBLAS solutions recorded in ``tuned_gemms_gfx950.csv``
    (PyTorch TunableOp): the library heuristic picks e.g. a 0.31 ms kernel for the [2e5, 256] x [256, 1536] projection where
    a 0.22 ms one exists.  Lookup only by default (an unknown shape, or a file recorded with other library versions, falls
    back to the heuristic); ``VSDE_TUNE_GEMMS=1`` also tunes new shapes on first use and appends them to the file (a few
    seconds per shape -- run the warm-up steps before capturing a HIP graph)."""
    if _tuned_state["enabled"] or not torch.cuda.is_available() or os.environ.get("VSDE_TUNED_GEMMS", "1") == "0":
        return _tuned_state["enabled"]
    try:
        import torch.cuda.tunable as tunable
        tune = os.environ.get("VSDE_TUNE_GEMMS") == "1"
        if not tune and not os.path.exists(TUNED_GEMMS_FILE):
            return False
        tunable.enable(True)
        # lookup mode must never rewrite the packaged file (several ranks may share it): point the writer at a scratch path
        tunable.set_filename(TUNED_GEMMS_FILE if tune else os.path.join(
            os.environ.get("TMPDIR", "/tmp"), f"vsde_tunableop_{os.getpid()}.csv"))
        tunable.tuning_enable(tune)
        if tune:
            tunable.set_max_tuning_duration(30)
            tunable.set_max_tuning_iterations(50)
        if os.path.exists(TUNED_GEMMS_FILE):
            tunable.read_file(TUNED_GEMMS_FILE)
        _tuned_state["enabled"] = True
    except Exception:  # an optimisation, never a requirement
        _tuned_state["enabled"] = False
    return _tuned_state["enabled"]


class CompileMode(Enum):
    DEFAULT = "default"
    REDUCE_OVERHEAD = "reduce-overhead"
    MAX_AUTOTUNE = "max-autotune"


@dataclass(frozen=True)
class Accelerator:
    compile: bool = False
    compile_mode: CompileMode = CompileMode.DEFAULT
    compile_fullgraph: bool = False       # accepted for configs written for the reference (accelerate.py:47-62); ignored
    compile_dynamic: Optional[bool] = None

    def optimize(self, module: nn.Module) -> nn.Module:
        if self.compile and not _tuned_state.get("compile_notice"):
            _tuned_state["compile_notice"] = True
            print("viforsdes_amd: Accelerator(compile=True) is ignored -- the hot path is hand-written HIP kernels, "
                  "no tracing compiler is used", file=sys.stderr)
        return module


@contextmanager
def suppress_torch_compile_output() -> Iterator[None]:
    yield
