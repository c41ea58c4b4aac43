"""Selects who executes the fused operators.

The shipped backend is the HIP library (``viforsdes_amd._hip``); it has no CPU fallback and
raises if the library is missing or tensors are not on a HIP device.  ``set_backend`` exists so
that *tests* can plug in the CPU oracle (``oracle/torch_backend.py``) to exercise host logic
(trainer loop, gloo data-parallel step) on machines without a GPU.  Nothing in this package
ever installs a non-HIP backend by itself.
"""
from __future__ import annotations

from typing import Any, Optional

_override: Optional[Any] = None


def get_backend() -> Any:
    if _override is not None:
        return _override
    from .. import _hip
    _hip.load()
    return _hip


def set_backend(backend: Optional[Any]) -> None:
    """Install (or with ``None`` remove) an alternative operator backend. Test use only."""
    global _override
    _override = backend
