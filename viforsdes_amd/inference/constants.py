"""Numerical constants (reference: inference/constants.py:5-7)."""
from typing import Final

LOSS_EMA_DECAY: Final = 0.98
DIAG_MIN: Final = 1e-2
DEFAULT_EMA_DECAY: Final = 0.999
