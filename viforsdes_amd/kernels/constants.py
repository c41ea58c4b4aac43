"""Gate order and limits shared with the HIP kernels (reference: kernels/constants.py:7-13)."""
from typing import Final

GATE_R: Final = 0
GATE_Z: Final = 1
GATE_N: Final = 2
NUM_GATES: Final = 3
MAX_LAYERS: Final = 4
