"""NumPy math utils (reference: src/rtgs/utils/math.py:8-14)."""

import numpy as np


def sigmoid(x: np.ndarray) -> np.ndarray:
    """1 / (1 + exp(-x)), the activation of the PLY loader (scene.py:113-114)."""
    return 1 / (1 + np.exp(-x))
