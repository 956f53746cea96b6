import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_mismatch(a, b, rel=1e-9, floor=1e-12):
    """Boolean mask of values that differ by more than `rel` relative (with an absolute floor)."""
    import numpy as np
    a = np.asarray(a)
    b = np.asarray(b)
    return np.abs(a - b) > rel * np.maximum(np.abs(a), np.abs(b)) + floor
