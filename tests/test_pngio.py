"""The stdlib PNG reader the GPU tests fall back to (tests/pngio.py) decodes the goldens as PIL does."""
import os

import numpy as np
import pytest

from pngio import GOLDEN, read_png_stdlib


@pytest.mark.parametrize("name", ["hexagon", "plane"])  # one 8-bit grey, one RGB
def test_stdlib_png_reader_agrees_with_pil(name):
    Image = pytest.importorskip("PIL.Image")
    path = os.path.join(GOLDEN, f"{name}.png")
    assert np.array_equal(read_png_stdlib(path), np.array(Image.open(path).convert("RGB")))
