"""The C-ABI libraries load and export every function include/*.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_hip_\w+|rth_\w+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("rt_hip.h", "librt_hip.so"), ("rt_host.h", "librt_host.so")])
def test_every_declared_symbol_is_exported(header, lib):
    names = [n for n in declared(header) if header == "rt_hip.h" or n.startswith("rth_")]
    assert len(names) >= 10
    handle = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "lib", lib))
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing


def test_binding_list_matches_header():
    from raytracer_amd import _native as N
    assert sorted(N.HIP_SYMBOLS) == declared("rt_hip.h")


def test_strerror_and_no_device_is_loud():
    """Without a GPU every compute entry point reports RT_ERR_NO_DEVICE; nothing falls back to a CPU path."""
    import numpy as np
    import torch

    import raytracer_amd as rt
    from raytracer_amd import _native as N
    assert N.hip.rt_hip_strerror(0) == b"ok"
    for code in range(-10, 0):
        assert N.hip.rt_hip_strerror(code) not in (b"", b"unknown error")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    world = rt.World.default()
    with pytest.raises(rt.RenderError) as e:
        rt.Camera(4, 4, 1.0).render(world)
    assert e.value.code == N.RT_ERR_NO_DEVICE
    with pytest.raises(rt.RenderError):
        world.color_at(np.array([0, 0, -5, 0, 0, 1.0]))
