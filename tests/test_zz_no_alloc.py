"""rt_hip_render_rows_noalloc: the entry point that allocates nothing (fixed device workspace in the library)
against rt_hip_render_rows, through the C ABI, bit for bit; two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def _both(scene, cam, depth=5, parts=((0, 1),)):
    import torch
    from raytracer_amd import tiling
    H, W = cam.v_size, cam.h_size
    cd = cam.desc()
    for part, n_parts in parts:
        band = tiling.DEFAULT_BAND_ROWS if n_parts > 1 else H
        rows = tiling.n_local_rows(H, part, n_parts, band)
        a = torch.full((rows, W, 3), -5.0, dtype=torch.float64, device="cuda")
        b = torch.full((rows, W, 3), -9.0, dtype=torch.float64, device="cuda")
        scene.render_rows(cd, depth, part, n_parts, band, rows, a.data_ptr())
        scene.render_rows_noalloc(cd, depth, part, n_parts, band, rows, b.data_ptr())
        torch.cuda.synchronize()
        scene.status()
        valid = tiling.image_rows(H, part, n_parts, band) >= 0
        a, b = a.cpu().numpy()[valid], b.cpu().numpy()[valid]
        assert (a != -5.0).all() and np.array_equal(a, b), (part, n_parts)


def test_noalloc_equals_render_rows_on_a_large_scene_then_a_smaller_one():
    # 1 500 primitives in nested groups, reflective and transparent materials: rt_hip_render_rows takes the kernel
    # pipeline with its allocated workspace, the no-alloc call the same kernels in the library's fixed array --
    # 960x540 needs about twice the array, so the rows go through in batches
    world, cam = rt.synthetic(960, 540, 1500)
    scene = rt.DeviceScene(world.flatten())
    assert "pipeline" in scene.pipeline_name(5)
    _both(scene, cam)
    # the second call, smaller: the array still holds the first call's queues and counters
    world2, cam2 = rt.synthetic(203, 117, 1500)
    _both(scene, cam2, parts=((0, 1), (1, 3)))
    _both(scene, cam2, depth=2)
    # a scene of another kind shares the same array: small scenes take the one-launch kernel (no workspace at all)
    world3, cam3 = rt.exercise("cover", 160, 90)
    scene3 = rt.DeviceScene(world3.flatten())
    _both(scene3, cam3)
    _both(scene, cam2)


def test_noalloc_refuses_a_frame_too_wide_for_its_workspace():
    import ctypes as C
    import torch
    from raytracer_amd import _native as N
    world, cam = rt.synthetic(64, 8, 400)
    scene = rt.DeviceScene(world.flatten())
    scene.set_wavefront(2)
    wide = rt.Camera(60000, 8, 0.8, cam.transform)
    out = torch.zeros((8, 60000, 3), dtype=torch.float64, device="cuda")
    cd = wide.desc()
    rc = N.hip.rt_hip_render_rows_noalloc(scene._p, C.byref(cd), 5, 0, 1, 8, 8, C.c_void_p(out.data_ptr()), None, None)
    assert rc == N.RT_ERR_NOALLOC_SIZE
    assert b"too wide" in N.hip.rt_hip_strerror(rc)
