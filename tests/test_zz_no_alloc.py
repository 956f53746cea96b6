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


def test_noalloc_with_three_lights():
    """The workspace holds one occlusion byte per light and hit, so the number of lights decides how many rows fit a batch
    (the synthetic scene above has cover.rs's two lights): 260 primitives in 13 groups under three lights."""
    from raytracer_amd import Material, Matrix, PointLight, Primitive, World
    rng = np.random.default_rng(17)
    groups = []
    for g in range(13):
        prims = []
        for i in range(20):
            kind = [Primitive.new_sphere, Primitive.new_cube][i % 2]
            p = rng.uniform(-1.5, 1.5, 3)
            s = rng.uniform(0.15, 0.4)
            prims.append(kind().with_transform(Matrix.chain(Matrix.translation(p[0], p[1], p[2]), Matrix.scaling(s, s, s)))
                         .with_material(Material(color=tuple(rng.uniform(0.2, 0.9, 3)), reflective=0.4 if i % 4 == 0 else 0.0,
                                                 transparency=0.6 if i % 7 == 0 else 0.0, refractive_index=1.4)))
        groups.append(rt.Compound.new_group(prims).with_transform(
            Matrix.chain(Matrix.translation(4.0 * (g % 5) - 8.0, 1.6, 4.0 * (g // 5)), Matrix.rotation_y(0.3 * g))))
    floor = Primitive.new_plane().with_material(Material(color=(0.8, 0.8, 0.75), reflective=0.15))
    world = World(groups + [floor], [PointLight((-10, 12, -10), (0.6, 0.6, 0.6)), PointLight((9, 7, -6), (0.3, 0.3, 0.4)),
                                    PointLight((0, 15, 12), (0.3, 0.2, 0.2))])
    cam = rt.Camera(400, 225, 1.0).with_transform(Matrix.view_transform((0, 6, -16), (0, 1, 3), (0, 1, 0)))
    scene = rt.DeviceScene(world.flatten("global"))
    scene.set_wavefront(2)
    assert "pipeline" in scene.pipeline_name(5)
    _both(scene, cam)
    _both(scene, cam, parts=((2, 3),))
