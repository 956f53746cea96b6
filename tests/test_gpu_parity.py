"""GPU parity: HIP render path (through the C ABI) vs the CPU oracle on the same scenes.

Bar (BASELINE.json north_star): pre-quantisation colours within 1e-9 relative in FP64;
8-bit image identical except <=1 LSB on <=0.1 % of pixels.  The oracle and the kernels
share no code but follow the same operation order, so in practice every pixel agrees to the
last bits except where pow()/hypot() (ocml vs glibc) differ by an ulp.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from conftest import rel_mismatch
from oracle import orc

pytestmark = pytest.mark.gpu

REL = 1e-9  # tolerance stated by north_star


def compare(name, gpu, ref):
    bad = rel_mismatch(gpu, ref, REL).any(axis=-1)
    q_gpu, q_ref = orc.quantize(gpu), orc.quantize(ref)
    dq = np.abs(q_gpu.astype(int) - q_ref.astype(int)).max(axis=-1)
    n = bad.size
    print(f"{name}: {bad.sum()}/{n} px beyond 1e-9 rel; 8-bit: {(dq > 0).sum()} differ, {(dq > 1).sum()} by >1 LSB; "
          f"max abs diff {np.abs(gpu - ref).max():.3e}")
    assert (dq > 1).sum() == 0, f"{name}: pixels differ by more than 1 LSB"
    assert (dq > 0).sum() <= 1e-3 * n, f"{name}: more than 0.1% of pixels differ in 8 bits"
    assert bad.sum() <= 1e-3 * n, f"{name}: more than 0.1% of pixels beyond 1e-9 relative"
    return bad.sum()


@pytest.mark.parametrize("name,width,height", [
    ("plane", 512, 270), ("scene", 512, 270), ("pattern", 512, 270), ("glass_bubble", 512, 270),
    ("hexagon", 512, 270), ("csg", 384, 200), ("cover", 512, 270), ("room", 512, 270), ("teapot", 256, 135),
])
def test_exercise_matches_oracle(name, width, height):
    world, cam = rt.exercise(name, width, height)
    gpu = cam.render(world).pixels
    ref, _ = orc.OracleWorld(world.describe()).render(cam)
    compare(name, gpu, ref)


def test_default_world_center_pixel():
    # camera.rs:149-161 render_world_with_camera
    world = rt.World.default()
    cam = rt.Camera(11, 11, np.pi / 2).with_transform(rt.Matrix.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    img = cam.render(world)
    assert np.allclose(img.pixel_at(5, 5), (0.38066, 0.47583, 0.2855), atol=1e-4)


@pytest.mark.parametrize("name,width,height", [("pattern", 517, 271), ("cover", 300, 170), ("csg", 200, 101)])
def test_render_ppm_is_the_text_of_the_rendered_canvas(name, width, height):
    """`camera.render(&world).to_ppm()` with the quantisation done on the device (rt_hip_render_rgb8)
    is byte-identical to the reference writer (canvas.rs:46-90, restated in oracle/) applied to the f64
    frame of the same device render, and the 8-bit samples equal the oracle frame's up to the stated bar."""
    world, cam = rt.exercise(name, width, height)
    frame = cam.render(world).pixels
    rgb8 = cam.render_rgb8(world)
    assert np.array_equal(rgb8, orc.quantize(frame))
    text = cam.render_ppm(world)
    assert text == orc.to_ppm(frame)
    ref, _ = orc.OracleWorld(world.describe()).render(cam)
    dq = np.abs(rgb8.astype(int) - orc.quantize(ref).astype(int)).max(axis=-1)
    assert (dq > 1).sum() == 0 and (dq > 0).sum() <= 1e-3 * dq.size


def test_synthetic_matches_oracle():
    world, cam = rt.synthetic(256, 144, 2000)
    gpu = cam.render(world).pixels
    ref, _ = orc.OracleWorld(world.describe()).render(cam)
    compare("synthetic-2000", gpu, ref)


@pytest.mark.parametrize("name", ["pattern", "glass_bubble", "csg", "teapot"])
def test_full_size_frame_properties(name):
    """BASELINE.json's size (4096x2160): the frame rendered in one launch equals the frame rendered
    as 8 interleaved row-band shards and reassembled (bit for bit), and every 54th row equals the
    oracle's render of that row."""
    import torch
    from raytracer_amd import tiling
    W, H = 4096, 2160
    world, cam = rt.exercise(name, W, H)
    scene = rt.DeviceScene(world.flatten())
    cd = cam.desc()
    full = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    scene.render_rows(cd, 5, 0, 1, H, H, full.data_ptr())
    n_parts, band = 8, tiling.DEFAULT_BAND_ROWS
    slots = tiling.padded_local_rows(H, n_parts, band)
    shards = []
    for part in range(n_parts):
        buf = torch.zeros((slots, W, 3), dtype=torch.float64, device="cuda")
        scene.render_rows(cd, 5, part, n_parts, band, tiling.n_local_rows(H, part, n_parts, band), buf.data_ptr())
        shards.append(buf)
    torch.cuda.synchronize()
    scene.status()
    assert torch.equal(tiling.assemble(shards, H, n_parts, band), full)
    step = 54
    ref, _ = orc.OracleWorld(world.describe()).render(cam, rows=(step // 2, H, step))
    gpu = full.cpu().numpy()[step // 2::step]
    compare(f"{name}-4K-rows", gpu, ref[step // 2::step])


def test_flat_scene_too_large_for_lds_staging():
    """Seven cubes: plain primitives and no acceleration structure (FLAT kernel), but up to 14 world
    intersections, more than the LDS-resident list holds -- the private-memory variant runs."""
    from raytracer_amd import Material, Matrix, PointLight, Primitive, World
    shapes = [Primitive.new_cube().with_transform(Matrix.chain(Matrix.translation(2.2 * (i - 3), 0.3 * i, 0.5 * i),
                                                               Matrix.rotation_y(0.3 * i), Matrix.scaling(0.9, 0.9, 0.9)))
              .with_material(Material(color=(0.2 + 0.1 * i, 0.5, 0.9 - 0.1 * i), reflective=0.3 if i % 2 else 0.0,
                                      transparency=0.6 if i == 3 else 0.0, refractive_index=1.4))
              for i in range(7)]
    world = World(shapes, [PointLight((-10, 10, -10), (1, 1, 1)), PointLight((6, 8, -4), (0.3, 0.3, 0.3))])
    cam = rt.Camera(320, 180, 1.0).with_transform(Matrix.view_transform((0, 3, -12), (0, 0, 0), (0, 1, 0)))
    gpu = cam.render(world).pixels
    ref, _ = orc.OracleWorld(world.describe()).render(cam)
    compare("seven-cubes", gpu, ref)


def test_conic_parallel_ray_branch_under_acceleration():
    """Rays (nearly) parallel to truncated cylinders and cones inside an accelerated group: the
    reference reports a wall hit with no y-range check (primitive.rs:486-490), anywhere along the
    infinite surface; the direction index must let those hits through the boxes."""
    from raytracer_amd import Conic, Material, Matrix, PointLight, Primitive, World
    prims = []
    for i in range(12):
        kind = Primitive.new_cylinder if i % 2 == 0 else Primitive.new_cone
        lim = Conic(-0.5, 0.5, i % 3 == 0) if i % 2 == 0 else Conic(-1.0, -0.2, i % 3 == 0)
        prims.append(kind(lim).with_transform(Matrix.chain(Matrix.translation(1.5 * (i % 4) - 2.2, 1.2 * (i // 4) - 1.0, 0.0),
                                                           Matrix.rotation_x(np.pi / 2), Matrix.scaling(0.5, 0.5, 0.5)))
                     .with_material(Material(color=(0.3 + 0.05 * i, 0.6, 0.4))))
    group = rt.Compound.new_group(prims)
    world = World([group, Primitive.new_plane().with_transform(Matrix.chain(Matrix.translation(0, 0, 30), Matrix.rotation_x(np.pi / 2)))],
                  [PointLight((-5, 8, -10), (1, 1, 1))])
    # an orthographic-like bundle along +z (the cylinders' axes after rotation_x(pi/2)) plus slight tilts
    rng = np.random.default_rng(9)
    n = 20000
    o = np.stack([rng.uniform(-3.5, 3.5, n), rng.uniform(-2.0, 2.5, n), np.full(n, -8.0)], axis=1)
    d = np.stack([rng.normal(0, 2e-3, n), rng.normal(0, 2e-3, n), np.ones(n)], axis=1)
    d45 = np.stack([rng.normal(0, 2e-3, n), np.sin(np.pi / 4) + rng.normal(0, 1e-4, n), np.cos(np.pi / 4) + rng.normal(0, 1e-4, n)], axis=1)
    d = np.concatenate([d, d45])   # cones: 45 degrees to the axis zeroes `a`
    o = np.concatenate([o, o])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1)
    ow = orc.OracleWorld(world.describe())
    got, ref = world.probe(rays), ow.probe(rays)
    assert np.array_equal(got[:, [0, 2]], ref[:, [0, 2]])
    assert not rel_mismatch(got, ref, 1e-12, 1e-15).any()
    # the branch really fires in this bundle: some hits lie outside the shape's y-range
    col = world.color_at(rays, 2)
    assert not rel_mismatch(col, ow.color_at(rays, 2)).any()


def test_identity_transform_shortcut_is_bit_exact_down_to_the_sign_of_zero(monkeypatch):
    """Primitives whose inverse transform is the identity matrix skip the ray transform (rt_trace.h: identity_ray).
    1*x + 0*y + 0*z (+ 0) differs from x only in the sign of a zero, which the shortcut reproduces: every combination of
    signed zeros and ordinary values in origin and direction, through every primitive kind, against the same scene with
    the shortcut switched off -- compared as bit patterns -- and against the oracle."""
    import itertools
    from raytracer_amd import Conic, Material, Matrix, PointLight, Primitive, World

    def build():
        tri = [Primitive.new_triangle((-1, -1, 0.5), (1, -1, 0.5), (0, 1, 0.5)),
               Primitive.new_smooth_triangle((-1, -1, -0.5), (1, -1, -0.5), (0, 1, -0.5), (0, 0, 1), (0, 1, 0), (1, 0, 0))]
        solids = [Primitive.new_sphere().with_material(Material(color=(0.9, 0.2, 0.2), reflective=0.3)),
                  Primitive.new_cube().with_material(Material(color=(0.2, 0.9, 0.2), transparency=0.5, refractive_index=1.3)),
                  Primitive.new_cylinder(Conic(-0.7, 0.7, True))]  # (no cone: rays through its apex, the origin, have no normal)
        group = rt.Compound.new_group(tri + [Primitive.new_plane()])
        return World(solids + [group], [PointLight((-5, 8, -10), (1, 1, 1))])

    vals_o = (-0.0, 0.0, -1.5, 0.7)
    vals_d = (-0.0, 0.0, -1.0, 1.0, -0.3, 0.3)
    rays = np.array([o + d for o in itertools.product(vals_o, repeat=3) for d in itertools.product(vals_d, repeat=3)
                     if any(c != 0.0 for c in d)], dtype=np.float64)
    assert len(rays) > 13000
    world = build()
    got_p, got_c = world.probe(rays), world.color_at(rays, 3)
    monkeypatch.setenv("RT_HIP_NO_IDENTITY", "1")
    plain = build()
    ref_p, ref_c = plain.probe(rays), plain.color_at(rays, 3)
    monkeypatch.delenv("RT_HIP_NO_IDENTITY")
    same_p = (got_p.view(np.uint64) == ref_p.view(np.uint64)) | (np.isnan(got_p) & np.isnan(ref_p))
    same_c = (got_c.view(np.uint64) == ref_c.view(np.uint64)) | (np.isnan(got_c) & np.isnan(ref_c))
    assert same_p.all() and same_c.all()
    ow = orc.OracleWorld(world.describe())
    ref = ow.probe(rays)
    ok = ~np.isnan(ref).any(axis=1)
    assert np.array_equal(got_p[ok][:, [0, 2]], ref[ok][:, [0, 2]])
    assert not rel_mismatch(got_p[ok], ref[ok], 1e-12, 1e-15).any()


def test_direction_cells_built_on_the_device_equal_host_built_ones():
    """From 64 candidate conics on the flattener lists the candidates only and rt_hip_scene_create rasterises the
    direction cells on the device (rt_dirindex.hip).  Same frames as with the host-built table, in every mode; and the
    parallel-ray bundle of the test above through 96 conics in one accelerated group."""
    from raytracer_amd import Conic, Material, Matrix, PointLight, Primitive, World
    world, cam = rt.synthetic(384, 216, 1500)
    flat_dev, flat_host = world.flatten(), world.flatten(host_cells=True)
    assert not flat_dev.desc.contents.dir_cell_start and flat_dev.desc.contents.n_dir_cands >= 64
    assert flat_host.desc.contents.dir_cell_start and flat_host.desc.contents.n_dir_entries > 0
    a, b = rt.DeviceScene(flat_dev), rt.DeviceScene(flat_host)
    ref = _render_resident(b, cam, 0)
    for mode in (0, 1, 2):
        assert np.array_equal(_render_resident(a, cam, mode), ref), mode
    a.close(), b.close()
    prims = []
    for i in range(96):
        kind = Primitive.new_cylinder if i % 2 == 0 else Primitive.new_cone
        lim = Conic(-0.5, 0.5, i % 3 == 0) if i % 2 == 0 else Conic(-1.0, -0.2, i % 3 == 0)
        prims.append(kind(lim).with_transform(Matrix.chain(Matrix.translation(0.9 * (i % 12) - 5.0, 0.9 * (i // 12) - 3.0, 0.0),
                                                           Matrix.rotation_x(np.pi / 2 + 0.01 * (i % 5)), Matrix.scaling(0.3, 0.3, 0.3)))
                     .with_material(Material(color=(0.3 + 0.005 * i, 0.6, 0.4))))
    world = World([rt.Compound.new_group(prims),
                   Primitive.new_plane().with_transform(Matrix.chain(Matrix.translation(0, 0, 30), Matrix.rotation_x(np.pi / 2)))],
                  [PointLight((-5, 8, -10), (1, 1, 1))])
    assert not world.flatten().desc.contents.dir_cell_start
    rng = np.random.default_rng(11)
    n = 20000
    o = np.stack([rng.uniform(-5.5, 5.5, n), rng.uniform(-3.5, 4.5, n), np.full(n, -8.0)], axis=1)
    d = np.stack([rng.normal(0, 6e-3, n), rng.normal(0, 6e-3, n), np.ones(n)], axis=1)
    d45 = np.stack([rng.normal(0, 2e-3, n), np.sin(np.pi / 4) + rng.normal(0, 3e-3, n), np.cos(np.pi / 4) + rng.normal(0, 3e-3, n)], axis=1)
    d = np.concatenate([d, d45])
    o = np.concatenate([o, o])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1)
    ow = orc.OracleWorld(world.describe())
    got, ref = world.probe(rays), ow.probe(rays)
    assert np.array_equal(got[:, [0, 2]], ref[:, [0, 2]])
    assert not rel_mismatch(got, ref, 1e-12, 1e-15).any()
    assert not rel_mismatch(world.color_at(rays, 2), ow.color_at(rays, 2)).any()


def _render_resident(scene, cam, mode, queue_items=0, node_items=0, with_stats=False):
    import ctypes as C
    import torch
    from raytracer_amd import _native as N
    scene.set_wavefront(mode, queue_items, node_items)
    H, W = cam.v_size, cam.h_size
    out = torch.full((H, W, 3), -7.0, dtype=torch.float64, device="cuda")
    d_stats = torch.zeros(256, dtype=torch.uint8, device="cuda") if with_stats else None
    scene.render_rows(cam.desc(), 5, 0, 1, H, H, out.data_ptr(), None, d_stats.data_ptr() if with_stats else None)
    torch.cuda.synchronize()
    scene.status()
    img = out.cpu().numpy()
    if not with_stats:
        return img
    st = N.RtStats.from_buffer_copy(bytes(d_stats.cpu().numpy()[:C.sizeof(N.RtStats)])).as_dict()
    return img, st


@pytest.mark.parametrize("name,width,height", [("glass_bubble", 320, 180), ("cover", 320, 180), ("room", 256, 144),
                                                ("csg", 200, 112), ("synthetic", 256, 144)])
def test_generation_per_launch_equals_one_launch(name, width, height):
    """color_at's recursion traced one generation of rays per launch (node pool + ray queues) gives the
    frame of the one-launch kernel bit for bit and traces the same rays -- also when the queue or the
    node pool is far too small and most hits have to keep their subtrees to themselves."""
    world, cam = rt.synthetic(width, height, 1500) if name == "synthetic" else rt.exercise(name, width, height)
    scene = rt.DeviceScene(world.flatten())
    mono, st_mono = _render_resident(scene, cam, 0, with_stats=True)
    wave, st_wave = _render_resident(scene, cam, 1, with_stats=True)
    assert np.array_equal(mono, wave)
    for key in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "hits_shaded", "container_passes"):
        assert st_mono[key] == st_wave[key], key
    assert st_mono["rays_reflect"] + st_mono["rays_refract"] > 0
    pixels = width * height
    for q_items, n_items in ((pixels // 7, pixels), (pixels, pixels // 11), (193, 61)):
        assert np.array_equal(mono, _render_resident(scene, cam, 1, q_items, n_items))
    assert np.array_equal(mono, _render_resident(scene, cam, -1))
    # mode 2, the kernel pipeline (scenes it does not cover are rendered as in mode 1): same frame, same rays
    pipe, st_pipe = _render_resident(scene, cam, 2, with_stats=True)
    assert np.array_equal(mono, pipe)
    for key in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "hits_shaded", "container_passes", "lightings"):
        assert st_mono[key] == st_pipe[key], key
    for q_items, n_items in ((pixels // 7, pixels), (pixels, pixels // 11), (193, 61)):
        assert np.array_equal(mono, _render_resident(scene, cam, 2, q_items, n_items))
    ref, _ = orc.OracleWorld(world.describe()).render(cam)
    compare(f"{name}-generations", wave, ref)


def test_resident_scene_renders_equal_one_shot_renders():
    """rt_hip_scene_render / _rgb8: a scene kept on the device, rendered from several cameras and at
    several sizes (the scene's framebuffers grow and are reused), gives what `camera.render(&world)`
    gives for each camera -- flatten, upload and all -- bit for bit."""
    world, cam0 = rt.exercise("cover", 160, 90)
    scene = rt.DeviceScene(world.flatten())
    cams = [cam0,
            rt.Camera(96, 54, 0.9).with_transform(rt.Matrix.view_transform((-4, 5, -9), (5, -1, 5), (-0.4, 1, 0))),
            rt.Camera(240, 135, 0.7).with_transform(rt.Matrix.view_transform((-7, 7, -11), (6, -2, 6), (-0.45, 1, 0))),
            rt.Camera(33, 17, 1.2).with_transform(rt.Matrix.view_transform((0, 2, -12), (0, 0, 0), (0, 1, 0)))]
    for cam in cams:
        one_shot = cam.render(world).pixels
        resident, st = scene.render(cam, stats=True)
        assert np.array_equal(resident.pixels, one_shot)
        assert st.rays_primary == cam.h_size * cam.v_size
        assert np.array_equal(scene.render_rgb8(cam), orc.quantize(one_shot))


def test_generations_with_row_bands_and_other_budgets():
    """The generation-per-launch kernels under what a multi-GPU job asks of them -- interleaved row bands
    rendered shard by shard on one scene (the workspace is reused, sized by the largest shard) -- and at
    recursion budgets other than camera.rs:59's 5: always the one-launch kernel's frame, bit for bit."""
    import torch
    from raytracer_amd import tiling
    W, H = 203, 117  # ragged on purpose: not a multiple of the 16x16 tiles nor of the band height
    world, cam = rt.synthetic(W, H, 1200)
    scene = rt.DeviceScene(world.flatten())
    cd = cam.desc()

    def frame(mode, depth, n_parts=1, band=tiling.DEFAULT_BAND_ROWS):
        scene.set_wavefront(mode)
        slots = tiling.padded_local_rows(H, n_parts, band)
        shards = []
        for part in range(n_parts):
            buf = torch.full((slots, W, 3), -3.0, dtype=torch.float64, device="cuda")
            scene.render_rows(cd, depth, part, n_parts, band, tiling.n_local_rows(H, part, n_parts, band), buf.data_ptr())
            shards.append(buf)
        torch.cuda.synchronize()
        scene.status()
        return tiling.assemble(shards, H, n_parts, band).cpu().numpy()

    for depth in (0, 1, 2, 5, 8):
        one = frame(0, depth)
        assert np.array_equal(one, frame(1, depth)), depth
        assert np.array_equal(one, frame(2, depth)), depth
    whole = frame(0, 5)
    for n_parts, band in ((3, 16), (8, 16), (2, 5)):
        assert np.array_equal(frame(1, 5, n_parts, band), whole), (n_parts, band)
        assert np.array_equal(frame(2, 5, n_parts, band), whole), (n_parts, band)


def test_pipeline_in_batches_of_rows():
    """The kernel pipeline renders frames beyond its workspace budget in batches of rows (RT_HIP_WF_BUDGET_MB):
    the same frame whatever the batch height."""
    import subprocess, sys, os
    code = (
        "import numpy as np, torch, raytracer_amd as rt\n"
        "world, cam = rt.synthetic(640, 360, 1500)\n"
        "scene = rt.DeviceScene(world.flatten())\n"
        "def f(mode):\n"
        "    scene.set_wavefront(mode)\n"
        "    out = torch.zeros((360, 640, 3), dtype=torch.float64, device='cuda')\n"
        "    scene.render_rows(cam.desc(), 5, 0, 1, 360, 360, out.data_ptr())\n"
        "    torch.cuda.synchronize(); scene.status(); return out.cpu().numpy()\n"
        "assert 'pipeline' in scene.pipeline_name(5)\n"
        "assert np.array_equal(f(0), f(2))\n"
        "print('ok')\n")
    env = dict(os.environ, RT_HIP_WF_BUDGET_MB="24")  # 640x360 needs ~105 MB: five batches
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_render_multi_equals_render():
    """rt_hip_render_multi: n_devices = 1 is rt_hip_render bit for bit (frame and counters); with the test knob that
    runs several logical devices on the one GPU here (RT_HIP_MULTI_ALIAS=1) the shards, their landing area on device 0
    and the strided scatter into the frame are exercised for 2, 3 and 8 parts, ragged last band included.  The RCCL
    gather itself needs as many GPUs as parts and is not run here (librccl.so must resolve, though)."""
    import subprocess, sys, os
    code = r'''
import sys
import numpy as np
import raytracer_amd as rt
for name, w, h in (("cover", 203, 117), ("synthetic", 256, 150)):
    world, cam = rt.synthetic(w, h, 1500) if name == "synthetic" else rt.exercise(name, w, h)
    ref, st = cam.render(world, stats=True)
    one, st1 = cam.render_multi(world, 1, stats=True)
    assert np.array_equal(one.pixels, ref.pixels), name
    a, b = st.as_dict(), st1.as_dict()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "hits_shaded", "lightings", "schlicks"):
        assert a[k] == b[k], (name, k, a[k], b[k])  # (counters of the acceleration structures depend on how waves are cut)
    for n in (2, 3, 8):
        got, stn = cam.render_multi(world, n, stats=True)
        assert np.array_equal(got.pixels, ref.pixels), (name, n)
        assert stn.rays_primary == st.rays_primary and stn.hits_shaded == st.hits_shaded, (name, n)
# a device that fails before the gather: every device skips the gather and the call returns that device's error (a
# receive posted for a send that never comes would block device 0 for good); the next call is a normal one again, and
# the devices' resident copies of the scene are reused (same tables) or replaced (other tables)
import ctypes as C, os, time
from raytracer_amd import _native as N
world, cam = rt.synthetic(203, 117, 1500)
ref = cam.render(world).pixels
flat, cd = world.flatten(), cam.desc()
out = np.empty((117, 203, 3))
for bad in (0, 2, 3):
    os.environ["RT_HIP_MULTI_FAIL_DEVICE"] = str(bad)
    t0 = time.time()
    rc = N.hip.rt_hip_render_multi(flat.desc, C.byref(cd), 5, 4, out.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == N.RT_ERR_INVALID_ARG and time.time() - t0 < 20, (bad, rc)
del os.environ["RT_HIP_MULTI_FAIL_DEVICE"]
for _ in range(2):
    out[:] = -1
    assert N.hip.rt_hip_render_multi(flat.desc, C.byref(cd), 5, 4, out.ctypes.data_as(C.POINTER(C.c_double)), None) == 0
    assert np.array_equal(out, ref)
world2, cam2 = rt.exercise("cover", 203, 117)
assert np.array_equal(cam2.render_multi(world2, 4).pixels, cam2.render(world2).pixels)
print("ok")
'''
    env = dict(os.environ, RT_HIP_MULTI_ALIAS="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    from raytracer_amd import _native as N
    import ctypes as C
    world, cam = rt.exercise("plane", 64, 36)
    out = np.empty((36, 64, 3))
    cd = cam.desc()
    flat = world.flatten()
    rc = N.hip.rt_hip_render_multi(flat.desc, C.byref(cd), 5, 4096, out.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == N.RT_ERR_INVALID_ARG  # more devices than the machine has
    assert N.hip.rt_hip_multi_transport() == b"rccl"  # /opt/rocm/lib/librccl.so is part of the image


def test_pipeline_containers_deeper_than_four():
    """Seven concentric glass shells among 60 other primitives, through the kernel pipeline: its containers loop keeps
    four objects in registers and answers deeper nestings from the sorted list alone (containers_from_list); both must
    give the n1 / n2 of intersection.rs:50-75 -- the frame equals the one-launch kernel's and the oracle's."""
    from raytracer_amd import Material, Matrix, PointLight, Primitive, World
    rng = np.random.default_rng(5)
    shapes = [Primitive.new_plane().with_material(Material(color=(0.8, 0.8, 0.7), reflective=0.1))]
    for k in range(7):
        r = 1.6 - 0.2 * k
        shapes.append(Primitive.new_sphere().with_transform(Matrix.chain(Matrix.translation(0, 1.7, 0), Matrix.scaling(r, r, r)))
                      .with_material(Material(color=(0.1, 0.1 + 0.1 * k, 0.3), diffuse=0.2, transparency=0.85, reflective=0.1,
                                              refractive_index=1.0 + 0.07 * (k + 1))))
    for i in range(60):
        kind = [Primitive.new_sphere, Primitive.new_cube][i % 2]
        p = rng.uniform(-6, 6, 3)
        s = rng.uniform(0.2, 0.5)
        shapes.append(kind().with_transform(Matrix.chain(Matrix.translation(p[0], abs(p[1]) * 0.5 + 0.3, p[2] + 3.0), Matrix.scaling(s, s, s)))
                      .with_material(Material(color=tuple(rng.uniform(0.2, 0.9, 3)), transparency=0.5 if i % 5 == 0 else 0.0,
                                              refractive_index=1.3)))
    world = World(shapes, [PointLight((-6, 9, -8), (1, 1, 1))])
    cam = rt.Camera(240, 135, 1.0).with_transform(Matrix.view_transform((0, 2.5, -7), (0, 1.5, 0), (0, 1, 0)))
    flat = world.flatten("global")
    scene = rt.DeviceScene(flat)
    one = _render_resident(scene, cam, 0)
    two = _render_resident(scene, cam, 2)
    scene.set_wavefront(2)
    assert "pipeline" in scene.pipeline_name(5)
    assert np.array_equal(one, two)
    ref, _ = orc.OracleWorld(world.describe()).render(cam)
    compare("glass-shells", two, ref)
    scene.close()


def _structural_csg_world():
    """CSG nodes whose l_hit = left.includes(shape) (compound.rs:154: structural equality, shape.rs:18-23) does not follow
    the operands: a copy of the left sphere sitting in the right operand counts as a LEFT hit, a plane in the left operand
    (never equal to itself: its bounds are infinite) as a RIGHT one.  The reference then reports hits the 'no left entry,
    nothing passes' shortcuts would drop: hits of the displaced copy on rays that miss the left sphere, hits of the left
    sphere on rays that miss the right cube."""
    from raytracer_amd import Material, Matrix, PointLight, Primitive, World
    red, green, blue = Material(color=(0.9, 0.2, 0.2)), Material(color=(0.2, 0.9, 0.2)), Material(color=(0.2, 0.3, 0.9), reflective=0.2)
    ball = lambda m: Primitive.new_sphere().with_material(m)
    # difference(S, group(translate 5){S})
    a = rt.Compound.new_csg(rt.Operation.Difference, ball(red),
                            rt.Compound.new_group([ball(red)]).with_transform(Matrix.translation(3.0, 0.0, 0.0)))
    # intersection(S, group(translate){S, cube})
    b = rt.Compound.new_csg(rt.Operation.Intersection, ball(green),
                            rt.Compound.new_group([ball(green), Primitive.new_cube().with_transform(Matrix.scaling(0.6, 0.6, 0.6))])
                            .with_transform(Matrix.translation(0.0, 2.6, 0.0))).with_transform(Matrix.translation(-4.0, 0.0, 0.0))
    # intersection(group{sphere, plane}, cube): the plane's entries toggle in_r
    c = rt.Compound.new_csg(rt.Operation.Intersection,
                            rt.Compound.new_group([ball(blue), Primitive.new_plane().with_transform(Matrix.translation(0.0, -0.3, 0.0))]),
                            Primitive.new_cube().with_transform(Matrix.chain(Matrix.translation(2.5, 0.0, 0.0), Matrix.scaling(0.5, 0.5, 0.5)))
                            ).with_transform(Matrix.translation(0.0, 0.0, 5.0))
    floor = Primitive.new_plane().with_transform(Matrix.translation(0.0, -3.0, 0.0)).with_material(Material(color=(0.7, 0.7, 0.7)))
    world = World([a, b, c, floor], [PointLight((-6.0, 9.0, -10.0), (1.0, 1.0, 1.0))])
    cam = rt.Camera(320, 200, 1.1).with_transform(Matrix.view_transform((0.5, 2.0, -11.0), (0.0, 0.5, 1.0), (0.0, 1.0, 0.0)))
    return world, cam


@pytest.mark.parametrize("hierarchy", [True, False, "nested", "global"])
def test_csg_left_hits_are_structural(hierarchy):
    """the right operand is skipped / a CSG node takes one operand's box only when no primitive of the right
    operand is a left hit (and, for the right operand's box, every primitive of the left operand is one)."""
    world, cam = _structural_csg_world()
    ow = orc.OracleWorld(world.describe())
    ref, _ = ow.render(cam)
    flat = world.flatten(hierarchy)
    scene = rt.DeviceScene(flat)
    gpu = _render_resident(scene, cam, 0)
    compare(f"structural-csg-{hierarchy}", gpu, ref)
    scene.close()
    # the displaced copies are really seen (the shortcut would have dropped them): rays aimed at them hit something
    rays = np.array([[3.0, 0.0, -10.0, 0.0, 0.0, 1.0],      # the copy of a's sphere, 3 to the right: misses the left sphere
                     [-4.0, 2.6, -10.0, 0.0, 0.0, 1.0],     # b's displaced group
                     [0.0, 0.5, -10.0, 0.0, 0.0, 1.0]])     # c's sphere above its plane, far from the cube
    got, want = world.probe(rays), ow.probe(rays)
    assert np.array_equal(got[:, [0, 2]], want[:, [0, 2]])
    assert not rel_mismatch(got, want, 1e-12, 1e-15).any()
    assert (want[:, 0] >= 0).any()
