"""Python face of the host mirror: the reference's scene-building API
(src/lib.rs:11-45) with `Camera.render` / `World.color_at` running on the GPU.

Names follow the reference: Matrix, Pattern, Material, Primitive, Compound, Operation,
Conic, PointLight, World, Camera, Canvas, obj.parse_file.  Builder methods return new
objects (`with_transform(self) -> Self` in the reference).  All scene logic lives in
librt_host.so (C++); this module only holds handles.
"""
import ctypes as C

import numpy as np

from . import _native as N


def _dp(a):
    return a.ctypes.data_as(N.c_double_p)


def _vec3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


class Matrix:
    """Matrix<4> builders (src/core/matrix.rs:153-230); values are 4x4 float64 arrays."""

    @staticmethod
    def _out(fn, *args):
        m = np.empty(16, dtype=np.float64)
        fn(*args, _dp(m))
        return m.reshape(4, 4)

    @staticmethod
    def identity():
        return Matrix._out(N.host.rth_matrix_identity)

    @staticmethod
    def translation(x, y, z):
        return Matrix._out(N.host.rth_matrix_translation, x, y, z)

    @staticmethod
    def scaling(x, y, z):
        return Matrix._out(N.host.rth_matrix_scaling, x, y, z)

    @staticmethod
    def rotation_x(r):
        return Matrix._out(N.host.rth_matrix_rotation_x, r)

    @staticmethod
    def rotation_y(r):
        return Matrix._out(N.host.rth_matrix_rotation_y, r)

    @staticmethod
    def rotation_z(r):
        return Matrix._out(N.host.rth_matrix_rotation_z, r)

    @staticmethod
    def shearing(xy, xz, yx, yz, zx, zy):
        return Matrix._out(N.host.rth_matrix_shearing, xy, xz, yx, yz, zx, zy)

    @staticmethod
    def view_transform(frm, to, up):
        return Matrix._out(N.host.rth_matrix_view_transform, _dp(_vec3(frm)), _dp(_vec3(to)), _dp(_vec3(up)))

    @staticmethod
    def mul(a, b):
        """a * b with the reference's i-k-j accumulation order (matrix.rs:268-284)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return Matrix._out(N.host.rth_matrix_mul, _dp(a), _dp(b))

    @staticmethod
    def chain(*ms):
        out = ms[0]
        for m in ms[1:]:
            out = Matrix.mul(out, m)
        return out

    @staticmethod
    def inverse(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        m = np.empty(16, dtype=np.float64)
        if N.host.rth_matrix_inverse(_dp(a), _dp(m)) != 0:
            raise ValueError("matrix is not invertible")  # the reference asserts (matrix.rs:94)
        return m.reshape(4, 4)

    @staticmethod
    def transpose(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return Matrix._out(N.host.rth_matrix_transpose, _dp(a))

    @staticmethod
    def determinant(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return N.host.rth_matrix_determinant(_dp(a))


class _Handle:
    _free = None

    def __init__(self, ptr, what):
        if not ptr:
            raise RuntimeError(f"{what}: {N.host.rth_last_error().decode()}")
        self._p = C.c_void_p(ptr)

    def __del__(self):
        p, self._p = getattr(self, "_p", None), None
        if p and type(self)._free is not None:
            type(self)._free(p)


class Pattern(_Handle):
    """src/graphics/pattern.rs:29-74"""
    _free = N.host.rth_pattern_free

    @staticmethod
    def new_solid(color):
        r, g, b = color
        return Pattern(N.host.rth_pattern_solid(r, g, b), "Pattern::new_solid")

    @staticmethod
    def _mix(kind, a, b):
        return Pattern(N.host.rth_pattern_mixture(kind, a._p, b._p), "Pattern::mixture")

    @staticmethod
    def new_blend(a, b):
        return Pattern._mix(N.RT_PAT_BLEND, a, b)

    @staticmethod
    def new_stripe(a, b):
        return Pattern._mix(N.RT_PAT_STRIPE, a, b)

    @staticmethod
    def new_gradient(a, b):
        return Pattern._mix(N.RT_PAT_GRADIENT, a, b)

    @staticmethod
    def new_radial_gradient(a, b):
        return Pattern._mix(N.RT_PAT_RADIAL_GRADIENT, a, b)

    @staticmethod
    def new_ring(a, b):
        return Pattern._mix(N.RT_PAT_RING, a, b)

    @staticmethod
    def new_checker(a, b):
        return Pattern._mix(N.RT_PAT_CHECKER, a, b)

    @staticmethod
    def new_perturb(p):
        return Pattern(N.host.rth_pattern_perturb(p._p), "Pattern::new_perturb")

    @staticmethod
    def new_test():
        return Pattern(N.host.rth_pattern_test(), "Pattern::new_test")

    def with_transform(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64)
        return Pattern(N.host.rth_pattern_with_transform(self._p, _dp(m)), "Pattern::with_transform")


class Material:
    """src/raytracer/material.rs:6-32 (same defaults)."""

    def __init__(self, color=(1.0, 1.0, 1.0), ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0, reflective=0.0,
                 transparency=0.0, refractive_index=1.0, pattern=None):
        self.color = tuple(color)
        self.ambient, self.diffuse, self.specular, self.shininess = ambient, diffuse, specular, shininess
        self.reflective, self.transparency, self.refractive_index = reflective, transparency, refractive_index
        self.pattern = pattern

    def _c(self):
        m = N.RthMaterial()
        m.color[:] = self.color
        m.ambient, m.diffuse, m.specular, m.shininess = self.ambient, self.diffuse, self.specular, self.shininess
        m.reflective, m.transparency, m.refractive_index = self.reflective, self.transparency, self.refractive_index
        return m


class Conic:
    """primitive.rs:453-471; default = unbounded, open."""

    def __init__(self, minimum=-np.inf, maximum=np.inf, closed=False):
        self.minimum, self.maximum, self.closed = minimum, maximum, closed


class Operation:
    """compound.rs:135-139"""
    Union, Intersection, Difference = 0, 1, 2


class Shape(_Handle):
    """Primitive or Compound (shape.rs:10-24) with the SetProperties builders (shape.rs:62-73)."""
    _free = N.host.rth_shape_free

    def with_transform(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64)
        return Shape(N.host.rth_shape_with_transform(self._p, _dp(m)), "with_transform")

    def with_material(self, material):
        cm = material._c()
        pat = material.pattern._p if material.pattern is not None else None
        return Shape(N.host.rth_shape_with_material(self._p, C.byref(cm), pat), "with_material")

    def with_shadow(self, has_shadow):
        return Shape(N.host.rth_shape_with_shadow(self._p, 1 if has_shadow else 0), "with_shadow")

    def to_shape(self):
        return self

    def includes(self, primitive):
        return N.host.rth_shape_includes(self._p, primitive._p) == 1

    @property
    def kind(self):
        return N.host.rth_shape_kind(self._p)

    @property
    def bounds(self):
        b = np.empty(6, dtype=np.float64)
        N.host.rth_shape_bounds(self._p, _dp(b))
        return b[:3].copy(), b[3:].copy()

    def shade_rays(self, light, rays):
        """exercises/shading.rs:44-57 per ray, on the GPU: `intersect` -> `hit` -> `normal_at` -> `lighting` of this
        shape (no shadow test, normal as normal_at returns it).  rays [n, 6] -> (colours [n, 3], hit mask [n])."""
        r = np.ascontiguousarray(np.asarray(rays, dtype=np.float64).reshape(-1, 6))
        l = np.array(list(light.position) + list(light.intensity), dtype=np.float64)
        out = np.zeros((r.shape[0], 3), dtype=np.float64)
        hit = np.zeros(r.shape[0], dtype=np.uint8)
        N.check(N.host.rth_shape_shade_rays(self._p, _dp(l), _dp(r), r.shape[0], _dp(out),
                                            hit.ctypes.data_as(C.POINTER(C.c_uint8))), "shade_rays")
        return out, hit.astype(bool)

    def children(self):
        return [Shape(N.host.rth_shape_child(self._p, i), "child") for i in range(N.host.rth_shape_num_children(self._p))]


class Primitive:
    """primitive.rs:52-108"""

    @staticmethod
    def new_sphere():
        return Shape(N.host.rth_sphere(), "new_sphere")

    @staticmethod
    def new_glass_sphere():
        return Shape(N.host.rth_glass_sphere(), "new_glass_sphere")

    @staticmethod
    def new_plane():
        return Shape(N.host.rth_plane(), "new_plane")

    @staticmethod
    def new_cube():
        return Shape(N.host.rth_cube(), "new_cube")

    @staticmethod
    def new_cylinder(conic=None):
        c = conic or Conic()
        return Shape(N.host.rth_cylinder(c.minimum, c.maximum, 1 if c.closed else 0), "new_cylinder")

    @staticmethod
    def new_cone(conic=None):
        c = conic or Conic()
        return Shape(N.host.rth_cone(c.minimum, c.maximum, 1 if c.closed else 0), "new_cone")

    @staticmethod
    def new_triangle(p1, p2, p3):
        return Shape(N.host.rth_triangle(_dp(_vec3(p1)), _dp(_vec3(p2)), _dp(_vec3(p3))), "new_triangle")

    @staticmethod
    def new_smooth_triangle(p1, p2, p3, n1, n2, n3):
        a = [_vec3(v) for v in (p1, p2, p3, n1, n2, n3)]
        return Shape(N.host.rth_smooth_triangle(*[_dp(v) for v in a]), "new_smooth_triangle")


class Compound:
    """compound.rs:52-62"""

    @staticmethod
    def new_group(children):
        arr = (C.c_void_p * len(children))(*[c._p for c in children])
        return Shape(N.host.rth_group(arr, len(children)), "new_group")

    @staticmethod
    def new_csg(operation, left, right):
        return Shape(N.host.rth_csg(operation, left._p, right._p), "new_csg")


class PointLight:
    """point_light.rs:5-17"""

    def __init__(self, position, intensity):
        self.position, self.intensity = tuple(position), tuple(intensity)


class Canvas:
    """src/graphics/canvas.rs: row-major colours, 8-bit P3 PPM writer."""

    def __init__(self, width, height, pixels=None):
        self.width, self.height = width, height
        self.pixels = np.zeros((height, width, 3)) if pixels is None else np.asarray(pixels).reshape(height, width, 3)

    def write_pixel(self, x, y, color):
        self.pixels[y, x] = color

    def pixel_at(self, x, y):
        return tuple(self.pixels[y, x])

    def quantized(self):
        """scale_to_ppm_data (canvas.rs:85-90) on every channel -> uint8 [h, w, 3]."""
        src = np.ascontiguousarray(self.pixels, dtype=np.float64)
        out = np.empty(src.size, dtype=np.uint8)
        N.host.rth_canvas_quantize(_dp(src), src.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out.reshape(self.height, self.width, 3)

    def to_ppm(self):
        src = np.ascontiguousarray(self.pixels, dtype=np.float64)
        n = N.host.rth_canvas_to_ppm(_dp(src), self.width, self.height, None, 0)
        buf = C.create_string_buffer(n)
        N.host.rth_canvas_to_ppm(_dp(src), self.width, self.height, buf, n)
        return buf.raw

    def write_ppm(self, path):
        with open(path, "wb") as f:
            f.write(self.to_ppm())

    @staticmethod
    def ppm_from_rgb8(rgb8):
        """Canvas::to_ppm (canvas.rs:46-83) from uint8 samples [h, w, 3] that scale_to_ppm_data already produced."""
        src = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w = src.shape[0], src.shape[1]
        p = src.ctypes.data_as(C.POINTER(C.c_uint8))
        buf = np.empty(64 + h * w * 12, dtype=np.uint8)  # header + at most 4 bytes per sample: one formatting pass
        n = N.host.rth_ppm_from_rgb8(p, w, h, buf.ctypes.data_as(C.c_char_p), buf.size)
        return buf[:n].tobytes()


class World(_Handle):
    """src/raytracer/world.rs:17-43,130-145"""
    _free = N.host.rth_world_free

    def __init__(self, shapes=(), lights=(), _ptr=None):
        if _ptr is not None:
            super().__init__(_ptr, "World")
            return
        shapes = list(shapes)
        arr = (C.c_void_p * max(1, len(shapes)))(*[s._p for s in shapes])
        l6 = np.ascontiguousarray([list(l.position) + list(l.intensity) for l in lights], dtype=np.float64).reshape(-1)
        if l6.size == 0:
            l6 = np.zeros(6)
        super().__init__(N.host.rth_world_new(arr, len(shapes), _dp(l6), len(lights)), "World::new")

    @staticmethod
    def new(shapes, lights):
        return World(shapes, lights)

    @staticmethod
    def empty():
        return World((), ())

    @staticmethod
    def default():
        return World(_ptr=N.host.rth_world_default())

    def color_at(self, rays, remaining_recursions=5):
        """World::color_at (world.rs:32) for one ray (6 values) or a batch [n, 6]; runs on the GPU."""
        r = np.ascontiguousarray(rays, dtype=np.float64)
        single = r.ndim == 1
        r = r.reshape(-1, 6)
        out = np.empty((r.shape[0], 3), dtype=np.float64)
        N.check(N.host.rth_world_color_at(self._p, _dp(r), r.shape[0], remaining_recursions, _dp(out)), "World::color_at")
        return out[0] if single else out

    def probe(self, rays):
        """Computations of the first stage of color_at per ray (rt_hip_probe); runs on the GPU."""
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.empty((r.shape[0], 16), dtype=np.float64)
        N.check(N.host.rth_world_probe(self._p, _dp(r), r.shape[0], _dp(out)), "World::probe")
        return out

    def describe(self):
        """The object tree as the user built it, for an independent checker (tests only)."""
        n = N.host.rth_world_describe(self._p, None, 0)
        out = np.empty(n, dtype=np.float64)
        N.host.rth_world_describe(self._p, _dp(out), n)
        return out

    def flatten(self, build_bvh=True, host_cells=False):
        return FlatScene(self, build_bvh, host_cells)


class FlatScene(_Handle):
    """The flat record tables of include/rt_hip.h for one world (host memory)."""
    _free = N.host.rth_flat_free

    def __init__(self, world, build_bvh=True, host_cells=False):
        # build_bvh: False / True (by policy), or "nested" / "global" to force one form of hierarchy
        # host_cells: the direction index's cells rasterised by the flattener even for many candidates (otherwise
        # scenes with >= 64 candidate conics leave them to rt_hip_scene_create, which builds them on the device)
        mode = {"nested": 2, "global": 3}.get(build_bvh, 1 if build_bvh else 0) | (8 if host_cells else 0)
        super().__init__(N.host.rth_world_flatten(world._p, mode), "flatten")
        self.desc = N.host.rth_flat_desc(self._p)  # POINTER(RtSceneDesc), valid while self lives


class DeviceScene:
    """A scene resident on the current HIP device (rt_hip_scene_create / _destroy)."""

    def __init__(self, flat):
        self._flat = flat
        p = C.c_void_p()
        N.check(N.hip.rt_hip_scene_create(flat.desc, C.byref(p)), "rt_hip_scene_create")
        self._p = p

    def close(self):
        if self._p and N is not None and getattr(N, "hip", None) is not None:  # (module globals are gone at interpreter exit)
            N.hip.rt_hip_scene_destroy(self._p)
        self._p = None

    def __del__(self):
        self.close()

    def render_rows(self, cam_desc, depth, part, n_parts, band_rows, n_local_rows, d_out_ptr, stream=None, d_stats_ptr=None):
        """rt_hip_render_rows: asynchronous, device output pointer (an int address)."""
        N.check(N.hip.rt_hip_render_rows(self._p, C.byref(cam_desc), depth, part, n_parts, band_rows, n_local_rows,
                                         C.c_void_p(d_out_ptr), C.c_void_p(stream or 0), C.c_void_p(d_stats_ptr or 0)),
                "rt_hip_render_rows")

    def render_rows_noalloc(self, cam_desc, depth, part, n_parts, band_rows, n_local_rows, d_out_ptr, stream=None, d_stats_ptr=None):
        """rt_hip_render_rows_noalloc: the same call with no device allocation on any path."""
        N.check(N.hip.rt_hip_render_rows_noalloc(self._p, C.byref(cam_desc), depth, part, n_parts, band_rows, n_local_rows,
                                                 C.c_void_p(d_out_ptr), C.c_void_p(stream or 0), C.c_void_p(d_stats_ptr or 0)),
                "rt_hip_render_rows_noalloc")

    def render(self, camera, stats=False):
        """`camera.render(&world)` on the resident scene (rt_hip_scene_render): Canvas of f64 colours."""
        out = np.empty((camera.v_size, camera.h_size, 3), dtype=np.float64)
        cd = camera.desc()
        st = N.RtStats() if stats else None
        N.check(N.hip.rt_hip_scene_render(self._p, C.byref(cd), Camera.DEPTH, _dp(out), C.byref(st) if stats else None),
                "rt_hip_scene_render")
        canvas = Canvas(camera.h_size, camera.v_size, out)
        return (canvas, st) if stats else canvas

    def render_rgb8(self, camera):
        """... as the 8-bit samples of Canvas::to_ppm, quantised on the device -> uint8 [v, h, 3]."""
        out = np.empty((camera.v_size, camera.h_size, 3), dtype=np.uint8)
        cd = camera.desc()
        N.check(N.hip.rt_hip_scene_render_rgb8(self._p, C.byref(cd), Camera.DEPTH, out.ctypes.data_as(C.POINTER(C.c_uint8)), None),
                "rt_hip_scene_render_rgb8")
        return out

    def shade_rays(self, light, rays):
        """rt_hip_shade_rays on the resident scene (exercises/shading.rs:44-57 per ray) -> (colours [n, 3], hit mask [n])."""
        r = np.ascontiguousarray(np.asarray(rays, dtype=np.float64).reshape(-1, 6))
        l = N.RtLight()
        l.position[:] = list(light.position)
        l.intensity[:] = list(light.intensity)
        out = np.zeros((r.shape[0], 3), dtype=np.float64)
        hit = np.zeros(r.shape[0], dtype=np.uint8)
        N.check(N.hip.rt_hip_shade_rays(self._p, C.byref(l), _dp(r), r.shape[0], _dp(out),
                                        hit.ctypes.data_as(C.POINTER(C.c_uint8))), "rt_hip_shade_rays")
        return out, hit.astype(bool)

    def pipeline_name(self, depth=5):
        """rt_hip_scene_kernels: which kernels a render of this scene runs."""
        return N.hip.rt_hip_scene_kernels(self._p, depth).decode()

    def set_wavefront(self, mode=-1, queue_items=0, node_items=0):
        """rt_hip_scene_set_wavefront: -1 by policy, 0 one launch, 1 one launch per generation of rays, 2 the kernel pipeline."""
        N.check(N.hip.rt_hip_scene_set_wavefront(self._p, mode, queue_items, node_items), "rt_hip_scene_set_wavefront")

    def status(self, stream=None):
        N.check(N.hip.rt_hip_scene_status(self._p, C.c_void_p(stream or 0)), "rt_hip_scene_status")


class Camera:
    """src/raytracer/camera.rs:13-66; `render` runs on the GPU (recursion budget 5, camera.rs:59)."""
    DEPTH = 5

    def __init__(self, h_size, v_size, field_of_view, transform=None):
        self.h_size, self.v_size, self.field_of_view = int(h_size), int(v_size), float(field_of_view)
        self.transform = None if transform is None else np.ascontiguousarray(transform, dtype=np.float64)

    def with_transform(self, m):
        return Camera(self.h_size, self.v_size, self.field_of_view, m)

    def _c(self):
        c = N.RthCamera()
        c.h_size, c.v_size, c.field_of_view = self.h_size, self.v_size, self.field_of_view
        t = np.eye(4) if self.transform is None else self.transform
        c.transform[:] = list(np.asarray(t, dtype=np.float64).reshape(16))
        c.transform_set = 0 if self.transform is None else 1
        return c

    @staticmethod
    def _from_c(c):
        t = np.array(list(c.transform), dtype=np.float64).reshape(4, 4) if c.transform_set else None
        return Camera(c.h_size, c.v_size, c.field_of_view, t)

    def desc(self):
        """Camera::new / with_transform cached values as the RtCameraDesc the kernels take."""
        cd = N.RtCameraDesc()
        cc = self._c()
        N.check(N.host.rth_camera_desc(C.byref(cc), C.byref(cd)), "Camera::desc")
        return cd

    def render(self, world, stats=False):
        out = np.empty((self.v_size, self.h_size, 3), dtype=np.float64)
        cc = self._c()
        st = N.RtStats() if stats else None
        N.check(N.host.rth_camera_render(C.byref(cc), world._p, _dp(out), C.byref(st) if stats else None),
                "Camera::render")
        canvas = Canvas(self.h_size, self.v_size, out)
        return (canvas, st) if stats else canvas

    def render_multi(self, world, n_devices, stats=False):
        """`render(world)` with the rows dealt over the first n_devices HIP devices of this process (rt_hip_render_multi)."""
        out = np.empty((self.v_size, self.h_size, 3), dtype=np.float64)
        flat = world.flatten()
        cd = self.desc()
        st = N.RtStats() if stats else None
        N.check(N.hip.rt_hip_render_multi(flat.desc, C.byref(cd), Camera.DEPTH, n_devices, _dp(out), C.byref(st) if stats else None),
                "rt_hip_render_multi")
        canvas = Canvas(self.h_size, self.v_size, out)
        return (canvas, st) if stats else canvas

    def render_rgb8(self, world):
        """`render(world)` + scale_to_ppm_data per channel (canvas.rs:85-90) on the device -> uint8 [v, h, 3]."""
        out = np.empty((self.v_size, self.h_size, 3), dtype=np.uint8)
        cc = self._c()
        N.check(N.host.rth_camera_render_rgb8(C.byref(cc), world._p, out.ctypes.data_as(C.POINTER(C.c_uint8)), None),
                "Camera::render")
        return out

    def render_ppm(self, world):
        """`camera.render(&world).to_ppm()` as bytes; only the 8-bit samples leave the device."""
        return Canvas.ppm_from_rgb8(self.render_rgb8(world))


class obj:
    """src/io/obj.rs:55-112"""

    @staticmethod
    def parse_file(path):
        return Shape(N.host.rth_obj_parse_file(str(path).encode()), "obj::parse_file")

    @staticmethod
    def parse_string(text):
        ignored = C.c_size_t(0)
        s = Shape(N.host.rth_obj_parse_string(text.encode(), C.byref(ignored)), "obj::parse_string")
        return s, ignored.value


EXERCISES = ("plane", "scene", "pattern", "glass_bubble", "teapot", "hexagon", "csg", "cover", "room")


def exercise(name, width=2048, height=1080, resource_dir=None):
    """World and Camera of one of the reference's render exercises (exercises/<name>.rs)."""
    import os
    rd = resource_dir or os.path.join(N.REPO_ROOT, "tests", "golden")
    w = C.c_void_p()
    cam = N.RthCamera()
    N.check(N.host.rth_exercise(name.encode(), width, height, rd.encode(), C.byref(w), C.byref(cam)), f"exercise {name}")
    return World(_ptr=w.value), Camera._from_c(cam)


def synthetic(width=7680, height=4320, n_primitives=10000, seed=0x5EED5EED):
    """BASELINE.json config 5: cover.rs/room.rs-style scene, SplitMix64-seeded."""
    w = C.c_void_p()
    cam = N.RthCamera()
    N.check(N.host.rth_synthetic(width, height, n_primitives, seed, C.byref(w), C.byref(cam)), "synthetic")
    return World(_ptr=w.value), Camera._from_c(cam)


def exercise_shading():
    """exercises/shading.rs whole (its own 500 x 500 pixel loop over intersect / normal_at / lighting): Canvas."""
    out = np.empty((500, 500, 3), dtype=np.float64)
    N.check(N.host.rth_exercise_shading(_dp(out)), "exercise shading")
    return Canvas(500, 500, out)
