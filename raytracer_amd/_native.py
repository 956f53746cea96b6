"""ctypes bindings of the two in-tree libraries.

librt_hip.so  -- HIP kernels + the C ABI of include/rt_hip.h (the drop-in boundary)
librt_host.so -- C++ mirror of the reference's scene-building API (include/rt_host.h)

Nothing here computes anything: if the libraries are missing the import fails loudly,
and every render call goes to the GPU through librt_hip.so (there is no CPU path).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB_DIR: another build of the two libraries (kernel variants compiled with other -D settings, for studies)
LIB_DIR = os.environ.get("RT_HIP_LIB_DIR") or os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make` (or __graft_entry__.build()) -- "
            "raytracer_amd has no fallback path"
        )
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load("librt_hip.so")
host = _load("librt_host.so")

RT_MAX_NEST = 8
RT_MAX_DEPTH = 8

(RT_SPHERE, RT_PLANE, RT_CUBE, RT_CYLINDER, RT_CONE, RT_TRIANGLE, RT_SMOOTH_TRIANGLE, RT_GROUP, RT_CSG) = range(9)
(RT_PAT_SOLID, RT_PAT_PERTURB, RT_PAT_BLEND, RT_PAT_STRIPE, RT_PAT_GRADIENT, RT_PAT_RADIAL_GRADIENT, RT_PAT_RING,
 RT_PAT_CHECKER, RT_PAT_TEST) = range(9)

RT_OK = 0
RT_ERR_NO_DEVICE = -1
RT_ERR_HIP = -2
RT_ERR_INVALID_SCENE = -3
RT_ERR_NEST_TOO_DEEP = -4
RT_ERR_LIST_BOUND = -5
RT_ERR_LIST_OVERFLOW = -6
RT_ERR_NAN_T = -7
RT_ERR_INVALID_ARG = -8
RT_ERR_DEPTH = -9
RT_ERR_NOALLOC_SIZE = -10

c_double_p = C.POINTER(C.c_double)


class RtShape(C.Structure):
    _fields_ = [("inv", C.c_double * 12), ("kind_flags", C.c_uint32), ("parent", C.c_int32), ("skip", C.c_uint32),
                ("material", C.c_uint32), ("param", C.c_uint32), ("depth", C.c_uint32), ("pad", C.c_uint32 * 2)]


class RtConic(C.Structure):
    _fields_ = [("minimum", C.c_double), ("maximum", C.c_double)]


class RtTriangle(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("p1", "e1", "e2", "n1", "n2", "n3")]


class RtGroup(C.Structure):
    _fields_ = [("bmin", C.c_double * 3), ("bmax", C.c_double * 3), ("bvh_root", C.c_int32), ("n_children", C.c_uint32)]


class RtCsg(C.Structure):
    _fields_ = [("op", C.c_uint32), ("lhit_offset", C.c_uint32)]


class RtMaterial(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double), ("pattern", C.c_int32), ("pad", C.c_int32), ("pad2", C.c_double)]


class RtPattern(C.Structure):
    _fields_ = [("inv", C.c_double * 12), ("color", C.c_double * 3), ("kind", C.c_uint32), ("a", C.c_int32),
                ("b", C.c_int32), ("pad", C.c_uint32)]


class RtLight(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("intensity", C.c_double * 3)]


class RtBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("miss", C.c_int32), ("shape", C.c_int32)]


class RtDirCand(C.Structure):
    _fields_ = [("q", C.c_double * 6), ("thr", C.c_double), ("shape", C.c_uint32), ("visit", C.c_uint32)]


class RtSceneDesc(C.Structure):
    _fields_ = [("n_shapes", C.c_uint32), ("shapes", C.POINTER(RtShape)),
                ("n_conics", C.c_uint32), ("conics", C.POINTER(RtConic)),
                ("n_triangles", C.c_uint32), ("triangles", C.POINTER(RtTriangle)),
                ("n_groups", C.c_uint32), ("groups", C.POINTER(RtGroup)),
                ("n_csgs", C.c_uint32), ("csgs", C.POINTER(RtCsg)),
                ("n_lhit", C.c_uint32), ("lhit", C.POINTER(C.c_uint8)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(RtMaterial)),
                ("n_patterns", C.c_uint32), ("patterns", C.POINTER(RtPattern)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(RtLight)),
                ("n_bvh_nodes", C.c_uint32), ("bvh_nodes", C.POINTER(RtBvhNode)),
                ("world_bvh_root", C.c_int32), ("dir_res", C.c_uint32),
                ("n_dir_cands", C.c_uint32), ("dir_cands", C.POINTER(RtDirCand)),
                ("n_dir_entries", C.c_uint32), ("dir_cell_start", C.POINTER(C.c_uint32)),
                ("dir_entries", C.POINTER(C.c_uint32))]


class RtCameraDesc(C.Structure):
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("half_width", C.c_double), ("half_height", C.c_double),
                ("pixel_size", C.c_double), ("inv", C.c_double * 16)]


class RtStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("prim_tests", C.c_uint64 * 7), ("ray_transforms", C.c_uint64),
                ("group_gates", C.c_uint64), ("bvh_box_tests", C.c_uint64), ("hits_shaded", C.c_uint64),
                ("normal_levels", C.c_uint64), ("lightings", C.c_uint64), ("pow_calls", C.c_uint64),
                ("pattern_nodes", C.c_uint64), ("container_passes", C.c_uint64), ("list_entries", C.c_uint64),
                ("schlicks", C.c_uint64), ("accel_f32_tests", C.c_uint64), ("status_flags", C.c_uint64), ("kernel_ms", C.c_double)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "prim_tests" else v
        return d


class RthMaterial(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double)]


class RthCamera(C.Structure):
    _fields_ = [("h_size", C.c_uint64), ("v_size", C.c_uint64), ("field_of_view", C.c_double),
                ("transform", C.c_double * 16), ("transform_set", C.c_int32), ("pad", C.c_int32)]


# Every symbol include/rt_hip.h declares (checked by tests/test_capi_symbols.py).
HIP_SYMBOLS = ["rt_hip_init", "rt_hip_shutdown", "rt_hip_strerror", "rt_hip_last_hip_error", "rt_hip_scene_create",
               "rt_hip_scene_destroy", "rt_hip_scene_kernels", "rt_hip_scene_render", "rt_hip_scene_render_rgb8", "rt_hip_scene_set_wavefront", "rt_hip_scene_status", "rt_hip_render", "rt_hip_render_multi", "rt_hip_multi_transport", "rt_hip_render_rgb8", "rt_hip_render_rows", "rt_hip_render_rows_noalloc",
               "rt_hip_color_at", "rt_hip_shade_rays",
               "rt_hip_probe", "rt_hip_quantize"]

_vp = C.c_void_p


def _sig(lib, name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig(hip, "rt_hip_init", C.c_int, C.c_int)
_sig(hip, "rt_hip_shutdown", None)
_sig(hip, "rt_hip_strerror", C.c_char_p, C.c_int)
_sig(hip, "rt_hip_last_hip_error", C.c_char_p)
_sig(hip, "rt_hip_scene_create", C.c_int, C.POINTER(RtSceneDesc), C.POINTER(_vp))
_sig(hip, "rt_hip_scene_destroy", None, _vp)
_sig(hip, "rt_hip_scene_set_wavefront", C.c_int, _vp, C.c_int, C.c_uint64, C.c_uint64)
_sig(hip, "rt_hip_scene_status", C.c_int, _vp, _vp)
_sig(hip, "rt_hip_scene_kernels", C.c_char_p, _vp, C.c_uint32)
_sig(hip, "rt_hip_render", C.c_int, C.POINTER(RtSceneDesc), C.POINTER(RtCameraDesc), C.c_uint32, c_double_p,
     C.POINTER(RtStats))
_sig(hip, "rt_hip_render_multi", C.c_int, C.POINTER(RtSceneDesc), C.POINTER(RtCameraDesc), C.c_uint32, C.c_uint32, c_double_p,
     C.POINTER(RtStats))
_sig(hip, "rt_hip_multi_transport", C.c_char_p)
_sig(hip, "rt_hip_render_rgb8", C.c_int, C.POINTER(RtSceneDesc), C.POINTER(RtCameraDesc), C.c_uint32,
     C.POINTER(C.c_uint8), C.POINTER(RtStats))
_sig(hip, "rt_hip_scene_render", C.c_int, _vp, C.POINTER(RtCameraDesc), C.c_uint32, c_double_p, C.POINTER(RtStats))
_sig(hip, "rt_hip_scene_render_rgb8", C.c_int, _vp, C.POINTER(RtCameraDesc), C.c_uint32, C.POINTER(C.c_uint8),
     C.POINTER(RtStats))
_sig(hip, "rt_hip_render_rows", C.c_int, _vp, C.POINTER(RtCameraDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
     C.c_uint32, _vp, _vp, _vp)
_sig(hip, "rt_hip_render_rows_noalloc", C.c_int, _vp, C.POINTER(RtCameraDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
     C.c_uint32, _vp, _vp, _vp)
_sig(hip, "rt_hip_color_at", C.c_int, _vp, c_double_p, C.c_uint32, C.c_uint32, c_double_p)
_sig(hip, "rt_hip_shade_rays", C.c_int, _vp, C.POINTER(RtLight), c_double_p, C.c_uint32, c_double_p, C.POINTER(C.c_uint8))
_sig(hip, "rt_hip_probe", C.c_int, _vp, c_double_p, C.c_uint32, c_double_p)
_sig(hip, "rt_hip_quantize", C.c_int, _vp, C.c_uint64, _vp, _vp)

_sig(host, "rth_last_error", C.c_char_p)
for _n in ("identity",):
    _sig(host, "rth_matrix_" + _n, None, c_double_p)
_sig(host, "rth_matrix_translation", None, C.c_double, C.c_double, C.c_double, c_double_p)
_sig(host, "rth_matrix_scaling", None, C.c_double, C.c_double, C.c_double, c_double_p)
_sig(host, "rth_matrix_rotation_x", None, C.c_double, c_double_p)
_sig(host, "rth_matrix_rotation_y", None, C.c_double, c_double_p)
_sig(host, "rth_matrix_rotation_z", None, C.c_double, c_double_p)
_sig(host, "rth_matrix_shearing", None, *([C.c_double] * 6), c_double_p)
_sig(host, "rth_matrix_view_transform", None, c_double_p, c_double_p, c_double_p, c_double_p)
_sig(host, "rth_matrix_mul", None, c_double_p, c_double_p, c_double_p)
_sig(host, "rth_matrix_inverse", C.c_int, c_double_p, c_double_p)
_sig(host, "rth_matrix_transpose", None, c_double_p, c_double_p)
_sig(host, "rth_matrix_determinant", C.c_double, c_double_p)

_sig(host, "rth_pattern_solid", _vp, C.c_double, C.c_double, C.c_double)
_sig(host, "rth_pattern_mixture", _vp, C.c_uint32, _vp, _vp)
_sig(host, "rth_pattern_perturb", _vp, _vp)
_sig(host, "rth_pattern_test", _vp)
_sig(host, "rth_pattern_with_transform", _vp, _vp, c_double_p)
_sig(host, "rth_pattern_free", None, _vp)
_sig(host, "rth_material_default", None, C.POINTER(RthMaterial))

_sig(host, "rth_sphere", _vp)
_sig(host, "rth_glass_sphere", _vp)
_sig(host, "rth_plane", _vp)
_sig(host, "rth_cube", _vp)
_sig(host, "rth_cylinder", _vp, C.c_double, C.c_double, C.c_int)
_sig(host, "rth_cone", _vp, C.c_double, C.c_double, C.c_int)
_sig(host, "rth_triangle", _vp, c_double_p, c_double_p, c_double_p)
_sig(host, "rth_smooth_triangle", _vp, *([c_double_p] * 6))
_sig(host, "rth_group", _vp, C.POINTER(_vp), C.c_size_t)
_sig(host, "rth_csg", _vp, C.c_uint32, _vp, _vp)
_sig(host, "rth_shape_with_transform", _vp, _vp, c_double_p)
_sig(host, "rth_shape_with_material", _vp, _vp, C.POINTER(RthMaterial), _vp)
_sig(host, "rth_shape_with_shadow", _vp, _vp, C.c_int)
_sig(host, "rth_shape_includes", C.c_int, _vp, _vp)
_sig(host, "rth_shape_kind", C.c_int, _vp)
_sig(host, "rth_shape_num_children", C.c_size_t, _vp)
_sig(host, "rth_shape_child", _vp, _vp, C.c_size_t)
_sig(host, "rth_shape_bounds", None, _vp, c_double_p)
_sig(host, "rth_shape_free", None, _vp)
_sig(host, "rth_obj_parse_file", _vp, C.c_char_p)
_sig(host, "rth_obj_parse_string", _vp, C.c_char_p, C.POINTER(C.c_size_t))

_sig(host, "rth_world_new", _vp, C.POINTER(_vp), C.c_size_t, c_double_p, C.c_size_t)
_sig(host, "rth_world_default", _vp)
_sig(host, "rth_world_free", None, _vp)
_sig(host, "rth_world_color_at", C.c_int, _vp, c_double_p, C.c_uint32, C.c_uint32, c_double_p)
_sig(host, "rth_world_probe", C.c_int, _vp, c_double_p, C.c_uint32, c_double_p)
_sig(host, "rth_shape_shade_rays", C.c_int, _vp, c_double_p, c_double_p, C.c_uint32, c_double_p, C.POINTER(C.c_uint8))
_sig(host, "rth_exercise_shading", C.c_int, c_double_p)
_sig(host, "rth_world_describe", C.c_size_t, _vp, c_double_p, C.c_size_t)
_sig(host, "rth_camera_desc", C.c_int, C.POINTER(RthCamera), C.POINTER(RtCameraDesc))
_sig(host, "rth_camera_render", C.c_int, C.POINTER(RthCamera), _vp, c_double_p, C.POINTER(RtStats))
_sig(host, "rth_camera_render_rgb8", C.c_int, C.POINTER(RthCamera), _vp, C.POINTER(C.c_uint8), C.POINTER(RtStats))
_sig(host, "rth_world_flatten", _vp, _vp, C.c_int)
_sig(host, "rth_flat_desc", C.POINTER(RtSceneDesc), _vp)
_sig(host, "rth_flat_free", None, _vp)
_sig(host, "rth_canvas_to_ppm", C.c_size_t, c_double_p, C.c_size_t, C.c_size_t, C.c_char_p, C.c_size_t)
_sig(host, "rth_ppm_from_rgb8", C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.c_size_t, C.c_char_p, C.c_size_t)
_sig(host, "rth_canvas_quantize", None, c_double_p, C.c_size_t, C.POINTER(C.c_uint8))
_sig(host, "rth_exercise", C.c_int, C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p, C.POINTER(_vp),
     C.POINTER(RthCamera))
_sig(host, "rth_synthetic", C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.POINTER(_vp),
     C.POINTER(RthCamera))


class RenderError(RuntimeError):
    """A call into librt_hip.so failed (mirrors rt::RenderError; the reference panics)."""

    def __init__(self, code, where):
        self.code = code
        text = hip.rt_hip_strerror(code).decode() if -20 < code <= 0 else "host error"
        detail = hip.rt_hip_last_hip_error().decode()
        host_detail = host.rth_last_error().decode()
        msg = f"{where}: {text} (code {code})"
        if detail:
            msg += f" [{detail}]"
        if host_detail:
            msg += f" [{host_detail}]"
        super().__init__(msg)


def check(code, where):
    if code != RT_OK:
        raise RenderError(code, where)
