// C handle API (include/rt_host.h) over the C++ host mirror.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rt_host.h"
#include "rt_exercises.h"
#include "rt_flatten.h"
#include "rt_scene.h"

using namespace rt;

struct RthPattern {
    Pattern p;
};
struct RthShape {
    Shape s;
};
struct RthWorld {
    World w;
};
struct RthFlat {
    FlatScene flat;
    RtSceneDesc desc;
};

namespace {
thread_local std::string g_error;

Matrix4 to_matrix(const double* m16) {
    Matrix4 m;
    std::memcpy(m.m, m16, sizeof m.m);
    return m;
}
void from_matrix(const Matrix4& m, double* out16) { std::memcpy(out16, m.m, sizeof m.m); }
Point pt(const double* p) { return Point(p[0], p[1], p[2]); }
Vector vec(const double* p) { return Vector(p[0], p[1], p[2]); }

template <class F>
auto guard(F&& f) -> decltype(f()) {
    try {
        g_error.clear();
        return f();
    } catch (const std::exception& e) {
        g_error = e.what();
    } catch (...) {
        g_error = "unknown error";
    }
    return decltype(f())();
}
template <class F>
int guard_int(F&& f) {
    try {
        g_error.clear();
        return f();
    } catch (const RenderError& e) {
        g_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_error = e.what();
    } catch (...) {
        g_error = "unknown error";
    }
    return -100;
}
Camera make_camera(const RthCamera& c) {
    Camera cam(static_cast<size_t>(c.h_size), static_cast<size_t>(c.v_size), c.field_of_view);
    return c.transform_set ? cam.with_transform(to_matrix(c.transform)) : cam;
}
void fill_camera(const Camera& cam, RthCamera* out) {
    std::memset(out, 0, sizeof *out);
    out->h_size = cam.h_size();
    out->v_size = cam.v_size();
    out->field_of_view = cam.field_of_view();
    from_matrix(cam.transform(), out->transform);
    out->transform_set = cam.transform_was_set() ? 1 : 0;
}
}  // namespace

extern "C" {

const char* rth_last_error(void) { return g_error.c_str(); }

void rth_matrix_identity(double* out16) { from_matrix(Matrix4::identity(), out16); }
void rth_matrix_translation(double x, double y, double z, double* o) { from_matrix(Matrix4::translation(x, y, z), o); }
void rth_matrix_scaling(double x, double y, double z, double* o) { from_matrix(Matrix4::scaling(x, y, z), o); }
void rth_matrix_rotation_x(double r, double* o) { from_matrix(Matrix4::rotation_x(r), o); }
void rth_matrix_rotation_y(double r, double* o) { from_matrix(Matrix4::rotation_y(r), o); }
void rth_matrix_rotation_z(double r, double* o) { from_matrix(Matrix4::rotation_z(r), o); }
void rth_matrix_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double* o) {
    from_matrix(Matrix4::shearing(xy, xz, yx, yz, zx, zy), o);
}
void rth_matrix_view_transform(const double* from3, const double* to3, const double* up3, double* o) {
    from_matrix(Matrix4::view_transform(pt(from3), pt(to3), vec(up3)), o);
}
void rth_matrix_mul(const double* a, const double* b, double* o) { from_matrix(to_matrix(a) * to_matrix(b), o); }
int rth_matrix_inverse(const double* a, double* o) {
    return guard_int([&] {
        from_matrix(to_matrix(a).inverse(), o);
        return 0;
    }) == 0 ? 0 : -1;
}
void rth_matrix_transpose(const double* a, double* o) { from_matrix(to_matrix(a).transpose(), o); }
double rth_matrix_determinant(const double* a) { return to_matrix(a).determinant(); }

RthPattern* rth_pattern_solid(double r, double g, double b) {
    return guard([&] { return new RthPattern{Pattern::new_solid(Color(r, g, b))}; });
}
RthPattern* rth_pattern_mixture(uint32_t kind, const RthPattern* a, const RthPattern* b) {
    return guard([&]() -> RthPattern* {
        switch (kind) {
            case RT_PAT_BLEND: return new RthPattern{Pattern::new_blend(a->p, b->p)};
            case RT_PAT_STRIPE: return new RthPattern{Pattern::new_stripe(a->p, b->p)};
            case RT_PAT_GRADIENT: return new RthPattern{Pattern::new_gradient(a->p, b->p)};
            case RT_PAT_RADIAL_GRADIENT: return new RthPattern{Pattern::new_radial_gradient(a->p, b->p)};
            case RT_PAT_RING: return new RthPattern{Pattern::new_ring(a->p, b->p)};
            case RT_PAT_CHECKER: return new RthPattern{Pattern::new_checker(a->p, b->p)};
            default: throw std::invalid_argument("not a mixture kind");
        }
    });
}
RthPattern* rth_pattern_perturb(const RthPattern* p) {
    return guard([&] { return new RthPattern{Pattern::new_perturb(p->p)}; });
}
RthPattern* rth_pattern_test(void) {
    return guard([&] { return new RthPattern{Pattern::new_test()}; });
}
RthPattern* rth_pattern_with_transform(const RthPattern* p, const double* m16) {
    return guard([&] { return new RthPattern{p->p.with_transform(to_matrix(m16))}; });
}
void rth_pattern_free(RthPattern* p) { delete p; }

void rth_material_default(RthMaterial* out) {
    const Material m;
    out->color[0] = m.color.red, out->color[1] = m.color.green, out->color[2] = m.color.blue;
    out->ambient = m.ambient, out->diffuse = m.diffuse, out->specular = m.specular, out->shininess = m.shininess;
    out->reflective = m.reflective, out->transparency = m.transparency, out->refractive_index = m.refractive_index;
}

RthShape* rth_sphere(void) {
    return guard([&] { return new RthShape{Primitive::new_sphere().to_shape()}; });
}
RthShape* rth_glass_sphere(void) {
    return guard([&] { return new RthShape{Primitive::new_glass_sphere().to_shape()}; });
}
RthShape* rth_plane(void) {
    return guard([&] { return new RthShape{Primitive::new_plane().to_shape()}; });
}
RthShape* rth_cube(void) {
    return guard([&] { return new RthShape{Primitive::new_cube().to_shape()}; });
}
RthShape* rth_cylinder(double mn, double mx, int closed) {
    return guard([&] { return new RthShape{Primitive::new_cylinder(Conic(mn, mx, closed != 0)).to_shape()}; });
}
RthShape* rth_cone(double mn, double mx, int closed) {
    return guard([&] { return new RthShape{Primitive::new_cone(Conic(mn, mx, closed != 0)).to_shape()}; });
}
RthShape* rth_triangle(const double* p1, const double* p2, const double* p3) {
    return guard([&] { return new RthShape{Primitive::new_triangle(pt(p1), pt(p2), pt(p3)).to_shape()}; });
}
RthShape* rth_smooth_triangle(const double* p1, const double* p2, const double* p3, const double* n1,
                              const double* n2, const double* n3) {
    return guard([&] {
        return new RthShape{
            Primitive::new_smooth_triangle(pt(p1), pt(p2), pt(p3), vec(n1), vec(n2), vec(n3)).to_shape()};
    });
}
RthShape* rth_group(const RthShape* const* children, size_t n) {
    return guard([&] {
        std::vector<Shape> kids;
        for (size_t i = 0; i < n; ++i) kids.push_back(children[i]->s);
        return new RthShape{Compound::new_group(kids).to_shape()};
    });
}
RthShape* rth_csg(uint32_t operation, const RthShape* left, const RthShape* right) {
    return guard([&] {
        if (operation > 2) throw std::invalid_argument("bad CSG operation");
        return new RthShape{Compound::new_csg(static_cast<Operation>(operation), left->s, right->s).to_shape()};
    });
}
RthShape* rth_shape_with_transform(const RthShape* s, const double* m16) {
    return guard([&] {
        const Matrix4 m = to_matrix(m16);
        return new RthShape{s->s.is_primitive() ? s->s.primitive().with_transform(m).to_shape()
                                                : s->s.compound().with_transform(m).to_shape()};
    });
}
RthShape* rth_shape_with_material(const RthShape* s, const RthMaterial* m, const RthPattern* pattern) {
    return guard([&] {
        Material mat;
        mat.color = Color(m->color[0], m->color[1], m->color[2]);
        mat.ambient = m->ambient, mat.diffuse = m->diffuse, mat.specular = m->specular, mat.shininess = m->shininess;
        mat.reflective = m->reflective, mat.transparency = m->transparency;
        mat.refractive_index = m->refractive_index;
        if (pattern) mat.pattern = pattern->p;
        return new RthShape{s->s.is_primitive() ? s->s.primitive().with_material(mat).to_shape()
                                                : s->s.compound().with_material(mat).to_shape()};
    });
}
RthShape* rth_shape_with_shadow(const RthShape* s, int has_shadow) {
    return guard([&]() -> RthShape* {
        if (!s->s.is_primitive()) throw std::invalid_argument("with_shadow is a Primitive method");
        return new RthShape{s->s.primitive().with_shadow(has_shadow != 0).to_shape()};
    });
}
int rth_shape_includes(const RthShape* s, const RthShape* primitive) {
    if (!primitive->s.is_primitive()) return -1;
    return s->s.includes(primitive->s.primitive()) ? 1 : 0;
}
int rth_shape_kind(const RthShape* s) {
    return s->s.is_primitive() ? static_cast<int>(s->s.primitive().kind()) : static_cast<int>(s->s.compound().kind());
}
size_t rth_shape_num_children(const RthShape* s) {
    return s->s.is_primitive() ? 0 : s->s.compound().children().size();
}
RthShape* rth_shape_child(const RthShape* s, size_t i) {
    return guard([&] { return new RthShape{s->s.compound().children().at(i)}; });
}
void rth_shape_bounds(const RthShape* s, double* out6) {
    const Bounds& b = s->s.properties().bounds;
    out6[0] = b.minimum.x, out6[1] = b.minimum.y, out6[2] = b.minimum.z;
    out6[3] = b.maximum.x, out6[4] = b.maximum.y, out6[5] = b.maximum.z;
}
void rth_shape_free(RthShape* s) { delete s; }
RthShape* rth_obj_parse_file(const char* path) {
    return guard([&] { return new RthShape{obj::parse_file(path)}; });
}
RthShape* rth_obj_parse_string(const char* text, size_t* ignored_lines) {
    return guard([&] { return new RthShape{obj::parse_string(text, ignored_lines)}; });
}

RthWorld* rth_world_new(const RthShape* const* shapes, size_t n_shapes, const double* lights6, size_t n_lights) {
    return guard([&] {
        std::vector<Shape> ss;
        for (size_t i = 0; i < n_shapes; ++i) ss.push_back(shapes[i]->s);
        std::vector<PointLight> ls;
        for (size_t i = 0; i < n_lights; ++i)
            ls.emplace_back(pt(lights6 + 6 * i), Color(lights6[6 * i + 3], lights6[6 * i + 4], lights6[6 * i + 5]));
        return new RthWorld{World(std::move(ss), std::move(ls))};
    });
}
RthWorld* rth_world_default(void) {
    return guard([&] { return new RthWorld{World::default_world()}; });
}
void rth_world_free(RthWorld* w) { delete w; }
int rth_world_color_at(const RthWorld* w, const double* rays, uint32_t n, uint32_t remaining, double* out_rgb) {
    return guard_int([&] {
        std::vector<Ray> rs(n);
        std::memcpy(static_cast<void*>(rs.data()), rays, sizeof(double) * 6 * n);
        const std::vector<Color> out = w->w.color_at(rs, static_cast<uint8_t>(remaining));
        std::memcpy(out_rgb, out.data(), sizeof(double) * 3 * n);
        return 0;
    });
}
int rth_shape_shade_rays(const RthShape* s, const double* light6, const double* rays, uint32_t n, double* out_rgb,
                         uint8_t* out_hit) {
    return guard_int([&] {
        std::vector<Ray> rs(n);
        std::memcpy(static_cast<void*>(rs.data()), rays, sizeof(double) * 6 * n);
        const PointLight light(pt(light6), Color(light6[3], light6[4], light6[5]));
        const ShadedRays out = shade_rays(s->s, light, rs);
        std::memcpy(out_rgb, out.colors.data(), sizeof(double) * 3 * n);
        if (out_hit) std::memcpy(out_hit, out.hit.data(), n);
        return 0;
    });
}
int rth_exercise_shading(double* out_rgb) {
    return guard_int([&] {
        const Canvas c = shading_exercise();
        std::memcpy(out_rgb, c.pixels().data(), sizeof(double) * 3 * c.width() * c.height());
        return 0;
    });
}
int rth_world_probe(const RthWorld* w, const double* rays, uint32_t n, double* out) {
    return guard_int([&] {
        const FlatScene flat = flatten(w->w);
        const RtSceneDesc desc = flat.desc();
        RtScene* scene = nullptr;
        int rc = rt_hip_scene_create(&desc, &scene);
        if (rc != RT_OK) throw RenderError(rc, std::string("rt_hip_scene_create: ") + rt_hip_strerror(rc));
        rc = rt_hip_probe(scene, rays, n, out);
        rt_hip_scene_destroy(scene);
        if (rc != RT_OK) throw RenderError(rc, std::string("rt_hip_probe: ") + rt_hip_strerror(rc));
        return 0;
    });
}
size_t rth_world_describe(const RthWorld* w, double* out, size_t cap) {
    const std::vector<double> d = describe_world(w->w);
    if (out) std::memcpy(out, d.data(), sizeof(double) * (d.size() < cap ? d.size() : cap));
    return d.size();
}

int rth_camera_desc(const RthCamera* cam, RtCameraDesc* out) {
    return guard_int([&] {
        *out = camera_desc(make_camera(*cam));
        return 0;
    });
}
int rth_camera_render(const RthCamera* cam, const RthWorld* w, double* out_rgb, RtStats* stats) {
    return guard_int([&] {
        const Camera camera = make_camera(*cam);
        const FlatScene flat = flatten(w->w);
        const RtSceneDesc desc = flat.desc();
        const RtCameraDesc cd = camera_desc(camera);
        const int rc = rt_hip_render(&desc, &cd, Camera::kDepth, out_rgb, stats);
        if (rc != RT_OK) {
            std::string msg = std::string("Camera::render: ") + rt_hip_strerror(rc);
            const char* hip = rt_hip_last_hip_error();
            if (hip && *hip) msg += std::string(" (") + hip + ")";
            throw RenderError(rc, msg);
        }
        return 0;
    });
}

int rth_camera_render_rgb8(const RthCamera* cam, const RthWorld* w, uint8_t* out_rgb8, RtStats* stats) {
    return guard_int([&] {
        const Camera camera = make_camera(*cam);
        const FlatScene flat = flatten(w->w);
        const RtSceneDesc desc = flat.desc();
        const RtCameraDesc cd = camera_desc(camera);
        const int rc = rt_hip_render_rgb8(&desc, &cd, Camera::kDepth, out_rgb8, stats);
        if (rc != RT_OK) {
            std::string msg = std::string("Camera::render: ") + rt_hip_strerror(rc);
            const char* hip = rt_hip_last_hip_error();
            if (hip && *hip) msg += std::string(" (") + hip + ")";
            throw RenderError(rc, msg);
        }
        return 0;
    });
}

RthFlat* rth_world_flatten(const RthWorld* w, int build_bvh) {
    return guard([&] {
        FlattenOptions opt;
        const int form = build_bvh & 7;
        opt.build_bvh = form != 0;
        opt.hierarchy = form >= 2 ? form - 1 : 0;  // 2: nested, 3: global
        if (build_bvh & 8) opt.dir_device_min = static_cast<size_t>(-1);  // +8: the direction index's cells built here, whatever the count
        if (const char* e = std::getenv("RT_FLATTEN_DIR_RES")) opt.dir_res = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));  // study knob (0: no index, results may differ)
        auto* f = new RthFlat{flatten(w->w, opt), {}};
        f->desc = f->flat.desc();
        return f;
    });
}
const RtSceneDesc* rth_flat_desc(const RthFlat* f) { return &f->desc; }
void rth_flat_free(RthFlat* f) { delete f; }

size_t rth_canvas_to_ppm(const double* rgb, size_t width, size_t height, char* out, size_t cap) {
    std::vector<Color> px(width * height);
    std::memcpy(static_cast<void*>(px.data()), rgb, sizeof(double) * 3 * width * height);
    const std::string s = Canvas(width, height, std::move(px)).to_ppm();
    if (out) std::memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}
size_t rth_ppm_from_rgb8(const uint8_t* rgb8, size_t width, size_t height, char* out, size_t cap) {
    return Canvas::ppm_from_rgb8_into(width, height, rgb8, out, cap);
}
void rth_canvas_quantize(const double* rgb, size_t n_values, uint8_t* out) {
    for (size_t i = 0; i < n_values; ++i) out[i] = Canvas::scale_to_ppm_data(rgb[i]);
}

int rth_exercise(const char* name, size_t width, size_t height, const char* resource_dir, RthWorld** world,
                 RthCamera* camera) {
    return guard_int([&] {
        Exercise e = build_exercise(name, width, height, resource_dir ? resource_dir : ".");
        fill_camera(e.camera, camera);
        *world = new RthWorld{std::move(e.world)};
        return 0;
    });
}
int rth_synthetic(size_t width, size_t height, size_t n_primitives, uint64_t seed, RthWorld** world,
                  RthCamera* camera) {
    return guard_int([&] {
        Exercise e = synthetic_scene(width, height, n_primitives, seed);
        fill_camera(e.camera, camera);
        *world = new RthWorld{std::move(e.world)};
        return 0;
    });
}

}  // extern "C"
