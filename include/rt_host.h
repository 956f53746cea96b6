/* rt_host.h -- C handle API over the C++ host mirror of the reference's
 * scene-building interface (`librt_host.so`).  It exists so that bindings
 * (the Python package raytracer_amd, or any FFI) can drive the same
 * Camera / World / Primitive / Compound / Material / Pattern API the reference
 * exposes in Rust (src/lib.rs:11-45), with rendering going through the HIP
 * library of rt_hip.h.  One function per reference constructor/method; the
 * reference item each one mirrors is cited.
 *
 * Ownership: every function returning a handle returns a NEW object the caller
 * frees with the matching *_free.  Handles passed in are never consumed
 * (objects are immutable values, cloned as needed) -- the builder style of the
 * reference (`with_transform(self) -> Self`) becomes "returns a new handle".
 * Errors: NULL handle or negative int; rth_last_error() gives the text.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthPattern RthPattern;
typedef struct RthShape RthShape;
typedef struct RthWorld RthWorld;
typedef struct RthFlat RthFlat;

const char* rth_last_error(void);

/* ---- Matrix<4> (src/core/matrix.rs), row-major double[16] ------------------ */
void rth_matrix_identity(double* out16);                                    /* matrix.rs:15  */
void rth_matrix_translation(double x, double y, double z, double* out16);   /* matrix.rs:153 */
void rth_matrix_scaling(double x, double y, double z, double* out16);       /* matrix.rs:162 */
void rth_matrix_rotation_x(double radians, double* out16);                  /* matrix.rs:171 */
void rth_matrix_rotation_y(double radians, double* out16);                  /* matrix.rs:183 */
void rth_matrix_rotation_z(double radians, double* out16);                  /* matrix.rs:195 */
void rth_matrix_shearing(double xy, double xz, double yx, double yz, double zx, double zy,
                         double* out16);                                    /* matrix.rs:207 */
void rth_matrix_view_transform(const double* from3, const double* to3, const double* up3,
                               double* out16);                              /* matrix.rs:216 */
void rth_matrix_mul(const double* a16, const double* b16, double* out16);   /* matrix.rs:268 */
int rth_matrix_inverse(const double* a16, double* out16);                   /* matrix.rs:92; -1 if singular */
void rth_matrix_transpose(const double* a16, double* out16);                /* matrix.rs:25  */
double rth_matrix_determinant(const double* a16);                           /* matrix.rs:108 */

/* ---- Pattern (src/graphics/pattern.rs:29-74) ------------------------------ */
RthPattern* rth_pattern_solid(double r, double g, double b);
/* kind = RT_PAT_BLEND .. RT_PAT_CHECKER (new_blend/new_stripe/new_gradient/
 * new_radial_gradient/new_ring/new_checker) */
RthPattern* rth_pattern_mixture(uint32_t kind, const RthPattern* a, const RthPattern* b);
RthPattern* rth_pattern_perturb(const RthPattern* p);
RthPattern* rth_pattern_test(void);                                         /* pattern.rs:65 */
RthPattern* rth_pattern_with_transform(const RthPattern* p, const double* m16); /* pattern.rs:70 */
void rth_pattern_free(RthPattern* p);

/* ---- Material (src/raytracer/material.rs:6-32) ------------------------------ */
typedef struct RthMaterial {
    double color[3];
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
} RthMaterial;
void rth_material_default(RthMaterial* out);                                /* material.rs:18 */

/* ---- Primitive / Compound / Shape (src/raytracer/shapes) ------------------- */
RthShape* rth_sphere(void);                                                 /* primitive.rs:52 */
RthShape* rth_glass_sphere(void);                                           /* primitive.rs:56 */
RthShape* rth_plane(void);                                                  /* primitive.rs:64 */
RthShape* rth_cube(void);                                                   /* primitive.rs:68 */
RthShape* rth_cylinder(double minimum, double maximum, int closed);         /* primitive.rs:72, Conic::new :461 */
RthShape* rth_cone(double minimum, double maximum, int closed);             /* primitive.rs:76 */
RthShape* rth_triangle(const double* p1, const double* p2, const double* p3); /* primitive.rs:80 */
RthShape* rth_smooth_triangle(const double* p1, const double* p2, const double* p3, const double* n1,
                              const double* n2, const double* n3);          /* primitive.rs:87 */
RthShape* rth_group(const RthShape* const* children, size_t n);             /* compound.rs:52 */
RthShape* rth_csg(uint32_t operation, const RthShape* left, const RthShape* right); /* compound.rs:56 */
RthShape* rth_shape_with_transform(const RthShape* s, const double* m16);   /* shape.rs:63 */
RthShape* rth_shape_with_material(const RthShape* s, const RthMaterial* m, const RthPattern* pattern_or_null); /* shape.rs:69 */
RthShape* rth_shape_with_shadow(const RthShape* s, int has_shadow);         /* primitive.rs:105 */
int rth_shape_includes(const RthShape* s, const RthShape* primitive);       /* shape.rs:18; -1 if not a primitive */
int rth_shape_kind(const RthShape* s);                                      /* RT_SPHERE .. RT_CSG */
size_t rth_shape_num_children(const RthShape* s);
RthShape* rth_shape_child(const RthShape* s, size_t i);                     /* new handle */
void rth_shape_bounds(const RthShape* s, double* out6);                     /* Properties.bounds */
void rth_shape_free(RthShape* s);
RthShape* rth_obj_parse_file(const char* path);                             /* obj.rs:55 */
RthShape* rth_obj_parse_string(const char* text, size_t* ignored_lines);    /* obj.rs:62-112 */

/* ---- World (src/raytracer/world.rs) ---------------------------------------- */
/* lights6 = n_lights * {position xyz, intensity rgb} (point_light.rs:5-8) */
RthWorld* rth_world_new(const RthShape* const* shapes, size_t n_shapes, const double* lights6, size_t n_lights); /* world.rs:24 */
RthWorld* rth_world_default(void);                                          /* world.rs:130 */
void rth_world_free(RthWorld* w);
/* World::color_at over a batch, ON THE GPU (world.rs:32). rays n*6, out n*3 (host). */
int rth_world_color_at(const RthWorld* w, const double* rays, uint32_t n, uint32_t remaining, double* out_rgb);
/* rt_hip_probe on this world (GPU). out n*16. */
/* exercises/shading.rs:44-57 per ray for a batch, ON THE GPU (rt_hip_shade_rays): intersect, hit, normal_at and
 * lighting of ONE shape as user code calls them.  light6 = position xyz, intensity rgb; rays n*6; out_rgb n*3;
 * out_hit n (may be NULL). */
int rth_shape_shade_rays(const RthShape* s, const double* light6, const double* rays, uint32_t n, double* out_rgb,
                         uint8_t* out_hit);
/* exercises/shading.rs:17-63 whole: out_rgb[500*500*3], pixels the sphere does not cover stay BLACK. */
int rth_exercise_shading(double* out_rgb);
int rth_world_probe(const RthWorld* w, const double* rays, uint32_t n, double* out);
/* Object-tree description (transforms as given, nothing derived) for an independent checker.
 * Returns the number of doubles; writes at most `cap`. */
size_t rth_world_describe(const RthWorld* w, double* out, size_t cap);

/* ---- Camera (src/raytracer/camera.rs) -------------------------------------- */
typedef struct RthCamera {
    uint64_t h_size, v_size;
    double field_of_view;
    double transform[16];
    int32_t transform_set; /* 0: Camera::new only; 1: with_transform(transform) */
    int32_t pad;
} RthCamera;
/* Camera::new + with_transform cached values (camera.rs:24-49) -> RtCameraDesc. */
int rth_camera_desc(const RthCamera* cam, RtCameraDesc* out);
/* Camera::render(&world) ON THE GPU (camera.rs:51): out_rgb[v*h*3] host buffer. */
int rth_camera_render(const RthCamera* cam, const RthWorld* w, double* out_rgb, RtStats* stats_or_null);
/* Camera::render(&world) + Canvas::scale_to_ppm_data per channel (camera.rs:51, canvas.rs:85), quantised on
 * the device: out_rgb8[v*h*3] host buffer. */
int rth_camera_render_rgb8(const RthCamera* cam, const RthWorld* w, uint8_t* out_rgb8, RtStats* stats_or_null);

/* ---- flattened scene, for callers that keep it resident -------------------- */
/* build_bvh: 0 no acceleration structures, 1 by policy, 2 one hierarchy per group level, 3 one world-space hierarchy;
 * +8: the direction index's cells are rasterised here whatever the number of candidate conics (from 64 candidates on
 * they are otherwise left to rt_hip_scene_create, which builds them on the device: RtSceneDesc.dir_cell_start null) */
RthFlat* rth_world_flatten(const RthWorld* w, int build_bvh);
const RtSceneDesc* rth_flat_desc(const RthFlat* f);
void rth_flat_free(RthFlat* f);

/* ---- Canvas (src/graphics/canvas.rs) ---------------------------------------- */
size_t rth_canvas_to_ppm(const double* rgb, size_t width, size_t height, char* out, size_t cap); /* canvas.rs:46 */
void rth_canvas_quantize(const double* rgb, size_t n_values, uint8_t* out);  /* canvas.rs:85 */
/* Canvas::to_ppm (canvas.rs:46-83) from samples that are already 8-bit.  Returns the text length; writes <= cap. */
size_t rth_ppm_from_rgb8(const uint8_t* rgb8, size_t width, size_t height, char* out, size_t cap);

/* ---- exercises (the reference's exercises/ programs) + synthetic scene ---------------------------- */
/* name: plane scene pattern glass_bubble teapot hexagon csg cover room synthetic */
int rth_exercise(const char* name, size_t width, size_t height, const char* resource_dir, RthWorld** world,
                 RthCamera* camera);
int rth_synthetic(size_t width, size_t height, size_t n_primitives, uint64_t seed, RthWorld** world,
                  RthCamera* camera);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
