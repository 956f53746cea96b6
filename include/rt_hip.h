/* rt_hip.h -- C ABI of the MI355X (gfx950) render library `librt_hip.so`.
 *
 * This is the drop-in boundary for ONE path of rmehri01/raytracer: the
 * per-pixel render loop
 *     Camera::render -> ray_for_pixel -> World::color_at -> intersections ->
 *     prepare_computations -> shade_hit -> is_shadowed / reflected_color /
 *     refracted_color                       (reference: src/raytracer/camera.rs:51-80,
 *                                             src/raytracer/world.rs:32-127)
 * The reference has no FFI today; these are the entry points its
 * `Camera::render` / `World::color_at` would bind after a host-side
 * scene-flattening pass (INTEGRATION.md shows the Rust `extern "C"` block).
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; every table is a flat array
 *     of fixed-size records described below (row-major matrices, f64 maths);
 *   - caller owns all host buffers for the duration of a call; the library owns
 *     all device memory it allocates (scene tables) until *_destroy;
 *   - return value: 0 = RT_OK, negative = error (rt_hip_strerror); nothing
 *     unwinds across the boundary.  The reference panics/aborts instead
 *     (Cargo.toml:8 `panic = "abort"`); a Rust wrapper turns non-zero into panic!;
 *   - calls are blocking unless a stream argument says otherwise; the library is
 *     not re-entrant per device (serialise calls that share a scene).
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     returns RT_ERR_NO_DEVICE.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------- */
enum {
    RT_OK = 0,
    RT_ERR_NO_DEVICE = -1,     /* no HIP device / hipInit failed                    */
    RT_ERR_HIP = -2,           /* a HIP runtime call failed (see rt_hip_last_hip_error) */
    RT_ERR_INVALID_SCENE = -3, /* malformed tables (bad index, bad kind, cycle ...)  */
    RT_ERR_NEST_TOO_DEEP = -4, /* compound nesting deeper than RT_MAX_NEST           */
    RT_ERR_LIST_BOUND = -5,    /* a CSG subtree can yield more hits than RT_LIST_CAP */
    RT_ERR_LIST_OVERFLOW = -6, /* run time: per-lane hit list overflowed (containers pass) */
    RT_ERR_NAN_T = -7,         /* run time: a NaN intersection distance was dropped
                                  (the reference would panic: intersection.rs:119)   */
    RT_ERR_INVALID_ARG = -8,
    RT_ERR_DEPTH = -9,         /* recursion budget above RT_MAX_DEPTH                */
    RT_ERR_NOALLOC_SIZE = -10  /* rt_hip_render_rows_noalloc: the frame is too wide for the fixed workspace */
};

#define RT_MAX_NEST 8    /* max compound (group/CSG) nesting above a primitive      */
#define RT_LIST_CAP 96   /* per-lane sorted-hit list capacity (entries)             */
#define RT_MAX_DEPTH 8   /* max `remaining_recursions` (reference uses 5: camera.rs:59) */

/* ---- shape kinds (reference: primitive.rs:237-266, compound.rs:121-131) --- */
enum {
    RT_SPHERE = 0,
    RT_PLANE = 1,
    RT_CUBE = 2,
    RT_CYLINDER = 3,
    RT_CONE = 4,
    RT_TRIANGLE = 5,
    RT_SMOOTH_TRIANGLE = 6,
    RT_GROUP = 7,
    RT_CSG = 8
};
/* flags, stored in RtShape.kind_flags bits 8.. */
#define RT_FLAG_HAS_SHADOW (1u << 8)   /* primitive.rs:20,105                      */
#define RT_FLAG_CLOSED (1u << 9)       /* Conic.closed, primitive.rs:457           */
#define RT_FLAG_CSG_OP_SHIFT 10        /* 2 bits: 0 union, 1 intersection, 2 difference (compound.rs:135-139) */
#define RT_KIND(kf) ((kf)&0xffu)

/* ---- pattern node kinds (reference: pattern.rs:95-101,137-144) ------------ */
enum {
    RT_PAT_SOLID = 0,
    RT_PAT_PERTURB = 1,
    RT_PAT_BLEND = 2,
    RT_PAT_STRIPE = 3,
    RT_PAT_GRADIENT = 4,
    RT_PAT_RADIAL_GRADIENT = 5,
    RT_PAT_RING = 6,
    RT_PAT_CHECKER = 7,
    RT_PAT_TEST = 8 /* colour = point coordinates (pattern.rs:65-68,131), KATs only */
};

/* One record per shape (primitives AND compounds), 128 bytes = one L2 line.
 * The table is the scene forest in depth-first pre-order: the subtree of shape
 * i is the index range [i, skip), its first child is i+1, the next sibling of a
 * child c is c.skip.  (reference types: shape.rs:38-43 Properties,
 * primitive.rs:18-22, compound.rs:11-14)                                       */
typedef struct RtShape {
    double inv[12];      /* rows 0..2 of Properties.inverse_transform (row 3 is never read:
                            matrix.rs:286-308 use rows 0..2 only)                   */
    uint32_t kind_flags; /* RT_KIND in bits 0..7, RT_FLAG_* above                   */
    int32_t parent;      /* enclosing compound, -1 at world level                   */
    uint32_t skip;       /* first index after this subtree                          */
    uint32_t material;   /* primitives: index into materials (compounds: ignored, Q8) */
    uint32_t param;      /* cylinder/cone: conic index; (smooth)triangle: triangle index;
                            group: group index; csg: csg index                      */
    uint32_t depth;      /* number of enclosing compounds                           */
    uint32_t pad[2];
} RtShape;

/* Conic limits (primitive.rs:454-458); closed flag lives in kind_flags. */
typedef struct RtConic {
    double minimum, maximum;
} RtConic;

/* Triangle data (primitive.rs:395-409, 253-265). e1 = p2-p1, e2 = p3-p1 are
 * computed by the host exactly as Triangular::new does.  Flat triangles store
 * their cached normal in n1 (n2,n3 unused). */
typedef struct RtTriangle {
    double p1[3], e1[3], e2[3];
    double n1[3], n2[3], n3[3];
} RtTriangle;

/* Group: the box the reference tests per ray (compound.rs:93-95), i.e.
 * Properties.bounds AFTER `bounds.transform(&self.transform)`, precomputed once
 * on the host with the reference's own arithmetic (bounds.rs:19-35, Q5).
 * bvh_root/bvh_count describe an optional acceleration structure over the
 * group's direct children (acceleration only, never changes results). */
typedef struct RtGroup {
    double bmin[3], bmax[3];
    int32_t bvh_root;  /* index into bvh nodes, -1 = none (linear scan of children) */
    uint32_t n_children;
} RtGroup;

/* CSG node (compound.rs:126-130).  lhit[lhit_offset + (p - self - 1)] is 1 when
 * `left.includes(shape p)` holds for primitive p of this subtree (structural
 * equality, shape.rs:18-23, Q7), precomputed on the host. */
typedef struct RtCsg {
    uint32_t op; /* 0 union, 1 intersection, 2 difference */
    uint32_t lhit_offset;
} RtCsg;

/* Material (material.rs:6-16), 12 doubles. */
typedef struct RtMaterial {
    double color[3];
    double ambient, diffuse, specular, shininess;
    double reflective, transparency, refractive_index;
    int32_t pattern; /* root pattern node, -1 = none */
    int32_t pad;
    double pad2;
} RtMaterial;

/* Pattern tree node (pattern.rs:14-18,95-101).  Children by index. */
typedef struct RtPattern {
    double inv[12];  /* rows 0..2 of Pattern.inverse_transform */
    double color[3]; /* RT_PAT_SOLID                           */
    uint32_t kind;
    int32_t a; /* mixture: first child; perturb: the child */
    int32_t b; /* mixture: second child                    */
    uint32_t pad;
} RtPattern;

/* Point light (point_light.rs:5-8). */
typedef struct RtLight {
    double position[3];
    double intensity[3];
} RtLight;

/* Optional acceleration node over the direct children of one group (or over
 * the world's top-level shapes).  Stackless "skip link" layout: visit node i;
 * on box hit go to i+1, on miss go to `miss`.  Leaves name one child shape.
 * Boxes are padded outward so that no child the reference would hit is culled;
 * every accepted child is then tested with the reference's own arithmetic. */
typedef struct __attribute__((aligned(16))) RtBvhNode {
    float bmin[3], bmax[3]; /* the padded f64 box rounded OUTWARD to f32: half the bytes per step of the walk; the
                               test itself is done in f64 */
    int32_t miss;  /* next node when the box is missed (or after a leaf); -1 = end */
    int32_t shape; /* leaf: shape index, inner: -1 */
} RtBvhNode;

/* Direction index (acceleration only).  The reference's Conic::intersect has a branch that
 * reports a wall hit with NO y-range check when the ray is (nearly) parallel to the surface
 * (|a| <= 1e-4, primitive.rs:486-490), so a truncated cylinder/cone can be "hit" anywhere along
 * its infinite surface -- outside any box.  Whether the branch fires depends only on the ray
 * DIRECTION: a = d^T Q d with Q fixed per conic.  For every truncated conic that sits under an
 * acceleration structure the host stores Q (world space) and bins it into the cells of a cube
 * map over directions; a ray looks up its cell, evaluates a for the few candidates listed
 * there and, where |a| <= thr, tests that conic through its ancestor chain regardless of
 * boxes.  q = {xx, yy, zz, xy, xz, yz}. */
typedef struct RtDirCand {
    double q[6];
    double thr;        /* 1e-4 plus rounding slack */
    uint32_t shape;    /* the conic                                              */
    uint32_t visit;    /* what to evaluate: the conic, or its outermost CSG ancestor */
} RtDirCand;

typedef struct RtSceneDesc {
    uint32_t n_shapes;
    const RtShape* shapes;
    uint32_t n_conics;
    const RtConic* conics;
    uint32_t n_triangles;
    const RtTriangle* triangles;
    uint32_t n_groups;
    const RtGroup* groups;
    uint32_t n_csgs;
    const RtCsg* csgs;
    uint32_t n_lhit;
    const uint8_t* lhit;
    uint32_t n_materials;
    const RtMaterial* materials;
    uint32_t n_patterns;
    const RtPattern* patterns;
    uint32_t n_lights;
    const RtLight* lights;
    uint32_t n_bvh_nodes;
    const RtBvhNode* bvh_nodes;
    int32_t world_bvh_root; /* acceleration over top-level shapes, -1 = none */
    uint32_t dir_res;       /* direction index: cells per cube-face edge, 0 = no index */
    uint32_t n_dir_cands;
    const RtDirCand* dir_cands;
    uint32_t n_dir_entries;
    const uint32_t* dir_cell_start; /* 6*dir_res^2 + 1 offsets into dir_entries; NULL (with n_dir_entries 0): candidates
                                     * only -- rt_hip_scene_create builds the cells on the device, at a resolution of
                                     * its choosing */
    const uint32_t* dir_entries;    /* candidate indices, grouped by cell         */
} RtSceneDesc;

/* Camera (camera.rs:13-21): everything Camera::new / with_transform cached. */
typedef struct RtCameraDesc {
    uint32_t hsize, vsize;
    double half_width, half_height, pixel_size;
    double inv[16]; /* inverse_transform, row-major, all 4 rows */
} RtCameraDesc;

/* Event counters of one render (stats launches only).  Used for the FLOP
 * budget of SURVEY.md 8(d) and cross-checked against the CPU oracle. */
typedef struct RtStats {
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
    uint64_t prim_tests[7];   /* by RT_SPHERE..RT_SMOOTH_TRIANGLE */
    uint64_t ray_transforms;  /* ray -> object space transforms (ray.rs:19-24)  */
    uint64_t group_gates;     /* semantic AABB tests (bounds.rs:53-77)          */
    uint64_t bvh_box_tests;   /* acceleration-structure box tests               */
    uint64_t hits_shaded;     /* prepare_computations + shade_hit evaluations   */
    uint64_t normal_levels;   /* matrix levels walked in normal_at (both ways)  */
    uint64_t lightings;       /* Primitive::lighting evaluations                */
    uint64_t pow_calls;       /* powf in specular                               */
    uint64_t pattern_nodes;   /* pattern nodes evaluated                        */
    uint64_t container_passes;/* n1/n2 passes (second traversal)                */
    uint64_t list_entries;    /* hits materialised into per-lane lists          */
    uint64_t schlicks;
    uint64_t accel_f32_tests; /* acceleration-only tests done in f32: child boxes of the 4-wide hierarchy (4 per node
                                 visited) and the pre-tests of its leaves; NOT FP64 work (bvh_box_tests are f64)  */
    uint64_t status_flags;    /* OR of (1<<-err) run-time conditions            */
    double kernel_ms;         /* device time of the launch (hipEvent)           */
} RtStats;

typedef struct RtScene RtScene; /* opaque: device-resident scene tables */

/* Select `device` (hipSetDevice) and initialise the library for it.
 * Replaces: nothing in the reference (process start).  Idempotent. */
int rt_hip_init(int device);
void rt_hip_shutdown(void);
const char* rt_hip_strerror(int code);
/* Text of the last HIP runtime error seen by this thread ("" if none). */
const char* rt_hip_last_hip_error(void);

/* Validate the tables and copy them to the current device.
 * Replaces: the (implicit) borrow of `&World` by Camera::render (camera.rs:51). */
int rt_hip_scene_create(const RtSceneDesc* desc, RtScene** out);
void rt_hip_scene_destroy(RtScene* scene);

/* Drop-in for `Camera::render(&self, world) -> Canvas` (camera.rs:51-66):
 * renders hsize*vsize pixels with recursion budget `depth` (reference: 5,
 * camera.rs:59) into the caller's HOST buffer out_rgb[vsize*hsize*3]
 * (row-major, index = (y*hsize + x)*3, canvas.rs:30-38).  `stats` may be NULL. */
int rt_hip_render(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, double* out_rgb,
                  RtStats* stats);

/* rt_hip_render with the frame's rows dealt over the first `n_devices` HIP devices of this process (1 <= n_devices <=
 * the device count): the tables are copied to every device, each traces interleaved 8-row bands of the frame on a
 * stream and host thread of its own, the shards are gathered to device 0 -- one ncclSend per peer, the ncclRecvs grouped
 * on device 0 (RCCL, loaded from librccl.so at run time), or hipMemcpyPeerAsync when RCCL cannot be had or
 * RT_HIP_MULTI_GATHER=peer is set -- and the frame crosses PCIe once, from device 0.  Same pixels as rt_hip_render for
 * every n_devices (pixels are independent: camera.rs:54-62).  Communicators, streams and buffers are kept until
 * rt_hip_shutdown; one call at a time per process.  `stats`: counters summed over the devices, kernel_ms of the slowest.
 * Replaces: `Camera::render` where the reference would use all cores of the machine (rayon: camera.rs:54). */
int rt_hip_render_multi(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, uint32_t n_devices, double* out_rgb,
                        RtStats* stats);
/* How rt_hip_render_multi gathers shards in this process: "rccl" (librccl.so and its symbols resolved) or "peer"
 * (hipMemcpyPeerAsync).  A static string.  Replaces: nothing in the reference. */
const char* rt_hip_multi_transport(void);

/* `Camera::render(&world)` followed by the per-channel rule of `Canvas::to_ppm` (camera.rs:51-66 +
 * canvas.rs:85-90): the frame is quantised on the device and only the 8-bit samples cross PCIe.
 * out_rgb8[vsize*hsize*3] is a HOST buffer, index = (y*hsize + x)*3 + channel.  `stats` may be NULL. */
int rt_hip_render_rgb8(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, uint8_t* out_rgb8,
                       RtStats* stats);

/* The same two calls on a scene that stays resident (rt_hip_scene_create): camera-only changes between
 * frames -- a fly-through, a re-render at another size -- then cost the kernel and the copy out, not the
 * flattening and upload.  The device framebuffers belong to the scene and are
 * reused.  Replaces: `camera.render(&world)` called again with the same `world` (camera.rs:51). */
int rt_hip_scene_render(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, double* out_rgb, RtStats* stats);
int rt_hip_scene_render_rgb8(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, uint8_t* out_rgb8,
                             RtStats* stats);

/* Same path with everything resident: renders a subset of image rows of a
 * device-resident scene into DEVICE memory, asynchronously on `hip_stream`
 * (a hipStream_t, NULL = default stream).  Local row k (0 <= k < n_local_rows)
 * is image row ((k / band_rows) * n_parts + part) * band_rows + k % band_rows;
 * rows >= vsize are skipped.  part=0,n_parts=1 renders rows [0,n_local_rows).
 * d_out_rgb holds n_local_rows*hsize*3 doubles, local-row-major.
 * Row tiles are how the path shards across GPUs (camera.rs:52-54 treats rows as
 * independent rayon tasks).  *cam is read before the call returns.  The call only enqueues work,
 * except that the FIRST launch of a large scene at a given size allocates the wave-front workspace
 * (rt_hip_scene_set_wavefront) and, when that has to grow, waits for the device first.  If `d_stats` (device pointer to one RtStats,
 * zeroed by the caller) is non-NULL the counting variant of the kernel runs. */
int rt_hip_render_rows(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, uint32_t part,
                       uint32_t n_parts, uint32_t band_rows, uint32_t n_local_rows,
                       double* d_out_rgb, void* hip_stream, RtStats* d_stats);

/* rt_hip_render_rows for callers that cannot allocate device memory -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference.  Same arguments,
 * same frame bit for bit, but no hipMalloc / hipFree and no other allocating call on any path: where
 * rt_hip_render_rows would use the wave-front workspace (large scenes, mode 2 of rt_hip_scene_set_wavefront)
 * the kernels work in a fixed 128 MiB array in device memory that belongs to the library (one copy per GPU
 * the library runs on, held until the process ends), rendering the rows in batches that fit it; every other
 * scene is rendered by the one-launch kernel (mode 0), which needs no workspace.
 * Limit: one batch is at least 8 rows, so hsize can be at most about 30 000 pixels (fewer with many lights: a hit holds one
 * byte per light); wider frames are refused with RT_ERR_NOALLOC_SIZE -- a clear refusal rather than a silent change of
 * kernels: rt_hip_scene_set_wavefront(scene, 0, 0, 0) before the call renders the same frame bit for bit with the
 * one-launch kernel, which needs no workspace and has no such limit.  Every call on one GPU shares that GPU's array (each call clears the counters it
 * needs at zero): calls that overlap in time on one GPU -- other streams, other threads, other scenes --
 * race; order them.  rt_hip_scene_set_wavefront's capacities do not apply. */
int rt_hip_render_rows_noalloc(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, uint32_t part,
                               uint32_t n_parts, uint32_t band_rows, uint32_t n_local_rows,
                               double* d_out_rgb, void* hip_stream, RtStats* d_stats);

/* How launches on this scene organise color_at's recursion (world.rs:98-127); the image is the same
 * either way, bit for bit.  mode 0: one launch, every lane keeps the whole ray tree of its pixel.
 * mode 1: one launch per generation of rays -- a hit that needs reflected / refracted colours parks
 * itself in a node pool and queues its child rays, later launches fold the children's colours back
 * (dense waves at every depth).  mode 2: a pipeline of kernels per generation -- traversal (closest
 * hit, shadow feelers) in kernels of its own with persistent waves over a 4-wide hierarchy, shading
 * in between; for scenes with a world-level hierarchy over primitives, no CSG node and no shadowless
 * primitive (others are rendered as in mode 1).  mode -1 (default): the library chooses -- mode 2, else
 * mode 1, for scenes of >= 192 shapes with an acceleration structure in which some material is
 * reflective or transparent (the generations cost a few hundred bytes of queue traffic per ray, which
 * small scenes do not earn back), mode 0 otherwise.  queue_items / node_items: capacities of one
 * generation's ray queue and of the node pool, 0 = defaults in proportion to the pixels rendered at a
 * time; when they run out the affected hits keep their subtrees to themselves (mode-0 behaviour:
 * slower, same result).  The workspace belongs to the scene and is allocated -- with a device
 * synchronisation when it has to grow -- by the first launch that needs it: mode 1 about 640 B per
 * pixel of the frame (scaled down to fit the same bound as mode 2's); mode 2 about 680 B per pixel (6 GB at 4096 x 2160, 22 GB at 7680 x 4320), at most
 * 24 GB and at most half of the device's free memory (RT_HIP_WF_BUDGET_MB in the environment sets the
 * bound): larger frames are rendered in batches of rows.  If it cannot be allocated the
 * launch falls back to mode 0.
 * Mode 2 also launches on a second stream the scene owns (forked from and joined to `hip_stream` by events inside the
 * call: to the caller everything is ordered on `hip_stream` as before).
 * Replaces: nothing in the reference (tuning). */
int rt_hip_scene_set_wavefront(RtScene* scene, int mode, uint64_t queue_items, uint64_t node_items);

/* Which kernels a render of this scene with recursion budget `depth` runs under the mode in force (a static
 * string; for reports such as bench.py's roofline line).  Replaces: nothing in the reference. */
const char* rt_hip_scene_kernels(const RtScene* scene, uint32_t depth);

/* Run-time conditions raised by launches on this scene since the last call (RT_ERR_LIST_OVERFLOW,
 * RT_ERR_NAN_T) or RT_OK; synchronises `hip_stream` and clears them.  The blocking entry points
 * call this themselves; rt_hip_render_rows is asynchronous and leaves it to the caller.
 * Replaces: the reference's panics on the same conditions (intersection.rs:119). */
int rt_hip_scene_status(const RtScene* scene, void* hip_stream);

/* Drop-in for `World::color_at(&self, ray, remaining) -> Color` (world.rs:32-43)
 * over a batch: rays[n*6] = origin xyz, direction xyz (HOST), out_rgb[n*3] (HOST). */
int rt_hip_color_at(const RtScene* scene, const double* rays, uint32_t n, uint32_t depth,
                    double* out_rgb);

/* The per-pixel body of the reference's hand-rolled loops (exercises/shading.rs:44-57), which go
 * round Camera::render and call the shape API themselves, over a batch of rays:
 *     xs     = shape.intersect(&ray, &empty_trail)         shape.rs:28-32
 *     hit    = xs.hit(None)                                intersection.rs:154-158
 *     point  = ray.position(hit.t)                         ray.rs:15-17
 *     normal = shape.normal_at(&point, ..)                 primitive.rs:153-192
 *     colour = hit.shape.lighting(&point, &light, &-ray.direction, &normal, false, ..)   primitive.rs:111-151
 * The scene stands for the shape (flatten a world holding just that shape: its
 * World::intersections is shape.intersect with an empty trail); the scene's own lights are not
 * used.  Unlike color_at there is no shadow test, the normal is NOT turned towards the eye, and
 * the lighting is evaluated at `point` itself.  rays[n*6], out_rgb[n*3] (BLACK where nothing is
 * hit), out_hit[n] = 1 where something is hit (may be NULL): HOST buffers.   */
int rt_hip_shade_rays(const RtScene* scene, const RtLight* light, const double* rays, uint32_t n,
                      double* out_rgb, uint8_t* out_hit);

/* Debug/KAT probe of the first stage of color_at for a batch of rays (HOST
 * buffers): out[n*16] = { hit(0/1), t, shape index, n1, n2, inside(0/1),
 * point xyz(over_point), normal xyz, reflect xyz, is_shadowed(light 0) }.
 * Mirrors what the reference unit tests inspect through Computations
 * (intersection.rs:162-176). */
int rt_hip_probe(const RtScene* scene, const double* rays, uint32_t n, double* out);

/* Device-side 8-bit quantisation of a framebuffer, the rule of
 * Canvas::scale_to_ppm_data (canvas.rs:85-90): clamp(c*255,0,255).round() as u8,
 * NaN -> 0.  d_rgb and d_out are DEVICE pointers, n_values = pixels*3. */
int rt_hip_quantize(const double* d_rgb, uint64_t n_values, uint8_t* d_out, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
