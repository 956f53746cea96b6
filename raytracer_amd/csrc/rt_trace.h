// rt_trace.h -- the per-ray integrator on the flat tables of include/rt_hip.h:
// world traversal (closest hit / shadow any-hit), the sorted-hit-list evaluator
// used for CSG subtrees and the containers (n1/n2) pass, prepare_computations,
// lighting and the reflect/refract recursion unrolled into an explicit per-lane
// frame stack.  One lane traces one pixel/ray.
//
// How the reference's set semantics are kept without building its BTreeSet for
// every ray:
//   * closest hit = first t >= 0 of the merged sorted set.  Outside CSG nodes a
//     merge only reorders and, on a bit-equal t, keeps the entry of the shape
//     that comes LATER in traversal order (Q1).  So the hit is
//     argmin (t, then larger depth-first index) over each primitive's own
//     collected hits (Q2), which streams with no list at all;
//   * a CSG node needs its operands' sorted lists (filter toggles, post-filter
//     dedupe), so its subtree is evaluated by `list_eval` into a per-lane list
//     and the survivors are fed to the same reduction;
//   * n1/n2 need every entry up to the hit in sorted order: `list_eval` over the
//     whole world, run only for hits whose material is transparent (Q4);
//   * shadows: "first has_shadow entry with t >= 0 lies before the light"
//     == "some surviving has_shadow entry has 0 <= t < distance".  An entry only
//     fails to survive a merge when a later shape has the bit-equal t; if that
//     shape casts a shadow too the answer is unchanged, so with no shadowless
//     primitive in the scene any-hit with early out is exact.  Scenes that do
//     contain `with_shadow(false)` primitives evaluate the full list instead.
#pragma once
#include "rt_device.h"

namespace rtd {

#define RT_STACK (RT_MAX_NEST + 2)
// Rarely taken, register-hungry paths are real calls so that they do not inflate the
// register budget (and so the occupancy) of the traversal loop they are reached from.
#ifndef RT_COLD_CALLS
#define RT_DEV_COLD __device__ __forceinline__
#else
#define RT_DEV_COLD __device__ __noinline__
#endif
#define RT_CONTAINERS_CAP RT_LIST_CAP
#define RT_SCENE_HAS_SHADOWLESS 1u
#define RT_SCENE_FLAT 2u  // primitives only at world level, no acceleration structure
#define RT_SCENE_NESTED_LEAVES 16u  // the world-level hierarchy names shapes nested in groups directly (global hierarchy)
#define RT_SCENE_RECURSIVE 4u  // some material reflects or transmits: color_at can recurse (world.rs:98-127)

struct Wf2Accel;  // rt_bvh4.h
struct SceneDev {
    const RtShape* __restrict__ shapes;
    const RtConic* __restrict__ conics;
    const RtTriangle* __restrict__ tris;
    const RtGroup* __restrict__ groups;
    const RtCsg* __restrict__ csgs;
    const uint8_t* __restrict__ lhit;
    const RtMaterial* __restrict__ mats;
    const RtPattern* __restrict__ pats;
    const RtLight* __restrict__ lights;
    const RtBvhNode* __restrict__ bvh;
    const RtDirCand* __restrict__ dir_cands;
    const uint32_t* __restrict__ dir_cell_start;
    const uint32_t* __restrict__ dir_entries;
    uint32_t dir_res, n_dir_cands;
    uint32_t n_shapes, n_lights;
    int32_t world_bvh_root;
    uint32_t flags;
    // wave-uniform walks read records through the scalar cache: always the copies in global memory
    const RtShape* __restrict__ g_shapes;
    const RtConic* __restrict__ g_conics;
    const RtTriangle* __restrict__ g_tris;
    // LDS staging of small scenes: one packed image of shapes | conics | materials | patterns | lights
    // (the tables lanes index individually while shading); a workgroup copies it in at start and
    // repoints the five table pointers above at its LDS copy.
    const unsigned char* __restrict__ stage;
    uint32_t stage_bytes;
    uint32_t stage_off[10];  // shapes conics mats pats lights | scenes with compounds also: groups csgs lhit bvh tris
    uint32_t lds_list_cap;  // staged scenes: the most entries World::intersections can return = per-lane list slots in LDS
    const Wf2Accel* __restrict__ accel4;  // the 4-wide hierarchy of scenes that have one (rt_bvh4.h), else null
};

// ---- event counters (RtStats order) ------------------------------------------
enum {
    ST_RAY_PRIMARY = 0, ST_RAY_SHADOW, ST_RAY_REFLECT, ST_RAY_REFRACT,
    ST_PRIM0,  // .. ST_PRIM0 + 6
    ST_RAY_XFORM = ST_PRIM0 + 7, ST_GATES, ST_BVH_BOX, ST_HITS_SHADED, ST_NORMAL_LEVELS, ST_LIGHTINGS, ST_POW,
    ST_PATTERN_NODES, ST_CONTAINER_PASSES, ST_LIST_ENTRIES, ST_SCHLICK, ST_ACCEL_F32, ST_COUNT
};
#define RT_STATUS_LIST_OVERFLOW (1u << 6)  // 1 << -RT_ERR_LIST_OVERFLOW
#define RT_STATUS_NAN_T (1u << 7)          // 1 << -RT_ERR_NAN_T

// Scene class a kernel is instantiated for (the template parameter called FLAT for historical reasons):
// what the traversal has to be able to do.
enum { RT_CLASS_GENERAL = 0,        // groups / CSG, hierarchies per level
       RT_CLASS_FLAT = 1,           // RT_SCENE_FLAT: plain primitives, no hierarchy at all
       RT_CLASS_NESTED_LEAVES = 2   // general + RT_SCENE_NESTED_LEAVES
};
template <bool STATS, int FLAT>
struct Ctx {
    unsigned cnt[STATS ? ST_COUNT : 1];
    unsigned status;
};
template <bool STATS, int FLAT>
RT_DEV void count(Ctx<STATS, FLAT>& cx, int what, unsigned n = 1) {
    if constexpr (STATS) cx.cnt[what] += n;
}

// Per-lane working memory (private segment; lanes of a wave interleave, so equal
// indices coalesce).  Levels are absolute nesting depths, shared by the
// streaming traversal and the list evaluator it calls for CSG nodes.
struct LevelState {
    Ray rays[RT_STACK];   // ray in the local space of the compound open at each level
    int cur[RT_STACK];    // streaming traversal: saved cursor
    int end[RT_STACK];    //   linear scan: end index; -2: the level walks a BVH
    int fnode[RT_STACK];  // list evaluator: compound open at each level
    int fbase[RT_STACK];  //   start of its accumulated list
};
// ... with the sorted hit list and the containers set in private memory (any scene)
struct Scratch : LevelState {
    double lt[RT_LIST_CAP];
    uint32_t lid[RT_LIST_CAP];
    uint32_t ct[RT_LIST_CAP];
    RT_DEV double& LT(int i) { return lt[i]; }
    RT_DEV uint32_t& LID(int i) { return lid[i]; }
    RT_DEV uint32_t& CT(int i) { return ct[i]; }
    RT_DEV int cap() const { return RT_LIST_CAP; }
};
// ... or in LDS, for small scenes whose world list provably fits RT_LDS_LIST_CAP entries:
// entry i of lane l sits at [i][l], so a wave's accesses to one entry are conflict-free.
#ifndef RT_LDS_LIST_CAP
#define RT_LDS_LIST_CAP 12
#endif
#define RT_LDS_LIST_BYTES(threads, cap) ((threads) * (cap) * 16u)  // per block and entry: t (8) + id (4) + container (4)
struct ScratchLds : LevelState {
    double* lt;
    uint32_t* lid;
    uint32_t* ct;
    int cap_;  // what this scene needs (<= RT_LDS_LIST_CAP): a shorter list leaves LDS to more resident blocks
    int nt_;   // threads of the block (64, 128 or 256): the stride between a lane's entries
    RT_DEV void attach(unsigned char* block_base, unsigned tid, unsigned cap) {
        const unsigned nt = blockDim.x;
        lt = reinterpret_cast<double*>(block_base) + tid;
        lid = reinterpret_cast<uint32_t*>(block_base + nt * cap * 8u) + tid;
        ct = reinterpret_cast<uint32_t*>(block_base + nt * cap * 12u) + tid;
        cap_ = (int)cap;
        nt_ = (int)nt;
    }
    RT_DEV double& LT(int i) { return lt[i * nt_]; }
    RT_DEV uint32_t& LID(int i) { return lid[i * nt_]; }
    RT_DEV uint32_t& CT(int i) { return ct[i * nt_]; }
    RT_DEV int cap() const { return cap_; }
};

// ---- one primitive's own hit set ----------------------------------------------
// `.collect::<BTreeSet<_>>()` of primitive.rs:227-230 (Q2) on at most four values,
// with static indexing only: sorted t[0..n) plus a keep mask (bit i clear = entry i
// was dropped because it is ~equal to entry i+1).
struct PrimHits {
    double t[4];
    int n;
    unsigned keep;
};
RT_DEV void cswap(double& a, double& b) {
    const double lo = a <= b ? a : b;
    const double hi = a <= b ? b : a;
    a = lo;
    b = hi;
}
// N: the most entries the primitive kind can return (1 plane / triangle, 2 sphere / cube, 4 conic);
// slots >= N are left alone and their keep bits stay clear.
template <int N>
RT_DEV void finish_hits(PrimHits& h, unsigned& status) {
    const double INF = __builtin_inf();
    // NaN distances: the reference panics (intersection.rs:119); drop and flag
    int m = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool valid = i < h.n && h.t[i] == h.t[i];
        if (i < h.n && !valid) status |= RT_STATUS_NAN_T;
        h.t[i] = valid ? h.t[i] : INF;
        m += valid ? 1 : 0;
    }
    if constexpr (N == 2) {
        cswap(h.t[0], h.t[1]);
    } else if constexpr (N == 4) {
        if (h.n > 1) {
            cswap(h.t[0], h.t[1]);
            cswap(h.t[2], h.t[3]);
            cswap(h.t[0], h.t[2]);
            cswap(h.t[1], h.t[3]);
            cswap(h.t[1], h.t[2]);
        }
    }
    h.n = m;
    unsigned keep = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool drop = (i + 1 < N) && (i + 1 < m) && t_eq(h.t[i], h.t[i + 1 < N ? i + 1 : N - 1]);
        if (i < m && !drop) keep |= 1u << i;
    }
    h.keep = keep;
}

// Wave-uniform read of a table record through the constant address space: one scalar load
// per wave, the record then sits in SGPRs.  Only for addresses every lane agrees on; the
// tables are read-only for the lifetime of a launch.
template <class T>
RT_DEV T load_uniform(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) % 8 == 0, "records are whole quadwords");
    typedef const __attribute__((address_space(4))) unsigned long long* ConstQwords;
    ConstQwords q = (ConstQwords)(p);
    T out;
    unsigned long long* o = reinterpret_cast<unsigned long long*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; ++i) o[i] = q[i];
    return out;
#else
    return *p;
#endif
}

template <bool STATS, int FLAT>
RT_DEV void prim_intersect_rec(const double* __restrict__ inv, unsigned kf, const RtConic* __restrict__ conic,
                               const RtTriangle* __restrict__ tri, const Ray& parent_ray, PrimHits& h,
                               Ctx<STATS, FLAT>& cx) {
    // shape.rs:28-32, inside every branch (the kind is the same for the whole wave here): a kind that
    // reads only part of the local ray -- a plane: its y components -- pays only for that part
    const unsigned kind = RT_KIND(kf);
    count(cx, ST_RAY_XFORM);
    count(cx, ST_PRIM0 + (int)kind);
    h.n = 0;
    if (kind == RT_SPHERE) {
        const Ray r = transform_ray(inv, parent_ray);
        h.n = sphere_intersect(r, h.t);
        finish_hits<2>(h, cx.status);
    } else if (kind == RT_PLANE) {
        const Ray r = transform_ray(inv, parent_ray);
        h.n = plane_intersect(r, h.t);
        finish_hits<1>(h, cx.status);
    } else if (kind == RT_CUBE) {
        const Ray r = transform_ray(inv, parent_ray);
        h.n = cube_intersect(r, h.t);
        finish_hits<2>(h, cx.status);
    } else if (kind == RT_CYLINDER || kind == RT_CONE) {
        const Ray r = transform_ray(inv, parent_ray);
        const RtConic c = *conic;
        const bool closed = (kf & RT_FLAG_CLOSED) != 0;
        h.n = kind == RT_CYLINDER ? cylinder_intersect(r, c.minimum, c.maximum, closed, h.t)
                                  : cone_intersect(r, c.minimum, c.maximum, closed, h.t);
        finish_hits<4>(h, cx.status);
    } else {
        const Ray r = transform_ray(inv, parent_ray);
        double u, v;
        h.n = triangle_intersect(tri, r, h.t, u, v);
        finish_hits<1>(h, cx.status);
    }
}

// What transform_ray returns for the identity matrix (entries +1.0 and +0.0, as Matrix::inverse leaves them), bit for
// bit, without reading the matrix: a point's component is ((1*x + 0*y) + 0*z) + 0 = x, and +0 for x = +-0 (the last term
// is +0); a vector's has no last term, so a zero component keeps a minus sign exactly when all three products are -0,
// i.e. when all three components carry one.  (The caller has checked that every component is finite: 0 * inf is NaN.)
// An OBJ mesh is thousands of triangles with this transform (obj.rs: parse_file): the teapot's leaf tests read 72 bytes
// of triangle instead of 96 of matrix and 72 of triangle.
#define RT_FLAG_IDENTITY (1u << 16)  // kind_flags, set by the library in its own copy of the tables (rt_hip_scene_create)
// A CSG node one of whose RIGHT operand's primitives counts as a left hit (l_hit = left.includes(shape) is structural
// equality, compound.rs:154 / shape.rs:18-23: a copy of a left primitive sitting in the right operand): its right operand's
// entries can pass the filter without any left entry, so the "no left entry, nothing passes" shortcut does not hold for it.
#define RT_FLAG_CSG_RIGHT_LHIT (1u << 17)
RT_DEV Ray identity_ray(const Ray& r) {
    const bool all_neg = __builtin_signbit(r.d.x) && __builtin_signbit(r.d.y) && __builtin_signbit(r.d.z);
    const double z = all_neg ? -0.0 : 0.0;
    return Ray{v3(r.o.x + 0.0, r.o.y + 0.0, r.o.z + 0.0),
               v3(r.d.x != 0.0 ? r.d.x : z, r.d.y != 0.0 ? r.d.y : z, r.d.z != 0.0 ? r.d.z : z)};
}
RT_DEV bool ray_is_finite(const Ray& r) {
    const double sum = ((fabs(r.o.x) + fabs(r.o.y)) + (fabs(r.o.z) + fabs(r.d.x))) + (fabs(r.d.y) + fabs(r.d.z));
    return sum < __builtin_inf();  // (an overflowing sum only sends the ray the long way round)
}

template <bool STATS, int FLAT>
RT_DEV void prim_intersect(const SceneDev& S, const RtShape* __restrict__ sh, unsigned kf, const Ray& parent_ray,
                           PrimHits& h, Ctx<STATS, FLAT>& cx) {
    // shape.rs:28-32, ahead of the branches: lanes holding different kinds of primitive share it
    Ray r;
    if ((kf & RT_FLAG_IDENTITY) && ray_is_finite(parent_ray))
        r = identity_ray(parent_ray);
    else
        r = transform_ray(sh->inv, parent_ray);
    const unsigned kind = RT_KIND(kf);
    count(cx, ST_RAY_XFORM);
    count(cx, ST_PRIM0 + (int)kind);
    h.n = 0;
    if (kind == RT_SPHERE) {
        h.n = sphere_intersect(r, h.t);
        finish_hits<2>(h, cx.status);
    } else if (kind == RT_PLANE) {
        h.n = plane_intersect(r, h.t);
        finish_hits<1>(h, cx.status);
    } else if (kind == RT_CUBE) {
        h.n = cube_intersect(r, h.t);
        finish_hits<2>(h, cx.status);
    } else if (kind == RT_CYLINDER || kind == RT_CONE) {
        const RtConic c = S.conics[sh->param];
        const bool closed = (kf & RT_FLAG_CLOSED) != 0;
        h.n = kind == RT_CYLINDER ? cylinder_intersect(r, c.minimum, c.maximum, closed, h.t)
                                  : cone_intersect(r, c.minimum, c.maximum, closed, h.t);
        finish_hits<4>(h, cx.status);
    } else {
        double u, v;
        h.n = triangle_intersect(S.tris + sh->param, r, h.t, u, v);
        finish_hits<1>(h, cx.status);
    }
}

// ---- sorted hit list (stands for BTreeSet<Intersection>) -----------------------
// BTreeSet::append of one element (Q1): on a bit-equal t the entry of the shape that comes
// later in depth-first order survives (a linear scan inserts in that order anyway; a walk
// through an acceleration structure does not, hence the explicit comparison).
template <class SW>
RT_DEV int list_insert(SW& W, int base, int top, double t, uint32_t id, unsigned& status) {
    int pos = top;
    while (pos > base && W.LT(pos - 1) > t) --pos;
    if (pos > base && W.LT(pos - 1) == t) {
        if (id > W.LID(pos - 1)) W.LID(pos - 1) = id;
        return top;
    }
    if (top >= W.cap()) {
        status |= RT_STATUS_LIST_OVERFLOW;
        return top;
    }
    for (int j = top; j > pos; --j) {
        W.LT(j) = W.LT(j - 1);
        W.LID(j) = W.LID(j - 1);
    }
    W.LT(pos) = t;
    W.LID(pos) = id;
    return top + 1;
}
// self.append(other) for two adjacent runs [a,b) and [b,c): returns the new top.
template <class SW>
RT_DEV int list_merge(SW& W, int a, int b, int c, unsigned& status) {
    if (a == b) return c;
    int top = b;
    for (int r = b; r < c; ++r) {
        const double t = W.LT(r);
        const uint32_t id = W.LID(r);
        top = list_insert(W, a, top, t, id, status);
    }
    return top;
}
// Operation::filter_intersections (compound.rs:142-168) in place on [b, top), then the
// `.collect()` dedupe (Q2).  `g` is the CSG node's shape index.
template <class SW>
RT_DEV int csg_filter(const SceneDev& S, SW& W, int b, int top, int g, unsigned op, unsigned lhit_offset) {
    bool in_l = false, in_r = false;
    int k = b;
    for (int j = b; j < top; ++j) {
        const uint32_t id = W.LID(j);
        const bool l_hit = S.lhit[lhit_offset + (id - (uint32_t)g - 1u)] != 0;
        bool allowed;  // compound.rs:175-181
        if (op == 0u)
            allowed = (l_hit && !in_r) || (!l_hit && !in_l);
        else if (op == 1u)
            allowed = (l_hit && in_r) || (!l_hit && in_l);
        else
            allowed = (l_hit && !in_r) || (!l_hit && in_l);
        if (l_hit)
            in_l = !in_l;
        else
            in_r = !in_r;
        if (allowed) {
            W.LT(k) = W.LT(j);
            W.LID(k) = id;
            ++k;
        }
    }
    int m = b;
    for (int j = b; j < k; ++j) {
        if (j + 1 < k && t_eq(W.LT(j), W.LT(j + 1))) continue;
        W.LT(m) = W.LT(j);
        W.LID(m) = W.LID(j);
        ++m;
    }
    return m;
}

// Acceleration-only slab test of the whole LINE against a padded box, cut at t <= bound.
RT_DEV bool accel_line_box(const RtBvhNode& nd, const V3& o, const V3& inv_d, double bound) {
    const double ax = ((double)nd.bmin[0] - o.x) * inv_d.x, bx = ((double)nd.bmax[0] - o.x) * inv_d.x;
    const double ay = ((double)nd.bmin[1] - o.y) * inv_d.y, by = ((double)nd.bmax[1] - o.y) * inv_d.y;
    const double az = ((double)nd.bmin[2] - o.z) * inv_d.z, bz = ((double)nd.bmax[2] - o.z) * inv_d.z;
    const double t0 = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    const double t1 = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
    return !(t0 > t1) && !(t0 > bound);
}

RT_DEV int ancestor_chain(const SceneDev& S, int parent, int* chain) {  // innermost first == the trail
    int n = 0;
    for (int p = parent; p >= 0 && n < RT_STACK; p = S.shapes[p].parent) chain[n++] = p;
    return n;
}

// Direction index (RtDirCand): conics whose "ray parallel to the surface" branch
// (primitive.rs:486-490, no y-range check) fires for this ray direction must be evaluated
// whatever the acceleration boxes say.  The iterator yields each such conic (or its outermost
// CSG ancestor) that the reference's own group gates let the ray reach, with the ray in the
// space of the node's parent and the node's nesting level.  A shape evaluated this way and met
// again by the box walk contributes the same hits twice, which changes nothing.
struct CandIter {
    uint32_t k, e;
    bool scan_all;
};
RT_DEV void cand_begin(const SceneDev& S, const Ray& ray, CandIter& it) {
    const V3 d = ray.d;
    const double len2 = dot(d, d);
    // the cells are built for unit directions; a shorter direction widens the region
    it.scan_all = !(len2 >= 0.999999);
    it.k = 0;
    it.e = S.n_dir_cands;
    if (!it.scan_all) {
        const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
        int face;
        double m, u, v;
        if (ax >= ay && ax >= az) {
            face = d.x >= 0.0 ? 0 : 1, m = ax, u = d.y, v = d.z;
        } else if (ay >= az) {
            face = d.y >= 0.0 ? 2 : 3, m = ay, u = d.x, v = d.z;
        } else {
            face = d.z >= 0.0 ? 4 : 5, m = az, u = d.x, v = d.y;
        }
        const double R = (double)S.dir_res;
        int i = (int)((u / m + 1.0) * 0.5 * R);
        int j = (int)((v / m + 1.0) * 0.5 * R);
        const int hi = (int)S.dir_res - 1;
        i = i < 0 ? 0 : (i > hi ? hi : i);
        j = j < 0 ? 0 : (j > hi ? hi : j);
        const uint32_t cell = ((uint32_t)face * S.dir_res + (uint32_t)j) * S.dir_res + (uint32_t)i;
        it.k = S.dir_cell_start[cell];
        it.e = S.dir_cell_start[cell + 1];
    }
}
// The way down to a shape exactly as the nested traversal takes it: the ray through every ancestor's
// transform, outermost first, and past every group's gate (compound.rs:93-95).  `r` = the ray in the
// space of the shape's parent, `n` = the shape's nesting level; false when a gate turns the ray away.
template <bool STATS, int FLAT>
RT_DEV bool descend_to(const SceneDev& S, int parent, const Ray& world_ray, Ray& r, int& n, Ctx<STATS, FLAT>& cx) {
    int chain[RT_STACK];
    n = ancestor_chain(S, parent, chain);
    r = world_ray;
    bool reachable = true;
    for (int q = n - 1; q >= 0 && reachable; --q) {
        const RtShape* __restrict__ g = S.shapes + chain[q];
        r = transform_ray(g->inv, r);
        count(cx, ST_RAY_XFORM);
        if (RT_KIND(g->kind_flags) == RT_GROUP) {
            const RtGroup* __restrict__ gg = S.groups + g->param;
            count(cx, ST_GATES);
            reachable = gg->n_children != 0u && box_intersects(gg->bmin, gg->bmax, r);
        }
    }
    return reachable;
}

// Next reachable candidate: its shape index (or -1 when exhausted), `r` = ray at its level `n`.
template <bool STATS, int FLAT>
RT_DEV int cand_next(const SceneDev& S, const Ray& ray, CandIter& it, Ray& r, int& n, Ctx<STATS, FLAT>& cx) {
    const V3 d = ray.d;
    while (it.k < it.e) {
        const RtDirCand* __restrict__ c = S.dir_cands + (it.scan_all ? it.k : S.dir_entries[it.k]);
        ++it.k;
        const double a = c->q[0] * d.x * d.x + c->q[1] * d.y * d.y + c->q[2] * d.z * d.z +
                         2.0 * (c->q[3] * d.x * d.y + c->q[4] * d.x * d.z + c->q[5] * d.y * d.z);
        if (!(fabs(a) <= c->thr)) continue;
        // walk down to the shape exactly as the traversal would: transforms and group gates
        const int node = (int)c->visit;
        const bool reachable = descend_to(S, S.shapes[node].parent, ray, r, n, cx);
        if (reachable) return node;
    }
    return -1;
}

// Sorted intersections of the sibling shapes [begin, end) -- all at nesting level `level0`,
// whose ray is W.rays[level0] -- merged into the list W.lt/lid[base0, top0).  Entries with
// t > bound are not recorded (callers pass a bound that leaves a margin wider than every
// dedupe window between the entries they read and the cut).  Returns the new top.
// Group members go straight into the list of the nearest enclosing CSG node (or the root
// list): a group's merge is a set union with the tie rule of list_insert.  A CSG node gets a
// sub-list of its own, filtered when its subtree is done and then merged downwards.
// USE_BVH 1: the world level (its hierarchy and the direction candidates first) and groups that carry an acceleration
// structure are walked through it; 2: a subtree on its own, only the groups' structures (inside a CSG node a group has
// one only if no cylinder or cone sits below it: see the flattener); 0: linear scans throughout.
// Shape::intersect / Compound::local_intersect / World::intersections
// (shape.rs:28-32, compound.rs:87-117, world.rs:45-53).
template <int USE_BVH, bool STATS, int FLAT, class SW>
RT_DEV int list_eval_core(const SceneDev& S, SW& W, int begin, int end, int level0, int base0, int top0,
                          double bound, Ctx<STATS, FLAT>& cx) {
    constexpr bool WALK = USE_BVH != 0, WORLD = USE_BVH == 1;
    int level = level0;
    int floor = level0;  // the level whose exhaustion ends the current source (candidate or walk)
    int top = top0;
    int base = base0;  // start of the list entries currently go to
    const Ray ray0 = W.rays[level0];
    Ray ray = ray0;
    // with acceleration structures in play, the direction candidates come first, each as a
    // one-shape pseudo level; then the walk proper (USE_BVH is only used from the world level)
    bool in_cands = WORLD && S.dir_res != 0u;
    CandIter it;
    if (WORLD && in_cands) cand_begin(S, ray0, it);
    bool bvh = false;
    int cur = 0, lend = 0;  // nothing at hand: the first pass through the loop picks a source
    bool started = false;
    int resume = -2;  // >= -1: the world walk continues at this node once the pseudo level at hand is done
    V3 inv_d = v3(0.0, 0.0, 0.0);
    for (;;) {
        int s = -1;
        if (WALK && bvh) {
            while (cur >= 0) {
                const RtBvhNode nd = S.bvh[cur];  // 32 bytes: two 16-byte reads
                count(cx, ST_BVH_BOX);
                const int miss = nd.miss;
                const int leaf = nd.shape;
                if (!accel_line_box(nd, ray.o, inv_d, bound)) {
                    cur = miss;
                } else if (leaf < 0) {
                    ++cur;
                } else {
                    s = leaf;
                    cur = miss;
                    break;
                }
            }
        } else if (cur < lend) {
            s = cur;
            cur = (int)S.shapes[s].skip;
            // The right operand of an intersection or a difference whose LEFT operand has put nothing on the node's list: no
            // left entry means in_l never turns true, and without it (l_hit && ..) || (!l_hit && in_l) passes no entry of
            // either operand (compound.rs:142-181) -- the right operand need not be evaluated.  (Entries beyond `bound` are
            // not on the list, and need not be: toggles there do not reach the entries before them.)  Not for a node whose
            // right operand holds a primitive that counts as a LEFT hit (RT_FLAG_CSG_RIGHT_LHIT): its entries are "left entries".
            if (level > floor && top == base && s > W.fnode[level] + 1) {  // (level > floor: inside a compound this walk entered)
                const RtShape* __restrict__ gs = S.shapes + W.fnode[level];
                if (RT_KIND(gs->kind_flags) == RT_CSG && ((gs->kind_flags >> RT_FLAG_CSG_OP_SHIFT) & 3u) != 0u &&
                    !(gs->kind_flags & RT_FLAG_CSG_RIGHT_LHIT))
                    s = -2;
            }
        }
        if (s == -2) continue;  // (skipped: on to whatever comes after it on this level -- nothing, for a CSG node)
        if (s < 0) {  // level exhausted
            if (level == floor) {
                if (FLAT == RT_CLASS_NESTED_LEAVES && WORLD && resume >= -1) {  // back from a nested CSG node to the world walk
                    level = floor = level0;
                    ray = ray0;
                    bvh = true;
                    cur = resume;
                    resume = -2;
                    inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
                    continue;
                }
                if (WORLD && in_cands) {  // next direction candidate, or on to the walk
                    int n;
                    Ray r;
                    const int node = cand_next(S, ray0, it, r, n, cx);
                    if (node >= 0) {
                        level = floor = n;
                        ray = r;
                        bvh = false;
                        cur = node;
                        lend = node + 1;
                        continue;
                    }
                    in_cands = false;
                }
                if (started) break;
                started = true;
                level = floor = level0;
                ray = ray0;
                bvh = WORLD && S.world_bvh_root >= 0;
                cur = bvh ? S.world_bvh_root : begin;
                lend = end;
                if (WORLD && bvh) inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
                continue;
            }
            const int g = W.fnode[level];
            const RtShape* __restrict__ gs = S.shapes + g;
            if (RT_KIND(gs->kind_flags) == RT_CSG) {
                const RtCsg c = S.csgs[gs->param];
                top = csg_filter(S, W, base, top, g, c.op, c.lhit_offset);
                const int parent_base = W.fbase[level];
                top = list_merge(W, parent_base, base, top, cx.status);
                base = parent_base;
            }
            --level;
            ray = W.rays[level];
            cur = W.cur[level];
            lend = W.end[level];
            bvh = WALK && lend == -2;
            if (WALK && bvh) inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
            continue;
        }
        const RtShape* __restrict__ sh = S.shapes + s;
        const unsigned kf = sh->kind_flags;
        const unsigned kind = RT_KIND(kf);
        if (FLAT == RT_CLASS_NESTED_LEAVES && WORLD && level == level0 && floor == level0 && level0 == 0 &&
            sh->parent >= 0 && kind != RT_GROUP) {
            // a nested shape named directly by the world-level hierarchy (see trace): a primitive is met
            // through its ancestors right here; a CSG node becomes a one-shape pseudo level, like a
            // direction candidate, after which the walk picks up where it was
            Ray at;
            int n;
            if (!descend_to(S, sh->parent, ray0, at, n, cx)) continue;
            if (kind <= RT_SMOOTH_TRIANGLE) {
                PrimHits h;
                prim_intersect(S, sh, kf, at, h, cx);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (((h.keep >> k) & 1u) && h.t[k] <= bound) {
                        top = list_insert(W, base, top, h.t[k], (uint32_t)s, cx.status);
                        count(cx, ST_LIST_ENTRIES);
                    }
                }
                continue;
            }
            resume = cur;
            level = floor = n;
            ray = at;
            bvh = false;
            cur = s;
            lend = s + 1;
            continue;
        }
        if (kind <= RT_SMOOTH_TRIANGLE) {
            PrimHits h;
            prim_intersect(S, sh, kf, ray, h, cx);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (((h.keep >> k) & 1u) && h.t[k] <= bound) {
                    top = list_insert(W, base, top, h.t[k], (uint32_t)s, cx.status);
                    count(cx, ST_LIST_ENTRIES);
                }
            }
            continue;
        }
        const Ray lr = transform_ray(sh->inv, ray);
        count(cx, ST_RAY_XFORM);
        int root = -1;
        if (kind == RT_GROUP) {
            const RtGroup* __restrict__ g = S.groups + sh->param;
            count(cx, ST_GATES);
            if (g->n_children == 0u || !box_intersects(g->bmin, g->bmax, lr)) continue;  // compound.rs:93-95
            root = WALK ? g->bvh_root : -1;
        }
        W.rays[level] = ray;
        W.cur[level] = cur;
        W.end[level] = bvh ? -2 : lend;
        ++level;
        W.fnode[level] = s;
        W.fbase[level] = base;
        if (kind == RT_CSG) base = top;
        ray = lr;
        bvh = WALK && root >= 0;
        if (WALK && bvh) {
            cur = root;
            inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
        } else {
            cur = s + 1;
            lend = (int)sh->skip;
        }
    }
    return top;
}

// A compound subtree evaluated on its own (CSG nodes met by the streaming traversal).
template <bool STATS, int FLAT, class SW>
RT_DEV_COLD int list_eval(const SceneDev& S, SW& W, int begin, int end, int level0, int base0, double bound,
                     Ctx<STATS, FLAT>& cx) {
    return list_eval_core<2>(S, W, begin, end, level0, base0, base0, bound, cx);
}

// ---- streaming world traversal ---------------------------------------------------
struct Best {
    double t;
    int idx;
};
RT_DEV void offer(Best& best, double t, int idx) {  // first t >= 0; bit-equal t: later shape wins (Q1, Q3)
    if (t >= 0.0 && (t < best.t || (t == best.t && idx > best.idx))) {
        best.t = t;
        best.idx = idx;
    }
}

// Acceleration-only slab test against a padded box: can the ray meet it for some t in
// [0, t_far]?  Never used for a decision the reference makes.
RT_DEV bool accel_box(const RtBvhNode& nd, const V3& o, const V3& inv_d, double t_far) {
    const double ax = ((double)nd.bmin[0] - o.x) * inv_d.x, bx = ((double)nd.bmax[0] - o.x) * inv_d.x;
    const double ay = ((double)nd.bmin[1] - o.y) * inv_d.y, by = ((double)nd.bmax[1] - o.y) * inv_d.y;
    const double az = ((double)nd.bmin[2] - o.z) * inv_d.z, bz = ((double)nd.bmax[2] - o.z) * inv_d.z;
    const double t0 = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    const double t1 = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
    // the reciprocal costs an ulp or two; boxes are padded far wider than that
    return !(t0 > t1) && !(t1 < 0.0) && !(t0 > t_far);
}


// One non-group shape `s` met with the ray of its enclosing level (`level` = its nesting depth):
// a primitive is intersected, a CSG node is materialised through list_eval; the surviving hits go
// to the closest-hit reduction or, for a shadow feeler, to the "anything before the light" test.
// Returns true when a shadow hit was found.
template <bool STATS, int FLAT, class SW>
RT_DEV bool process_leaf(const SceneDev& S, SW& W, int s, const RtShape* __restrict__ sh, unsigned kf,
                         const Ray& ray, int level, bool shadow, double limit, Best& best, Ctx<STATS, FLAT>& cx) {
    if (RT_KIND(kf) <= RT_SMOOTH_TRIANGLE) {
        if (shadow && !(kf & RT_FLAG_HAS_SHADOW)) return false;
        PrimHits h;
        prim_intersect(S, sh, kf, ray, h, cx);
        bool found = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((h.keep >> k) & 1u) {
                if (shadow)
                    found = found || (h.t[k] >= 0.0 && h.t[k] < limit);
                else
                    offer(best, h.t[k], s);
            }
        }
        return found;
    }
    W.rays[level] = ray;
    const int n = list_eval(S, W, s, (int)sh->skip, level, 0, __builtin_inf(), cx);
    for (int j = 0; j < n; ++j) {
        const double t = W.LT(j);
        const int id = (int)W.LID(j);
        if (shadow) {
            if ((S.shapes[id].kind_flags & RT_FLAG_HAS_SHADOW) && t >= 0.0 && t < limit) return true;
        } else {
            offer(best, t, id);
        }
    }
    return false;
}

// World::intersections (world.rs:45-53) as a sorted list at W.lt/lid[0, n), cut at t <= bound.
template <bool STATS, int FLAT, class SW>
RT_DEV int world_list(const SceneDev& S, SW& W, const Ray& ray, double bound, Ctx<STATS, FLAT>& cx) {
    int top = 0;
    if constexpr (FLAT == RT_CLASS_FLAT) {
        const int n = (int)S.n_shapes;
        for (int s = 0; s < n; ++s) {
            const RtShape rec = load_uniform(S.g_shapes + s);
            const unsigned kf = rec.kind_flags;
            const unsigned kind = RT_KIND(kf);
            RtConic conic{0.0, 0.0};
            if (kind == RT_CYLINDER || kind == RT_CONE) conic = load_uniform(S.g_conics + rec.param);
            RtTriangle tri;
            if (kind == RT_TRIANGLE || kind == RT_SMOOTH_TRIANGLE) tri = load_uniform(S.g_tris + rec.param);
            PrimHits h;
            prim_intersect_rec(rec.inv, kf, &conic, &tri, ray, h, cx);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (((h.keep >> k) & 1u) && h.t[k] <= bound) {
                    top = list_insert(W, 0, top, h.t[k], (uint32_t)s, cx.status);
                    count(cx, ST_LIST_ENTRIES);
                }
            }
        }
        return top;
    } else {
    W.rays[0] = ray;
    return list_eval_core<1>(S, W, 0, (int)S.n_shapes, 0, 0, top, bound, cx);
    }
}

#include "rt_bvh4.h"

// One ray against the world.  shadow == false: closest hit into `best` (world.rs:32-36,
// intersection.rs:154-158).  shadow == true: returns true as soon as a shadow-casting
// primitive is met at 0 <= t < limit (world.rs:85-96).  Colour rays and shadow feelers share
// this one loop so that the lanes of a wave stay together whatever each one is tracing.
template <bool STATS, int FLAT, class SW>
RT_DEV bool trace(const SceneDev& S, SW& W, const Ray& world_ray, bool shadow, double limit, Best& best,
                  Ctx<STATS, FLAT>& cx) {
    if constexpr (FLAT == RT_CLASS_FLAT) {
        // only primitives, no acceleration structure: every lane walks the same shape sequence,
        // so the records are fetched once per wave through the scalar cache
        bool found = false;
        const int n = (int)S.n_shapes;
        for (int s = 0; s < n; ++s) {
            const RtShape rec = load_uniform(S.g_shapes + s);
            const unsigned kf = rec.kind_flags;
            const unsigned kind = RT_KIND(kf);
            RtConic conic{0.0, 0.0};
            if (kind == RT_CYLINDER || kind == RT_CONE) conic = load_uniform(S.g_conics + rec.param);
            RtTriangle tri;
            if (kind == RT_TRIANGLE || kind == RT_SMOOTH_TRIANGLE) tri = load_uniform(S.g_tris + rec.param);
            if (shadow && !(kf & RT_FLAG_HAS_SHADOW)) continue;
            PrimHits h;
            prim_intersect_rec(rec.inv, kf, &conic, &tri, world_ray, h, cx);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((h.keep >> k) & 1u) {
                    if (shadow)
                        found = found || (h.t[k] >= 0.0 && h.t[k] < limit);
                    else
                        offer(best, h.t[k], s);
                }
            }
            if (__builtin_amdgcn_ballot_w64(!found) == 0ull) break;  // every live lane is already in shadow
        }
        return found;
    } else {
    if constexpr (FLAT == RT_CLASS_NESTED_LEAVES) {
        if (S.accel4) return trace4(S, world_ray, shadow, limit, best, cx);
    }
    int level = 0;
    int floor = 0;  // level whose exhaustion ends the current source (a direction candidate, or the walk)
    Ray ray = world_ray;
    bool in_cands = S.dir_res != 0u;
    CandIter it;
    if (in_cands) cand_begin(S, world_ray, it);
    bool started = false;
    bool bvh = false;
    int cur = 0, end = 0;  // nothing at hand: the first pass through the loop picks a source
    V3 inv_d = v3(0.0, 0.0, 0.0);
    for (;;) {
        int s = -1;
        if (bvh) {
            // box tests only, until this lane holds a leaf whose box the ray meets (or the level
            // ends): lanes wait here for each other, so the expensive leaf code below runs with
            // most of the wave active instead of once per lane per step
            const double t_far = shadow ? limit : best.t;
            while (cur >= 0) {
                const RtBvhNode nd = S.bvh[cur];  // 32 bytes: two 16-byte reads
                count(cx, ST_BVH_BOX);
                const int miss = nd.miss;
                const int leaf = nd.shape;
                if (!accel_box(nd, ray.o, inv_d, t_far)) {
                    cur = miss;
                } else if (leaf < 0) {
                    ++cur;
                } else {
                    s = leaf;
                    cur = miss;
                    break;
                }
            }
        } else if (cur < end) {
            s = cur;
            cur = (int)S.shapes[s].skip;
        }
        if (s < 0) {  // this level is exhausted
            if (level == floor) {
                if (in_cands) {  // next direction candidate as a one-shape pseudo level, or on to the walk
                    int n;
                    Ray r;
                    const int node = cand_next(S, world_ray, it, r, n, cx);
                    if (node >= 0) {
                        level = floor = n;
                        ray = r;
                        bvh = false;
                        cur = node;
                        end = node + 1;
                        continue;
                    }
                    in_cands = false;
                }
                if (started) break;
                started = true;
                level = floor = 0;
                ray = world_ray;
                bvh = S.world_bvh_root >= 0;
                cur = bvh ? S.world_bvh_root : 0;
                end = (int)S.n_shapes;
                if (bvh) inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
                continue;
            }
            if constexpr (FLAT == RT_CLASS_NESTED_LEAVES) {
                break;  // no level is ever entered (see below)
            } else {
            --level;
            ray = W.rays[level];
            cur = W.cur[level];
            end = W.end[level];
            bvh = end == -2;
            if (bvh) inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
            continue;
            }
        }
        const RtShape* __restrict__ sh = S.shapes + s;
        const unsigned kf = sh->kind_flags;
        const unsigned kind = RT_KIND(kf);
        if (kind != RT_GROUP) {
            // The world-level hierarchy may name a primitive (or CSG node) nested in groups directly: it is
            // then reached the way the reference reaches it, through its ancestors' transforms and gates.
            // (The loop's own ray and level stand in for the descended ones and are put back afterwards, so
            // that the ordinary case carries nothing extra.)
            const bool nested = FLAT == RT_CLASS_NESTED_LEAVES && level == 0 && sh->parent >= 0;
            if (nested) {
                Ray at;
                int n;
                if (!descend_to(S, sh->parent, world_ray, at, n, cx)) continue;
                ray = at;
                level = n;
            }
            const bool done = process_leaf(S, W, s, sh, kf, ray, level, shadow, limit, best, cx);
            if (nested) {
                ray = world_ray;
                level = 0;
            }
            if (done) return true;
            continue;
        }
        // A scene whose world-level hierarchy names nested leaves names ONLY leaves (primitives and CSG nodes: checked
        // when the scene is created), and so do the direction candidates: its walk never enters a group.
        if constexpr (FLAT == RT_CLASS_NESTED_LEAVES) continue;
        const Ray lr = transform_ray(sh->inv, ray);
        const RtGroup* __restrict__ g = S.groups + sh->param;
        count(cx, ST_RAY_XFORM);
        count(cx, ST_GATES);
        if (g->n_children == 0u || !box_intersects(g->bmin, g->bmax, lr)) continue;  // compound.rs:93-95
        W.rays[level] = ray;
        W.cur[level] = cur;
        W.end[level] = bvh ? -2 : end;
        ++level;
        ray = lr;
        const int root = g->bvh_root;
        bvh = root >= 0;
        if (bvh) {
            cur = root;
            inv_d = v3(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
        } else {
            cur = s + 1;
            end = (int)sh->skip;
        }
    }
    return false;
    }
}

// ---- object space / normals (primitive.rs:153-192) -----------------------------------
RT_DEV V3 world_to_object(const SceneDev& S, const RtShape* __restrict__ sh, const int* chain, int n, V3 p) {
    for (int k = n - 1; k >= 0; --k) p = mul_point(S.shapes[chain[k]].inv, p);  // trail.iter().rev()
    return mul_point(sh->inv, p);
}
RT_DEV V3 normal_to_world(const SceneDev& S, const RtShape* __restrict__ sh, const int* chain, int n, V3 nrm) {
    V3 v = normalize(mul_vector_transposed(sh->inv, nrm));
    for (int k = 0; k < n; ++k) v = normalize(mul_vector_transposed(S.shapes[chain[k]].inv, v));
    return v;
}
RT_DEV V3 local_normal(const SceneDev& S, const RtShape* __restrict__ sh, unsigned kind, V3 p, const int* chain,
                       int n, const Ray& world_ray) {
    if (kind == RT_SPHERE) return p - v3(0.0, 0.0, 0.0);
    if (kind == RT_PLANE) return v3(0.0, 1.0, 0.0);
    if (kind == RT_CUBE) return cube_normal(p);
    if (kind == RT_CYLINDER) {
        const RtConic c = S.conics[sh->param];
        return conic_normal(p, 0.0, c.minimum, c.maximum);
    }
    if (kind == RT_CONE) {
        const RtConic c = S.conics[sh->param];
        double y = hypot(p.x, p.z);
        if (p.y > 0.0) y = -y;
        return conic_normal(p, y, c.minimum, c.maximum);
    }
    const RtTriangle* __restrict__ tri = S.tris + sh->param;
    const V3 n1 = v3(tri->n1[0], tri->n1[1], tri->n1[2]);
    if (kind == RT_TRIANGLE) return n1;
    // smooth triangle: the hit's (u, v) are those of the local intersection; rebuild the
    // local ray exactly as the traversal did and recompute them (primitive.rs:221-224,365-373)
    Ray r = world_ray;
    for (int k = n - 1; k >= 0; --k) r = transform_ray(S.shapes[chain[k]].inv, r);
    r = transform_ray(sh->inv, r);
    double ts[1], u, v;
    triangle_intersect(tri, r, ts, u, v);
    const V3 n2 = v3(tri->n2[0], tri->n2[1], tri->n2[2]);
    const V3 n3 = v3(tri->n3[0], tri->n3[1], tri->n3[2]);
    return n2 * u + n3 * v + n1 * (1.0 - u - v);
}

// ---- Computations (intersection.rs:45-99,162-200) ---------------------------------------
struct Comps {
    int idx;
    double t;
    V3 over_point, under_point, eye_v, normal_v, reflect_v;
    V3 object_over_point;  // world_to_object(over_point): what pattern_at_shape needs (pattern.rs:76-85)
    bool inside;
    double n1, n2;
};

// The containers walk of prepare_computations (intersection.rs:50-75) on the full sorted list.
// The walk stops at the first entry ~equal to the hit; entries this far beyond it cannot
// influence anything before it (dedupe windows are 1e-5 per nesting level).
RT_DEV double containers_bound(double hit_t) { return hit_t + 1e-3 + fabs(hit_t) * 1e-9; }

// The containers walk itself on the world list W.lt/lid[0, n).
template <bool STATS, int FLAT, class SW>
RT_DEV void containers_walk(const SceneDev& S, SW& W, int n, double hit_t, double& n1, double& n2,
                            Ctx<STATS, FLAT>& cx) {
    int nc = 0;
    n1 = 1.0;
    n2 = 1.0;
    for (int j = 0; j < n; ++j) {
        const bool is_self = t_eq(W.LT(j), hit_t);
        const uint32_t id = W.LID(j);
        if (is_self && nc > 0) n1 = S.mats[S.shapes[W.CT(nc - 1)].material].refractive_index;
        int pos = -1;
        for (int k = 0; k < nc; ++k)
            if (W.CT(k) == id && pos < 0) pos = k;
        if (pos >= 0) {
            for (int k = pos; k + 1 < nc; ++k) W.CT(k) = W.CT(k + 1);
            --nc;
        } else if (nc < W.cap()) {
            W.CT(nc++) = id;
        } else {
            cx.status |= RT_STATUS_LIST_OVERFLOW;
        }
        if (is_self) {
            if (nc > 0) n2 = S.mats[S.shapes[W.CT(nc - 1)].material].refractive_index;
            break;
        }
    }
}

template <bool STATS, int FLAT, class SW>
RT_DEV void containers_pass(const SceneDev& S, SW& W, const Ray& world_ray, double hit_t, double& n1, double& n2,
                            Ctx<STATS, FLAT>& cx) {
    count(cx, ST_CONTAINER_PASSES);
    const int n = world_list(S, W, world_ray, containers_bound(hit_t), cx);
    containers_walk(S, W, n, hit_t, n1, n2, cx);
}

template <bool STATS, int FLAT, class SW>
RT_DEV void prepare_computations(const SceneDev& S, SW& W, const Ray& ray, const Best& hit, bool want_n1n2,
                                 Comps& c, Ctx<STATS, FLAT>& cx) {
    count(cx, ST_HITS_SHADED);
    const RtShape* __restrict__ sh = S.shapes + hit.idx;
    const unsigned kind = RT_KIND(sh->kind_flags);
    int chain[RT_STACK];
    const int n = FLAT == RT_CLASS_FLAT ? 0 : ancestor_chain(S, sh->parent, chain);
    count(cx, ST_NORMAL_LEVELS, 2u * (unsigned)(n + 1));
    const V3 point = ray.o + ray.d * hit.t;  // ray.rs:15-17
    c.idx = hit.idx;
    c.t = hit.t;
    c.eye_v = -ray.d;
    const V3 local_point = world_to_object(S, sh, chain, n, point);
    V3 normal_v = normal_to_world(S, sh, chain, n, local_normal(S, sh, kind, local_point, chain, n, ray));
    c.inside = dot(normal_v, c.eye_v) < 0.0;
    if (c.inside) normal_v = -normal_v;
    c.normal_v = normal_v;
    c.over_point = point + normal_v * EPSILON;
    c.under_point = point - normal_v * EPSILON;
    c.reflect_v = reflect(ray.d, normal_v);
    c.object_over_point = world_to_object(S, sh, chain, n, c.over_point);
    c.n1 = 1.0;
    c.n2 = 1.0;
    if (want_n1n2) containers_pass(S, W, ray, hit.t, c.n1, c.n2, cx);
}

RT_DEV double schlick(const Comps& c) {  // intersection.rs:182-200
    double cosv = dot(c.eye_v, c.normal_v);
    if (c.n1 > c.n2) {
        const double n = c.n1 / c.n2;
        const double sin2_t = (n * n) * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin2_t);
    }
    const double r = (c.n1 - c.n2) / (c.n1 + c.n2);
    const double r0 = r * r;
    const double x = 1.0 - cosv;
    const double x2 = x * x;  // powi(5) == x^2 -> x^4 -> x^4 * x (Q11)
    const double x4 = x2 * x2;
    return r0 + (1.0 - r0) * (x4 * x);
}

// World::is_shadowed (world.rs:85-96)
template <bool STATS, int FLAT, class SW>
RT_DEV bool is_shadowed(const SceneDev& S, SW& W, V3 point, V3 light_position, Ctx<STATS, FLAT>& cx) {
    const V3 v = light_position - point;
    const double distance = magnitude(v);
    const Ray r{point, normalize(v)};
    count(cx, ST_RAY_SHADOW);
    if (S.flags & RT_SCENE_HAS_SHADOWLESS) {
        // exact-tie replacement between a shadowless and a shadow-casting primitive can
        // decide the answer: evaluate the set itself (see file header)
        const double bound = distance + 1e-3 + distance * 1e-9;  // margin: see containers_pass
        const int n = world_list(S, W, r, bound, cx);
        for (int j = 0; j < n; ++j) {
            const double t = W.LT(j);
            if (t >= 0.0 && (S.shapes[W.LID(j)].kind_flags & RT_FLAG_HAS_SHADOW)) return t < distance;
        }
        return false;
    }
    Best unused{0.0, -1};
    return trace(S, W, r, true, distance, unused, cx);
}

// Primitive::lighting (primitive.rs:111-151); `color` is the material or pattern colour.
template <bool STATS, int FLAT>
RT_DEV V3 lighting(const RtMaterial& m, V3 color, const RtLight& light, V3 position, V3 eye_v, V3 normal_v,
                   bool in_shadow, Ctx<STATS, FLAT>& cx) {
    count(cx, ST_LIGHTINGS);
    const V3 intensity = v3(light.intensity[0], light.intensity[1], light.intensity[2]);
    const V3 effective_color = hadamard(color, intensity);
    const V3 light_v = normalize(v3(light.position[0], light.position[1], light.position[2]) - position);
    const V3 ambient = effective_color * m.ambient;
    V3 diffuse = v3(0.0, 0.0, 0.0), specular = v3(0.0, 0.0, 0.0);
    const double light_dot_normal = dot(light_v, normal_v);
    if (!(light_dot_normal < 0.0 || in_shadow)) {
        diffuse = (effective_color * m.diffuse) * light_dot_normal;
        const V3 reflect_v = reflect(-light_v, normal_v);
        const double reflect_dot_eye = dot(reflect_v, eye_v);
        if (!(reflect_dot_eye <= 0.0)) {
            count(cx, ST_POW);
            const double factor = pow(reflect_dot_eye, m.shininess);
            specular = (intensity * m.specular) * factor;
        }
    }
    return (ambient + diffuse) + specular;
}

// ---- World::color_at with the recursion unrolled (world.rs:32-127) ----------------------
// A frame is one shade_hit waiting for its reflected and/or refracted child colour.
struct Frame {
    V3 surface;
    V3 reflected;
    Ray refract_ray;
    double reflective, transparency, reflectance;
    int state;  // see below
    int use_schlick;
};
enum { FR_REFLECT_THEN_REFRACT = 0, FR_REFLECT_ONLY = 1, FR_REFRACT = 2 };

RT_DEV V3 combine(V3 surface, V3 reflected, V3 refracted, bool use_schlick, double reflectance) {  // world.rs:76-82
    if (use_schlick) return (surface + reflected * reflectance) + refracted * (1.0 - reflectance);
    return (surface + reflected) + refracted;
}

template <bool STATS, int FLAT, class SW>
RT_DEV bool find_hit(const SceneDev& S, SW& W, const Ray& ray, Best& best, Ctx<STATS, FLAT>& cx) {
    best.t = __builtin_inf();
    best.idx = -1;
    trace(S, W, ray, false, 0.0, best, cx);
    return best.idx >= 0;
}

// ---- wave-front organisation of the recursion ---------------------------------------------
// In scenes with reflective / transparent materials the lanes of a wave disagree about how deep
// their recursion goes, and a wave that keeps every lane's whole ray tree to itself runs mostly
// empty below the first bounce.  The level kernels instead trace ONE generation of colour rays
// each: a shade_hit that needs child colours parks what it has in a WfNode, appends its
// reflected / refracted rays to the next generation's queue (dense again, whatever pixels they
// came from) and ends; combine kernels then fold the children's colours into their nodes from
// the deepest generation upwards with the arithmetic of world.rs:76-82, in the same order.
//
// Slots are handed out with ONE 64-bit atomic per wave and generation step (the lanes that need
// slots first finish their shade_hit, then ask together), on one of RT_WF_SHARDS counters chosen
// by the wave's number: device-scope atomics on a single address serialise at ~0.1 us each.
// Each shard owns a slice of the queue and of the node pool.  A request that does not fit is
// refused and the hit keeps its subtree to itself (color_at's frame stack), so capacity only
// ever costs speed, never the result.
struct WfItem {  // one colour ray of the next generation
    Ray ray;
    uint32_t parent;     // RT_WF_PIXEL | framebuffer slot, or (node << 1) | child slot
    uint32_t remaining;  // recursion budget of this ray; RT_WF_EMPTY: nothing here
    uint32_t pad[2];
};
struct WfNode {  // a shade_hit waiting for its children (world.rs:72-82)
    V3 surface;
    V3 child[2];  // color_at of the reflected / refracted ray, as delivered
    double reflective, transparency, reflectance;
    uint32_t parent;
    uint32_t flags;
    double pad[3];
};
static_assert(sizeof(WfItem) == 64 && sizeof(WfNode) == 128, "wave-front records are sized for aligned per-lane stores");
#define RT_WF_PIXEL 0x80000000u
#define RT_WF_EMPTY 0xffffffffu
enum { WF_USE_SCHLICK = 1u, WF_HAS_REFLECT = 2u, WF_HAS_REFRACT = 4u, WF_DEAD = 8u };
#define RT_WF_LEVELS (RT_MAX_DEPTH + 2)
#define RT_WF_SHARDS 256u
#ifndef RT_WF_BINS
#define RT_WF_BINS 8u  // a shard's slice of the queue is split by the direction of the ray: 8 = octants, 24 = octant x dominant axis
#endif
struct WfShard {  // the counters of one shard
    // asked of this shard while generation g was traced; may run past the capacity -- those requests were refused
    unsigned nodes_asked[RT_WF_LEVELS];
    unsigned node_begin[RT_WF_LEVELS + 1];  // generation g's nodes are [node_begin[g], node_begin[g + 1]) of the shard's slice
    // items asked per octant: [g & 1] is read by generation g while generation g + 1's requests count up in the other row
    // (zeroed again once read: wf_mark_kernel)
    unsigned items[2][RT_WF_BINS];
    unsigned pad[(RT_WF_BINS > 8u ? 128 : 64) - 2 * RT_WF_LEVELS - 1 - 2 * RT_WF_BINS];
};
static_assert(sizeof(WfShard) % 128 == 0, "a shard's counters have their 128-byte lines to themselves");
struct WfSpace {
    WfItem* queue[2];  // generation g reads queue[g & 1] and appends to queue[(g + 1) & 1]
    WfNode* nodes;
    WfShard* shards;
    const RtCameraDesc* cam;   // the launch's camera record, for the rare lane that has to start over
    unsigned q_cap, node_cap;  // per shard; q_cap is a multiple of RT_WF_BINS
};

RT_DEV unsigned wf_octant(const V3& d) {
    const unsigned oct = (d.x < 0.0 ? 1u : 0u) | (d.y < 0.0 ? 2u : 0u) | (d.z < 0.0 ? 4u : 0u);
    if constexpr (RT_WF_BINS == 24u) {
        const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
        const unsigned axis = ax >= ay && ax >= az ? 0u : (ay >= az ? 1u : 2u);
        return axis * 8u + oct;
    }
    return oct;
}

RT_DEV void wf_deliver(WfNode* __restrict__ nodes, double* __restrict__ out, uint32_t parent, V3 value) {
    if (parent & RT_WF_PIXEL) {
        double* o = out + (size_t)(parent & ~RT_WF_PIXEL) * 3u;
        o[0] = value.x, o[1] = value.y, o[2] = value.z;
    } else {
        nodes[parent >> 1].child[parent & 1u] = value;
    }
}

struct NoSpill {  // the whole ray tree stays with the lane
    static constexpr bool enabled = false;
    RT_DEV bool operator()(V3, double, double, double, bool, bool, const Ray&, bool, const Ray&, int) { return false; }
};
// color_at's switch in the wave-front kernels: open, the lane's first shade_hit returns to the kernel
// once it is lit; closed (a lane that found no room before), color_at keeps the subtree.
struct WfPending {
    static constexpr bool enabled = true;
    bool open;
};

// The hit being lit: what shade_hit keeps while its shadow feelers are traced.  (Kept in
// registers on purpose: parking it in the private frame stack cost 20-30 % on small scenes.)
struct Shade {
    V3 over, under, eye, normal, reflect_v, color, surface;
    double n1, n2;
    int mat;
    unsigned li;  // next light
    int want_n;
};

// reflected_color / refracted_color's decisions for a lit hit (world.rs:98-127, intersection.rs:182-200)
struct Children {
    bool do_reflect, do_refract, use_schlick;
    Ray refract_ray;
    double reflectance;
};
template <bool STATS, int FLAT>
RT_DEV void shade_children(const RtMaterial& m, const Shade& sd, const Ray& ray, int remaining, Children& ch,
                           Ctx<STATS, FLAT>& cx) {
    ch.do_reflect = !(m.reflective == 0.0 || remaining == 0);
    ch.do_refract = false;
    ch.refract_ray = ray;
    if (sd.want_n != 0) {
        const double n_ratio = sd.n1 / sd.n2;
        const double cos_i = dot(sd.eye, sd.normal);
        const double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (!(sin2_t > 1.0)) {
            const double cos_t = sqrt(1.0 - sin2_t);
            ch.do_refract = true;
            ch.refract_ray.o = sd.under;
            ch.refract_ray.d = sd.normal * (n_ratio * cos_i - cos_t) - sd.eye * n_ratio;
        }
    }
    ch.use_schlick = m.reflective > 0.0 && m.transparency > 0.0;
    ch.reflectance = 0.0;
    if (ch.use_schlick && remaining != 0) {
        count(cx, ST_SCHLICK);
        Comps c;
        c.eye_v = sd.eye, c.normal_v = sd.normal, c.n1 = sd.n1, c.n2 = sd.n2;
        ch.reflectance = schlick(c);
    }
}

// Called by every lane still in the kernel's item loop; `want`: this lane has children to place.
// Returns false for a lane whose request found no room.
RT_DEV bool wf_commit(const WfSpace& ws, unsigned shard, unsigned level, bool want, uint32_t parent, V3 surface,
                      double reflective, double transparency, const Children& ch, V3 reflect_o, V3 reflect_d,
                      int remaining_children) {
    const unsigned long long asking = __builtin_amdgcn_ballot_w64(want);
    if (asking == 0ull) return false;
    const unsigned lane = __lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    WfShard& sh = ws.shards[shard];
    // one node per asking lane
    const int leader = __builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
    unsigned base_nodes = 0u;
    if ((int)lane == leader) base_nodes = atomicAdd(&sh.nodes_asked[level], (unsigned)__builtin_popcountll(asking));
    base_nodes = (unsigned)__shfl((int)base_nodes, leader, 64);
    const unsigned long long ni =
        (unsigned long long)sh.node_begin[level] + base_nodes + (unsigned)__builtin_popcountll(asking & below);
    // The rays go to the part of the shard's slice that belongs to the octant of their direction, the wave's
    // reflected rays first, then its refracted ones: a slice is read back in order, 64 items to a wave, and rays
    // that start near each other and point the same way walk the same part of the hierarchy.  One atomic per
    // octant present in the wave.
    const unsigned sub_cap = ws.q_cap / RT_WF_BINS;
    unsigned slot[2] = {0u, 0u};
    bool given[2] = {false, false};
    bool fits = ni < ws.node_cap;
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
        const bool has = want && (kind == 0 ? ch.do_reflect : ch.do_refract);
        const unsigned oct = wf_octant(kind == 0 ? reflect_d : ch.refract_ray.d);
        unsigned long long todo = __builtin_amdgcn_ballot_w64(has);
        while (todo != 0ull) {
            const int first = __builtin_ctzll(todo);
            const unsigned o = (unsigned)__shfl((int)oct, first, 64);
            const unsigned long long group = __builtin_amdgcn_ballot_w64(has && oct == o);
            unsigned base = 0u;
            if ((int)lane == first) base = atomicAdd(&sh.items[(level + 1u) & 1u][o], (unsigned)__builtin_popcountll(group));
            base = (unsigned)__shfl((int)base, first, 64);
            if (has && oct == o) {
                const unsigned long long at = (unsigned long long)base + (unsigned)__builtin_popcountll(group & below);
                given[kind] = at < sub_cap;
                if (given[kind])
                    slot[kind] = o * sub_cap + (unsigned)at;
                else
                    fits = false;
            }
            todo &= ~group;
        }
    }
    if (!want) return false;
    WfItem* __restrict__ q = ws.queue[(level + 1u) & 1u] + (size_t)shard * ws.q_cap;
    WfNode* __restrict__ nodes = ws.nodes + (size_t)shard * ws.node_cap;
    if (!fits) {
        // no room: blank whatever part of the request did fit
        if (ni < ws.node_cap) nodes[ni].flags = WF_DEAD;
        if (given[0]) q[slot[0]].remaining = RT_WF_EMPTY;
        if (given[1]) q[slot[1]].remaining = RT_WF_EMPTY;
        return false;
    }
    WfNode& n = nodes[ni];
    n.surface = surface;
    n.child[0] = v3(0.0, 0.0, 0.0);
    n.child[1] = v3(0.0, 0.0, 0.0);
    n.reflective = reflective, n.transparency = transparency, n.reflectance = ch.reflectance;
    n.parent = parent;
    n.flags = (ch.use_schlick ? WF_USE_SCHLICK : 0u) | (ch.do_reflect ? WF_HAS_REFLECT : 0u) | (ch.do_refract ? WF_HAS_REFRACT : 0u);
    const uint32_t global_node = (uint32_t)((size_t)shard * ws.node_cap + ni);
    if (ch.do_reflect) {
        WfItem& it = q[slot[0]];
        it.ray = Ray{reflect_o, reflect_d}, it.parent = global_node << 1, it.remaining = (uint32_t)remaining_children;
    }
    if (ch.do_refract) {
        WfItem& it = q[slot[1]];
        it.ray = ch.refract_ray, it.parent = (global_node << 1) | 1u, it.remaining = (uint32_t)remaining_children;
    }
    return true;
}

// One loop iteration traces ONE ray per lane -- a colour ray (primary, reflected, refracted) or
// a shadow feeler -- and then advances that lane's shade_hit state machine; the wave never
// splits into "lanes tracing shadows" and "lanes tracing colour rays".
// `spill` (wave-front kernels): offered the children of the lane's FIRST shade_hit; when it takes
// them, `deferred` is set and the returned colour means nothing.
template <bool STATS, int FLAT, class SW, class SP>
RT_DEV V3 color_at(const SceneDev& S, SW& W, Frame* frames, Ray ray, int remaining, Ctx<STATS, FLAT>& cx, SP& spill,
                   bool& deferred, Shade& sd) {
    int sp = 0;
    bool shadow = false;
    double limit = 0.0;
    sd.li = 0u;
    sd.mat = 0;
    sd.want_n = 0;
    for (;;) {
        Best hit;
        hit.t = __builtin_inf();
        hit.idx = -1;
        const bool occluded = trace(S, W, ray, shadow, limit, hit, cx);
        bool have_value = false;
        bool need_containers = false;
        double hit_t = 0.0;
        V3 value = v3(0.0, 0.0, 0.0);
        if (shadow) {  // light sd.li is answered (world.rs:58-70)
            const RtLight light = S.lights[sd.li];
            const RtMaterial m = S.mats[sd.mat];
            sd.surface = sd.surface + lighting(m, sd.color, light, sd.over, sd.eye, sd.normal, occluded, cx);
            ++sd.li;
        } else if (hit.idx < 0) {  // world.rs:41: a miss is BLACK
            have_value = true;
        } else {
            const RtShape* __restrict__ sh = S.shapes + hit.idx;
            const RtMaterial m = S.mats[sh->material];
            // n1/n2 are consumed only by refracted_color past its `transparency == 0 ||
            // remaining == 0` exit and by schlick(), whose value multiplies BLACK when
            // remaining == 0 -- skip the pass where nothing can read it.
            const bool want_n = m.transparency != 0.0 && remaining != 0;
            Comps c;
            prepare_computations(S, W, ray, hit, false, c, cx);
            V3 color = v3(m.color[0], m.color[1], m.color[2]);
            if (m.pattern >= 0) {
                unsigned nodes = 0;
                color = pattern_eval(S.pats, m.pattern, c.object_over_point, nodes);
                count(cx, ST_PATTERN_NODES, nodes);
            }
            sd.over = c.over_point, sd.under = c.under_point, sd.eye = c.eye_v, sd.normal = c.normal_v;
            sd.reflect_v = c.reflect_v, sd.color = color, sd.surface = v3(0.0, 0.0, 0.0);
            sd.n1 = 1.0, sd.n2 = 1.0, sd.mat = (int)sh->material, sd.li = 0u, sd.want_n = want_n ? 1 : 0;
            need_containers = want_n;
            hit_t = hit.t;
        }
        bool retrace = false;
        if (!have_value) {
            // Work that needs the sorted world set -- the containers walk of a transparent hit, and
            // shadow tests in scenes with shadowless primitives -- goes through ONE world_list call
            // site; shadow feelers otherwise go back to the top of the loop to be traced.
            for (;;) {
                const bool list_shadow = !need_containers && sd.li < S.n_lights && (S.flags & RT_SCENE_HAS_SHADOWLESS);
                if (!need_containers && !list_shadow) break;
                Ray lray = ray;
                double bound, distance = 0.0;
                if (need_containers) {
                    count(cx, ST_CONTAINER_PASSES);
                    bound = containers_bound(hit_t);
                } else {
                    const RtLight light = S.lights[sd.li];
                    const V3 v = v3(light.position[0], light.position[1], light.position[2]) - sd.over;
                    distance = magnitude(v);
                    lray = Ray{sd.over, normalize(v)};
                    bound = distance + 1e-3 + distance * 1e-9;
                    count(cx, ST_RAY_SHADOW);
                }
                const int n = world_list(S, W, lray, bound, cx);
                if (need_containers) {
                    containers_walk(S, W, n, hit_t, sd.n1, sd.n2, cx);
                    need_containers = false;
                } else {
                    bool in_shadow = false;  // first shadow-casting entry with t >= 0 (intersection.rs:154-158)
                    for (int j = 0; j < n; ++j) {
                        const double t = W.LT(j);
                        if (t >= 0.0 && (S.shapes[W.LID(j)].kind_flags & RT_FLAG_HAS_SHADOW)) {
                            in_shadow = t < distance;
                            break;
                        }
                    }
                    const RtLight light = S.lights[sd.li];
                    const RtMaterial m = S.mats[sd.mat];
                    sd.surface = sd.surface + lighting(m, sd.color, light, sd.over, sd.eye, sd.normal, in_shadow, cx);
                    ++sd.li;
                }
            }
            if (sd.li < S.n_lights) {  // next light: one shadow feeler (world.rs:86-89)
                const RtLight light = S.lights[sd.li];
                const V3 v = v3(light.position[0], light.position[1], light.position[2]) - sd.over;
                limit = magnitude(v);
                ray = Ray{sd.over, normalize(v)};
                shadow = true;
                count(cx, ST_RAY_SHADOW);
                retrace = true;
            }
        }
        if (retrace) continue;
        if (!have_value) {  // all lights in: reflected_color / refracted_color (world.rs:72-82,98-127)
            if constexpr (SP::enabled) {
                // wave-front kernels: the lane's first shade_hit ends here; its children are worked out
                // from `sd` by the kernel once the whole wave is through (nothing else has to stay live)
                if (sp == 0 && spill.open) {
                    deferred = true;
                    return v3(0.0, 0.0, 0.0);
                }
            }
            const RtMaterial m = S.mats[sd.mat];
            Children ch;
            shade_children(m, sd, ray, remaining, ch, cx);
            const bool do_reflect = ch.do_reflect, do_refract = ch.do_refract, use_schlick = ch.use_schlick;
            const Ray refract_ray = ch.refract_ray;
            const double reflectance = ch.reflectance;
            if (do_reflect || do_refract) {
                Frame& f = frames[sp];
                f.surface = sd.surface;
                f.reflected = v3(0.0, 0.0, 0.0);
                f.refract_ray = refract_ray;
                f.reflective = m.reflective;
                f.transparency = m.transparency;
                f.reflectance = reflectance;
                f.use_schlick = use_schlick ? 1 : 0;
                f.state = do_reflect ? (do_refract ? FR_REFLECT_THEN_REFRACT : FR_REFLECT_ONLY) : FR_REFRACT;
                ++sp;
                --remaining;
                shadow = false;
                if (do_reflect) {
                    count(cx, ST_RAY_REFLECT);
                    ray = Ray{sd.over, sd.reflect_v};
                } else {
                    count(cx, ST_RAY_REFRACT);
                    ray = refract_ray;
                }
                continue;
            }
            value = combine(sd.surface, v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 0.0), use_schlick, reflectance);
        }
        // ---- hand `value` to the waiting frames
        for (;;) {
            if (sp == 0) return value;
            Frame& f = frames[sp - 1];
            if (f.state == FR_REFLECT_THEN_REFRACT) {
                f.reflected = value * f.reflective;
                f.state = FR_REFRACT;
                ray = f.refract_ray;
                shadow = false;
                count(cx, ST_RAY_REFRACT);
                break;
            }
            V3 reflected, refracted;
            if (f.state == FR_REFLECT_ONLY) {
                reflected = value * f.reflective;
                refracted = v3(0.0, 0.0, 0.0);
            } else {
                reflected = f.reflected;
                refracted = value * f.transparency;
            }
            value = combine(f.surface, reflected, refracted, f.use_schlick != 0, f.reflectance);
            --sp;
            ++remaining;
        }
    }
}

template <bool STATS, int FLAT, class SW>
RT_DEV V3 color_at(const SceneDev& S, SW& W, Frame* frames, Ray ray, int remaining, Ctx<STATS, FLAT>& cx) {
    NoSpill none;
    bool deferred = false;
    Shade sd;
    return color_at(S, W, frames, ray, remaining, cx, none, deferred, sd);
}

// Camera::ray_for_pixel (camera.rs:68-80)
RT_DEV Ray ray_for_pixel(const RtCameraDesc& cam, unsigned px, unsigned py) {
    const double x_offset = ((double)px + 0.5) * cam.pixel_size;
    const double y_offset = ((double)py + 0.5) * cam.pixel_size;
    const double world_x = cam.half_width - x_offset;
    const double world_y = cam.half_height - y_offset;
    const V3 pixel = mul_point(cam.inv, v3(world_x, world_y, -1.0));
    const V3 origin = mul_point(cam.inv, v3(0.0, 0.0, 0.0));
    return Ray{origin, normalize(pixel - origin)};
}

}  // namespace rtd
