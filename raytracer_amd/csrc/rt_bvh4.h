// rt_bvh4.h -- the 4-wide hierarchy of large scenes whose world-level hierarchy names leaves directly (acceleration
// only: never changes a result): nodes, the f32 side of a ray, the walk, the f32 pre-test and the f64 evaluation of a
// leaf, and trace4 -- one ray through it, the drop-in for trace()'s skip-link walk.  Included by rt_trace.h between the
// primitives' intersection code and trace(); the kernel pipeline (rt_wf2.h) is built from the same pieces.
#pragma once

// ---- 4-wide hierarchy (acceleration only; built by the library from the skip-link nodes of the scene) -----------
struct __attribute__((aligned(64))) Bvh4Node {
    float org[3];       // lower corner of the box around all children
    uint8_t exp[3];     // per axis: the grid step as the exponent byte of a float, step = 2^(exp - 127)
    uint8_t pad0;
    int32_t child[4];   // inner node index, RT_BVH4_LEAF | shape index, or RT_BVH4_DONE for an empty slot
    uint8_t qlo[3][4];  // qlo[axis][child]: the child's padded box on that grid, lower corner rounded down ...
    uint8_t qhi[3][4];  // ... upper corner rounded up: box = org + q * step
    uint32_t pad1[2];
};
static_assert(sizeof(Bvh4Node) == 64, "one node = four 16-byte words: half the words of f32 boxes (the walk is bound by its vector memory instructions)");
#define RT_BVH4_LEAF 0x80000000u
#define RT_BVH4_DONE 0x7fffffffu
#define RT_BVH4_NOPRE 0x40000000u  // in a leaf code: the leaf has no f32 pre-test (planes, triangles): its record is read for the ancestors only
// bits 24-29 of a leaf code are zero in the hierarchy: the traversal kernel's leaf exchange puts the owner lane there (rt_wf2.h),
// so the pipeline takes scenes of fewer than 2^24 shapes
#define RT_BVH4_SHAPE_BITS 24
#define RT_BVH4_SHAPE(code) ((code) & 0x00ffffffu)
#ifndef RT_WF2_LDS_STACK
#define RT_WF2_LDS_STACK 24   // stack entries per lane kept in LDS; deeper ones in private memory
#endif
#ifndef RT_WF2_MAX_STACK
#define RT_WF2_MAX_STACK 96   // scenes whose 4-wide tree could need more keep the older kernels
#endif

// Conservative f32 test of a leaf in its own object space, ahead of the reference's f64 intersection (acceleration only):
// the composed world-to-object transform as one f32 matrix, and what bounds the error of using it.  A ray that misses the
// primitive's unit shape INFLATED by eps = k1 * (max |o| + extent) + k2 certainly has no hit to report; everything else
// goes through the ancestors' matrices one by one in f64 as the reference does (wf2_leaf).
struct LeafPre {
    float m[12];     // rows 0..2 of inverse(primitive) * inverse(parent) * ... * inverse(outermost group), rounded to f32
    float k1, k2;    // see above (host: leaf_pre_of)
    uint32_t kind;   // RT_SPHERE / RT_CUBE / RT_CYLINDER / RT_CONE, or RT_PRE_NONE: no test for this leaf
    float ymid;      // conics: centre and half height of the y range
    float yhalf;
    float radius;    // conics: 1 (cylinder) or max(|min|, |max|) (cone)
    uint32_t anc[2];  // the enclosing groups, outermost first (RT_PRE_NO_ANC: none / no more); RT_PRE_DEEP in anc[0]: nested deeper,
                      // walk the parent links instead
};
#define RT_PRE_NO_ANC 0xffffffffu
#define RT_PRE_DEEP 0xfffffffeu
static_assert(sizeof(LeafPre) == 80, "five 16-byte words");
#define RT_PRE_NONE 0xffu

struct Wf2Accel {
    const Bvh4Node* nodes;
    uint32_t root;       // inner node index, or a leaf / DONE code when the tree is that small
    uint32_t root_open;  // the same over the OPEN primitives only (planes, triangles, uncapped or unbounded conics): see WF2_CONTAINERS
    uint32_t n_always;   // leaves with unbounded boxes (planes ...): every ray tests them, outside the hierarchy
    uint32_t always[8];
    float extent[3];     // largest |coordinate| of any finite box, per axis: scales the f32 slack of the box test
    uint32_t has_cands;  // the scene has a direction index
    uint32_t chunk;      // rays a wave takes from the queue at a time
    uint32_t refill;     // idle lanes that make a wave go back to the queue
    float slack;         // slack of the f32 box test, as a fraction of |o| + extent (2^-20; RT_HIP_WF2_SLACK scales it: studies)
    float seg_near;      // containers walk: how far behind the origin the segment walk starts (RT_HIP_WF2_SEG_NEAR: studies)
    const LeafPre* pre;  // by shape index; null: no f32 test ahead of the leaves (RT_HIP_WF2_PRE=0: studies)
    float extent_max;
    uint32_t list_share;  // containers walks: chunks per resident wave a launch is cut into at least (RT_HIP_WF2_LIST_SHARE)
};

// ---- the f32 side of a ray: conservative slab tests ------------------------------------------------------------
// t_lo(axis, child) = lo * inv + cl, t_hi = hi * inv + ch with cl / ch = -o * inv shifted outward by the slack
// K = |inv| * 2^-20 * (|o| + extent): every rounding of the f32 evaluation (o and inv rounded to f32, the product-sum,
// and a relative 2^-23 of t, whose spatial size along the axis is bounded by |o| + extent) moves a plane by less.
struct RayF32 {
    float inv[3], cn[3], cf[3];  // slope; the constant of the plane met first (shifted down), of the one met second (shifted up)
};
RT_DEV float f32_up(double x) { return __double2float_ru(x); }
RT_DEV float f32_down(double x) { return __double2float_rd(x); }
RT_DEV void ray_f32(const Ray& r, const float extent[3], float slack, RayF32& f) {
    const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // the slope in f32 from the start (d rounded, then an IEEE f32 division: 2^-23 relative together, inside the slack's
        // budget; three f64 divisions per ray less).  A direction component of (almost) zero: any huge finite slope says
        // the same about every box
        float invf = 1.0f / (float)d[a];
        if (!(fabsf(invf) <= 1e18f)) invf = (d[a] < 0.0 || (d[a] == 0.0 && __double_as_longlong(d[a]) < 0)) ? -1e18f : 1e18f;
        const double c = -o[a] * (double)invf;
        const double k = fabs((double)invf) * (double)slack * (fabs(o[a]) + (double)extent[a]);
        f.inv[a] = invf;
        // the plane met first is lo when the slope is positive, hi when it is negative (walk_step picks by the slope's
        // sign); first planes move down, second up
        f.cn[a] = f32_down(c - k);
        f.cf[a] = f32_up(c + k);
    }
}

struct Walk {
    uint32_t cur;  // node at hand: inner index, RT_BVH4_LEAF | shape, or RT_BVH4_DONE
    int sp;        // stack entries (entry 0 is the DONE sentinel)
};

// The four child boxes of one node against the ray segment [t_near, t_far]: which of them the ray meets, and where it
// enters them.  `org` / `expw` / `qlo` / `qhi`: the node's words as walk_step reads them per lane, or as a wave reads them
// through scalar loads when all its lanes test the same node (wf2_packet_kernel).
// (t_near = 0 for a ray; the containers walk also wants what lies behind the origin: a negative t_near, or -inf.)
RT_DEV void node_children(const float org[3], uint32_t expw, const uint32_t kid[4], const uint32_t qlo[3], const uint32_t qhi[3],
                          const RayF32& f, float t_near, float t_far, bool met[4], float tn_out[4]) {
    // plane(axis, child) = org + q * step, so t = q * (step * inv) + (org * inv + c): the slack in c (ray_f32) also covers
    // the one extra rounding of org * inv + c.  Which of an axis' two planes the ray meets first is the sign of its slope:
    // the four children's lo or hi bytes are picked as one word per axis, and entry / exit are a max3 / min3 of products --
    // no per-child min / max of the two planes; the twelve products per side go two to an instruction (v_pk_fma_f32).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 n01[3], n23[3], f01[3], f23[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float step = __uint_as_float(((expw >> (8 * a)) & 0xffu) << 23);
        const float sinv = step * f.inv[a];
        const float cn = __builtin_fmaf(org[a], f.inv[a], f.cn[a]), cf = __builtin_fmaf(org[a], f.inv[a], f.cf[a]);
        const bool neg = f.inv[a] < 0.f;
        const uint32_t qn = neg ? qhi[a] : qlo[a], qf = neg ? qlo[a] : qhi[a];
        const f2 s2 = {sinv, sinv}, cn2 = {cn, cn}, cf2 = {cf, cf};
        const f2 bn01 = {(float)(qn & 0xffu), (float)((qn >> 8) & 0xffu)}, bn23 = {(float)((qn >> 16) & 0xffu), (float)(qn >> 24)};
        const f2 bf01 = {(float)(qf & 0xffu), (float)((qf >> 8) & 0xffu)}, bf23 = {(float)((qf >> 16) & 0xffu), (float)(qf >> 24)};
        n01[a] = __builtin_elementwise_fma(bn01, s2, cn2), n23[a] = __builtin_elementwise_fma(bn23, s2, cn2);
        f01[a] = __builtin_elementwise_fma(bf01, s2, cf2), f23[a] = __builtin_elementwise_fma(bf23, s2, cf2);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float nx = c < 2 ? n01[0][c & 1] : n23[0][c & 1], ny = c < 2 ? n01[1][c & 1] : n23[1][c & 1], nz = c < 2 ? n01[2][c & 1] : n23[2][c & 1];
        const float fx = c < 2 ? f01[0][c & 1] : f23[0][c & 1], fy = c < 2 ? f01[1][c & 1] : f23[1][c & 1], fz = c < 2 ? f01[2][c & 1] : f23[2][c & 1];
        const float tn = fmaxf(fmaxf(nx, ny), nz);
        const float tf = fminf(fminf(fx, fy), fz);
        // (negated comparisons: a NaN anywhere lets the box through, as the f64 walk does)
        met[c] = kid[c] != RT_BVH4_DONE && !(tn > tf) && !(tf < t_near) && !(tn > t_far);
        tn_out[c] = tn;
    }
}

// One step of the walk at an inner node: the four child boxes against the ray segment [t_near, t_far]; the nearest
// child met becomes the node at hand, the others go on the stack.
template <int LDS_N = RT_WF2_LDS_STACK, int STRIDE = 64, bool STATS, int FLAT>
RT_DEV void walk_step(const Wf2Accel& A, const RayF32& f, float t_near, float t_far, Walk& w, uint32_t* lds_stack, uint32_t* deep,
                      Ctx<STATS, FLAT>& cx) {
    const uint4* __restrict__ q = reinterpret_cast<const uint4*>(A.nodes + w.cur);
    const uint4 w0 = q[0], w1 = q[1], w2 = q[2];  // org + exponents | children | qlo[3], qhi[0]
    const uint2 w3 = *reinterpret_cast<const uint2*>(q + 3);  // qhi[1], qhi[2]
    count(cx, ST_ACCEL_F32, 4u);
    const uint32_t kid[4] = {w1.x, w1.y, w1.z, w1.w};
    const uint32_t qlo[3] = {w2.x, w2.y, w2.z}, qhi[3] = {w2.w, w3.x, w3.y};
    const float org[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
    bool met_c[4];
    float tn_c[4];
    node_children(org, w0.w, kid, qlo, qhi, f, t_near, t_far, met_c, tn_c);
    uint32_t next = RT_BVH4_DONE;
    float t_next = __builtin_inff();
    bool have = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tn = tn_c[c];
        if (met_c[c]) {
            uint32_t other = kid[c];
            if (!have || tn < t_next) {
                other = next;
                next = kid[c];
                t_next = tn;
            }
            if (have) {
                if (w.sp < LDS_N)
                    lds_stack[w.sp * STRIDE] = other;
                else
                    deep[w.sp - LDS_N] = other;
                ++w.sp;
            }
            have = true;
        }
    }
    if (have) {
        w.cur = next;
    } else {
        --w.sp;
        w.cur = w.sp < LDS_N ? lds_stack[w.sp * STRIDE] : deep[w.sp - LDS_N];
    }
}
template <int LDS_N = RT_WF2_LDS_STACK, int STRIDE = 64>
RT_DEV void walk_pop(Walk& w, const uint32_t* lds_stack, const uint32_t* deep) {
    --w.sp;
    w.cur = w.sp < LDS_N ? lds_stack[w.sp * STRIDE] : deep[w.sp - LDS_N];
}

// false: the ray's LINE certainly misses the leaf's primitive (see LeafPre).  `reach` = max |o| + extent of the ray.
// (r0, r1, r2 = the rows of the record's matrix, kk = k1, k2, kind, ymid, cc = yhalf, radius, anc[0], anc[1])
RT_DEV bool leaf_may_hit_core(float4 r0, float4 r1, float4 r2, float4 kk, float4 cc, const Ray& ray, float reach) {
    const unsigned kind = __float_as_uint(kk.z);
    if (kind == RT_PRE_NONE) return true;
    const float ox = (float)ray.o.x, oy = (float)ray.o.y, oz = (float)ray.o.z;
    const float dx = (float)ray.d.x, dy = (float)ray.d.y, dz = (float)ray.d.z;
    const float px0 = __builtin_fmaf(r0.x, ox, __builtin_fmaf(r0.y, oy, __builtin_fmaf(r0.z, oz, r0.w)));
    const float py0 = __builtin_fmaf(r1.x, ox, __builtin_fmaf(r1.y, oy, __builtin_fmaf(r1.z, oz, r1.w)));
    const float pz0 = __builtin_fmaf(r2.x, ox, __builtin_fmaf(r2.y, oy, __builtin_fmaf(r2.z, oz, r2.w)));
    const float ux = __builtin_fmaf(r0.x, dx, __builtin_fmaf(r0.y, dy, r0.z * dz));
    const float uy = __builtin_fmaf(r1.x, dx, __builtin_fmaf(r1.y, dy, r1.z * dz));
    const float uz = __builtin_fmaf(r2.x, dx, __builtin_fmaf(r2.y, dy, r2.z * dz));
    const float uu = ux * ux + uy * uy + uz * uz;
    if (!(uu > 1e-30f) || !(uu < 1e30f)) return true;
    // the point of the line nearest the object's origin (any point of the line would do: this one keeps the numbers small)
    const float sft = (px0 * ux + py0 * uy + pz0 * uz) / uu;
    const float px = __builtin_fmaf(-sft, ux, px0), py = __builtin_fmaf(-sft, uy, py0), pz = __builtin_fmaf(-sft, uz, pz0);
    const float eps = __builtin_fmaf(kk.x, reach, kk.y);
    const float grow = 1.00002f;  // the comparisons' own rounding
    if (kind == RT_SPHERE) {
        const float r = 1.f + eps;
        return !(px * px + py * py + pz * pz > r * r * grow);
    }
    // line against a box (centre (0, ymid, 0), half extents h): the separating axes of a line and a box are u x e_i
    float hx = 1.f + eps, hy = 1.f + eps, hz = 1.f + eps, wy = py;
    if (kind != RT_CUBE) {
        hx = hz = cc.y + eps;  // radius
        hy = cc.x + eps;       // half height (inf: unbounded)
        wy = py - kk.w;        // centre of the y range
    }
    const float ax = fabsf(ux), ay = fabsf(uy), az = fabsf(uz);
    const float cx_ = wy * uz - pz * uy, cy_ = pz * ux - px * uz, cz_ = px * uy - wy * ux;
    if (fabsf(cx_) > (hy * az + hz * ay) * grow) return false;
    if (fabsf(cy_) > (hz * ax + hx * az) * grow) return false;
    if (fabsf(cz_) > (hx * ay + hy * ax) * grow) return false;
    if (kind == RT_CYLINDER) {  // and the line against the infinite cylinder of radius 1 + eps about the y axis
        const float den = ux * ux + uz * uz;
        if (den > 1e-6f * uu) {
            const float r = 1.f + eps;
            if (cy_ * cy_ > r * r * den * grow) return false;
        }
    }
    return true;
}
RT_DEV bool leaf_may_hit(const LeafPre* __restrict__ pre, const Ray& ray, float reach, uint32_t& anc0, uint32_t& anc1) {
    const float4* __restrict__ q = reinterpret_cast<const float4*>(pre);
    const float4 r0 = q[0], r1 = q[1], r2 = q[2], kk = q[3], cc = q[4];
    anc0 = __float_as_uint(cc.z), anc1 = __float_as_uint(cc.w);
    return leaf_may_hit_core(r0, r1, r2, kk, cc, ray, reach);
}

// The ray in the space of a leaf's parent: through the enclosing groups' transforms, outermost first (shape.rs:28-32 per
// level).  anc0 / anc1 name them when the leaf's record does (at most two levels: no dependent reads of parent links);
// RT_PRE_DEEP: walk the links.  `gates`: also ask every group's gate (compound.rs:93-95); false when one is closed.
template <bool GATES, bool STATS, int FLAT>
RT_DEV bool leaf_way_down(const SceneDev& S, int parent, uint32_t anc0, uint32_t anc1, Ray& r, Ctx<STATS, FLAT>& cx) {
    auto level = [&](int g_idx) -> bool {
        const RtShape* __restrict__ g = S.shapes + g_idx;
        r = transform_ray(g->inv, r);
        if (!GATES) {
            count(cx, ST_RAY_XFORM);  // (the second pass, with the gates, repeats transforms already counted)
            return true;
        }
        if (RT_KIND(g->kind_flags) != RT_GROUP) return true;
        const RtGroup* __restrict__ gg = S.groups + g->param;
        count(cx, ST_GATES);
        return gg->n_children != 0u && box_intersects(gg->bmin, gg->bmax, r);
    };
    if (anc0 != RT_PRE_DEEP) {
        if (anc0 != RT_PRE_NO_ANC && !level((int)anc0)) return false;
        if (anc1 != RT_PRE_NO_ANC && !level((int)anc1)) return false;
        return true;
    }
    if (parent < 0) return true;
    int chain[RT_STACK];
    const int n = ancestor_chain(S, parent, chain);
    for (int q = n - 1; q >= 0; --q)
        if (!level(chain[q])) return false;
    return true;
}

// A leaf of the world-level hierarchy (a primitive, possibly nested in groups; scenes with CSG nodes keep the older
// kernels): the ray goes through the ancestors' transforms, outermost first, and the primitive is intersected with
// the reference's arithmetic (shape.rs:28-32, primitive.rs:210-233).  The ancestors' gates (compound.rs:93-95)
// decide whether the reference reaches the primitive at all; they are evaluated only when the primitive has a hit
// to offer that could change the answer -- a hit behind a closed gate and no hit at all leave the same result.
template <bool SHADOW, bool STATS, int FLAT>
RT_DEV bool wf2_leaf(const SceneDev& S, int s, const Ray& world_ray, double limit, Best& best, Ctx<STATS, FLAT>& cx,
                     uint32_t anc0 = RT_PRE_DEEP, uint32_t anc1 = RT_PRE_NO_ANC) {
    const RtShape* __restrict__ sh = S.shapes + s;
    const unsigned kf = sh->kind_flags;
    if (SHADOW && !(kf & RT_FLAG_HAS_SHADOW)) return false;
    const int parent = anc0 == RT_PRE_DEEP ? sh->parent : -1;
    Ray at = world_ray;
    leaf_way_down<false>(S, parent, anc0, anc1, at, cx);
    PrimHits h;
    prim_intersect(S, sh, kf, at, h, cx);
    bool useful = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((h.keep >> k) & 1u) {
            const double t = h.t[k];
            if (SHADOW)
                useful = useful || (t >= 0.0 && t < limit);
            else
                useful = useful || (t >= 0.0 && (t < best.t || (t == best.t && s > best.idx)));
        }
    }
    if (!useful) return false;
    {  // the way down again, this time past the gates
        Ray r = world_ray;
        if (!leaf_way_down<true>(S, parent, anc0, anc1, r, cx)) return false;
    }
    if (SHADOW) return true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((h.keep >> k) & 1u) offer(best, h.t[k], s);
    return false;
}

// wf2_leaf for a leaf that the lanes of a wave test TOGETHER (wf2_packet_kernel): `s`, `anc0` and `anc1` are the same in every
// lane, so the shape's record, its enclosing groups' records and their gates come through scalar loads (one per wave, the
// values in scalar registers) instead of one vector read per lane and word; each lane brings its own ray.  Same operations
// in the same order per lane as wf2_leaf (prim_intersect_rec is prim_intersect with the record handed in).  For leaves nested
// at most two groups deep (anc0 != RT_PRE_DEEP).
template <bool SHADOW, bool STATS, int FLAT>
RT_DEV bool wf2_leaf_uniform(const SceneDev& S, int s, uint32_t anc0, uint32_t anc1, const Ray& world_ray, double limit, Best& best,
                             Ctx<STATS, FLAT>& cx) {
    const RtShape rec = load_uniform(S.g_shapes + s);
    const unsigned kf = rec.kind_flags;
    if (SHADOW && !(kf & RT_FLAG_HAS_SHADOW)) return false;
    Ray at = world_ray;
    if (anc0 != RT_PRE_NO_ANC) {
        const RtShape g = load_uniform(S.g_shapes + anc0);
        at = transform_ray(g.inv, at);
        count(cx, ST_RAY_XFORM);
    }
    if (anc1 != RT_PRE_NO_ANC) {
        const RtShape g = load_uniform(S.g_shapes + anc1);
        at = transform_ray(g.inv, at);
        count(cx, ST_RAY_XFORM);
    }
    const unsigned kind = RT_KIND(kf);
    RtConic conic{0.0, 0.0};
    if (kind == RT_CYLINDER || kind == RT_CONE) conic = load_uniform(S.g_conics + rec.param);
    RtTriangle tri;
    if (kind == RT_TRIANGLE || kind == RT_SMOOTH_TRIANGLE) tri = load_uniform(S.g_tris + rec.param);
    PrimHits h;
    prim_intersect_rec(rec.inv, kf, &conic, &tri, at, h, cx);
    bool useful = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((h.keep >> k) & 1u) {
            const double t = h.t[k];
            if (SHADOW)
                useful = useful || (t >= 0.0 && t < limit);
            else
                useful = useful || (t >= 0.0 && (t < best.t || (t == best.t && s > best.idx)));
        }
    }
    // the way down again, this time past the gates (compound.rs:93-95), for the lanes that have a hit to offer
    bool open = true;
    if (useful) {
        Ray r = world_ray;
#pragma unroll
        for (int level = 0; level < 2; ++level) {
            const uint32_t anc = level == 0 ? anc0 : anc1;
            if (anc == RT_PRE_NO_ANC) continue;
            const RtShape g = load_uniform(S.g_shapes + anc);
            r = transform_ray(g.inv, r);
            if (RT_KIND(g.kind_flags) != RT_GROUP) continue;
            const RtGroup gg = load_uniform(S.groups + g.param);
            count(cx, ST_GATES);
            open = open && gg.n_children != 0u && box_intersects(gg.bmin, gg.bmax, r);
        }
    }
    if (!useful || !open) return false;
    if (SHADOW) return true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((h.keep >> k) & 1u) offer(best, h.t[k], s);
    return false;
}

// The sorted hit list of one lane (stands for BTreeSet<Intersection>; see list_insert) and the containers set.
struct LaneList {
    double lt[RT_LIST_CAP];
    uint32_t lid[RT_LIST_CAP];
    RT_DEV double& LT(int i) { return lt[i]; }
    RT_DEV uint32_t& LID(int i) { return lid[i]; }
    RT_DEV int cap() const { return RT_LIST_CAP; }
};
// The containers loop of prepare_computations (intersection.rs:50-75) on the sorted list.  containers_in_registers: the
// reference's loop as it stands, the `containers` vector in four registers (it rarely holds more than two objects; a
// per-lane array in private memory is shifted per entry by containers_walk of rt_trace.h); false when a fifth object would
// have to be pushed.  containers_from_list then answers from the list alone, for any depth:
// what the loop holds when it reaches the hit entry j* (the first entry ~equal to the hit) is the objects with an ODD number
// of entries before j*, in the order of their last entries (an object is pushed by its 1st, 3rd, ... entry and removed by
// its 2nd, 4th, ...; removing one does not reorder the others).  So `containers.last()` is found by going back from j*:
// the first entry that is its object's last one before j* and whose object has an odd count there.
//   n1 = last before the hit entry is processed (1.0 if none); the hit's object is then removed (odd count) or pushed
//   (even); n2 = last after that.
// The loop's state, so that it can run over a list that comes in pieces (a line walked by several lanes, each holding the
// entries of its own range of the ray parameter: see wf2_trace_kernel): phase 0 = still looking for the hit entry,
// 1 = past it (n1, n2 final), 2 = a fifth object would have to be pushed (the list-only form has to answer).
struct ContainersState {
    uint32_t c0, c1, c2, c3;
    int nc, phase;
    double n1, n2;
};
RT_DEV void containers_begin(ContainersState& st) {
    st.c0 = st.c1 = st.c2 = st.c3 = 0u;
    st.nc = 0, st.phase = 0;
    st.n1 = 1.0, st.n2 = 1.0;
}
RT_DEV void containers_run(const SceneDev& S, LaneList& W, int n, double hit_t, ContainersState& st) {
    if (st.phase != 0) return;
    uint32_t c0 = st.c0, c1 = st.c1, c2 = st.c2, c3 = st.c3;
    int nc = st.nc;
    auto last = [&]() { return nc == 1 ? c0 : (nc == 2 ? c1 : (nc == 3 ? c2 : c3)); };
    for (int j = 0; j < n; ++j) {
        const bool is_self = t_eq(W.LT(j), hit_t);
        const uint32_t id = W.LID(j);
        if (is_self && nc > 0) st.n1 = S.mats[S.shapes[last()].material].refractive_index;
        const int pos = (nc > 0 && c0 == id) ? 0 : ((nc > 1 && c1 == id) ? 1 : ((nc > 2 && c2 == id) ? 2 : ((nc > 3 && c3 == id) ? 3 : -1)));
        if (pos >= 0) {  // remove: the later ones move up
            if (pos <= 0) c0 = c1;
            if (pos <= 1) c1 = c2;
            if (pos <= 2) c2 = c3;
            --nc;
        } else {
            if (nc == 4) {
                st.phase = 2;
                return;
            }
            if (nc == 0) c0 = id;
            else if (nc == 1) c1 = id;
            else if (nc == 2) c2 = id;
            else c3 = id;
            ++nc;
        }
        if (is_self) {
            if (nc > 0) st.n2 = S.mats[S.shapes[last()].material].refractive_index;
            st.phase = 1;
            break;
        }
    }
    st.c0 = c0, st.c1 = c1, st.c2 = c2, st.c3 = c3, st.nc = nc;
}
RT_DEV bool containers_in_registers(const SceneDev& S, LaneList& W, int n, double hit_t, double& n1, double& n2) {
    ContainersState st;
    containers_begin(st);
    containers_run(S, W, n, hit_t, st);
    n1 = st.n1, n2 = st.n2;
    return st.phase != 2;
}
RT_DEV void containers_from_list(const SceneDev& S, LaneList& W, int n, double hit_t, double& n1, double& n2) {
    n1 = 1.0, n2 = 1.0;
    int js = -1;
    for (int j = 0; j < n && js < 0; ++j)
        if (t_eq(W.LT(j), hit_t)) js = j;
    if (js < 0) return;  // (no entry at the hit: the reference's loop runs off the list with both still 1.0)
    const uint32_t hit_id = W.LID(js);
    // the last container among the entries [0, js), leaving out object `skip` (0xffffffff: nobody)
    auto last_container = [&](uint32_t skip, bool& any) -> uint32_t {
        for (int j = js - 1; j >= 0; --j) {
            const uint32_t id = W.LID(j);
            if (id == skip) continue;
            bool later = false;
            for (int k = j + 1; k < js; ++k) later = later || W.LID(k) == id;
            if (later) continue;  // not this object's last entry
            unsigned count = 0;
            for (int k = 0; k <= j; ++k) count += W.LID(k) == id ? 1u : 0u;
            if (count & 1u) {
                any = true;
                return id;
            }
        }
        any = false;
        return 0u;
    };
    bool any = false;
    const uint32_t before = last_container(0xffffffffu, any);
    if (any) n1 = S.mats[S.shapes[before].material].refractive_index;
    unsigned hit_count = 0;
    for (int k = 0; k < js; ++k) hit_count += W.LID(k) == hit_id ? 1u : 0u;
    if (hit_count & 1u) {  // the hit leaves its object: whoever else is last
        const uint32_t after = last_container(hit_id, any);
        if (any) n2 = S.mats[S.shapes[after].material].refractive_index;
    } else {  // the hit enters its object: that object is last
        n2 = S.mats[S.shapes[hit_id].material].refractive_index;
    }
}

// wf2_leaf for World::intersections as a sorted list cut at t <= bound (world_list), the half that does not touch the list:
// the primitive's collected hits in `h`, -> the mask of those the list takes (none when a gate on the way down is closed).
template <bool STATS, int FLAT>
RT_DEV unsigned wf2_leaf_hits(const SceneDev& S, int s, const Ray& world_ray, double bound, PrimHits& h, Ctx<STATS, FLAT>& cx,
                              uint32_t anc0 = RT_PRE_DEEP, uint32_t anc1 = RT_PRE_NO_ANC) {
    const RtShape* __restrict__ sh = S.shapes + s;
    const unsigned kf = sh->kind_flags;
    const int parent = anc0 == RT_PRE_DEEP ? sh->parent : -1;
    Ray at = world_ray;
    if (!leaf_way_down<true>(S, parent, anc0, anc1, at, cx)) return 0u;
    prim_intersect(S, sh, kf, at, h, cx);
    unsigned keep = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (((h.keep >> k) & 1u) && h.t[k] <= bound) keep |= 1u << k;
    return keep;
}

// ... and the whole: the primitive's surviving hits are inserted with BTreeSet::append's tie rule.  Returns the new length.
template <bool STATS, int FLAT>
// (keep_lo <= t < keep_hi: the range of the ray parameter whose entries this lane keeps, when a line is shared out among lanes)
RT_DEV int wf2_leaf_list(const SceneDev& S, LaneList& W, int top, int s, const Ray& world_ray, double bound, Ctx<STATS, FLAT>& cx,
                         uint32_t anc0 = RT_PRE_DEEP, uint32_t anc1 = RT_PRE_NO_ANC, double keep_lo = -__builtin_inf(),
                         double keep_hi = __builtin_inf()) {
    const RtShape* __restrict__ sh = S.shapes + s;
    const unsigned kf = sh->kind_flags;
    const int parent = anc0 == RT_PRE_DEEP ? sh->parent : -1;
    // (the gates on the way down, not after the fact as wf2_leaf has them: nearly every hit on the line counts here, and the
    // second way down costs more than the gates of the leaves that yield nothing)
    Ray at = world_ray;
    if (!leaf_way_down<true>(S, parent, anc0, anc1, at, cx)) return top;
    PrimHits h;
    prim_intersect(S, sh, kf, at, h, cx);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (((h.keep >> k) & 1u) && h.t[k] <= bound && h.t[k] >= keep_lo && h.t[k] < keep_hi) {
            top = list_insert(W, 0, top, h.t[k], (uint32_t)s, cx.status);
            count(cx, ST_LIST_ENTRIES);
        }
    }
    return top;
}

// Direction-index candidates of one ray, evaluated like the older kernels evaluate them (cand_next + process_leaf).
template <bool SHADOW, bool STATS, int FLAT>
RT_DEV bool wf2_candidates(const SceneDev& S, const Ray& ray, double limit, Best& best, Ctx<STATS, FLAT>& cx) {
    CandIter it;
    cand_begin(S, ray, it);
    const V3 d = ray.d;
    bool found = false;
    while (it.k < it.e && !found) {
        const RtDirCand* __restrict__ c = S.dir_cands + (it.scan_all ? it.k : S.dir_entries[it.k]);
        ++it.k;
        const double a = c->q[0] * d.x * d.x + c->q[1] * d.y * d.y + c->q[2] * d.z * d.z +
                         2.0 * (c->q[3] * d.x * d.y + c->q[4] * d.x * d.z + c->q[5] * d.y * d.z);
        if (!(fabs(a) <= c->thr)) continue;
        found = wf2_leaf<SHADOW>(S, (int)c->visit, ray, limit, best, cx) && SHADOW;
    }
    return found;
}

// One ray through the 4-wide hierarchy: what trace() below does over the skip-link nodes, for the scenes that have one
// (S.accel4: world-level hierarchy over leaves, no CSG node, no shadowless primitive).  shadow == false: closest hit into
// `best`; shadow == true: true as soon as a shadow-casting primitive is met at 0 <= t < limit.  The stack's first
// RT_TRACE4_LDS_STACK entries per lane live in LDS (blocks of up to 256 threads), deeper ones in private memory.
#ifndef RT_TRACE4_LDS_STACK
#define RT_TRACE4_LDS_STACK 8
#endif
template <bool STATS, int FLAT>
RT_DEV bool trace4(const SceneDev& S, const Ray& ray, bool shadow, double limit, Best& best, Ctx<STATS, FLAT>& cx) {
    const Wf2Accel& A = *S.accel4;
    __shared__ uint32_t lds_stack_all[RT_TRACE4_LDS_STACK * 256];
    uint32_t* lds_stack = lds_stack_all + (threadIdx.x & 255u);
    uint32_t deep[RT_WF2_MAX_STACK - RT_TRACE4_LDS_STACK];
    if (A.has_cands) {  // the direction index's candidates first
        if (shadow ? wf2_candidates<true>(S, ray, limit, best, cx) : wf2_candidates<false>(S, ray, limit, best, cx)) return true;
    }
    RayF32 f;
    ray_f32(ray, A.extent, A.slack, f);
    float t_far = shadow ? f32_up(limit) : f32_up(best.t);
    Walk w;
    lds_stack[0] = RT_BVH4_DONE;
    w.sp = 1;
    w.cur = A.root;
    for (unsigned k = 0; k < A.n_always; ++k) {  // (n_always <= 8 - 1 < RT_TRACE4_LDS_STACK is not assumed: the general push)
        if (w.sp < RT_TRACE4_LDS_STACK) lds_stack[w.sp * 256] = w.cur;
        else deep[w.sp - RT_TRACE4_LDS_STACK] = w.cur;
        ++w.sp;
        w.cur = RT_BVH4_LEAF | A.always[k];  // (always[] carries RT_BVH4_NOPRE where it applies)
    }
    for (;;) {
        // box tests only, until this lane holds a leaf or is through: the lanes of a wave meet again at the leaves
        while (w.cur < RT_BVH4_DONE) walk_step<RT_TRACE4_LDS_STACK, 256>(A, f, 0.f, t_far, w, lds_stack, deep, cx);
        if (w.cur == RT_BVH4_DONE) break;
        const int s = (int)RT_BVH4_SHAPE(w.cur);
        bool worth = true;
        uint32_t anc0 = RT_PRE_DEEP, anc1 = RT_PRE_NO_ANC;
        if (A.pre) {
            if (w.cur & RT_BVH4_NOPRE) {
                const uint2 anc = *reinterpret_cast<const uint2*>(&A.pre[s].anc[0]);
                anc0 = anc.x, anc1 = anc.y;
            } else {
                const float reach = fmaxf(fmaxf(fabsf((float)ray.o.x), fabsf((float)ray.o.y)), fabsf((float)ray.o.z)) * 1.000001f + A.extent_max;
                worth = leaf_may_hit(A.pre + s, ray, reach, anc0, anc1);
                count(cx, ST_ACCEL_F32);
            }
        }
        if (worth) {
            if (shadow) {
                if (wf2_leaf<true>(S, s, ray, limit, best, cx, anc0, anc1)) return true;
            } else {
                wf2_leaf<false>(S, s, ray, limit, best, cx, anc0, anc1);
                t_far = f32_up(best.t);
            }
        }
        walk_pop<RT_TRACE4_LDS_STACK, 256>(w, lds_stack, deep);
    }
    return false;
}
