// rt_kernels.h -- the gfx950 kernels of the render path (one lane = one pixel or one ray, FP64, no FMA contraction):
//   render_kernel   Camera::render        (camera.rs:51-80)  8x8-pixel tile per wave
//   wf_level_kernel / wf_mark_kernel / wf_combine_kernel   the same, one launch per generation of rays
//   color_at_kernel World::color_at       (world.rs:32-43)   batch of rays
//   probe_kernel    prepare_computations  (intersection.rs:45-99) for the KATs
//   shade_rays_kernel  intersect / hit / normal_at / lighting as user code calls them (exercises/shading.rs:44-57)
//   quantize_kernel Canvas::scale_to_ppm_data (canvas.rs:85-90)
// The templates are instantiated explicitly, a few variants per translation unit (rt_inst.hip, compiled once per
// scene class x counting variant so that the library builds in parallel); every other unit sees them `extern`.
#pragma once
#include <type_traits>

#include "rt_trace.h"

#ifndef RT_STAGE_MAX_BYTES
#define RT_STAGE_MAX_BYTES 16384u  // table image; with the 48 KB LDS list the block stays within 64 KB of dynamic LDS
#endif
#ifndef RT_STAGE_MAX_BYTES_COMPOUND
#define RT_STAGE_MAX_BYTES_COMPOUND 49152u  // scenes with groups / CSG (their lists stay in private memory): two blocks per CU keep 80 KB each
#endif
#ifndef RT_WAVES_PER_SIMD_FLAT
#define RT_WAVES_PER_SIMD_FLAT 2
#endif
#ifndef RT_WAVES_PER_SIMD_WF
#define RT_WAVES_PER_SIMD_WF 3  // generation kernels of large scenes: incoherent rays wait on dependent reads, a third wave hides more of it than ~60 spilled VGPRs cost
#endif
#ifndef RT_WAVES_PER_SIMD_DENSE_OPAQUE
#define RT_WAVES_PER_SIMD_DENSE_OPAQUE 3  // staged flat scenes that cannot recurse
#endif
#ifndef RT_WAVES_PER_SIMD_WF0
#define RT_WAVES_PER_SIMD_WF0 3  // ... and the camera-ray generation
#endif
#ifndef RT_WAVES_PER_SIMD_OPAQUE
#define RT_WAVES_PER_SIMD_OPAQUE 3  // general kernels of scenes that cannot recurse (no frame stack, no n1/n2): 168 VGPRs, ~57 spilled, still faster
#endif
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 2  // register budget of the tracing kernels: 512 / this many VGPR+AGPR per lane
#endif

// ============================================================== kernels ===
namespace rtk {
using namespace rtd;

struct RowMap {  // local row k -> image row (rt_hip_render_rows)
    uint32_t part, n_parts, band_rows, n_local_rows;
    uint32_t row0, batch_rows;  // kernel pipeline of large scenes (rt_wf2.h): the local rows [row0, row0 + batch_rows) at hand
};

template <bool STATS, int FLAT>
__device__ void flush_stats(Ctx<STATS, FLAT>& cx, RtStats* __restrict__ stats, unsigned* __restrict__ status) {
    if (cx.status) atomicOr(status, cx.status);
    if constexpr (STATS) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(stats);
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) {
            unsigned v = cx.cnt[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63u) == 0u && v) atomicAdd(out + i, (unsigned long long)v);
        }
    }
}

// FLAT: the scene holds only primitives and no acceleration structure (RT_SCENE_FLAT); that
// variant carries none of the group / CSG / BVH machinery and so needs far fewer registers.
//
// STAGED (FLAT scenes whose tables fit RT_STAGE_MAX_BYTES): the block first copies the packed
// table image into LDS and shades out of that copy; the per-lane, dependent record reads of
// prepare_computations / lighting / pattern evaluation then cost an LDS access, not an L2 one.
#ifdef RT_HOST_EMU  // (tests/emu: the sources compiled for the host under sanitizers; staged scenes are not run there)
inline __align__(16) unsigned char rt_lds[160u * 1024u];
#else
extern __shared__ __align__(16) unsigned char rt_lds[];
#endif

// RECURSE == false: no material of the scene reflects or transmits (world.rs:99,110 always take their
// early exits), so the kernel is instantiated with a recursion budget of zero: no frames, no n1 / n2,
// and under_point / reflect_v / the containers walk fall away as dead code.
// DENSE: 3 waves/SIMD (168 VGPRs, a few spilled) instead of 2 -- FLAT scenes whose LDS share lets three
// blocks live on a CU.  Measured on pattern.rs (no recursion) at 2 / 3 / 4 waves/SIMD: 0.97 / 0.83 /
// 1.02 ms; glass_bubble.rs 4.83 / 4.50 / -; the general kernel gains nothing (profiles/r01_variants.txt).
template <bool STATS, int FLAT, bool STAGED, bool RECURSE = true, bool DENSE = false>
__global__ __launch_bounds__(256, DENSE ? (RECURSE ? 3 : RT_WAVES_PER_SIMD_DENSE_OPAQUE) : (FLAT == RT_CLASS_FLAT ? RT_WAVES_PER_SIMD_FLAT : (RECURSE ? RT_WAVES_PER_SIMD : RT_WAVES_PER_SIMD_OPAQUE))) void render_kernel(const SceneDev S_in, RtCameraDesc cam, int depth, RowMap rows,
                                                     double* __restrict__ out, RtStats* __restrict__ stats,
                                                     unsigned* __restrict__ status) {
    SceneDev S = S_in;
    if constexpr (STAGED) {
        S.accel4 = nullptr;  // staged scenes walk the skip-link nodes in their LDS copy (a constant here: trace4 and its LDS stack drop out)
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(S_in.stage);
        uint4* dst = reinterpret_cast<uint4*>(rt_lds);
        for (unsigned i = threadIdx.x; i < S_in.stage_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        S.shapes = reinterpret_cast<const RtShape*>(rt_lds + S_in.stage_off[0]);
        S.conics = reinterpret_cast<const RtConic*>(rt_lds + S_in.stage_off[1]);
        S.mats = reinterpret_cast<const RtMaterial*>(rt_lds + S_in.stage_off[2]);
        S.pats = reinterpret_cast<const RtPattern*>(rt_lds + S_in.stage_off[3]);
        S.lights = reinterpret_cast<const RtLight*>(rt_lds + S_in.stage_off[4]);
        if constexpr (FLAT != RT_CLASS_FLAT) {  // small scenes with compounds: the traversal's tables too
            S.groups = reinterpret_cast<const RtGroup*>(rt_lds + S_in.stage_off[5]);
            S.csgs = reinterpret_cast<const RtCsg*>(rt_lds + S_in.stage_off[6]);
            S.lhit = reinterpret_cast<const uint8_t*>(rt_lds + S_in.stage_off[7]);
            S.bvh = reinterpret_cast<const RtBvhNode*>(rt_lds + S_in.stage_off[8]);
            S.tris = reinterpret_cast<const RtTriangle*>(rt_lds + S_in.stage_off[9]);
        }
    }
    // one 8x8 tile of pixels per wave (neighbouring pixels share a wave); a block is 1, 2x1 or 2x2 waves, as
    // launched: the fewer, the sooner a CU gets new work (a block's slot is free only when its slowest wave is done)
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned wx = blockDim.x >= 128u ? 2u : 1u, wy = (blockDim.x >> 6) / wx;
    const unsigned x = (blockIdx.x * wx + wave % wx) * 8u + (lane & 7u);
    const unsigned k = (blockIdx.y * wy + wave / wx) * 8u + (lane >> 3);
    Ctx<STATS, FLAT> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    bool active = x < cam.hsize && k < rows.n_local_rows;
    unsigned y = 0;
    if (active) {
        y = ((k / rows.band_rows) * rows.n_parts + rows.part) * rows.band_rows + k % rows.band_rows;
        active = y < cam.vsize;
    }
    if (active) {
        // STAGED scenes are small enough for the per-lane sorted list to live in LDS too
        typename std::conditional<STAGED && FLAT == RT_CLASS_FLAT, ScratchLds, Scratch>::type W;
        if constexpr (STAGED && FLAT == RT_CLASS_FLAT) W.attach(rt_lds + ((S_in.stage_bytes + 15u) & ~15u), threadIdx.x, S_in.lds_list_cap);
        Frame frames[RT_MAX_DEPTH + 1];
        const Ray ray = ray_for_pixel(cam, x, y);
        count(cx, ST_RAY_PRIMARY);
        const V3 c = color_at(S, W, frames, ray, RECURSE ? depth : 0, cx);
        double* o = out + ((size_t)k * cam.hsize + x) * 3u;
        o[0] = c.x;
        o[1] = c.y;
        o[2] = c.z;
    }
    flush_stats(cx, stats, status);
}

#ifndef RT_WF_BLOCKS_SHIFT
#define RT_WF_BLOCKS_SHIFT 2  // 2^2 neighbouring blocks of camera rays share a shard
#endif
#ifndef RT_WF_CHUNK_SHIFT
#define RT_WF_CHUNK_SHIFT 9   // 2^9 consecutive items of a slice send their children to one shard
#endif
// One generation of colour rays (see "wave-front organisation" in rt_trace.h).  PRIMARY: generation
// 0, rays from the camera, pixel tiles as in render_kernel.  Otherwise generation g > 0, rays from the
// queue slices generation g - 1 filled, whose lengths only the device knows: block b strides over
// slice b % RT_WF_SHARDS together with the other blocks of that slice.
// (Nothing the cold no-room path needs is kept in registers across color_at: it re-reads its item,
// or the camera record from the workspace.)
template <bool STATS, int FLAT, bool STAGED, bool PRIMARY>
__global__ __launch_bounds__(256, FLAT == RT_CLASS_FLAT ? RT_WAVES_PER_SIMD_FLAT : (PRIMARY ? RT_WAVES_PER_SIMD_WF0 : RT_WAVES_PER_SIMD_WF)) void wf_level_kernel(
    const SceneDev S_in, RtCameraDesc cam, int depth, RowMap rows, double* __restrict__ out, RtStats* __restrict__ stats,
    unsigned* __restrict__ status, WfSpace ws, unsigned level) {
    SceneDev S = S_in;
    if constexpr (STAGED) {
        S.accel4 = nullptr;  // staged scenes walk the skip-link nodes in their LDS copy (a constant here: trace4 and its LDS stack drop out)
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(S_in.stage);
        uint4* dst = reinterpret_cast<uint4*>(rt_lds);
        for (unsigned i = threadIdx.x; i < S_in.stage_bytes / 16u; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        S.shapes = reinterpret_cast<const RtShape*>(rt_lds + S_in.stage_off[0]);
        S.conics = reinterpret_cast<const RtConic*>(rt_lds + S_in.stage_off[1]);
        S.mats = reinterpret_cast<const RtMaterial*>(rt_lds + S_in.stage_off[2]);
        S.pats = reinterpret_cast<const RtPattern*>(rt_lds + S_in.stage_off[3]);
        S.lights = reinterpret_cast<const RtLight*>(rt_lds + S_in.stage_off[4]);
        if constexpr (FLAT != RT_CLASS_FLAT) {  // small scenes with compounds: the traversal's tables too
            S.groups = reinterpret_cast<const RtGroup*>(rt_lds + S_in.stage_off[5]);
            S.csgs = reinterpret_cast<const RtCsg*>(rt_lds + S_in.stage_off[6]);
            S.lhit = reinterpret_cast<const uint8_t*>(rt_lds + S_in.stage_off[7]);
            S.bvh = reinterpret_cast<const RtBvhNode*>(rt_lds + S_in.stage_off[8]);
            S.tris = reinterpret_cast<const RtTriangle*>(rt_lds + S_in.stage_off[9]);
        }
    }
    Ctx<STATS, FLAT> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    typename std::conditional<STAGED && FLAT == RT_CLASS_FLAT, ScratchLds, Scratch>::type W;
    if constexpr (STAGED && FLAT == RT_CLASS_FLAT) W.attach(rt_lds + ((S_in.stage_bytes + 15u) & ~15u), threadIdx.x, S_in.lds_list_cap);
    Frame frames[RT_MAX_DEPTH + 1];
    // Where this wave's children go.  A shard's slice is read back in arrival order, 64 items to a wave, so
    // what arrives together should belong together: four neighbouring blocks of camera rays (64 x 16 pixels),
    // or eight consecutive 64-item chunks of a slice, share one shard -- picked by a hash, so that every shard
    // sees the same load and few writers at a time.
    unsigned out_shard = (((blockIdx.y * gridDim.x + blockIdx.x) >> RT_WF_BLOCKS_SHIFT) * 2654435761u) >> 24;
    static_assert(RT_WF_SHARDS == 256, "the hash above keeps 8 bits");
    // One pass of one item: color_at up to the end of the lane's first shade_hit, then -- the wave
    // together -- its children into the next generation's queue, or its colour to whoever waits for it.
    // Returns false when the lane found no room and has to go again with `pending.open` cleared.
    WfPending pending;
    pending.open = true;
    auto pass = [&](bool active, const Ray& ray, int remaining, uint32_t parent) -> bool {
        bool deferred = false;
        Shade sd;
        V3 c = v3(0.0, 0.0, 0.0);
        const Ctx<STATS, FLAT> before = cx;
        if (active && PRIMARY) count(cx, ST_RAY_PRIMARY);
        if (active) c = color_at(S, W, frames, ray, remaining, cx, pending, deferred, sd);
        Children ch;
        ch.do_reflect = ch.do_refract = ch.use_schlick = false;
        double reflective = 0.0, transparency = 0.0;
        bool want = false;
        if (active && deferred) {
            const RtMaterial m = S.mats[sd.mat];
            reflective = m.reflective, transparency = m.transparency;
            shade_children(m, sd, Ray{sd.over, sd.reflect_v}, remaining, ch, cx);
            want = ch.do_reflect || ch.do_refract;
            if (!want) c = combine(sd.surface, v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 0.0), ch.use_schlick, ch.reflectance);
        }
        const bool placed = wf_commit(ws, out_shard, level, want, parent, sd.surface, reflective, transparency, ch, sd.over,
                                      sd.reflect_v, remaining - 1);
        if (want && !placed) {  // no room: forget this pass, the next one keeps the subtree
            const unsigned flags = cx.status;
            cx = before;
            cx.status |= flags;
            pending.open = false;
            return false;
        }
        if (placed) {
            if (ch.do_reflect) count(cx, ST_RAY_REFLECT);
            if (ch.do_refract) count(cx, ST_RAY_REFRACT);
        } else if (active) {
            wf_deliver(ws.nodes, out, parent, c);
        }
        pending.open = true;
        return true;
    };
    if constexpr (PRIMARY) {
        const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const unsigned x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
        const unsigned k = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
        bool active = x < cam.hsize && k < rows.n_local_rows;
        unsigned y = 0;
        if (active) {
            y = ((k / rows.band_rows) * rows.n_parts + rows.part) * rows.band_rows + k % rows.band_rows;
            active = y < cam.vsize;
        }
        while (__builtin_amdgcn_ballot_w64(active) != 0ull) {  // at most twice
            // (the camera record is read from the workspace so that it need not stay in registers)
            const Ray ray = active ? ray_for_pixel(*ws.cam, x, y) : Ray{v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};
            if (pass(active, ray, depth, RT_WF_PIXEL | (k * ws.cam->hsize + x))) active = false;
        }
    } else {
        const unsigned in_shard = blockIdx.x % RT_WF_SHARDS;
        // the shard's slice, octant by octant, as one sequence
        const unsigned sub_cap = ws.q_cap / RT_WF_BINS;
        unsigned upto[RT_WF_BINS];
        unsigned n_items = 0;
#pragma unroll
        for (unsigned o = 0; o < RT_WF_BINS; ++o) {
            const unsigned n = ws.shards[in_shard].items[level & 1u][o];
            n_items += n < sub_cap ? n : sub_cap;
            upto[o] = n_items;
        }
        const unsigned stride = (gridDim.x / RT_WF_SHARDS) * blockDim.x;
        const WfItem* q_in = ws.queue[level & 1u] + (size_t)in_shard * ws.q_cap;
        unsigned i = (blockIdx.x / RT_WF_SHARDS) * blockDim.x + threadIdx.x;
        while (i < n_items) {
            unsigned o = 0, first = 0;
#pragma unroll
            for (unsigned b = 0; b + 1u < RT_WF_BINS; ++b)
                if (i >= upto[b]) o = b + 1u, first = upto[b];
            const WfItem* it = q_in + (size_t)o * sub_cap + (i - first);
            out_shard = ((in_shard + RT_WF_SHARDS * ((unsigned)__builtin_amdgcn_readfirstlane((int)i) >> RT_WF_CHUNK_SHIFT)) * 2654435761u) >> 24;
            if (pass(it->remaining != RT_WF_EMPTY, it->ray, (int)it->remaining, it->parent)) i += stride;
        }
    }
    flush_stats(cx, stats, status);
}

template <int FLAT>
__global__ __launch_bounds__(256, FLAT == RT_CLASS_FLAT ? RT_WAVES_PER_SIMD_FLAT : RT_WAVES_PER_SIMD) void color_at_kernel(const SceneDev S, const double* __restrict__ rays,
                                                       unsigned n, int depth, double* __restrict__ out,
                                                       unsigned* __restrict__ status) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    Ctx<false, FLAT> cx;
    cx.status = 0;
    if (i < n) {
        Scratch W;
        Frame frames[RT_MAX_DEPTH + 1];
        const double* r = rays + (size_t)i * 6u;
        const Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])};
        const V3 c = color_at(S, W, frames, ray, depth, cx);
        out[(size_t)i * 3u] = c.x;
        out[(size_t)i * 3u + 1u] = c.y;
        out[(size_t)i * 3u + 2u] = c.z;
    }
    flush_stats(cx, nullptr, status);
}

template <int FLAT>
__global__ __launch_bounds__(256) void probe_kernel(const SceneDev S, const double* __restrict__ rays, unsigned n,
                                                    double* __restrict__ out, unsigned* __restrict__ status) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    Ctx<false, FLAT> cx;
    cx.status = 0;
    if (i < n) {
        Scratch W;
        const double* r = rays + (size_t)i * 6u;
        double* o = out + (size_t)i * 16u;
        for (int k = 0; k < 16; ++k) o[k] = 0.0;
        const Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])};
        Best hit;
        if (find_hit(S, W, ray, hit, cx)) {
            Comps c;
            prepare_computations(S, W, ray, hit, true, c, cx);
            o[0] = 1.0, o[1] = hit.t, o[2] = (double)hit.idx, o[3] = c.n1, o[4] = c.n2, o[5] = c.inside ? 1.0 : 0.0;
            o[6] = c.over_point.x, o[7] = c.over_point.y, o[8] = c.over_point.z;
            o[9] = c.normal_v.x, o[10] = c.normal_v.y, o[11] = c.normal_v.z;
            o[12] = c.reflect_v.x, o[13] = c.reflect_v.y, o[14] = c.reflect_v.z;
            if (S.n_lights > 0u) {
                const RtLight l = S.lights[0];
                o[15] = is_shadowed(S, W, c.over_point, v3(l.position[0], l.position[1], l.position[2]), cx) ? 1.0 : 0.0;
            }
        }
    }
    flush_stats(cx, nullptr, status);
}

// exercises/shading.rs:44-57 per ray: closest hit of the scene, the hit shape's normal there and its
// lighting at the point itself (no shadow test, normal not turned towards the eye).
template <int FLAT>
__global__ __launch_bounds__(256) void shade_rays_kernel(const SceneDev S, const RtLight light,
                                                         const double* __restrict__ rays, unsigned n,
                                                         double* __restrict__ out, uint8_t* __restrict__ out_hit,
                                                         unsigned* __restrict__ status) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    Ctx<false, FLAT> cx;
    cx.status = 0;
    if (i < n) {
        Scratch W;
        const double* r = rays + (size_t)i * 6u;
        const Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])};
        V3 c = v3(0.0, 0.0, 0.0);
        Best hit;
        const bool found = find_hit(S, W, ray, hit, cx);
        if (found) {
            const RtShape* __restrict__ sh = S.shapes + hit.idx;
            const unsigned kind = RT_KIND(sh->kind_flags);
            int chain[RT_STACK];
            const int nc = FLAT == RT_CLASS_FLAT ? 0 : ancestor_chain(S, sh->parent, chain);
            const V3 point = ray.o + ray.d * hit.t;
            const V3 local_point = world_to_object(S, sh, chain, nc, point);
            const V3 normal = normal_to_world(S, sh, chain, nc, local_normal(S, sh, kind, local_point, chain, nc, ray));
            const RtMaterial m = S.mats[sh->material];
            V3 color = v3(m.color[0], m.color[1], m.color[2]);
            if (m.pattern >= 0) {
                unsigned nodes = 0;
                color = pattern_eval(S.pats, m.pattern, local_point, nodes);
            }
            c = lighting(m, color, light, point, -ray.d, normal, false, cx);
        }
        out[(size_t)i * 3u] = c.x;
        out[(size_t)i * 3u + 1u] = c.y;
        out[(size_t)i * 3u + 2u] = c.z;
        if (out_hit) out_hit[i] = found ? 1 : 0;
    }
    flush_stats(cx, nullptr, status);
}


// ---- explicit instantiation lists ----------------------------------------------------------------------------
// X(template-id) for every variant launch_render can pick, per scene class and counting variant.
#define RT_RENDER_ARGS const SceneDev, RtCameraDesc, int, RowMap, double*, RtStats*, unsigned*
#define RT_WF_ARGS const SceneDev, RtCameraDesc, int, RowMap, double*, RtStats*, unsigned*, WfSpace, unsigned
#define RT_RENDER_LIST_COMPOUND(X, ST, CLS)          \
    X(render_kernel<ST, CLS, true, false, false>)    \
    X(render_kernel<ST, CLS, true, true, false>)     \
    X(render_kernel<ST, CLS, false, false, false>)   \
    X(render_kernel<ST, CLS, false, true, false>)
#define RT_RENDER_LIST_FLAT(X, ST)                             \
    RT_RENDER_LIST_COMPOUND(X, ST, RT_CLASS_FLAT)              \
    X(render_kernel<ST, RT_CLASS_FLAT, true, false, true>)     \
    X(render_kernel<ST, RT_CLASS_FLAT, true, true, true>)
#define RT_WF_LIST_COMPOUND(X, ST, CLS)          \
    X(wf_level_kernel<ST, CLS, false, true>)     \
    X(wf_level_kernel<ST, CLS, false, false>)
#define RT_WF_LIST_FLAT(X, ST)                              \
    RT_WF_LIST_COMPOUND(X, ST, RT_CLASS_FLAT)               \
    X(wf_level_kernel<ST, RT_CLASS_FLAT, true, true>)       \
    X(wf_level_kernel<ST, RT_CLASS_FLAT, true, false>)
#define RT_MISC_LIST(CLS)                                                                                              \
    RT_MISC_X(color_at_kernel<CLS>, (const SceneDev, const double*, unsigned, int, double*, unsigned*))                  \
    RT_MISC_X(probe_kernel<CLS>, (const SceneDev, const double*, unsigned, double*, unsigned*))                          \
    RT_MISC_X(shade_rays_kernel<CLS>, (const SceneDev, const RtLight, const double*, unsigned, double*, uint8_t*, unsigned*))

#define RT_EXTERN_RENDER(...) extern template __global__ void __VA_ARGS__(RT_RENDER_ARGS);
#define RT_EXTERN_WF(...) extern template __global__ void __VA_ARGS__(RT_WF_ARGS);
#define RT_MISC_X(ID, ARGS) extern template __global__ void ID ARGS;
RT_RENDER_LIST_FLAT(RT_EXTERN_RENDER, false)
RT_RENDER_LIST_FLAT(RT_EXTERN_RENDER, true)
RT_RENDER_LIST_COMPOUND(RT_EXTERN_RENDER, false, RT_CLASS_GENERAL)
RT_RENDER_LIST_COMPOUND(RT_EXTERN_RENDER, true, RT_CLASS_GENERAL)
RT_RENDER_LIST_COMPOUND(RT_EXTERN_RENDER, false, RT_CLASS_NESTED_LEAVES)
RT_RENDER_LIST_COMPOUND(RT_EXTERN_RENDER, true, RT_CLASS_NESTED_LEAVES)
RT_WF_LIST_FLAT(RT_EXTERN_WF, false)
RT_WF_LIST_FLAT(RT_EXTERN_WF, true)
RT_WF_LIST_COMPOUND(RT_EXTERN_WF, false, RT_CLASS_GENERAL)
RT_WF_LIST_COMPOUND(RT_EXTERN_WF, true, RT_CLASS_GENERAL)
RT_WF_LIST_COMPOUND(RT_EXTERN_WF, false, RT_CLASS_NESTED_LEAVES)
RT_WF_LIST_COMPOUND(RT_EXTERN_WF, true, RT_CLASS_NESTED_LEAVES)
RT_MISC_LIST(RT_CLASS_FLAT)
RT_MISC_LIST(RT_CLASS_GENERAL)
RT_MISC_LIST(RT_CLASS_NESTED_LEAVES)
#undef RT_MISC_X

}  // namespace rtk
