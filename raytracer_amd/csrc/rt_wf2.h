// rt_wf2.h -- the render path of LARGE scenes as a pipeline of kernels per generation of rays
// ("wave-front" organisation, second form).  Same functions, same arithmetic and therefore the same
// frame, bit for bit, as render_kernel / wf_level_kernel; what changes is who does what when:
//
//   wf2_trace_kernel     World::intersections + hit (world.rs:45-53, intersection.rs:154-158), closest hit or
//                        shadow any-hit (world.rs:85-96): NOTHING but traversal.  One-wave blocks that stay resident
//                        and take 64 rays at a time from the launch's items (counters split 32 ways: see Wf2Counters);
//                        a wave first evaluates its rays' direction-index candidates (primitive.rs:486-490's branch,
//                        see RtDirCand; -DRT_WF2_CANDS_INLINE=0: wf2_cands_kernel ahead of every trace launch instead).
//                        The walk is over a 4-wide hierarchy with conservative f32 boxes (rt_bvh4.h: one 64-byte node
//                        read yields four child boxes: a quarter of the dependent reads of the 2-wide skip-link walk,
//                        box arithmetic in f32 instead of f64); every leaf the walk reaches is tested with the
//                        reference's own f64 arithmetic, so no decision changes.
//   wf2_prepare_kernel   prepare_computations + the pattern colour (intersection.rs:45-99, pattern.rs:76-91) of
//                        every hit into the ray's shading context (slot = ray index); one shadow feeler per light each.
//                        A third mode of the trace kernel builds the sorted list of the world's intersections up to a
//                        transparent hit and walks it for n1 / n2 (intersection.rs:50-75).  The reference lists what the whole LINE meets,
//                        behind the origin too, and so does this walk.  (Measured and not adopted: closed primitives
//                        -- sphere, cube, capped conic -- looked for only along the segment from the origin to the
//                        hit and only the open ones along the whole line, on the grounds that an entry PAIR wholly
//                        before the hit pushes and removes its primitive and leaves no trace in `containers`.  A third
//                        of the walk's time, 7.9 -> 2.6 ms per 4K frame of the 10k scene -- but a ray that clips the
//                        edge of a cube behind the origin with a chord under 1e-5 has its two entries merged into
//                        ONE by the set's dedupe (primitive.rs:227-230), an odd count that the reference keeps in
//                        `containers`: one pixel of 1.1 M differed at 7680x4320.  RT_HIP_WF2_SEG_NEAR=1 runs it.)
//   wf2_spawn_kernel     reflected / refracted children (world.rs:98-127, schlick intersection.rs:182-200) into the node
//                        pool and the next generation's queue: the next generation starts from here
//   wf2_light_kernel     lighting per light (primitive.rs:111-151) with the feelers' answers -- with the feelers on the
//                        scene's second stream, off the critical path (contexts are double-buffered by generation)
//   wf2_overflow_kernel  hits whose children found no room keep their subtree (color_at's frame stack)
//   wf2_scan_kernel, wf2_compact_kernel   the next generation's queue made dense, every slice sorted by where its rays
//                        start and the octant they point into
// and wf_combine_kernel folds children into parents from the deepest generation upwards (world.rs:76-82).
#pragma once
#include "rt_kernels.h"

namespace rtk {

struct Wf2Hit {  // closest hit of one colour ray
    double t;
    int32_t idx;
    uint32_t pad;
};

// Shading contexts of one generation's hits, one slot per hit, structure-of-arrays so that the lanes of a
// wave read and write consecutive words.
// What a hit keeps between the kernels is what cannot be had again bit for bit: over_point, eye, the normal, the pattern
// colour and t.  reflect_v = reflect(-eye, normal) and under_point = position(ray, t) - normal * EPSILON are worked out
// again, by the same expressions on the same bits, by the few hits that need them (wf2_spawn / wf2_overflow: the ray is
// the item's, slot = ray index); n1 / n2 exist only for transparent hits (written by the containers walk); CX_SURFACE only
// and CX_UNDER only for hits whose children found no room (wf2_overflow_kernel runs on the scene's second stream, beside the
// compaction that overwrites the generation's dense queue: it cannot go back to the ray, so wf2_spawn leaves it under_point).
// 13 doubles + 4 words written per hit (120 B; 188 B in round 2).
enum { CX_OVER = 0, CX_EYE = 3, CX_NORMAL = 6, CX_COLOR = 9, CX_T = 12, CX_N1 = 13, CX_N2 = 14, CX_SURFACE = 15, CX_UNDER = 18,
       CX_DOUBLES = 21 };
enum { CU_MAT = 0, CU_PARENT = 1, CU_REMAINING = 2, CU_NODE = 3, CU_UINTS = 4 };
// CU_NODE: where the hit's lit surface colour goes -- the node wf2_spawn_kernel made for it, or one of
#define RT_WF2_NODE_NONE 0xffffffffu     // no children: the colour straight to whoever waits for this ray
#define RT_WF2_NODE_REFUSED 0xfffffffeu  // children found no room: into the context (CX_SURFACE), for wf2_overflow_kernel

// Counters many waves bump at once are split RT_WF2_PARTS ways, one 128-byte line each: atomics on ONE address are
// served one after the other (~11 ns each, measured: a second per-wave atomic doubled wf2_prepare_kernel's time), and a
// frame has half a million 64-ray chunks to hand out.
#ifndef RT_WF2_PARTS
#define RT_WF2_PARTS 32u  // a multiple of the 8 XCDs: block b's home part b % PARTS is then always served by the same L2
#endif
struct alignas(128) Wf2Padded {
    unsigned v;
    unsigned pad[31];
};
struct Wf2Counters {
    unsigned n_rays[RT_WF_LEVELS + 1];  // colour rays of generation g (dense queue length)
    unsigned n_overflow[RT_WF_LEVELS + 1];  // per generation: hits whose children found no room
    Wf2Padded n_creq[RT_WF2_PARTS];     // containers requests (transparent hits), per part of the request list
    Wf2Padded chunk[3 * (RT_WF_LEVELS + 1)][RT_WF2_PARTS];  // next chunk of each trace launch (generation x {colour, shadow, containers}) per part of its items
    unsigned offsets[RT_WF_SHARDS * RT_WF_BINS + 1];  // compaction: start of each (shard, octant) sub-slice in the dense queue
    unsigned long long prof[4][8];  // RT_WF2_PROFILE builds: shader cycles of the traversal kernel by mode and phase, summed over waves
};

struct Wf2Space {
    WfSpace ws;        // node pool, shard counters and the sliced queue wf_commit fills (both queue pointers name it)
    WfItem* qd;        // the generation's colour rays, dense
    Wf2Hit* hits;      // one per colour ray
    double* ctx_d;     // [CX_DOUBLES][ctx_cap]
    uint32_t* ctx_u;   // [CU_UINTS][ctx_cap]
    uint8_t* occ;      // [n_lights][ctx_cap]: light l is hidden from hit `slot`
    uint32_t* creq;    // [RT_WF2_PARTS][creq_part_cap]: slots whose hit is transparent, by part (see wf2_prepare_kernel)
    uint32_t* overflow;  // [ctx_cap]
    // Generation g works in buffer set g & 1 of these four (the host swaps the pointers above per generation): its feelers
    // and lighting run on a second stream while generation g + 1 is already traced and prepared into the other set.
    double* alt_ctx_d;
    uint32_t* alt_ctx_u;
    uint8_t* alt_occ;
    uint32_t* alt_overflow;
    Wf2Counters* cnt;
    unsigned qd_cap, ctx_cap, creq_part_cap;
};
// slots per part of the request list: a part takes the requests of every RT_WF2_PARTS-th group of four waves
__host__ __device__ inline unsigned wf2_creq_part_cap(size_t ctx_cap) { return (unsigned)(((ctx_cap + 255u) / 256u + RT_WF2_PARTS - 1u) / RT_WF2_PARTS * 256u); }


#ifndef RT_WF2_CHUNK
#define RT_WF2_CHUNK 64u   // rays a wave takes from the queue at a time (RT_HIP_WF2_CHUNK in the environment: studies)
#endif
#ifndef RT_WF2_REFILL
#define RT_WF2_REFILL 64u  // idle lanes that make a wave go back to the queue (RT_HIP_WF2_REFILL); measured: partial refills (8-32) cost more than the idle lanes they fill (10k scene 33.0-33.4 ms against 31.6)
#endif
enum { WF2_PRIMARY = 0, WF2_QUEUE = 1, WF2_SHADOW = 2, WF2_CONTAINERS = 3 };

// Item i of generation 0 = one pixel, tiles of 8x8 in row-major tile order (neighbouring rays to one wave).
struct PixelMap {
    unsigned tiles_x, n_items;
};
RT_DEV PixelMap pixel_map(const RtCameraDesc& cam, const RowMap& rows) {
    PixelMap m;
    m.tiles_x = (cam.hsize + 7u) / 8u;
    m.n_items = m.tiles_x * ((rows.batch_rows + 7u) / 8u) * 64u;
    return m;
}
// shading context slots of generation `gen` = its colour rays (slot i is ray i's)
RT_DEV unsigned wf2_n_slots(const Wf2Space& sp, unsigned gen, const PixelMap& pm) { return gen == 0u ? pm.n_items : sp.cnt->n_rays[gen]; }
// -> false for the slots of a tile that fall outside the frame; k = local row, y = image row
RT_DEV bool pixel_of_item(const RtCameraDesc& cam, const RowMap& rows, const PixelMap& m, unsigned i, unsigned& x, unsigned& k,
                          unsigned& y) {
    const unsigned tile = i >> 6, within = i & 63u;
    x = (tile % m.tiles_x) * 8u + (within & 7u);
    const unsigned kb = (tile / m.tiles_x) * 8u + (within >> 3);  // row of the batch
    k = rows.row0 + kb;
    y = 0;
    if (!(x < cam.hsize && kb < rows.batch_rows && k < rows.n_local_rows)) return false;
    y = ((k / rows.band_rows) * rows.n_parts + rows.part) * rows.band_rows + k % rows.band_rows;
    return y < cam.vsize;
}

// The ray of item `i` of a launch: camera ray, queued colour ray, or the shadow feeler of (hit slot, light):
// feelers are numbered light-major, so a wave's feelers go to ONE light from neighbouring hits.
template <int MODE>
RT_DEV bool wf2_item_ray(const SceneDev& S, const Wf2Space& sp, const RtCameraDesc& cam, const RowMap& rows, const PixelMap& pm,
                         unsigned i, unsigned n_hits, Ray& ray, double& limit) {
    limit = 0.0;
    if constexpr (MODE == WF2_PRIMARY) {
        unsigned x, k, y;
        if (!pixel_of_item(cam, rows, pm, i, x, k, y)) return false;
        ray = ray_for_pixel(cam, x, y);
        return true;
    } else if constexpr (MODE == WF2_QUEUE) {
        const WfItem* it = sp.qd + i;
        if (it->remaining == RT_WF_EMPTY) return false;
        ray = it->ray;
        return true;
    } else {
        const unsigned l = i / n_hits, slot = i - l * n_hits;
        const double* cd = sp.ctx_d + slot;
        const size_t st = sp.ctx_cap;
        const V3 over = v3(cd[(CX_OVER + 0) * st], cd[(CX_OVER + 1) * st], cd[(CX_OVER + 2) * st]);
        if (over.x != over.x) return false;  // no hit in this slot (wf2_prepare_kernel's mark)
        const RtLight light = S.lights[l];
        const V3 v = v3(light.position[0], light.position[1], light.position[2]) - over;  // world.rs:86-89
        limit = magnitude(v);
        ray = Ray{over, normalize(v)};
        return true;
    }
}

// ---- shading contexts -----------------------------------------------------------------------------------------
RT_DEV V3 ctx_v3(const Wf2Space& sp, unsigned slot, int field) {
    const double* p = sp.ctx_d + (size_t)field * sp.ctx_cap + slot;
    return v3(p[0], p[sp.ctx_cap], p[2 * (size_t)sp.ctx_cap]);
}
RT_DEV void ctx_put(const Wf2Space& sp, unsigned slot, int field, V3 v) {
    double* p = sp.ctx_d + (size_t)field * sp.ctx_cap + slot;
    p[0] = v.x, p[sp.ctx_cap] = v.y, p[2 * (size_t)sp.ctx_cap] = v.z;
}
RT_DEV double& ctx_f(const Wf2Space& sp, unsigned slot, int field) { return sp.ctx_d[(size_t)field * sp.ctx_cap + slot]; }
RT_DEV uint32_t& ctx_u(const Wf2Space& sp, unsigned slot, int field) { return sp.ctx_u[(size_t)field * sp.ctx_cap + slot]; }

template <bool STATS, int MODE>
__global__ __launch_bounds__(256) void wf2_cands_kernel(const SceneDev S, Wf2Space sp, RtCameraDesc cam, RowMap rows, unsigned gen,
                                                        RtStats* __restrict__ stats, unsigned* __restrict__ status) {
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    const PixelMap pm = pixel_map(cam, rows);
    const unsigned n_hits = MODE == WF2_SHADOW ? wf2_n_slots(sp, gen, pm) : 0u;
    const unsigned n = MODE == WF2_PRIMARY ? pm.n_items : (MODE == WF2_QUEUE ? sp.cnt->n_rays[gen] : n_hits * S.n_lights);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Ray ray;
        double limit;
        Best best{__builtin_inf(), -1};
        bool found = false;
        if (wf2_item_ray<MODE>(S, sp, cam, rows, pm, i, n_hits, ray, limit))
            found = wf2_candidates<MODE == WF2_SHADOW>(S, ray, limit, best, cx);
        if constexpr (MODE == WF2_SHADOW) {
            sp.occ[i] = found ? 1 : 0;  // feelers are numbered light * n_hits + slot, and so are their flags
        } else {
            Wf2Hit h;
            h.t = best.t, h.idx = best.idx, h.pad = 0;
            sp.hits[i] = h;
        }
    }
    flush_stats(cx, stats, status);
}

// what a wave's lanes have written to LDS is there for the others to read
RT_DEV void wf2_lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The traversal kernel.  One-wave blocks; every lane holds at most one ray.
#ifndef RT_WF2_CANDS_INLINE
#define RT_WF2_CANDS_INLINE 1  // the traversal kernel evaluates the direction candidates when it takes a ray (a whole wave refills at once); 0: wf2_cands_kernel ahead of every trace launch, 21.8 ms against 21.4
#endif
#ifndef RT_WF2_MIN_CHUNK
#define RT_WF2_MIN_CHUNK 8u
#endif
#ifndef RT_WF2_PEND
#define RT_WF2_PEND 4
#endif
#ifndef RT_WF2_PEND_COLOUR
#define RT_WF2_PEND_COLOUR 0  // the same for colour rays (closest hit): every leaf can shorten the rest of their walk
#endif
#ifndef RT_WF2_LIST_SPLIT
#define RT_WF2_LIST_SPLIT 8u  // containers walks of small launches: lanes a line is shared out among at most (0: never)
#endif
#ifndef RT_WF2_LIST_SHARE
#define RT_WF2_LIST_SHARE 1u
#endif
#ifndef RT_WF2_TRACE_WAVES
#define RT_WF2_TRACE_WAVES 4  // waves per SIMD the traversal kernel is compiled for (<= 128 VGPRs)
#endif
#ifndef RT_WF2_SHARED_LEAVES
// Feelers / containers walks: 1 = the wave's set-aside leaves dealt out to ALL its lanes (below: "leaf exchange"), 0 = every lane
// tests its own, one a round.  Measured (profiles/r03_variants.txt): the exchange takes the lanes busy in a leaf round from
// 0.56 to 0.72 (feelers) and from 0.27 to 0.63 (containers walks) -- and the leaf phases take as long as before, the frame
// 17.64 ms against 17.23: a leaf round's cost is the vector memory transactions of its ACTIVE lanes (each leaf is ~25
// scattered 16-byte reads), not the instructions a round issues.  Kept as a study knob.
#define RT_WF2_SHARED_LEAVES 0
#endif
// Stack entries per lane in LDS by kind of walk (deeper ones in private memory): what is left of a wave's 10 KB share of the
// CU's 160 KB (4 waves / SIMD) once the feelers' and containers walks' leaf exchange -- rays, work list, results -- has its room.
#define RT_WF2_LDS_STACK_SHADOW 16
#define RT_WF2_LDS_STACK_LIST 6
template <bool STATS, int MODE>
__global__ __launch_bounds__(64, RT_WF2_TRACE_WAVES) void wf2_trace_kernel(const SceneDev S, const Wf2Accel A, Wf2Space sp, RtCameraDesc cam, RowMap rows,
                                                       unsigned gen, RtStats* __restrict__ stats, unsigned* __restrict__ status) {
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    constexpr bool SHADOW = MODE == WF2_SHADOW;
    constexpr bool LIST = MODE == WF2_CONTAINERS;  // n1 / n2 of the transparent hits: the sorted world list up to the hit
    constexpr int PEND = (SHADOW || LIST) ? RT_WF2_PEND : RT_WF2_PEND_COLOUR;  // leaves a lane may set aside while other lanes look for theirs
    constexpr bool EXCHANGE = PEND > 0 && RT_WF2_SHARED_LEAVES != 0;           // ... and those leaves are dealt out to all lanes of the wave
    constexpr int LDS_N = !EXCHANGE ? RT_WF2_LDS_STACK : (SHADOW ? RT_WF2_LDS_STACK_SHADOW : RT_WF2_LDS_STACK_LIST);
    constexpr int RING = PEND > 0 ? PEND : 1;
    __shared__ uint32_t lds_stack_all[LDS_N * 64];
    __shared__ uint32_t pend_all[RING * 64];
    // The leaf exchange (EXCHANGE): every ray of the wave's chunk in LDS -- a leaf is tested by whichever lane it is dealt to --,
    // the work list (leaf code | owner lane << 24, so shapes < 2^24: wf2_create), and what comes back: a flag per ray
    // (feelers) or the primitive's hits (containers walks).
    __shared__ double x_ray[EXCHANGE ? 6 * 64 : 1];
    __shared__ double x_lim[EXCHANGE ? 64 : 1];
    __shared__ uint32_t x_work[EXCHANGE ? RING * 64 : 1];
    __shared__ uint32_t x_hidden[EXCHANGE && SHADOW ? 64 : 1];
    __shared__ double x_t[EXCHANGE && LIST ? 4 * 64 : 1];
    __shared__ uint32_t x_sk[EXCHANGE && LIST ? 2 * 64 : 1];  // shape | keep mask
    uint32_t* lds_stack = lds_stack_all + threadIdx.x;
    uint32_t* pend = pend_all + threadIdx.x;
    unsigned p_head = 0, p_n = 0;
    uint32_t deep[RT_WF2_MAX_STACK - LDS_N];
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    const PixelMap pm = pixel_map(cam, rows);
    const unsigned n_hits = SHADOW ? wf2_n_slots(sp, gen, pm) : 0u;
    // The launch's items in RT_WF2_PARTS parts, each with a counter of its own: equal stretches of the item range, or (LIST)
    // the parts of the request list.  A wave starts at its block's home part and moves on when a part is used up.
    const unsigned n = MODE == WF2_PRIMARY ? pm.n_items : (MODE == WF2_QUEUE ? sp.cnt->n_rays[gen] : (LIST ? 0u : n_hits * S.n_lights));
    Wf2Padded* chunk_counters = sp.cnt->chunk[3u * gen + (SHADOW ? 1u : (LIST ? 2u : 0u))];
    // Rays a wave takes at a time: A.chunk (64) when the launch has that many for every resident wave, fewer when it has
    // not -- a wave runs as long as its slowest ray, so a small launch (a late generation, a narrow band of a multi-GPU
    // frame) is through sooner as many waves of 8-32 rays than as a few full ones with the rest of the device idle.
    unsigned chunk = A.chunk;
    {
        unsigned total = n;
        if constexpr (LIST) {
            total = 0;
            for (unsigned p = 0; p < RT_WF2_PARTS; ++p) total += sp.cnt->n_creq[p].v;
        }
        // (LIST: A.list_share chunks per resident wave at least -- lines differ a lot in what they meet, and a wave lasts as
        // long as its longest)
        // With fewer rays than that: ONE chunk per wave, as even as it comes (any size, not a power of two) -- a wave that
        // takes two chunks of 16 one after the other lasts twice as long as one that takes its 25 at once.
        const unsigned long long slots = (unsigned long long)gridDim.x * (LIST ? A.list_share : 1u);
        if (slots * chunk > total) {
            const unsigned even = (unsigned)((total + slots - 1u) / slots);
            chunk = even < RT_WF2_MIN_CHUNK ? RT_WF2_MIN_CHUNK : (even < chunk ? even : chunk);
        }
    }
    // LIST, small launches: a line is shared out among `split` neighbouring lanes by ranges of the ray parameter.  A launch
    // lasts as long as its slowest ray, a containers walk goes through everything its whole line meets, and with fewer rays
    // than lanes the other lanes have nothing to do: lane k of a ray's `split` walks the k-th stretch of the line inside the
    // scene's extent (the first also everything before it, the last everything after) and keeps the entries whose t falls
    // into its own half-open range -- every entry ends up in exactly one lane, each lane's list is sorted, and the lists
    // one after the other ARE the sorted list (equal t's share a range).  The containers loop then runs through them in
    // order, its state handed from lane to lane.
    unsigned split = 1u;
#if RT_WF2_LIST_SPLIT
    if constexpr (LIST) {
        if (A.refill == 64u && !(A.seg_near > -__builtin_inff()))
            while (split < RT_WF2_LIST_SPLIT && chunk * split * 2u <= 64u) split *= 2u;
    }
#endif
    double keep_lo = -__builtin_inf(), keep_hi = __builtin_inf();
    const unsigned per_part = LIST ? sp.creq_part_cap : ((n + chunk - 1u) / chunk + RT_WF2_PARTS - 1u) / RT_WF2_PARTS * chunk;
    unsigned part = blockIdx.x % RT_WF2_PARTS, parts_left = RT_WF2_PARTS;
    unsigned part_lo = part * per_part;
    unsigned part_hi = LIST ? part_lo + sp.cnt->n_creq[part].v : (part_lo + per_part < n ? part_lo + per_part : n);
    typename std::conditional<LIST, LaneList, int>::type W;
    int top = 0;       // LIST: entries in W
    unsigned slot = 0;  // LIST: the hit's shading context
    const unsigned lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned chunk_next = 0, chunk_end = 0;  // wave-uniform
    bool dry = false;                        // the queue has no chunk left
    bool busy = false;
    unsigned item = 0;
    Ray ray{v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};  // (EXCHANGE: only while a lane takes its ray; the leaf tests read x_ray)
    RayF32 f{};
    Best best{__builtin_inf(), -1};
    double limit = 0.0;
    bool found = false;
    float t_far = 0.f, t_near = 0.f;
    bool second_walk = false;  // LIST: the walk over the open primitives is still to come
    Walk w{RT_BVH4_DONE, 0};
    auto push = [&](uint32_t code) {
        if (w.sp < LDS_N)
            lds_stack[w.sp * 64] = code;
        else
            deep[w.sp - LDS_N] = code;
        ++w.sp;
    };
    // the ray and the limit (feelers: distance to the light; containers walks: where the list is cut) of the wave's lane `l`
    auto ray_of = [&](unsigned l) {
        if constexpr (EXCHANGE)
            return Ray{v3(x_ray[0 * 64 + l], x_ray[1 * 64 + l], x_ray[2 * 64 + l]), v3(x_ray[3 * 64 + l], x_ray[4 * 64 + l], x_ray[5 * 64 + l])};
        else
            return ray;
    };
    auto limit_of = [&](unsigned l) {
        if constexpr (EXCHANGE)
            return x_lim[l];
        else
            return limit;
    };
    // a leaf ahead of its f64 test: the f32 test of the ray's line against the inflated primitive (LeafPre), and the enclosing
    // groups' indices on the way
    auto leaf_worth = [&](uint32_t leaf, const Ray& r, uint32_t& anc0, uint32_t& anc1) {
        anc0 = RT_PRE_DEEP, anc1 = RT_PRE_NO_ANC;
        if (!A.pre) return true;
        const int s = (int)RT_BVH4_SHAPE(leaf);
        if (leaf & RT_BVH4_NOPRE) {  // no pre-test for this kind: the record is read for the ancestors only
            const uint2 anc = *reinterpret_cast<const uint2*>(&A.pre[s].anc[0]);
            anc0 = anc.x, anc1 = anc.y;
            return true;
        }
        const float reach = fmaxf(fmaxf(fabsf((float)r.o.x), fabsf((float)r.o.y)), fabsf((float)r.o.z)) * 1.000001f + A.extent_max;
        count(cx, ST_ACCEL_F32);  // one more acceleration-only test
        return leaf_may_hit(A.pre + s, r, reach, anc0, anc1);
    };
#ifdef RT_WF2_PROFILE
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pn[2] = {0, 0};
#define RT_STAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pt[k] += now_ - stamp_; stamp_ = now_; }
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#else
#define RT_STAMP(k)
#endif
    for (;;) {
        // ---- refill: idle lanes take the next rays of the wave's chunk
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const int n_idle = __builtin_popcountll(idle);
        if (n_idle == 64 || (n_idle >= (int)A.refill && !dry)) {
            while (chunk_next >= chunk_end && !dry) {
                unsigned c = 0xffffffffu;
                if (part_lo < part_hi) {
                    if (lane == 0) c = atomicAdd(&chunk_counters[part].v, chunk);
                    c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                }
                if (c < part_hi - part_lo && part_lo < part_hi) {
                    chunk_next = part_lo + c;
                    chunk_end = chunk_next + chunk < part_hi ? chunk_next + chunk : part_hi;
                } else if (--parts_left == 0u) {
                    dry = true;
                } else {  // this part is used up: on to the next
                    part = (part + 1u) % RT_WF2_PARTS;
                    part_lo = part * per_part;
                    part_hi = LIST ? part_lo + sp.cnt->n_creq[part].v : (part_lo + per_part < n ? part_lo + per_part : n);
                    if (!LIST && part_lo > n) part_lo = part_hi = n;
                }
            }
            const unsigned avail = chunk_end - chunk_next;
            if (avail != 0u) {
                // (split > 1: the whole wave refills at once, `split` neighbouring lanes take one ray)
                const unsigned rank = (unsigned)__builtin_popcountll(idle & below) / (LIST ? split : 1u);
                if (!busy && rank < avail) {
                    item = chunk_next + rank;
                    busy = true;
                    best.t = __builtin_inf(), best.idx = -1;
                    found = false;
                    p_head = p_n = 0u;
                    bool valid;
                    if constexpr (LIST) {
                        slot = sp.creq[item];
                        const unsigned of = slot;  // (a hit's shading context is its ray's index)
                        valid = gen == 0u ? wf2_item_ray<WF2_PRIMARY>(S, sp, cam, rows, pm, of, 0u, ray, limit)
                                          : wf2_item_ray<WF2_QUEUE>(S, sp, cam, rows, pm, of, 0u, ray, limit);
                        const unsigned sub = lane & (split - 1u);
                        if (sub == 0u) count(cx, ST_CONTAINER_PASSES);
                        limit = containers_bound(ctx_f(sp, slot, CX_T));  // the list is cut here
                        top = 0;
                        keep_lo = -__builtin_inf(), keep_hi = __builtin_inf();
                        if (split > 1u) {
                            // the stretch [a, b] of the line inside the scene's extent, cut at the list's bound, in `split` parts
                            double a = -__builtin_inf(), b = limit;
                            const double o3[3] = {ray.o.x, ray.o.y, ray.o.z}, d3[3] = {ray.d.x, ray.d.y, ray.d.z};
#pragma unroll
                            for (int ax = 0; ax < 3; ++ax) {
                                const double e = (double)A.extent[ax] * 1.0001 + 1e-9;
                                if (d3[ax] != 0.0) {
                                    const double t1 = (-e - o3[ax]) / d3[ax], t2 = (e - o3[ax]) / d3[ax];
                                    a = fmax(a, fmin(t1, t2)), b = fmin(b, fmax(t1, t2));
                                }
                            }
                            if (!(a < b) || !(b - a < 1e300)) a = b = 0.0;  // (off the scene, or nothing to go by: the first and the last lane share the line)
                            const double step = (b - a) / (double)split;
                            if (sub != 0u) keep_lo = a + step * (double)sub;
                            if (sub != split - 1u) keep_hi = a + step * (double)(sub + 1u);
                        }
                        if (A.has_cands) {  // the direction index's candidates first, as world_list has them
                            CandIter it;
                            cand_begin(S, ray, it);
                            const V3 d = ray.d;
                            while (it.k < it.e) {
                                const RtDirCand* __restrict__ c = S.dir_cands + (it.scan_all ? it.k : S.dir_entries[it.k]);
                                ++it.k;
                                const double a = c->q[0] * d.x * d.x + c->q[1] * d.y * d.y + c->q[2] * d.z * d.z +
                                                 2.0 * (c->q[3] * d.x * d.y + c->q[4] * d.x * d.z + c->q[5] * d.y * d.z);
                                if (fabs(a) <= c->thr) top = wf2_leaf_list(S, W, top, (int)c->visit, ray, limit, cx, RT_PRE_DEEP, RT_PRE_NO_ANC, keep_lo, keep_hi);
                            }
                        }
                    } else {
                        valid = wf2_item_ray<MODE>(S, sp, cam, rows, pm, item, n_hits, ray, limit);
#if RT_WF2_CANDS_INLINE
                        // the direction index's candidates first (a whole wave refills at once: the lanes are together here)
                        if (A.has_cands && valid) found = wf2_candidates<SHADOW>(S, ray, limit, best, cx);
#else
                        if (A.has_cands) {  // what the direction index already found for this ray (wf2_cands_kernel)
                            if constexpr (SHADOW) {
                                found = sp.occ[item] != 0;
                            } else {
                                const Wf2Hit h0 = sp.hits[item];
                                best.t = h0.t, best.idx = h0.idx;
                            }
                        }
#endif
                    }
                    ray_f32(ray, A.extent, A.slack, f);
                    t_far = (SHADOW || LIST) ? f32_up(limit) : f32_up(best.t);
                    if constexpr (EXCHANGE) {
                        x_ray[0 * 64 + lane] = ray.o.x, x_ray[1 * 64 + lane] = ray.o.y, x_ray[2 * 64 + lane] = ray.o.z;
                        x_ray[3 * 64 + lane] = ray.d.x, x_ray[4 * 64 + lane] = ray.d.y, x_ray[5 * 64 + lane] = ray.d.z;
                        x_lim[lane] = limit;
                        if constexpr (SHADOW) x_hidden[lane] = 0u;
                    }
                    // LIST: the list is what the whole LINE meets up to the cut, behind the origin too (seg_near = -inf).
                    // (Study, RT_HIP_WF2_SEG_NEAR=1: a finite seg_near and a second walk over the open primitives only; see
                    // the header of this file for why that is not the default.)
                    t_near = LIST ? A.seg_near : 0.f;
                    second_walk = LIST && A.seg_near > -__builtin_inff();
                    if constexpr (LIST) {  // (a lane's share of the line, rounded outwards)
                        if (keep_lo > -__builtin_inf()) t_near = f32_down(keep_lo);
                        if (keep_hi < __builtin_inf()) t_far = fminf(t_far, f32_up(keep_hi));
                    }
                    // the stack: the sentinel, the hierarchy's root, the unbounded leaves -- those are tested first (a plane's
                    // hit bounds the walk), through the same leaf code as every other leaf
                    lds_stack[0] = RT_BVH4_DONE;
                    w.sp = 1;
                    w.cur = RT_BVH4_DONE;
                    if (valid && !found) {
                        w.cur = A.root;
                        for (unsigned k = 0; k < A.n_always; ++k) {
                            push(w.cur);
                            w.cur = RT_BVH4_LEAF | A.always[k];
                        }
                    }
                    // (reflected / refracted rays are counted where they are queued, as in wf_level_kernel)
                    if (valid && (MODE == WF2_PRIMARY || SHADOW)) count(cx, SHADOW ? ST_RAY_SHADOW : ST_RAY_PRIMARY);
                }
                {
                    const unsigned takers = (unsigned)n_idle / (LIST ? split : 1u);
                    chunk_next += takers < avail ? takers : avail;
                }
            } else if (n_idle == 64) {
                break;  // nothing in flight, nothing left
            }
        }
        RT_STAMP(0)
        // ---- box tests only, until every busy lane holds a leaf or is through.  (Measured and rejected: leaving the loop
        // for the leaves at hand once only 4-48 lanes are still walking -- 30.7 ms at 0 and 4, 31.3-36.4 ms at 8-48.)
        // Feelers and the containers walk (PEND > 0): a lane that has its leaf does not idle while others still look for
        // theirs -- it sets the leaf aside (a ring of PEND per lane in LDS) and walks on.  Nothing a leaf yields could have
        // shortened those walks much: a feeler that is lit goes through the whole segment anyway, the containers list is
        // cut at a fixed bound.
        if constexpr (PEND > 0) {
            for (;;) {
                const bool inner = busy && w.cur < RT_BVH4_DONE;
                if (__builtin_amdgcn_ballot_w64(inner && p_n == 0u) == 0ull) break;
#ifdef RT_WF2_PROFILE
                pn[0] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(inner || (busy && (w.cur & RT_BVH4_LEAF) && p_n < (unsigned)PEND)));
                pn[1] += 64u;
#endif
                if (inner) {
                    walk_step<LDS_N>(A, f, t_near, t_far, w, lds_stack, deep, cx);
                } else if (busy && (w.cur & RT_BVH4_LEAF) && p_n < (unsigned)PEND) {
                    pend[((p_head + p_n) % (unsigned)RING) * 64u] = w.cur;
                    ++p_n;
                    walk_pop<LDS_N>(w, lds_stack, deep);
                }
            }
        } else {
#ifdef RT_WF2_PROFILE
            while (__builtin_amdgcn_ballot_w64(busy && w.cur < RT_BVH4_DONE) != 0ull) {
                pn[0] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(busy && w.cur < RT_BVH4_DONE));
                pn[1] += 64u;
                if (busy && w.cur < RT_BVH4_DONE) walk_step<LDS_N>(A, f, t_near, t_far, w, lds_stack, deep, cx);
            }
#else
            while (busy && w.cur < RT_BVH4_DONE) walk_step<LDS_N>(A, f, t_near, t_far, w, lds_stack, deep, cx);
#endif
        }
        RT_STAMP(1)
        // ---- leaves
        bool own_leaves = true;  // every lane tests its own leaf, one a round (colour rays; LIST with a line shared among lanes)
        if constexpr (EXCHANGE) {
            if (split == 1u) {
                // The wave's set-aside leaves dealt out to ALL its lanes.  Lines and feelers differ a lot in how many leaves
                // they meet: worked off one per lane and round, a round ran with a quarter (containers walks) to a half
                // (feelers) of the lanes, the others waiting for the lane with the most.  Here the leaves of all lanes go on
                // one list and lane i tests item base + i, whose ever it is -- with that ray, read from LDS -- and hands the
                // answer back through LDS: a flag (feelers: any occluder will do, world.rs:85-96) or the primitive's hits, which
                // the owner then puts on its sorted list (the list's merge rule does not depend on the order of arrival:
                // list_insert).  Nothing a test yields depends on which lane ran it.
                own_leaves = false;
                if (busy && (w.cur & RT_BVH4_LEAF) && p_n < (unsigned)PEND) {  // (the leaf at hand joins the others when there is room)
                    pend[((p_head + p_n) % (unsigned)RING) * 64u] = w.cur;
                    ++p_n;
                    walk_pop<LDS_N>(w, lds_stack, deep);
                }
                const unsigned cnt = busy ? p_n : 0u;
                unsigned off[RING];  // where this lane's k-th leaf goes: all lanes' first leaves, then their second ones, ...
                unsigned total = 0u;
#pragma unroll
                for (unsigned k = 0; k < (unsigned)RING; ++k) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(cnt > k);
                    off[k] = total + (unsigned)__builtin_popcountll(m & below);
                    total += (unsigned)__builtin_popcountll(m);
                }
                if (total != 0u) {
#pragma unroll
                    for (unsigned k = 0; k < (unsigned)RING; ++k)
                        if (k < cnt) x_work[off[k]] = (pend[((p_head + k) % (unsigned)RING) * 64u] & 0xc0ffffffu) | (lane << 24);
                    p_head = p_n = 0u;
                    wf2_lds_wave_sync();
                    for (unsigned base = 0u; base < total; base += 64u) {
                        const unsigned idx = base + lane;
                        bool take = idx < total;
                        const uint32_t it = take ? x_work[idx] : 0u;
                        const unsigned owner = (it >> 24) & 63u;
                        const uint32_t leaf = it & 0xc0ffffffu;
                        if constexpr (SHADOW) take = take && x_hidden[owner] == 0u;  // (that ray already has its occluder)
#ifdef RT_WF2_PROFILE
                        pt[4] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(take));
                        pt[5] += 64u;
#endif
                        const int s = (int)RT_BVH4_SHAPE(leaf);
                        bool hit = false;
                        unsigned keep = 0u;
                        PrimHits h;
                        h.t[0] = h.t[1] = h.t[2] = h.t[3] = 0.0;
                        if (take) {
                            const Ray r = ray_of(owner);
                            const double lim = limit_of(owner);
                            uint32_t anc0, anc1;
                            if (leaf_worth(leaf, r, anc0, anc1)) {
                                if constexpr (LIST) {
                                    keep = wf2_leaf_hits(S, s, r, lim, h, cx, anc0, anc1);
                                } else {
                                    Best none{__builtin_inf(), -1};
                                    hit = wf2_leaf<true>(S, s, r, lim, none, cx, anc0, anc1);
                                }
                            }
                        }
                        if constexpr (SHADOW) {
                            if (hit) x_hidden[owner] = 1u;
                        } else {
                            x_sk[lane] = (uint32_t)s, x_sk[64 + lane] = keep;
                            x_t[0 * 64 + lane] = h.t[0], x_t[1 * 64 + lane] = h.t[1], x_t[2 * 64 + lane] = h.t[2], x_t[3 * 64 + lane] = h.t[3];
                        }
                        wf2_lds_wave_sync();
                        if constexpr (LIST) {  // the owners take what came back for them in this batch
#pragma unroll
                            for (unsigned k = 0; k < (unsigned)RING; ++k) {
                                const unsigned j = off[k] - base;  // (wraps for an item of an earlier batch)
                                if (k < cnt && j < 64u) {
                                    const unsigned kept = x_sk[64 + j];
                                    const uint32_t shape = x_sk[j];
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        if ((kept >> q) & 1u) {
                                            top = list_insert(W, 0, top, x_t[q * 64 + j], shape, cx.status);
                                            count(cx, ST_LIST_ENTRIES);
                                        }
                                }
                            }
                        }
                        wf2_lds_wave_sync();  // (the next batch writes the same words)
                    }
                    if constexpr (SHADOW) {
                        if (busy && x_hidden[lane] != 0u) {
                            found = true;
                            w.cur = RT_BVH4_DONE;
                        }
                    }
                }
            }
        }
        if (own_leaves) {
            // the oldest leaf set aside, else the one at hand
            const bool from_ring = PEND > 0 && busy && p_n != 0u;
            const bool take = from_ring || (busy && (w.cur & RT_BVH4_LEAF));
            uint32_t leaf = w.cur;
            if constexpr (PEND > 0) {
                if (from_ring) {
                    leaf = pend[p_head * 64u];
                    p_head = (p_head + 1u) % (unsigned)RING;
                    --p_n;
                }
            }
#ifdef RT_WF2_PROFILE
            pt[4] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(take));
            pt[5] += 64u;
#endif
            if (take) {
                const int s = (int)RT_BVH4_SHAPE(leaf);
                const Ray r = ray_of(lane);
                const double lim = limit_of(lane);
                bool hit_shadow = false;
                uint32_t anc0, anc1;
                if (leaf_worth(leaf, r, anc0, anc1)) {
                    if constexpr (LIST)
                        top = wf2_leaf_list(S, W, top, s, r, lim, cx, anc0, anc1, keep_lo, keep_hi);
                    else
                        hit_shadow = wf2_leaf<SHADOW>(S, s, r, lim, best, cx, anc0, anc1);
                }
                if (SHADOW && hit_shadow) {
                    found = true;
                    w.cur = RT_BVH4_DONE;
                    p_n = 0u;
                } else if (!from_ring) {
                    if (!LIST && !SHADOW) t_far = f32_up(best.t);
                    walk_pop<LDS_N>(w, lds_stack, deep);
                }
            }
        }
        RT_STAMP(2)
        // ---- rays that are through
        if (LIST && busy && w.cur == RT_BVH4_DONE && p_n == 0u && second_walk) {
            second_walk = false;
            t_near = -__builtin_inff();
            w.sp = 1;
            w.cur = A.root_open;
        }
        if constexpr (LIST) {
            // the containers loop over a finished list: once per wave and chunk, when its last walk is through -- lane by
            // lane as the walks end it ran with one or two lanes at a time and was a quarter of this kernel
            const bool through = busy && w.cur == RT_BVH4_DONE && p_n == 0u;
            const bool all_through = __builtin_amdgcn_ballot_w64(busy && !through) == 0ull;
            if (all_through && split == 1u && through) {
                double n1, n2;
                if (!containers_in_registers(S, W, top, ctx_f(sp, slot, CX_T), n1, n2)) containers_from_list(S, W, top, ctx_f(sp, slot, CX_T), n1, n2);
                ctx_f(sp, slot, CX_N1) = n1;
                ctx_f(sp, slot, CX_N2) = n2;
                busy = false;
            } else if (all_through && split > 1u && __builtin_amdgcn_ballot_w64(through) != 0ull) {
                // a line's lists in the order of their ranges, the loop's state from lane to lane (every lane of the wave
                // takes part in the exchanges; those without a ray hold nothing)
                const unsigned sub = lane & (split - 1u);
                const double hit_t = through ? ctx_f(sp, slot, CX_T) : 0.0;
                ContainersState st;
                containers_begin(st);
                for (unsigned k = 0; k < split; ++k) {
                    if (k != 0u) {
                        ContainersState in;
                        in.c0 = (uint32_t)__shfl_up((int)st.c0, 1, 64), in.c1 = (uint32_t)__shfl_up((int)st.c1, 1, 64);
                        in.c2 = (uint32_t)__shfl_up((int)st.c2, 1, 64), in.c3 = (uint32_t)__shfl_up((int)st.c3, 1, 64);
                        in.nc = __shfl_up(st.nc, 1, 64), in.phase = __shfl_up(st.phase, 1, 64);
                        in.n1 = __shfl_up(st.n1, 1, 64), in.n2 = __shfl_up(st.n2, 1, 64);
                        if (sub == k) st = in;
                    }
                    if (through && sub == k) containers_run(S, W, top, hit_t, st);
                }
                // a line with more than four objects open at once: its first lane gathers the lists (ascending ranges: every
                // entry is appended) and answers from the whole list
                const int last_phase = __shfl(st.phase, (int)(lane | (split - 1u)), 64);
                const bool gather = through && last_phase == 2;
                if (__builtin_amdgcn_ballot_w64(gather) != 0ull) {
                    const int first = (int)(lane & ~(split - 1u));
                    int all = top;
                    for (unsigned k = 1; k < split; ++k) {
                        const int n_k = __shfl(top, first + (int)k, 64);
                        for (int j = 0; __builtin_amdgcn_ballot_w64(gather && j < n_k) != 0ull; ++j) {
                            const bool mine = through && sub == k && j < top;
                            const double t_j = mine ? W.LT(j) : 0.0;
                            const uint32_t id_j = mine ? W.LID(j) : 0u;
                            const double t_in = __shfl(t_j, first + (int)k, 64);
                            const uint32_t id_in = (uint32_t)__shfl((int)id_j, first + (int)k, 64);
                            if (gather && sub == 0u && j < n_k) all = list_insert(W, 0, all, t_in, id_in, cx.status);
                        }
                    }
                    if (gather && sub == 0u) {
                        double n1, n2;
                        containers_from_list(S, W, all, hit_t, n1, n2);
                        ctx_f(sp, slot, CX_N1) = n1;
                        ctx_f(sp, slot, CX_N2) = n2;
                    }
                }
                if (through && sub == split - 1u && st.phase != 2) {
                    ctx_f(sp, slot, CX_N1) = st.n1;
                    ctx_f(sp, slot, CX_N2) = st.n2;
                }
                busy = false;
            }
        } else if (busy && w.cur == RT_BVH4_DONE && p_n == 0u) {
            if constexpr (SHADOW) {
                sp.occ[item] = found ? 1 : 0;
            } else {
                Wf2Hit h;
                h.t = best.t, h.idx = best.idx, h.pad = 0;
                sp.hits[item] = h;
            }
            busy = false;
        }
        RT_STAMP(3)
    }
#ifdef RT_WF2_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&sp.cnt->prof[MODE][k], pt[k]);
        atomicAdd(&sp.cnt->prof[MODE][4], pn[0]);
        atomicAdd(&sp.cnt->prof[MODE][5], pn[1]);
        atomicAdd(&sp.cnt->prof[MODE][6], pt[4]);
        atomicAdd(&sp.cnt->prof[MODE][7], pt[5]);
    }
#endif
    flush_stats(cx, stats, status);
}

// The traversal kernel for COHERENT launches: camera rays (a wave = an 8 x 8 tile of pixels) and the feelers of the first
// generation (numbered light-major from neighbouring hits: a wave's feelers go to one light from one patch of the scene).
// Such rays meet the same nodes and mostly the same leaves, and wf2_trace_kernel -- every lane its own walk -- spends its
// time on vector memory transactions: 4 scattered 16-byte reads per lane and node, ~25 per lane and leaf (TA busy 0.66-0.72,
// VALU busy 0.2; the leaf exchange above showed that it is the reads of the ACTIVE lanes that cost, not the instructions).
// Here the WAVE walks: one stack for all 64 rays (LDS: node code + the lanes that met its box), one node at a time read
// through scalar loads (64 bytes, once per wave, into scalar registers), its four child boxes tested by every lane that is
// still in it with its own ray, ballots say which children any lane met, and they go on the stack nearest first (by the
// first such lane's distances).  A leaf is tested by the lanes that met its box, together, its records again through
// scalar loads (wf2_leaf_uniform).  Every lane thus tests exactly the leaves whose chain of boxes its own ray meets within
// its own current reach -- what the per-lane walk guarantees --, so the closest hit / the occlusion answer is the same
// (neither depends on the order of the tests).
#ifndef RT_WF2_PACKET
#define RT_WF2_PACKET 1  // 0: wf2_trace_kernel for every launch (studies)
#endif
#ifndef RT_WF2_PACKET_WAVES
#define RT_WF2_PACKET_WAVES RT_WF2_TRACE_WAVES  // waves per SIMD the wave walk is compiled for.  Measured 4 / 5 / 6 / 8 (128 / 96 / 80 / 64 VGPRs): 16.17 / 16.16 / 16.31 / 16.74 ms
#endif
template <bool STATS, int MODE>
__global__ __launch_bounds__(64, RT_WF2_PACKET_WAVES) void wf2_packet_kernel(const SceneDev S, const Wf2Accel A, Wf2Space sp, RtCameraDesc cam, RowMap rows,
                                                        unsigned gen, RtStats* __restrict__ stats, unsigned* __restrict__ status) {
    static_assert(MODE == WF2_PRIMARY || MODE == WF2_SHADOW, "closest hits of camera rays, or feelers");
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    constexpr bool SHADOW = MODE == WF2_SHADOW;
    // the wave's stack: a walk pushes at most three entries per level it descends (wf2_create bounds the per-lane walk's
    // need by RT_WF2_MAX_STACK in the same way), the unbounded leaves and the root
    __shared__ uint32_t pk_code[RT_WF2_MAX_STACK + 12];
    __shared__ unsigned long long pk_mask[RT_WF2_MAX_STACK + 12];
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    const PixelMap pm = pixel_map(cam, rows);
    const unsigned n_hits = SHADOW ? wf2_n_slots(sp, gen, pm) : 0u;
    const unsigned n = SHADOW ? n_hits * S.n_lights : pm.n_items;
    Wf2Padded* chunk_counters = sp.cnt->chunk[3u * gen + (SHADOW ? 1u : 0u)];
    unsigned chunk = A.chunk > 64u ? 64u : A.chunk;  // (one ray per lane)
    {
        const unsigned long long slots = gridDim.x;  // fewer rays than 64 for every resident wave: one even chunk each (see wf2_trace_kernel)
        if (slots * chunk > n) {
            const unsigned even = (unsigned)((n + slots - 1u) / slots);
            chunk = even < RT_WF2_MIN_CHUNK ? RT_WF2_MIN_CHUNK : (even < chunk ? even : chunk);
        }
    }
    const unsigned per_part = ((n + chunk - 1u) / chunk + RT_WF2_PARTS - 1u) / RT_WF2_PARTS * chunk;
    unsigned part = blockIdx.x % RT_WF2_PARTS, parts_left = RT_WF2_PARTS;
    unsigned part_lo = part * per_part;
    unsigned part_hi = part_lo + per_part < n ? part_lo + per_part : n;
    if (part_lo > n) part_lo = part_hi = n;
    const unsigned lane = threadIdx.x;
#ifdef RT_WF2_PROFILE
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pn[2] = {0, 0};
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        // ---- the wave's next chunk of rays
        unsigned c = 0xffffffffu;
        for (;;) {
            if (part_lo < part_hi) {
                if (lane == 0) c = atomicAdd(&chunk_counters[part].v, chunk);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                if (c < part_hi - part_lo) break;
            }
            c = 0xffffffffu;
            if (--parts_left == 0u) break;
            part = (part + 1u) % RT_WF2_PARTS;
            part_lo = part * per_part;
            part_hi = part_lo + per_part < n ? part_lo + per_part : n;
            if (part_lo > n) part_lo = part_hi = n;
        }
        if (c == 0xffffffffu) break;  // (the same in every lane)
        const unsigned first = part_lo + c;
        const unsigned avail = first + chunk < part_hi ? chunk : part_hi - first;
        const unsigned item = first + lane;
        Ray ray{v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};
        double limit = 0.0;
        Best best{__builtin_inf(), -1};
        bool found = false;
        bool valid = lane < avail && wf2_item_ray<MODE>(S, sp, cam, rows, pm, item, n_hits, ray, limit);
        if (valid && (MODE == WF2_PRIMARY || SHADOW)) count(cx, SHADOW ? ST_RAY_SHADOW : ST_RAY_PRIMARY);
        // the direction index's candidates first, every lane its own (primitive.rs:486-490's branch: see RtDirCand)
        if (A.has_cands && valid) found = wf2_candidates<SHADOW>(S, ray, limit, best, cx);
        RayF32 f;
        ray_f32(ray, A.extent, A.slack, f);
        float t_far = SHADOW ? f32_up(limit) : f32_up(best.t);
        bool alive = valid && !found;  // the lane still has a question (feelers: no occluder yet)
        RT_STAMP(0)
        // ---- the wave's walk.  (LDS traffic between the lanes of ONE wave needs no fence: a wave's LDS instructions are
        // carried out in the order issued; the wave barriers below only keep the compiler from moving them.)
        // (Keeping the nearest child in registers instead of pushing and popping it was measured, twice, in two forms: 0.6-0.8 ms
        // slower -- camera rays 1.48 -> 1.77 ms -- with fewer spilled SGPRs and one LDS round trip less per node; not understood.)
        unsigned sp_top = 0u;  // entries on the wave's stack (the same in every lane)
        {
            const unsigned long long all = __builtin_amdgcn_ballot_w64(alive);
            if (all != 0ull) {
                if (lane == 0) {
                    pk_code[0] = A.root, pk_mask[0] = all;
                    for (unsigned k = 0; k < A.n_always; ++k) pk_code[1u + k] = RT_BVH4_LEAF | A.always[k], pk_mask[1u + k] = all;
                }
                sp_top = 1u + A.n_always;  // (the unbounded leaves on top: tested first, a plane's hit bounds the walk)
            }
            __builtin_amdgcn_wave_barrier();
        }
        for (;;) {
            if (sp_top == 0u) break;
            --sp_top;
            const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk_code[sp_top]);
            const unsigned long long m = pk_mask[sp_top];
            const unsigned long long mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) |
                                            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
            const bool act = alive && ((mask >> lane) & 1ull) != 0ull;
            const unsigned long long amask = __builtin_amdgcn_ballot_w64(act);
            if (amask == 0ull || code == RT_BVH4_DONE) continue;
            if (code & RT_BVH4_LEAF) {
                RT_STAMP(1)
#ifdef RT_WF2_PROFILE
                pt[4] += (unsigned)__builtin_popcountll(amask);
                pt[5] += 64u;
#endif
                const int s = (int)RT_BVH4_SHAPE(code);
                uint32_t anc0 = RT_PRE_DEEP, anc1 = RT_PRE_NO_ANC;
                bool worth = act;
                if (A.pre) {
                    const LeafPre pre = load_uniform(A.pre + s);
                    anc0 = pre.anc[0], anc1 = pre.anc[1];
                    if (!(code & RT_BVH4_NOPRE) && act) {
                        const float reach = fmaxf(fmaxf(fabsf((float)ray.o.x), fabsf((float)ray.o.y)), fabsf((float)ray.o.z)) * 1.000001f + A.extent_max;
                        const float4 r0 = {pre.m[0], pre.m[1], pre.m[2], pre.m[3]}, r1 = {pre.m[4], pre.m[5], pre.m[6], pre.m[7]};
                        const float4 r2 = {pre.m[8], pre.m[9], pre.m[10], pre.m[11]};
                        const float4 kk = {pre.k1, pre.k2, __uint_as_float(pre.kind), pre.ymid};
                        const float4 cc = {pre.yhalf, pre.radius, 0.f, 0.f};
                        worth = leaf_may_hit_core(r0, r1, r2, kk, cc, ray, reach);
                        count(cx, ST_ACCEL_F32);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(worth) == 0ull) {
                    RT_STAMP(2)
                    continue;
                }
                bool hit = false;
                if (anc0 == RT_PRE_DEEP) {  // nested deeper than the record says: every lane the long way round
                    if (worth) hit = wf2_leaf<SHADOW>(S, s, ray, limit, best, cx);
                } else if (worth) {
                    hit = wf2_leaf_uniform<SHADOW>(S, s, anc0, anc1, ray, limit, best, cx);
                }
                if constexpr (SHADOW) {
                    if (hit) found = true, alive = false;
                } else {
                    t_far = f32_up(best.t);
                }
                RT_STAMP(2)
                continue;
            }
#ifdef RT_WF2_PROFILE
            pn[0] += (unsigned)__builtin_popcountll(amask);
            pn[1] += 64u;
#endif
            // an inner node: its 64 bytes once per wave
            const Bvh4Node nd = load_uniform(A.nodes + code);
            bool met[4] = {false, false, false, false};
            float tn[4] = {0.f, 0.f, 0.f, 0.f};
            const uint32_t kid[4] = {(uint32_t)nd.child[0], (uint32_t)nd.child[1], (uint32_t)nd.child[2], (uint32_t)nd.child[3]};
            if (act) {
                const uint32_t expw = (uint32_t)nd.exp[0] | ((uint32_t)nd.exp[1] << 8) | ((uint32_t)nd.exp[2] << 16);
                uint32_t qlo[3], qhi[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    qlo[a] = (uint32_t)nd.qlo[a][0] | ((uint32_t)nd.qlo[a][1] << 8) | ((uint32_t)nd.qlo[a][2] << 16) | ((uint32_t)nd.qlo[a][3] << 24);
                    qhi[a] = (uint32_t)nd.qhi[a][0] | ((uint32_t)nd.qhi[a][1] << 8) | ((uint32_t)nd.qhi[a][2] << 16) | ((uint32_t)nd.qhi[a][3] << 24);
                }
                node_children(nd.org, expw, kid, qlo, qhi, f, 0.f, t_far, met, tn);
                count(cx, ST_ACCEL_F32, 4u);
            }
            // which children any lane met, and where the first lane of the node meets them (children it does not meet: at
            // infinity, so they come first in the order below)
            unsigned long long cm[4];
            uint32_t key[4];  // (distances are >= 0: their bit patterns order as the numbers do)
            uint32_t kc[4] = {kid[0], kid[1], kid[2], kid[3]};
            const int rep = __builtin_ctzll(amask);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cm[k] = __builtin_amdgcn_ballot_w64(act && met[k]);
                key[k] = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(met[k] ? tn[k] : __builtin_inff()), rep);
            }
            // the four in descending order of that distance (a network of five exchanges on wave-uniform values): the far
            // ones go on the stack in that order, the nearest becomes the node at hand; children nobody met are skipped
            auto exchange = [&](int i, int j) {
                if (key[i] < key[j]) {
                    const uint32_t t = key[i];
                    key[i] = key[j], key[j] = t;
                    const unsigned long long m = cm[i];
                    cm[i] = cm[j], cm[j] = m;
                    const uint32_t c2 = kc[i];
                    kc[i] = kc[j], kc[j] = c2;
                }
            };
            exchange(0, 1), exchange(2, 3), exchange(0, 2), exchange(1, 3), exchange(1, 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (cm[q] != 0ull) {
                    if (lane == 0) pk_code[sp_top] = kc[q], pk_mask[sp_top] = cm[q];
                    ++sp_top;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        RT_STAMP(1)
        // ---- the answers
        if (valid) {
            if constexpr (SHADOW) {
                sp.occ[item] = found ? 1 : 0;
            } else {
                Wf2Hit h;
                h.t = best.t, h.idx = best.idx, h.pad = 0;
                sp.hits[item] = h;
            }
        } else if (lane < avail) {  // an item without a ray (a tile's slot outside the frame, a hit-less context): as wf2_trace_kernel leaves it
            if constexpr (SHADOW) {
                sp.occ[item] = 0;
            } else {
                Wf2Hit h;
                h.t = __builtin_inf(), h.idx = -1, h.pad = 0;
                sp.hits[item] = h;
            }
        }
        RT_STAMP(3)
    }
#ifdef RT_WF2_PROFILE
    if (lane == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&sp.cnt->prof[MODE][k], pt[k]);
        atomicAdd(&sp.cnt->prof[MODE][4], pn[0]);
        atomicAdd(&sp.cnt->prof[MODE][5], pn[1]);
        atomicAdd(&sp.cnt->prof[MODE][6], pt[4]);
        atomicAdd(&sp.cnt->prof[MODE][7], pt[5]);
    }
#endif
    flush_stats(cx, stats, status);
}

// Slots for the lanes of a wave that ask, in lane order, from one counter: one atomic per wave.
RT_DEV unsigned wave_slots(unsigned* counter, bool want) {
    const unsigned long long asking = __builtin_amdgcn_ballot_w64(want);
    if (asking == 0ull) return 0u;
    const unsigned lane = __lane_id();
    const int leader = __builtin_ctzll(asking);
    unsigned base = 0u;
    if ((int)lane == leader) base = atomicAdd(counter, (unsigned)__builtin_popcountll(asking));
    base = (unsigned)__shfl((int)base, leader, 64);
    return base + (unsigned)__builtin_popcountll(asking & ((1ull << lane) - 1ull));
}

// prepare_computations (intersection.rs:45-99 without n1 / n2) for a leaf nested in at most two groups, as the leaf
// records name them: the (up to) three inverse matrices are read ONCE and stay in registers for the three passes the
// reference makes over them (point to object space, normal to world space, over_point to object space), where
// world_to_object / normal_to_world of rt_trace.h walk the parent links and read every matrix again per pass -- the
// kernel is bound by its vector memory instructions.  Same operations in the same order.  False: not this leaf
// (nested deeper, a smooth triangle, no leaf records): the caller uses prepare_computations.
template <bool STATS, int FLAT>
RT_DEV bool prepare_in_registers(const SceneDev& S, const LeafPre* __restrict__ pre, const Ray& ray, const Best& hit, Comps& c,
                                 Ctx<STATS, FLAT>& cx) {
    if (!pre) return false;
    const RtShape* __restrict__ sh = S.shapes + hit.idx;
    const unsigned kind = RT_KIND(sh->kind_flags);
    const uint2 anc = *reinterpret_cast<const uint2*>(&pre[hit.idx].anc[0]);
    if (anc.x == RT_PRE_DEEP || kind == RT_SMOOTH_TRIANGLE) return false;
    const bool has_o = anc.x != RT_PRE_NO_ANC, has_i = anc.y != RT_PRE_NO_ANC;  // outermost group, the one inside it
    double ms[12], mo[12], mi[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) ms[k] = sh->inv[k], mo[k] = 0.0, mi[k] = 0.0;
    if (has_o) {
        const RtShape* __restrict__ g = S.shapes + anc.x;
#pragma unroll
        for (int k = 0; k < 12; ++k) mo[k] = g->inv[k];
    }
    if (has_i) {
        const RtShape* __restrict__ g = S.shapes + anc.y;
#pragma unroll
        for (int k = 0; k < 12; ++k) mi[k] = g->inv[k];
    }
    count(cx, ST_HITS_SHADED);
    count(cx, ST_NORMAL_LEVELS, 2u * (1u + (has_o ? 1u : 0u) + (has_i ? 1u : 0u)));
    auto to_object = [&](V3 p) {  // primitive.rs:179-185: trail.iter().rev(), then the primitive itself
        if (has_o) p = mul_point(mo, p);
        if (has_i) p = mul_point(mi, p);
        return mul_point(ms, p);
    };
    const V3 point = ray.o + ray.d * hit.t;  // ray.rs:15-17
    c.idx = hit.idx;
    c.t = hit.t;
    c.eye_v = -ray.d;
    const V3 local_point = to_object(point);
    const V3 local_n = local_normal(S, sh, kind, local_point, nullptr, 0, ray);  // (chain and ray: smooth triangles only)
    V3 normal_v = normalize(mul_vector_transposed(ms, local_n));  // primitive.rs:187-192, inside out
    if (has_i) normal_v = normalize(mul_vector_transposed(mi, normal_v));
    if (has_o) normal_v = normalize(mul_vector_transposed(mo, normal_v));
    c.inside = dot(normal_v, c.eye_v) < 0.0;
    if (c.inside) normal_v = -normal_v;
    c.normal_v = normal_v;
    c.over_point = point + normal_v * EPSILON;
    c.under_point = point - normal_v * EPSILON;
    c.reflect_v = reflect(ray.d, normal_v);
    c.object_over_point = to_object(c.over_point);
    c.n1 = 1.0;
    c.n2 = 1.0;
    return true;
}

// prepare_computations + pattern colour of every hit of the generation (world.rs:32-43 up to shade_hit's first
// line); a miss is BLACK, delivered at once (world.rs:41).  PRIMARY: generation 0, items are pixels.
#ifndef RT_WF2_PREP_WAVES
#define RT_WF2_PREP_WAVES 1
#endif
#ifndef RT_WF2_FINISH_WAVES
#define RT_WF2_FINISH_WAVES 1
#endif
template <bool STATS, bool PRIMARY>
__global__ __launch_bounds__(256, RT_WF2_PREP_WAVES) void wf2_prepare_kernel(const SceneDev S, const LeafPre* __restrict__ pre, Wf2Space sp, RtCameraDesc cam, RowMap rows, int depth,
                                                          unsigned gen, double* __restrict__ out, RtStats* __restrict__ stats,
                                                          unsigned* __restrict__ status) {
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    const PixelMap pm = pixel_map(cam, rows);
    const unsigned n = PRIMARY ? pm.n_items : sp.cnt->n_rays[gen];
    const unsigned n_round = (n + 63u) & ~63u;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
        bool valid = i < n;
        Ray ray{v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};
        uint32_t parent = 0, remaining = 0;
        if (valid) {
            if constexpr (PRIMARY) {
                unsigned x, k, y;
                valid = pixel_of_item(cam, rows, pm, i, x, k, y);
                if (valid) {
                    ray = ray_for_pixel(cam, x, y);
                    parent = RT_WF_PIXEL | (k * cam.hsize + x);
                    remaining = (uint32_t)depth;
                }
            } else {
                const WfItem* it = sp.qd + i;
                remaining = it->remaining;
                valid = remaining != RT_WF_EMPTY;
                if (valid) ray = it->ray, parent = it->parent;
            }
        }
        Best hit{__builtin_inf(), -1};
        if (valid) {
            const Wf2Hit h = sp.hits[i];
            hit.t = h.t, hit.idx = h.idx;
        }
        const bool has = valid && hit.idx >= 0;
        if (valid && !has) wf_deliver(sp.ws.nodes, out, parent, v3(0.0, 0.0, 0.0));
        // The shading context of ray i is slot i: nine rays in ten hit something, so packing the hits would save little,
        // and the one counter every wave would bump for its slots was this kernel's whole time (see Wf2Counters).  A slot
        // without a hit is marked (NaN in over_point.x, no material) and skipped by whoever reads slots.
        const unsigned slot = i;
        if (i < n && !has) {
            ctx_f(sp, slot, CX_OVER) = __builtin_nan("");
            ctx_u(sp, slot, CU_MAT) = 0xffffffffu;
        }
        bool want_n = false;
        if (has) {
            const RtShape* __restrict__ sh = S.shapes + hit.idx;
            const RtMaterial m = S.mats[sh->material];
            want_n = m.transparency != 0.0 && remaining != 0u;
            Comps c;
            if (!prepare_in_registers(S, pre, ray, hit, c, cx)) {
                Scratch unused_list;  // prepare_computations touches the list only for n1 / n2, which come later (dead here)
                prepare_computations(S, unused_list, ray, hit, false, c, cx);
            }
            V3 color = v3(m.color[0], m.color[1], m.color[2]);
            if (m.pattern >= 0) {
                unsigned nodes = 0;
                color = pattern_eval(S.pats, m.pattern, c.object_over_point, nodes);
                count(cx, ST_PATTERN_NODES, nodes);
            }
            ctx_put(sp, slot, CX_OVER, c.over_point);
            ctx_put(sp, slot, CX_EYE, c.eye_v);
            ctx_put(sp, slot, CX_NORMAL, c.normal_v);
            ctx_put(sp, slot, CX_COLOR, color);
            ctx_f(sp, slot, CX_T) = hit.t;
            ctx_u(sp, slot, CU_MAT) = sh->material;
            ctx_u(sp, slot, CU_PARENT) = parent;
            ctx_u(sp, slot, CU_REMAINING) = remaining;
            ctx_u(sp, slot, CU_NODE) = RT_WF2_NODE_NONE;
        }
        // requests for n1 / n2 go to the part of the list that belongs to this group of four waves
        const unsigned creq_part = (i >> 8) % RT_WF2_PARTS;
        const unsigned r = wave_slots(&sp.cnt->n_creq[creq_part].v, want_n);
        if (want_n) sp.creq[(size_t)creq_part * sp.creq_part_cap + r] = slot;
    }
    flush_stats(cx, stats, status);
}

// The Shade record of a hit whose material lets it have children (reflective or transparent, recursion budget left), for
// shade_children: everything the context does not keep is worked out again as prepare_computations had it
// (intersection.rs:76-98) -- reflect_v from -eye = the ray's direction and the normal; for a transparent hit n1 / n2 as the
// containers walk left them and under_point from the ray (`gen`'s item `slot`: a pixel's camera ray, or the queued one)
// and t.  surface and color are not read by shade_children and are left alone.
RT_DEV void ctx_shade_children(const SceneDev& S, const Wf2Space& sp, const RtCameraDesc& cam, const RowMap& rows, const PixelMap& pm,
                               unsigned gen, unsigned slot, const RtMaterial& m, uint32_t remaining, Shade& sd) {
    sd.over = ctx_v3(sp, slot, CX_OVER), sd.eye = ctx_v3(sp, slot, CX_EYE), sd.normal = ctx_v3(sp, slot, CX_NORMAL);
    sd.reflect_v = reflect(-sd.eye, sd.normal);  // intersection.rs:91: ray.direction.reflect(&normal_v), eye_v = -direction
    sd.under = sd.over;
    sd.n1 = 1.0, sd.n2 = 1.0;
    sd.li = S.n_lights;
    sd.want_n = (m.transparency != 0.0 && remaining != 0u) ? 1 : 0;
    if (sd.want_n != 0) {
        sd.n1 = ctx_f(sp, slot, CX_N1), sd.n2 = ctx_f(sp, slot, CX_N2);
        Ray ray{v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};
        double unused;
        if (gen == 0u)
            (void)wf2_item_ray<WF2_PRIMARY>(S, sp, cam, rows, pm, slot, 0u, ray, unused);
        else
            (void)wf2_item_ray<WF2_QUEUE>(S, sp, cam, rows, pm, slot, 0u, ray, unused);
        const V3 point = ray.o + ray.d * ctx_f(sp, slot, CX_T);  // ray.rs:15-17
        sd.under = point - sd.normal * EPSILON;                  // intersection.rs:90
    }
}
// can the hit have children at all (world.rs:99,110: the early exits of reflected_color / refracted_color)?
RT_DEV bool may_have_children(const RtMaterial& m, uint32_t remaining) {
    return remaining != 0u && (m.reflective != 0.0 || m.transparency != 0.0);
}

// wf_commit for the pipeline: a node per asking lane and its child rays into the shard's slice of the sliced queue,
// the wave's reflected rays first, then its refracted ones; two atomics per wave.  Every lane of the wave calls it.
// Returns false for a lane whose request found no room (its pieces that did fit are blanked).
// The wave's nodes are consecutive in the pool and its items consecutive in the slice, but a lane's 128-byte node written
// with per-lane stores touches 64 different lines per instruction (the kernel was bound by exactly that: 75 % of its time
// in the texture-address units, 1.4 % in the ALUs).  So the records are put together in LDS -- `stage`: RT_WF2_STAGE_WORDS
// 8-byte words of the calling wave's own, record strides of 17 and 9 words so that the lanes' field stores spread over the
// banks -- and leave in whole 512-byte rows.
#define RT_WF2_STAGE_WORDS 1152u  // max(64 nodes x 17, 128 items x 9) 8-byte words
RT_DEV bool wf2_commit(const WfSpace& ws, unsigned shard, unsigned level, bool want, uint32_t parent, V3 surface,
                       double reflective, double transparency, const Children& ch, V3 reflect_o, V3 reflect_d,
                       int remaining_children, uint32_t& node_out, uint2* stage) {
    node_out = RT_WF2_NODE_NONE;
    const unsigned long long asking = __builtin_amdgcn_ballot_w64(want);
    if (asking == 0ull) return false;
    const unsigned lane = __lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    WfShard& sh = ws.shards[shard];
    const bool has0 = want && ch.do_reflect, has1 = want && ch.do_refract;
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(has0), m1 = __builtin_amdgcn_ballot_w64(has1);
    const unsigned n0 = (unsigned)__builtin_popcountll(m0), n1 = (unsigned)__builtin_popcountll(m1);
    const unsigned n_ask = (unsigned)__builtin_popcountll(asking);
    const int leader = __builtin_ctzll(asking);
    unsigned base_nodes = 0u, base_items = 0u;
    if ((int)lane == leader) {
        base_nodes = atomicAdd(&sh.nodes_asked[level], n_ask);
        base_items = atomicAdd(&sh.items[(level + 1u) & 1u][0], n0 + n1);
    }
    base_nodes = (unsigned)__shfl((int)base_nodes, leader, 64);
    base_items = (unsigned)__shfl((int)base_items, leader, 64);
    const unsigned rank = (unsigned)__builtin_popcountll(asking & below);
    const unsigned i0 = (unsigned)__builtin_popcountll(m0 & below), i1 = n0 + (unsigned)__builtin_popcountll(m1 & below);  // items of the wave
    const unsigned long long ni0 = (unsigned long long)sh.node_begin[level] + base_nodes;
    const unsigned long long ni = ni0 + rank;
    const unsigned long long s0 = (unsigned long long)base_items + i0, s1 = (unsigned long long)base_items + i1;
    const bool fit0 = !has0 || s0 < ws.q_cap, fit1 = !has1 || s1 < ws.q_cap;
    const bool placed = want && ni < ws.node_cap && fit0 && fit1;
    WfItem* __restrict__ q = ws.queue[(level + 1u) & 1u] + (size_t)shard * ws.q_cap;
    WfNode* __restrict__ nodes = ws.nodes + (size_t)shard * ws.node_cap;
    const uint32_t global_node = (uint32_t)((size_t)shard * ws.node_cap + ni);
    // ---- nodes
    if (want && ni < ws.node_cap) {
        double* d = reinterpret_cast<double*>(stage + rank * 17u);
        uint32_t* u = reinterpret_cast<uint32_t*>(d + 12);
        d[0] = surface.x, d[1] = surface.y, d[2] = surface.z;
        d[3] = 0.0, d[4] = 0.0, d[5] = 0.0, d[6] = 0.0, d[7] = 0.0, d[8] = 0.0;  // child[0], child[1]
        d[9] = reflective, d[10] = transparency, d[11] = ch.reflectance;
        u[0] = parent;
        u[1] = !placed ? (uint32_t)WF_DEAD
                       : (ch.use_schlick ? WF_USE_SCHLICK : 0u) | (ch.do_reflect ? WF_HAS_REFLECT : 0u) | (ch.do_refract ? WF_HAS_REFRACT : 0u);
        d[13] = 0.0, d[14] = 0.0, d[15] = 0.0;  // pad
    }
    wf2_lds_wave_sync();
    {
        const unsigned n_valid = ni0 >= ws.node_cap ? 0u : (n_ask < ws.node_cap - ni0 ? n_ask : (unsigned)(ws.node_cap - ni0));
        uint2* __restrict__ g = reinterpret_cast<uint2*>(nodes + ni0);
        for (unsigned i = lane; i < n_valid * 16u; i += 64u) g[i] = stage[(i >> 4) * 17u + (i & 15u)];
    }
    wf2_lds_wave_sync();
    // ---- items
    if (has0 && fit0) {
        double* d = reinterpret_cast<double*>(stage + i0 * 9u);
        uint32_t* u = reinterpret_cast<uint32_t*>(d + 6);
        d[0] = reflect_o.x, d[1] = reflect_o.y, d[2] = reflect_o.z, d[3] = reflect_d.x, d[4] = reflect_d.y, d[5] = reflect_d.z;
        u[0] = global_node << 1, u[1] = placed ? (uint32_t)remaining_children : RT_WF_EMPTY;
        d[7] = 0.0;  // pad
    }
    if (has1 && fit1) {
        double* d = reinterpret_cast<double*>(stage + i1 * 9u);
        uint32_t* u = reinterpret_cast<uint32_t*>(d + 6);
        const Ray r = ch.refract_ray;
        d[0] = r.o.x, d[1] = r.o.y, d[2] = r.o.z, d[3] = r.d.x, d[4] = r.d.y, d[5] = r.d.z;
        u[0] = (global_node << 1) | 1u, u[1] = placed ? (uint32_t)remaining_children : RT_WF_EMPTY;
        d[7] = 0.0;
    }
    wf2_lds_wave_sync();
    {
        const unsigned n_items = n0 + n1;
        const unsigned n_valid = base_items >= ws.q_cap ? 0u : (n_items < ws.q_cap - base_items ? n_items : ws.q_cap - base_items);
        uint2* __restrict__ g = reinterpret_cast<uint2*>(q + base_items);
        for (unsigned i = lane; i < n_valid * 8u; i += 64u) g[i] = stage[(i >> 3) * 9u + (i & 7u)];
    }
    wf2_lds_wave_sync();  // (the next call's stores must not overtake these reads)
    if (placed) node_out = global_node;
    return placed;
}

// shade_hit (world.rs:55-83) of every hit of the generation, in two kernels on two streams:
//   wf2_spawn_kernel (the caller's stream, after the containers walks): the children -- reflected / refracted rays,
//       Schlick (world.rs:72-82 as far as it needs no colour yet) -- into the node pool and the next generation's queue
//       (wf2_commit); the hit's context remembers its node.  The next generation starts from here.
//   wf2_light_kernel (the scene's second stream, after the generation's feelers): lighting per light (world.rs:58-70) with the
//       feelers' answers; the surface colour goes into the hit's node, or -- a hit without children -- straight to whoever
//       waits for it, or -- children refused for want of room -- into the context for wf2_overflow_kernel.
// Feelers are two thirds of all rays and nothing downstream of a generation needs their answers before the frame's colours
// are folded together at the very end: off the critical path they fill the device while the colour rays' launches, each as
// long as its slowest wave, come and go.
template <bool STATS>
__global__ __launch_bounds__(256, RT_WF2_FINISH_WAVES) void wf2_spawn_kernel(const SceneDev S, Wf2Space sp, RtCameraDesc cam, RowMap rows, unsigned gen,
                                                                            unsigned stretch_shift, unsigned inlane_below,
                                                                            RtStats* __restrict__ stats, unsigned* __restrict__ status) {
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    __shared__ uint2 stage_all[4u * RT_WF2_STAGE_WORDS];  // (256-thread blocks)
    uint2* stage = stage_all + (threadIdx.x >> 6) * RT_WF2_STAGE_WORDS;
    const PixelMap pm = pixel_map(cam, rows);
    const unsigned n = wf2_n_slots(sp, gen, pm);
    const unsigned n_round = (n + 63u) & ~63u;
    // A generation that has thinned out below what the device holds at once (a launch that began with more): its hits keep
    // their subtrees in their lanes (wf2_overflow_kernel, as for children that find no room) -- the generations that would
    // follow are launches as long as their slowest wave with most of the device idle.  See RT_WF2_INLANE_BELOW.
    const bool inlane = pm.n_items >= inlane_below && n < inlane_below;
    for (unsigned slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_round; slot += gridDim.x * blockDim.x) {
        const uint32_t mat = slot < n ? ctx_u(sp, slot, CU_MAT) : 0xffffffffu;
        Shade sd;
        sd.surface = v3(0.0, 0.0, 0.0);
        sd.over = sd.reflect_v = v3(0.0, 0.0, 0.0);
        uint32_t remaining = 0, parent = 0;
        Children ch;
        ch.do_reflect = ch.do_refract = ch.use_schlick = false;
        ch.reflectance = 0.0;
        double reflective = 0.0, transparency = 0.0;
        bool want = false;
        if (mat != 0xffffffffu) {  // a hit: only one whose material and budget allow children reads the rest of its context
            remaining = ctx_u(sp, slot, CU_REMAINING);
            const RtMaterial m = S.mats[mat];
            if (may_have_children(m, remaining)) {
                sd.mat = (int)mat;
                ctx_shade_children(S, sp, cam, rows, pm, gen, slot, m, remaining, sd);
                parent = ctx_u(sp, slot, CU_PARENT);
                reflective = m.reflective, transparency = m.transparency;
                shade_children(m, sd, Ray{sd.over, sd.reflect_v}, (int)remaining, ch, cx);
                want = ch.do_reflect || ch.do_refract;
            }
        }
        // a wave's children go to the shard picked by where they come from: a stretch of consecutive hits
        const unsigned out_shard = (((unsigned)__builtin_amdgcn_readfirstlane((int)slot) >> stretch_shift) * 2654435761u) >> 24;
        uint32_t node;
        const bool placed = wf2_commit(sp.ws, out_shard, gen, want && !inlane, parent, v3(0.0, 0.0, 0.0), reflective, transparency, ch, sd.over,
                                      sd.reflect_v, (int)remaining - 1, node, stage);
        if (placed) {
            if (ch.do_reflect) count(cx, ST_RAY_REFLECT);
            if (ch.do_refract) count(cx, ST_RAY_REFRACT);
            ctx_u(sp, slot, CU_NODE) = node;
        }
        const bool refused = want && !placed;
        const unsigned r = wave_slots(&sp.cnt->n_overflow[gen], refused);
        if (refused) {
            ctx_u(sp, slot, CU_NODE) = RT_WF2_NODE_REFUSED;
            ctx_put(sp, slot, CX_UNDER, sd.under);  // (what wf2_overflow_kernel cannot work out again: see CX_UNDER)
            sp.overflow[r] = slot;
        }
    }
    flush_stats(cx, stats, status);
}

template <bool STATS>
__global__ __launch_bounds__(256, RT_WF2_FINISH_WAVES) void wf2_light_kernel(const SceneDev S, Wf2Space sp, RtCameraDesc cam, RowMap rows, unsigned gen,
                                                                            double* __restrict__ out, RtStats* __restrict__ stats,
                                                                            unsigned* __restrict__ status) {
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    const unsigned n = wf2_n_slots(sp, gen, pixel_map(cam, rows));
    for (unsigned slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        const uint32_t mat = ctx_u(sp, slot, CU_MAT);
        if (mat == 0xffffffffu) continue;  // no hit in this slot
        const V3 over = ctx_v3(sp, slot, CX_OVER), eye = ctx_v3(sp, slot, CX_EYE), normal = ctx_v3(sp, slot, CX_NORMAL);
        const V3 color = ctx_v3(sp, slot, CX_COLOR);
        V3 surface = v3(0.0, 0.0, 0.0);
        const RtMaterial m = S.mats[mat];
        for (unsigned l = 0; l < S.n_lights; ++l) {  // world.rs:58-70, in light order
            const RtLight light = S.lights[l];
            const bool occluded = sp.occ[(size_t)l * n + slot] != 0;
            surface = surface + lighting(m, color, light, over, eye, normal, occluded, cx);
        }
        const uint32_t node = ctx_u(sp, slot, CU_NODE);
        if (node == RT_WF2_NODE_REFUSED) {
            ctx_put(sp, slot, CX_SURFACE, surface);
        } else if (node != RT_WF2_NODE_NONE) {
            sp.ws.nodes[node].surface = surface;
        } else {
            // No children: world.rs:76-82 with both of them black.  The Schlick mix is still the one the material asks for
            // (reflective > 0 && transparency > 0), with the reflectance shade_children leaves at 0.0 for a hit without
            // recursion budget -- and a hit WITH budget and a reflective material has children (world.rs:99), so it is not here.
            const bool use_schlick = m.reflective > 0.0 && m.transparency > 0.0;
            wf_deliver(sp.ws.nodes, out, ctx_u(sp, slot, CU_PARENT),
                       combine(surface, v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 0.0), use_schlick, 0.0));
        }
    }
    flush_stats(cx, stats, status);
}

// Hits whose children found no room in the queue or the node pool: the lane keeps the subtree (color_at's frame
// stack), exactly as world.rs:98-127 recurse.  Capacity costs speed, never the result.
template <bool STATS>
__global__ __launch_bounds__(64, RT_WAVES_PER_SIMD) void wf2_overflow_kernel(const SceneDev S, Wf2Space sp, unsigned gen, double* __restrict__ out,
                                                                             RtStats* __restrict__ stats, unsigned* __restrict__ status) {
    constexpr int CLS = RT_CLASS_NESTED_LEAVES;
    Ctx<STATS, CLS> cx;
    cx.status = 0;
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < ST_COUNT; ++i) cx.cnt[i] = 0;
    }
    const unsigned n = sp.cnt->n_overflow[gen];
    for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const unsigned slot = sp.overflow[r];
        Shade sd;
        const uint32_t remaining = ctx_u(sp, slot, CU_REMAINING);
        sd.mat = (int)ctx_u(sp, slot, CU_MAT);
        const RtMaterial m = S.mats[sd.mat];
        // (as ctx_shade_children has it, except that under_point is the one wf2_spawn_kernel left: this kernel runs beside
        // the compaction of the next generation's queue, which overwrites the rays of this one)
        sd.over = ctx_v3(sp, slot, CX_OVER), sd.eye = ctx_v3(sp, slot, CX_EYE), sd.normal = ctx_v3(sp, slot, CX_NORMAL);
        sd.reflect_v = reflect(-sd.eye, sd.normal);
        sd.under = ctx_v3(sp, slot, CX_UNDER);
        sd.li = S.n_lights;
        sd.want_n = (m.transparency != 0.0 && remaining != 0u) ? 1 : 0;
        sd.n1 = sd.want_n != 0 ? ctx_f(sp, slot, CX_N1) : 1.0, sd.n2 = sd.want_n != 0 ? ctx_f(sp, slot, CX_N2) : 1.0;
        sd.surface = ctx_v3(sp, slot, CX_SURFACE);
        Children ch;
        Ctx<false, CLS> quiet;  // (the Schlick evaluation was counted when the hit was lit)
        quiet.status = 0;
        shade_children(m, sd, Ray{sd.over, sd.reflect_v}, (int)remaining, ch, quiet);
        Scratch W;
        Frame frames[RT_MAX_DEPTH + 1];
        V3 child[2] = {v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 0.0)};
        for (int kind = 0; kind < 2; ++kind) {  // reflected first, then refracted (world.rs:72-73)
            if (!(kind == 0 ? ch.do_reflect : ch.do_refract)) continue;
            count(cx, kind == 0 ? ST_RAY_REFLECT : ST_RAY_REFRACT);
            const Ray ray = kind == 0 ? Ray{sd.over, sd.reflect_v} : ch.refract_ray;
            child[kind] = color_at(S, W, frames, ray, (int)remaining - 1, cx) * (kind == 0 ? m.reflective : m.transparency);
        }
        wf_deliver(sp.ws.nodes, out, ctx_u(sp, slot, CU_PARENT), combine(sd.surface, child[0], child[1], ch.use_schlick, ch.reflectance));
    }
    flush_stats(cx, stats, status);
}

// Generation `gen` is through: its nodes end here (what wf_mark_kernel does), and the rays it queued for generation
// gen + 1 -- one slice of the sliced queue per shard, filled by wf2_commit -- get their places in the dense queue.
__global__ __launch_bounds__(RT_WF_SHARDS) void wf2_scan_kernel(Wf2Space sp, unsigned gen) {
    __shared__ unsigned totals[RT_WF_SHARDS];
    const unsigned s = threadIdx.x;
    WfShard& sh = sp.ws.shards[s];
    const unsigned long long end = (unsigned long long)sh.node_begin[gen] + sh.nodes_asked[gen];
    sh.node_begin[gen + 1u] = end < sp.ws.node_cap ? (unsigned)end : sp.ws.node_cap;
    const unsigned asked = sh.items[(gen + 1u) & 1u][0];
    sh.items[(gen + 1u) & 1u][0] = 0u;  // the generation after next counts here
    totals[s] = asked < sp.ws.q_cap ? asked : sp.ws.q_cap;
    __syncthreads();
    if (s == 0) {  // 256 values: a serial scan is a microsecond
        unsigned run = 0;
        for (unsigned k = 0; k < RT_WF_SHARDS; ++k) {
            sp.cnt->offsets[k] = run;
            run += totals[k];
        }
        sp.cnt->offsets[RT_WF_SHARDS] = run;
        sp.cnt->n_rays[gen + 1u] = run;
    }
    if (s < RT_WF2_PARTS) sp.cnt->n_creq[s].v = 0u;
}
// One block per shard: the shard's slice into the dense queue, counting-sorted by where the ray starts (a grid of
// RT_WF2_SORT_CELLS^3 cells over the scene's extent, in Morton order) and the octant of its direction: rays that start near
// each other and point the same way walk the same part of the hierarchy, and 64 neighbours of the dense queue are one wave.
#ifndef RT_WF2_SORT_CELLS
#define RT_WF2_SORT_CELLS 4u  // cells per axis (a power of two); 1: by octant only.  Measured 1 / 2 / 4 / 8: 21.3 / 21.0 / 20.85 / 20.84 ms; octant-major keys 21.0; one sort over all slices instead of one per slice 20.95
#endif
RT_DEV unsigned wf2_sort_key(const WfItem& it, float ex, float ey, float ez) {
    unsigned key = wf_octant(it.ray.d);
    if constexpr (RT_WF2_SORT_CELLS > 1u) {
        const float n = (float)RT_WF2_SORT_CELLS;
        auto cell = [n](double o, float e) {
            const float u = ((float)o / e + 1.f) * 0.5f * n;
            return (unsigned)(u > 0.f ? (u < n ? u : n - 1.f) : 0.f);  // (NaN -> 0)
        };
        const unsigned cx = cell(it.ray.o.x, ex), cy = cell(it.ray.o.y, ey), cz = cell(it.ray.o.z, ez);
        unsigned morton = 0;
#pragma unroll
        for (unsigned b = 0; (1u << b) < RT_WF2_SORT_CELLS; ++b)
            morton |= (((cx >> b) & 1u) << (3u * b)) | (((cy >> b) & 1u) << (3u * b + 1u)) | (((cz >> b) & 1u) << (3u * b + 2u));
        key = morton * RT_WF_BINS + key;
    }
    return key;
}
// (Round 3.)  A STABLE counting sort, 1024 threads per slice: the order of a slice is the order its hits were in, rays that
// were neighbours there walk the same part of the hierarchy, and every kernel of the next generation works in the dense
// queue's order (slot = ray index) -- measured: with places handed out by plain atomics, in whatever order sixteen waves
// arrive, the next generation's walks are 8 % longer than with the slice's order kept.  So: wave w of the block owns the
// w-th sixteenth of the slice; the first pass counts per (key, wave); ONE exclusive scan over the counters laid out key-major,
// wave-minor turns each into the place of the first item of that key in that wave's part; the second pass goes through the
// wave's part in order, 16 items a step, ranks the items of equal key among themselves (lane order), and the first of each
// key takes the places for all of them from the wave's OWN counter -- no wave waits for another.  An item is read and moved
// by four lanes, 16 bytes each (whole 64-byte items per instruction), which OR their parts of the key together.
#ifndef RT_WF2_COMPACT_THREADS
#ifdef RT_HOST_EMU
#define RT_WF2_COMPACT_THREADS 256u
#else
#define RT_WF2_COMPACT_THREADS 1024u
#endif
#endif
#ifndef RT_WF2_COMPACT_AHEAD
#define RT_WF2_COMPACT_AHEAD 4u  // steps of 16 items whose loads a wave has in flight together
#endif
// the part of wf2_sort_key() that piece `piece` (16 bytes) of an item decides: ray.o.x, o.y | o.z, d.x | d.y, d.z | the rest
RT_DEV unsigned wf2_sort_key_part(unsigned piece, const uint4& v, float ex, float ey, float ez) {
    static_assert(RT_WF_BINS == 8u, "the key's parts OR together only with plain octants");
    const double a = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
    const double b = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
    const float n = (float)RT_WF2_SORT_CELLS;
    auto cell = [n](double o, float e) {
        const float u = ((float)o / e + 1.f) * 0.5f * n;
        return (unsigned)(u > 0.f ? (u < n ? u : n - 1.f) : 0.f);  // (NaN -> 0)
    };
    auto spread = [](unsigned c, unsigned axis) {
        unsigned m = 0;
#pragma unroll
        for (unsigned bit = 0; (1u << bit) < RT_WF2_SORT_CELLS; ++bit) m |= ((c >> bit) & 1u) << (3u * bit + axis);
        return m;
    };
    unsigned key = 0;
    if (piece == 0u) {
        if constexpr (RT_WF2_SORT_CELLS > 1u) key = (spread(cell(a, ex), 0u) | spread(cell(b, ey), 1u)) * RT_WF_BINS;
    } else if (piece == 1u) {
        if constexpr (RT_WF2_SORT_CELLS > 1u) key = spread(cell(a, ez), 2u) * RT_WF_BINS;
        key |= b < 0.0 ? 1u : 0u;
    } else if (piece == 2u) {
        key = (a < 0.0 ? 2u : 0u) | (b < 0.0 ? 4u : 0u);
    }
    return key;
}
__global__ __launch_bounds__(RT_WF2_COMPACT_THREADS) void wf2_compact_kernel(Wf2Space sp, unsigned gen, float ex, float ey, float ez) {
    constexpr unsigned KEYS = RT_WF_BINS * RT_WF2_SORT_CELLS * RT_WF2_SORT_CELLS * RT_WF2_SORT_CELLS;
    constexpr unsigned NW = RT_WF2_COMPACT_THREADS / 64u;          // waves of the block
    constexpr unsigned CELLS = KEYS * NW;                          // counters, [key][wave]
    constexpr unsigned PER = CELLS / RT_WF2_COMPACT_THREADS;       // counters per thread in the scan
    constexpr unsigned AHEAD = RT_WF2_COMPACT_AHEAD;
    static_assert(CELLS % RT_WF2_COMPACT_THREADS == 0u && CELLS * 4u <= 48u * 1024u, "the counters: a whole number per thread, in LDS");
    static_assert(sizeof(WfItem) == 64 && offsetof(WfItem, ray) == 0, "an item is four 16-byte pieces, the ray first");
    __shared__ unsigned cells[CELLS];
    __shared__ unsigned wave_sum[NW];
    const unsigned tid = threadIdx.x;  // (blockDim.x == RT_WF2_COMPACT_THREADS: the one launch site says so)
    const unsigned shard = blockIdx.x;
    const unsigned begin = sp.cnt->offsets[shard], n = sp.cnt->offsets[shard + 1u] - begin;
    if (n == 0u) return;  // (the whole block: nothing in this slice)
    const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(sp.ws.queue[(gen + 1u) & 1u] + (size_t)shard * sp.ws.q_cap);
    uint4* __restrict__ dst4 = reinterpret_cast<uint4*>(sp.qd + begin);
    const unsigned piece = tid & 3u, lane = tid & 63u, quad = lane >> 2, wave = tid >> 6;
    // the wave's part of the slice: whole steps of 16 items
    const unsigned part = ((n + NW - 1u) / NW + 15u) & ~15u;
    const unsigned lo = wave * part < n ? wave * part : n, hi = lo + part < n ? lo + part : n;
    // the key of the quad's item, in all four lanes
    auto quad_key = [&](unsigned p) {
        p |= (unsigned)__shfl_xor((int)p, 1, 64);
        p |= (unsigned)__shfl_xor((int)p, 2, 64);
        return p;
    };
    for (unsigned k = tid; k < CELLS; k += RT_WF2_COMPACT_THREADS) cells[k] = 0u;
    __syncthreads();
    for (unsigned base = lo; base < hi; base += AHEAD * 16u) {  // (wave-uniform bounds: whole waves reach the shuffles)
        uint4 v[AHEAD];
#pragma unroll
        for (unsigned u = 0; u < AHEAD; ++u) {
            const unsigned i = base + u * 16u + quad;
            v[u] = uint4{0u, 0u, 0u, 0u};
            if (i < hi) v[u] = src4[(size_t)i * 4u + piece];
        }
#pragma unroll
        for (unsigned u = 0; u < AHEAD; ++u) {
            const unsigned i = base + u * 16u + quad;
            const unsigned key = quad_key(wf2_sort_key_part(piece, v[u], ex, ey, ez));
#ifdef RT_HOST_EMU  // (tests/emu: the quad's key is the whole item's key)
            if (i < hi && key != wf2_sort_key(reinterpret_cast<const WfItem*>(src4)[i], ex, ey, ez)) std::abort();
#endif
            if (i < hi && piece == 0u) atomicAdd(&cells[key * NW + wave], 1u);
        }
    }
    __syncthreads();
    {  // exclusive scan of the counters as laid out: thread t's PER consecutive ones, the wave's threads, the block's waves
        unsigned mine[PER], sum = 0u;
#pragma unroll
        for (unsigned j = 0; j < PER; ++j) {
            mine[j] = cells[tid * PER + j];
            sum += mine[j];
        }
        unsigned incl = sum;
#pragma unroll
        for (unsigned off = 1u; off < 64u; off <<= 1) {
            const unsigned up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63u) wave_sum[wave] = incl;
        __syncthreads();
        unsigned run = incl - sum;
        for (unsigned w = 0; w < wave; ++w) run += wave_sum[w];
#pragma unroll
        for (unsigned j = 0; j < PER; ++j) {
            cells[tid * PER + j] = run;
            run += mine[j];
        }
    }
    __syncthreads();
    for (unsigned base = lo; base < hi; base += AHEAD * 16u) {
        uint4 v[AHEAD];
#pragma unroll
        for (unsigned u = 0; u < AHEAD; ++u) {
            const unsigned i = base + u * 16u + quad;
            v[u] = uint4{0u, 0u, 0u, 0u};
            if (i < hi) v[u] = src4[(size_t)i * 4u + piece];
        }
#pragma unroll
        for (unsigned u = 0; u < AHEAD; ++u) {
            const unsigned i = base + u * 16u + quad;
            const bool valid = i < hi;
            const unsigned item_key = quad_key(wf2_sort_key_part(piece, v[u], ex, ey, ez));  // (every lane: a shuffle)
            const unsigned key = valid ? item_key : 0xffffffffu;
            unsigned rank = 0u, total = 0u, first_quad = quad;
#pragma unroll
            for (unsigned j = 0; j < 16u; ++j) {
                const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, (int)(j * 4u));
                const bool same = kj == key;
                if (same && j < quad) ++rank;
                if (same && j < first_quad) first_quad = j;
                if (same) ++total;
            }
            unsigned to = 0u;
            if (valid && piece == 0u && rank == 0u) to = atomicAdd(&cells[key * NW + wave], total);
            __builtin_amdgcn_wave_barrier();  // (nothing on the device; tests/emu: the next step's atomics come after this one's)
            to = (unsigned)__shfl((int)to, (int)(first_quad * 4u), 64) + rank;
            if (valid) dst4[(size_t)to * 4u + piece] = v[u];
        }
    }
#ifdef RT_HOST_EMU  // (tests/emu: the dense queue is the slice sorted by key, equal keys in the slice's order)
    __syncthreads();
    if (tid == 0u) {
        const WfItem* src = reinterpret_cast<const WfItem*>(src4);
        const WfItem* dst = reinterpret_cast<const WfItem*>(dst4);
        unsigned at = 0u;
        for (unsigned k = 0; k < KEYS; ++k)
            for (unsigned i = 0; i < n; ++i)
                if (wf2_sort_key(src[i], ex, ey, ez) == k && std::memcmp(&dst[at++], &src[i], sizeof(WfItem)) != 0) std::abort();
        if (at != n) std::abort();
    }
#endif
}

}  // namespace rtk
