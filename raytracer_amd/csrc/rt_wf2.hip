// rt_wf2.hip -- host side of the kernel pipeline for large scenes (rt_wf2.h): the 4-wide hierarchy built from the
// scene's skip-link nodes, the workspace, and the launch sequence of one frame.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <thread>

#include "rt_host_internal.h"
#include "rt_wf2.h"

using namespace rtd;
using namespace rtk;

#ifndef RT_WF2_SIDE_BLOCKS
#define RT_WF2_SIDE_BLOCKS 0  // feelers on the second stream: one-wave blocks per CU (0: as many as fit; RT_HIP_WF2_SIDE_BLOCKS)
#endif

struct Wf2 {
    size_t off_nodes = 0, off_pre = 0, off_accel = 0;  // the hierarchy, the leaf records and the accel record inside the scene's block of tables
    bool has_pre = false;
    Wf2Accel accel{};
    uint32_t n_nodes = 0, stack_need = 0;
    // workspace, grown to the largest batch asked for; launches on one scene are ordered by the caller
    Wf2Space sp{};
    void* blocks[16] = {};
    size_t batch_pixels = 0, q_items = 0, node_items = 0;
    int trace_blocks_per_cu[2][4] = {};  // [stats][mode], from the occupancy query
    int packet_blocks_per_cu[2][2] = {};  // wf2_packet_kernel: [stats][camera rays, feelers]
    unsigned packet = 0;                  // which launches the wave-walk kernel takes: 1 camera rays, 2 the first generation's feelers
    unsigned inlane_below = 0;            // wf2_spawn_kernel: generations with fewer hits keep their subtrees in-lane (0: never)
    int n_cus = 0;
    int side_blocks = 0;  // feelers on the second stream: one-wave blocks per CU (0: as many as fit)
    // a generation's containers walks run beside its feelers (neither reads what the other writes): a stream of the
    // scene's own, forked from and joined to the caller's by these events (created with the scene: a launch creates nothing)
    hipStream_t side = nullptr;
    hipEvent_t prepared[RT_WF_LEVELS + 1] = {}, spawned[RT_WF_LEVELS + 1] = {}, lit[RT_WF_LEVELS + 1] = {};
};

namespace {

// ---- 4-wide hierarchy ------------------------------------------------------------------------------------------
struct TNode {
    float lo[3], hi[3];
    int32_t shape;  // leaf: shape index; inner: -1
    std::vector<uint32_t> kids;
};

// The skip-link nodes [k, ...) up to `after` as a sequence of sibling subtrees (an inner node's children are the
// sequence from k + 1 to its miss link).  False when the links do not nest that way.
bool parse_seq(const RtSceneDesc& d, int32_t k, int32_t after, std::vector<TNode>& tree, std::vector<uint32_t>& out, int depth) {
    if (depth > 4096) return false;
    while (k != after) {
        if (k < 0 || k >= (int32_t)d.n_bvh_nodes) return false;
        if (after >= 0 && k > after) return false;
        const RtBvhNode& nd = d.bvh_nodes[k];
        TNode t;
        for (int a = 0; a < 3; ++a) t.lo[a] = nd.bmin[a], t.hi[a] = nd.bmax[a];
        t.shape = nd.shape;
        const uint32_t self = (uint32_t)tree.size();
        tree.push_back(t);
        if (nd.shape < 0) {
            std::vector<uint32_t> kids;
            if (!parse_seq(d, k + 1, nd.miss, tree, kids, depth + 1) || kids.empty()) return false;
            tree[self].kids = std::move(kids);
        }
        out.push_back(self);
        k = nd.miss;
    }
    return true;
}

double half_area(const TNode& t) {
    const double x = (double)t.hi[0] - t.lo[0], y = (double)t.hi[1] - t.lo[1], z = (double)t.hi[2] - t.lo[2];
    if (!(x >= 0.0 && y >= 0.0 && z >= 0.0)) return 0.0;
    const double a = x * y + y * z + z * x;
    return std::isfinite(a) ? a : INFINITY;
}

struct Builder {
    std::vector<TNode>& tree;
    std::vector<Bvh4Node> nodes;
    uint32_t stack_need = 0;
    bool ok = true;  // false: some box cannot be put on a node's 8-bit grid (not finite): the scene keeps the older kernels

    // at most four subtrees out of `list`: inner ones are opened up (largest first) while the result still fits four;
    // longer lists are cut into four runs, each under a synthetic inner node
    std::vector<uint32_t> four(std::vector<uint32_t> list) {
        while (list.size() > 4) {
            std::vector<uint32_t> grouped;
            const size_t n = list.size();
            for (size_t g = 0; g < 4; ++g) {
                const size_t a = n * g / 4, b = n * (g + 1) / 4;
                if (b - a == 1) {
                    grouped.push_back(list[a]);
                    continue;
                }
                TNode t;
                for (int ax = 0; ax < 3; ++ax) t.lo[ax] = INFINITY, t.hi[ax] = -INFINITY;
                t.shape = -1;
                for (size_t i = a; i < b; ++i) {
                    for (int ax = 0; ax < 3; ++ax) {
                        t.lo[ax] = std::fmin(t.lo[ax], tree[list[i]].lo[ax]);
                        t.hi[ax] = std::fmax(t.hi[ax], tree[list[i]].hi[ax]);
                    }
                    t.kids.push_back(list[i]);
                }
                grouped.push_back((uint32_t)tree.size());
                tree.push_back(t);
            }
            list = grouped;
        }
        for (;;) {
            int best = -1;
            double best_area = -1.0;
            for (size_t i = 0; i < list.size(); ++i) {
                const TNode& t = tree[list[i]];
                if (t.shape >= 0 || list.size() - 1 + t.kids.size() > 4) continue;
                const double a = half_area(t);
                if (a > best_area) best_area = a, best = (int)i;
            }
            if (best < 0) break;
            const std::vector<uint32_t> kids = tree[list[(size_t)best]].kids;
            list.erase(list.begin() + best);
            list.insert(list.end(), kids.begin(), kids.end());
        }
        return list;
    }

    // emits the node whose children are `list` (<= 4 after four()); returns its index and the stack entries a walk
    // below it can need
    uint32_t emit(const std::vector<uint32_t>& list_in, uint32_t& need) {
        const std::vector<uint32_t> list = four(list_in);
        const uint32_t self = (uint32_t)nodes.size();
        nodes.emplace_back();
        Bvh4Node nd;
        std::memset(&nd, 0, sizeof nd);
        uint32_t deepest = 0;
        // the grid: origin = the children's common lower corner, step = the power of two that brings the common upper
        // corner within 255 steps
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t id : list)
            for (int a = 0; a < 3; ++a) {
                if (!std::isfinite(tree[id].lo[a]) || !std::isfinite(tree[id].hi[a]) || tree[id].lo[a] > tree[id].hi[a]) ok = false;
                lo[a] = std::fmin(lo[a], tree[id].lo[a]), hi[a] = std::fmax(hi[a], tree[id].hi[a]);
            }
        double step[3];
        for (int a = 0; a < 3; ++a) {
            nd.org[a] = (float)lo[a];  // exact: the corner is one of the f32 box values
            int e = 0;
            std::frexp(std::fmax((hi[a] - lo[a]) / 255.0, 1e-37), &e);  // (hi - lo) / 255 <= 2^e
            int biased = e + 127;
            if (biased < 1) biased = 1;
            if (biased > 254) ok = false, biased = 254;
            nd.exp[a] = (uint8_t)biased;
            step[a] = std::ldexp(1.0, biased - 127);
        }
        for (int c = 0; c < 4; ++c) nd.child[c] = (int32_t)RT_BVH4_DONE;
        for (size_t c = 0; c < list.size(); ++c) {
            const TNode t = tree[list[c]];
            for (int a = 0; a < 3; ++a) {
                const double ql = std::floor(((double)t.lo[a] - lo[a]) / step[a]), qh = std::ceil(((double)t.hi[a] - lo[a]) / step[a]);
                if (!(ql >= 0.0 && qh <= 255.0 && ql <= qh)) ok = false;
                nd.qlo[a][c] = (uint8_t)std::fmax(0.0, std::fmin(255.0, ql));
                nd.qhi[a][c] = (uint8_t)std::fmax(0.0, std::fmin(255.0, qh));
            }
            if (t.shape >= 0) {
                nd.child[c] = (int32_t)(RT_BVH4_LEAF | (uint32_t)t.shape);
            } else {
                uint32_t below = 0;
                nd.child[c] = (int32_t)emit(t.kids, below);
                deepest = std::max(deepest, below);
            }
        }
        nodes[self] = nd;
        need = (uint32_t)list.size() - 1u + deepest;
        return self;
    }
};

// A copy of the subtree `id` holding only leaves for which `keep` says yes (boxes re-fitted); -1 when nothing is left.
template <class F>
int filtered(std::vector<TNode>& tree, uint32_t id, F&& keep) {
    if (tree[id].shape >= 0) return keep(tree[id].shape) ? (int)id : -1;
    TNode t;
    for (int a = 0; a < 3; ++a) t.lo[a] = INFINITY, t.hi[a] = -INFINITY;
    t.shape = -1;
    const std::vector<uint32_t> kids = tree[id].kids;
    for (uint32_t k : kids) {
        const int f = filtered(tree, k, keep);
        if (f < 0) continue;
        t.kids.push_back((uint32_t)f);
        for (int a = 0; a < 3; ++a) {
            t.lo[a] = std::fmin(t.lo[a], tree[(size_t)f].lo[a]);
            t.hi[a] = std::fmax(t.hi[a], tree[(size_t)f].hi[a]);
        }
    }
    if (t.kids.empty()) return -1;
    if (t.kids.size() == 1) return (int)t.kids[0];
    tree.push_back(t);
    return (int)tree.size() - 1;
}

// ---- the f32 test ahead of a leaf (LeafPre) ---------------------------------------------------------------------
// C = inverse(primitive) * inverse(parent) * ... * inverse(outermost group) in f64 (what the reference applies one matrix
// at a time), rounded to f32, with the constants that bound what the f32 evaluation can be off by in OBJECT space:
//   position: every term of C32 * o32 + c32 and of the shift along the line rounds by <= 2^-23 of its size, and the sizes
//             add up to at most (sum_j |C_ij| + |c_i| / E) * (max |o| + E) with E = the scene's extent: k1 = 2^-17 of that
//             row norm (16 roundings' worth);
//   direction: |du| <= 2^-22 |C| |d|, which moves the line by at most rho * |du| / |u| <= rho * 2^-22 * cond(C) * sqrt(3) over
//             the rho = the shape's bounding radius that matters: into k2, with a base of 1e-3 that also covers the f64
//             side's own rounding for origins up to ~1e5 object sizes away (cf. ABox::pad in the flattener).
// Shapes whose bound is not small and finite in object space, or whose constants come out large, get no test.
LeafPre leaf_pre_of(const RtSceneDesc& d, uint32_t shape, double extent) {
    LeafPre out;
    std::memset(&out, 0, sizeof out);
    out.kind = RT_PRE_NONE;
    const RtShape& sh = d.shapes[shape];
    const unsigned kind = RT_KIND(sh.kind_flags);
    double rho = 1.7320508075688772;  // cube; sphere 1
    double ymid = 0.0, yhalf = 1.0, radius = 1.0;
    if (kind == RT_CYLINDER || kind == RT_CONE) {
        const RtConic& c = d.conics[sh.param];
        if (!(c.minimum <= c.maximum)) return out;
        const bool finite = std::isfinite(c.minimum) && std::isfinite(c.maximum);
        if (kind == RT_CONE && !finite) return out;  // its radius grows with |y|
        radius = kind == RT_CONE ? std::fmax(std::fabs(c.minimum), std::fabs(c.maximum)) : 1.0;
        ymid = finite ? 0.5 * (c.minimum + c.maximum) : 0.0;
        yhalf = finite ? 0.5 * (c.maximum - c.minimum) : INFINITY;
        rho = std::sqrt(radius * radius + (finite ? std::fmax(c.minimum * c.minimum, c.maximum * c.maximum) : 0.0));
        if (!finite) rho = 64.0;  // (only the radial part of the test is finite; the direction term below is then void: no test)
        if (!finite) return out;
    } else if (kind != RT_SPHERE && kind != RT_CUBE) {
        return out;  // planes and triangles: always the f64 test
    }
    if (!(rho <= 64.0)) return out;
    // compose, innermost last: p_obj = inv_prim * (inv_parent * (... * (inv_outermost * p)))
    double C[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    int32_t chain[64];  // (validate() holds the nesting far below this)
    size_t n_chain = 0;
    for (int32_t p = sh.parent; p >= 0; p = d.shapes[p].parent) {
        if (n_chain == 64) return out;
        chain[n_chain++] = p;
    }
    auto apply = [&](const double* m) {  // C = M * C
        double r[3][4];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                double v = j == 3 ? m[i * 4 + 3] : 0.0;
                for (int k = 0; k < 3; ++k) v += m[i * 4 + k] * C[k][j];
                r[i][j] = v;
            }
        std::memcpy(C, r, sizeof r);
    };
    for (size_t q = n_chain; q-- > 0;) apply(d.shapes[chain[q]].inv);
    apply(sh.inv);
    // norms of the linear part and of its inverse
    const double a = C[0][0], b = C[0][1], c = C[0][2], e = C[1][0], f = C[1][1], g = C[1][2], h = C[2][0], i2 = C[2][1], j2 = C[2][2];
    const double det = a * (f * j2 - g * i2) - b * (e * j2 - g * h) + c * (e * i2 - f * h);
    if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) return out;
    const double inv3[3][3] = {{(f * j2 - g * i2) / det, (c * i2 - b * j2) / det, (b * g - c * f) / det},
                               {(g * h - e * j2) / det, (a * j2 - c * h) / det, (c * e - a * g) / det},
                               {(e * i2 - f * h) / det, (b * h - a * i2) / det, (a * f - b * e) / det}};
    double norm = 0.0, norm_inv = 0.0, row = 0.0;
    const double E = std::fmax(extent, 1.0);
    for (int r = 0; r < 3; ++r) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < 3; ++k) s1 += std::fabs(C[r][k]), s2 += std::fabs(inv3[r][k]);
        norm = std::fmax(norm, s1), norm_inv = std::fmax(norm_inv, s2);
        row = std::fmax(row, s1 + std::fabs(C[r][3]) / E);
    }
    const double k1 = row / 131072.0;                                                  // 2^-17
    const double k2 = 1e-3 + rho * norm * norm_inv * 3.0 * 1.7320508075688772 / 4194304.0;  // 2^-22, infinity norms: sqrt(3) each way
    if (!(k1 * 2.0 * E <= 0.125) || !(k2 <= 0.125) || !std::isfinite(k1) || !std::isfinite(k2)) return out;  // too loose to be worth it
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) {
            out.m[r * 4 + k] = (float)C[r][k];
            if (!std::isfinite(out.m[r * 4 + k])) return out;
        }
    out.k1 = (float)(k1 * 1.0001), out.k2 = (float)(k2 * 1.0001);
    out.ymid = (float)ymid, out.yhalf = (float)(yhalf * 1.0001), out.radius = (float)(radius * 1.0001);
    out.kind = kind;
    return out;
}

template <class K>
int occupancy(K kernel, int* out) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(out, kernel, 64, 0));
    if (*out < 1) *out = 1;
    return RT_OK;
}

#ifndef RT_WF2_QUEUE_PER_PIXEL
#define RT_WF2_QUEUE_PER_PIXEL 0.75  // queue slots of one generation and node slots of a batch, per pixel of the batch;
#define RT_WF2_NODES_PER_PIXEL 1.0   // past that a hit keeps its subtree to itself (slower, same result)
#endif
#ifndef RT_WF2_BUDGET_MB
// workspace bound: larger frames are rendered in batches of rows (RT_HIP_WF_BUDGET_MB overrides).  Sized for a 288 GB
// device: a 7680 x 4320 frame (18 GB of workspace) goes through in one batch, 94 ms against 106 ms in four; never more
// than half of what the device has free when the workspace is first laid out.
#define RT_WF2_BUDGET_MB 24576
#endif

// Generations with fewer hits than this finish in-lane (wf2_spawn_kernel); RT_HIP_WF2_INLANE_BELOW in the environment: studies.
// Measured on rank 0's share of the 4K frame of the 10k scene (tools/gpu_shard_time.py), threshold 0 / 131072 / 262144 / 524288:
// 1/4 share 6.17 / 6.16 / 6.14 / 10.77 ms, 1/8 share 4.11 / 4.12 / 7.30 / 10.26 ms -- wherever it triggers it loses: a lane that
// keeps a subtree runs color_at's frame stack at 256 VGPRs, two waves per SIMD, its rays one after the other.  Off.
#ifndef RT_WF2_INLANE_BELOW
#define RT_WF2_INLANE_BELOW 0u
#endif

size_t budget_bytes() {
    static const size_t v = [] {
        const char* e = std::getenv("RT_HIP_WF_BUDGET_MB");
        const long mb = e && *e ? std::atol(e) : 0;
        size_t bytes = (size_t)(mb > 0 ? mb : RT_WF2_BUDGET_MB) << 20;
        size_t free_b = 0, total_b = 0;
        if (mb <= 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 < bytes) bytes = free_b / 2;
        if (bytes < ((size_t)64 << 20)) bytes = (size_t)64 << 20;
        return bytes;
    }();
    return v;
}

struct Plan {
    size_t q_cap, node_cap;  // per shard
    size_t qd_cap, ctx_cap, bytes;
};
Plan plan_for(size_t items, size_t q_items, size_t node_items, uint32_t n_lights) {
    Plan p;
    const size_t qi = q_items ? q_items : (size_t)(RT_WF2_QUEUE_PER_PIXEL * (double)items);
    const size_t ni = node_items ? node_items : (size_t)(RT_WF2_NODES_PER_PIXEL * (double)items);
    p.q_cap = (qi + RT_WF_SHARDS - 1) / RT_WF_SHARDS;
    p.node_cap = (ni + RT_WF_SHARDS - 1) / RT_WF_SHARDS;
    if (!q_items && p.q_cap < 128) p.q_cap = 128;
    if (!node_items && p.node_cap < 64) p.node_cap = 64;
    if (p.node_cap == 0) p.node_cap = 1;
    p.q_cap = (p.q_cap + RT_WF_BINS - 1) / RT_WF_BINS * RT_WF_BINS;  // a slice is split evenly over the direction octants
    if (p.q_cap == 0) p.q_cap = RT_WF_BINS;
    p.qd_cap = p.q_cap * RT_WF_SHARDS;
    p.ctx_cap = std::max(items, p.qd_cap);
    p.bytes = 2 * p.qd_cap * sizeof(WfItem) + p.node_cap * RT_WF_SHARDS * sizeof(WfNode) + p.ctx_cap * sizeof(Wf2Hit) +
              2 * p.ctx_cap * (CX_DOUBLES * sizeof(double) + CU_UINTS * sizeof(uint32_t) + sizeof(uint32_t) + n_lights) + p.ctx_cap * sizeof(uint32_t) +
              (size_t)RT_WF2_PARTS * 256u * sizeof(uint32_t) + sizeof(Wf2Counters) + 16u * 256u;
    return p;
}

// The workspace of a plan laid out in one block of device memory (256-byte aligned pieces); false when it does not fit.
bool carve_space(Wf2Space& s, unsigned char* base, size_t bytes, const Plan& p, uint32_t n_lights, size_t* used = nullptr) {
    size_t at = 0;
    bool ok = true;
    auto take = [&](auto** ptr, size_t n) {
        at = (at + 255u) & ~(size_t)255u;
        if (at + n > bytes) ok = false;
        *ptr = base ? reinterpret_cast<std::remove_reference_t<decltype(**ptr)>*>(base + at) : nullptr;  // (no base: measuring)
        at += n;
    };
    std::memset(&s, 0, sizeof s);
    take(&s.ws.queue[0], p.qd_cap * sizeof(WfItem));
    s.ws.queue[1] = s.ws.queue[0];
    take(&s.ws.nodes, p.node_cap * RT_WF_SHARDS * sizeof(WfNode));
    take(&s.ws.shards, RT_WF_SHARDS * sizeof(WfShard));
    take(&s.qd, p.qd_cap * sizeof(WfItem));
    take(&s.hits, p.ctx_cap * sizeof(Wf2Hit));
    take(&s.ctx_d, p.ctx_cap * CX_DOUBLES * sizeof(double));
    take(&s.ctx_u, p.ctx_cap * CU_UINTS * sizeof(uint32_t));
    take(&s.occ, p.ctx_cap * (n_lights ? n_lights : 1u));
    take(&s.creq, (size_t)RT_WF2_PARTS * wf2_creq_part_cap(p.ctx_cap) * sizeof(uint32_t));
    take(&s.overflow, p.ctx_cap * sizeof(uint32_t));
    take(&s.alt_ctx_d, p.ctx_cap * CX_DOUBLES * sizeof(double));
    take(&s.alt_ctx_u, p.ctx_cap * CU_UINTS * sizeof(uint32_t));
    take(&s.alt_occ, p.ctx_cap * (n_lights ? n_lights : 1u));
    take(&s.alt_overflow, p.ctx_cap * sizeof(uint32_t));
    take(&s.cnt, sizeof(Wf2Counters));
    s.ws.q_cap = (unsigned)p.q_cap, s.ws.node_cap = (unsigned)p.node_cap;
    s.qd_cap = (unsigned)p.qd_cap, s.ctx_cap = (unsigned)p.ctx_cap;
    s.creq_part_cap = wf2_creq_part_cap(p.ctx_cap);
    if (used) *used = at;
    return ok;
}

void free_space(Wf2* w) {
    for (void*& p : w->blocks) {
        if (p) rth::block_free(p);
        p = nullptr;
    }
    std::memset(&w->sp, 0, sizeof w->sp);
    w->batch_pixels = 0;
}

// One block from the library's cache of device blocks (rt_hostmem.hip), laid out by carve_space: a scene created, rendered
// once and destroyed -- what rt_hip_render does per call -- takes the workspace the scene before it gave back instead of
// fifteen hipMalloc / hipFree pairs (the 8K frame's 21 GB: 450 of the call's 530 ms).
int ensure_space(Wf2* w, size_t items, size_t q_items, size_t node_items, uint32_t n_lights) {
    if (w->sp.cnt && w->batch_pixels >= items && w->q_items == q_items && w->node_items == node_items) return RT_OK;
    if (w->sp.cnt) {
        HIP_TRY(hipDeviceSynchronize());  // grow: the old space is idle by contract once the device is
        free_space(w);
    }
    const Plan p = plan_for(items, q_items, node_items, n_lights);
    if (p.node_cap * RT_WF_SHARDS >= 0x40000000ull || p.qd_cap >= 0x80000000ull || p.ctx_cap >= 0x80000000ull) return RT_ERR_INVALID_ARG;
    size_t bytes = 0;
    {
        Wf2Space measure;
        (void)carve_space(measure, nullptr, ~(size_t)0, p, n_lights, &bytes);  // (what the layout comes to, alignment included)
    }
    void* block = nullptr;
    const int rc = rth::block_alloc(&block, bytes);
    if (rc != RT_OK) return rc;
    w->blocks[0] = block;
    if (!carve_space(w->sp, static_cast<unsigned char*>(block), bytes, p, n_lights)) {
        free_space(w);
        return RT_ERR_INVALID_ARG;  // (cannot happen: plan_for sized the block)
    }
    w->sp.ws.cam = nullptr;
    w->batch_pixels = items, w->q_items = q_items, w->node_items = node_items;
    return RT_OK;
}

}  // namespace

int rth::wf2_create(const RtSceneDesc& d, uint32_t scene_flags, Arena& arena, Wf2** out) {
    *out = nullptr;
    // What the pipeline covers: a world-level hierarchy whose leaves are primitives (nested in groups or not), no CSG
    // node and no shadowless primitive anywhere (those need the sorted world list per ray: the older kernels have it).
    if (d.world_bvh_root < 0 || (scene_flags & RT_SCENE_HAS_SHADOWLESS) || d.n_csgs != 0 || d.n_lights == 0) return RT_OK;
    if (d.n_shapes >= (1u << RT_BVH4_SHAPE_BITS)) return RT_OK;  // (a leaf code's upper bits: flags, and the owner lane in the leaf exchange)
    const bool timing = std::getenv("RT_HIP_CREATE_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  wf2_create: %-32s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<TNode> tree;
    std::vector<uint32_t> top;
    if (!parse_seq(d, d.world_bvh_root, -1, tree, top, 0) || top.empty()) return RT_OK;
    stage("skip links -> tree");
    for (const TNode& t : tree)
        if (t.shape >= 0 && RT_KIND(d.shapes[t.shape].kind_flags) > RT_SMOOTH_TRIANGLE) return RT_OK;
    Wf2* w = new Wf2();
    Builder b{tree, {}, 0};
    {  // leaves with unbounded boxes (the flattener puts them after the tree) stay outside the hierarchy: every ray tests them
        std::vector<uint32_t> bounded;
        for (uint32_t id : top) {
            const TNode& t = tree[id];
            bool finite = true;
            for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(t.lo[a]) && std::isfinite(t.hi[a]);
            if (finite) {
                bounded.push_back(id);
            } else if (t.shape >= 0 && w->accel.n_always < 8u) {
                w->accel.always[w->accel.n_always++] = (uint32_t)t.shape;
            } else {  // an unbounded inner node, or too many unbounded leaves
                delete w;
                return RT_OK;
            }
        }
        top = bounded;
    }
    while (top.size() == 1 && tree[top[0]].shape < 0) top = tree[top[0]].kids;
    if (top.empty()) {
        w->accel.root = RT_BVH4_DONE;
        w->stack_need = 1;
    } else if (top.size() == 1) {
        w->accel.root = RT_BVH4_LEAF | (uint32_t)tree[top[0]].shape;
        w->stack_need = 1;
    } else {
        uint32_t need = 0;
        w->accel.root = b.emit(top, need);
        w->stack_need = need + 1u;  // + the sentinel
    }
    stage("4-wide nodes");
    {  // the hierarchy over the open primitives (see WF2_CONTAINERS in rt_wf2.h)
        auto open = [&](int32_t shape) {
            const RtShape& sh = d.shapes[shape];
            const unsigned kind = RT_KIND(sh.kind_flags);
            if (kind == RT_SPHERE || kind == RT_CUBE) return false;
            if (kind == RT_CYLINDER || kind == RT_CONE) {
                const RtConic& c = d.conics[sh.param];
                return !((sh.kind_flags & RT_FLAG_CLOSED) && std::isfinite(c.minimum) && std::isfinite(c.maximum));
            }
            return true;
        };
        std::vector<uint32_t> open_top;
        for (uint32_t id : top) {
            const int f = filtered(tree, id, open);
            if (f >= 0) open_top.push_back((uint32_t)f);
        }
        while (open_top.size() == 1 && tree[open_top[0]].shape < 0) open_top = tree[open_top[0]].kids;
        if (open_top.empty()) {
            w->accel.root_open = RT_BVH4_DONE;
        } else if (open_top.size() == 1) {
            w->accel.root_open = RT_BVH4_LEAF | (uint32_t)tree[open_top[0]].shape;
        } else {
            uint32_t need = 0;
            w->accel.root_open = b.emit(open_top, need);
            w->stack_need = std::max(w->stack_need, need + 1u);
        }
    }
    stage("open primitives' hierarchy");
    w->stack_need += w->accel.n_always;
    if (w->stack_need > RT_WF2_MAX_STACK || !b.ok) {
        delete w;
        return RT_OK;
    }
    for (int a = 0; a < 3; ++a) w->accel.extent[a] = 0.f;
    for (const TNode& t : tree)
        for (int a = 0; a < 3; ++a) {
            if (std::isfinite(t.lo[a])) w->accel.extent[a] = std::fmax(w->accel.extent[a], std::fabs(t.lo[a]));
            if (std::isfinite(t.hi[a])) w->accel.extent[a] = std::fmax(w->accel.extent[a], std::fabs(t.hi[a]));
        }
    w->accel.has_cands = d.dir_res != 0 && d.n_dir_cands != 0 ? 1u : 0u;
    w->accel.extent_max = std::fmax(w->accel.extent[0], std::fmax(w->accel.extent[1], w->accel.extent[2]));
    std::vector<LeafPre> pre(d.n_shapes);
    for (LeafPre& p : pre) std::memset(&p, 0, sizeof p), p.kind = RT_PRE_NONE, p.anc[0] = RT_PRE_DEEP, p.anc[1] = RT_PRE_NO_ANC;
    {  // one record per leaf, each on its own: on a few host threads when there are many
        std::vector<int32_t> leaf_shapes;
        leaf_shapes.reserve(tree.size() / 2 + 1);
        for (const TNode& t : tree)
            if (t.shape >= 0) leaf_shapes.push_back(t.shape);
        const double extent_max = w->accel.extent_max;
        auto work = [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; ++k) {
                const int32_t shape = leaf_shapes[k];
                LeafPre& p = pre[(size_t)shape];
                p = leaf_pre_of(d, (uint32_t)shape, extent_max);
                // the enclosing groups, outermost first, when there are at most two
                int32_t chain[3];
                size_t n_chain = 0;
                for (int32_t a = d.shapes[shape].parent; a >= 0 && n_chain < 3; a = d.shapes[a].parent) chain[n_chain++] = a;
                p.anc[0] = p.anc[1] = RT_PRE_NO_ANC;
                if (n_chain > 2) p.anc[0] = RT_PRE_DEEP;
                else for (size_t q = 0; q < n_chain; ++q) p.anc[q] = (uint32_t)chain[n_chain - 1 - q];
            }
        };
        unsigned n_threads = leaf_shapes.size() >= 4096 ? std::min(4u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (n_threads <= 1u) {
            work(0, leaf_shapes.size());
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < n_threads; ++t)
                pool.emplace_back(work, leaf_shapes.size() * t / n_threads, leaf_shapes.size() * (t + 1) / n_threads);
            work(0, leaf_shapes.size() / n_threads);
            for (std::thread& th : pool) th.join();
        }
    }
    stage("leaf records");
    {  // leaves without a pre-test say so in their code (the walk then reads two of the record's five words)
        auto flag = [&](uint32_t code) {
            if ((code & RT_BVH4_LEAF) && code != RT_BVH4_DONE) {
                const uint32_t shape = RT_BVH4_SHAPE(code);
                if (pre[shape].kind == RT_PRE_NONE) code |= RT_BVH4_NOPRE;
            }
            return code;
        };
        for (Bvh4Node& nd : b.nodes)
            for (int c = 0; c < 4; ++c) nd.child[c] = (int32_t)flag((uint32_t)nd.child[c]);
        w->accel.root = flag(w->accel.root), w->accel.root_open = flag(w->accel.root_open);
        for (uint32_t k = 0; k < w->accel.n_always; ++k) w->accel.always[k] = flag(RT_BVH4_LEAF | w->accel.always[k]) & ~RT_BVH4_LEAF;
    }
    auto env_or = [](const char* name, unsigned dflt, unsigned lo, unsigned hi) {
        const char* e = std::getenv(name);
        const long v = e && *e ? std::atol(e) : 0;
        return v >= (long)lo && v <= (long)hi ? (unsigned)v : dflt;
    };
    w->accel.chunk = env_or("RT_HIP_WF2_CHUNK", RT_WF2_CHUNK, 1u, 1u << 20);
    w->accel.refill = env_or("RT_HIP_WF2_REFILL", RT_WF2_REFILL, 1u, 64u);
    w->accel.list_share = env_or("RT_HIP_WF2_LIST_SHARE", RT_WF2_LIST_SHARE, 1u, 64u);
    w->accel.slack = (1.0f / 1048576.0f) * (float)env_or("RT_HIP_WF2_SLACK", 1u, 1u, 1u << 20);
    w->accel.seg_near = env_or("RT_HIP_WF2_SEG_NEAR", 0u, 0u, 1u) ? -1e-2f : -INFINITY;
    w->n_nodes = (uint32_t)b.nodes.size();
    w->off_nodes = arena.add(b.nodes.data(), b.nodes.size() * sizeof(Bvh4Node));
    arena.reserve(sizeof(Bvh4Node));  // (one node of slack after the table)
    const char* pre_env = std::getenv("RT_HIP_WF2_PRE");
    w->has_pre = !(pre_env && *pre_env == '0');
    if (w->has_pre) {
        w->off_pre = arena.add(pre.data(), pre.size() * sizeof(LeafPre));
        arena.reserve(sizeof(LeafPre));
    }
    w->off_accel = arena.reserve(sizeof(Wf2Accel));
    auto body = [&]() -> int {
        // the device's compute units and how many one-wave blocks of each traversal kernel one of them holds: asked once per device
        static std::mutex m;
        struct PerDevice {
            bool known = false;
            int n_cus = 0;
            int blocks[2][4] = {};
            int packet[2][2] = {};
        };
        static PerDevice known[64];
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(m);
        PerDevice local;
        PerDevice& pd = dev >= 0 && dev < 64 ? known[dev] : local;
        if (!pd.known) {
            HIP_TRY(hipDeviceGetAttribute(&pd.n_cus, hipDeviceAttributeMultiprocessorCount, dev));
            int rc = RT_OK;
#define RT_OCC(ST, MODE) if (rc == RT_OK) rc = occupancy(wf2_trace_kernel<ST, MODE>, &pd.blocks[ST ? 1 : 0][MODE]);
            // (what the kernels' LDS use is sized for: 16 one-wave blocks per compute unit -- rt_wf2.h, RT_WF2_LDS_STACK_*)
            RT_OCC(false, WF2_PRIMARY) RT_OCC(false, WF2_QUEUE) RT_OCC(false, WF2_SHADOW)
            RT_OCC(true, WF2_PRIMARY) RT_OCC(true, WF2_QUEUE) RT_OCC(true, WF2_SHADOW)
            RT_OCC(false, WF2_CONTAINERS) RT_OCC(true, WF2_CONTAINERS)
#undef RT_OCC
            if (rc == RT_OK) rc = occupancy(wf2_packet_kernel<false, WF2_PRIMARY>, &pd.packet[0][0]);
            if (rc == RT_OK) rc = occupancy(wf2_packet_kernel<false, WF2_SHADOW>, &pd.packet[0][1]);
            if (rc == RT_OK) rc = occupancy(wf2_packet_kernel<true, WF2_PRIMARY>, &pd.packet[1][0]);
            if (rc == RT_OK) rc = occupancy(wf2_packet_kernel<true, WF2_SHADOW>, &pd.packet[1][1]);
            if (rc != RT_OK) return rc;
            pd.known = true;
        }
        w->n_cus = pd.n_cus;
        if (const char* e = std::getenv("RT_HIP_WF2_SIDE_BLOCKS")) w->side_blocks = std::atoi(e);
        else w->side_blocks = RT_WF2_SIDE_BLOCKS;
        std::memcpy(w->trace_blocks_per_cu, pd.blocks, sizeof pd.blocks);
        std::memcpy(w->packet_blocks_per_cu, pd.packet, sizeof pd.packet);
        // RT_HIP_WF2_PACKET (studies): bit 0 = camera rays, bit 1 = the first generation's feelers walked by the wave together
        w->packet = RT_WF2_PACKET ? 3u : 0u;
        if (const char* e = std::getenv("RT_HIP_WF2_PACKET")) w->packet = (unsigned)std::atoi(e) & 3u;
        w->inlane_below = RT_WF2_INLANE_BELOW;
        if (const char* e = std::getenv("RT_HIP_WF2_INLANE_BELOW")) w->inlane_below = (unsigned)std::atol(e);
        const char* side_env = std::getenv("RT_HIP_WF2_SIDE");
        if (!(side_env && *side_env == '0')) {
            HIP_TRY(hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking));
            for (int g = 0; g <= RT_WF_LEVELS; ++g) {
                HIP_TRY(hipEventCreateWithFlags(&w->prepared[g], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&w->spawned[g], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&w->lit[g], hipEventDisableTiming));
            }
        }
        return RT_OK;
    };
    stage("into the arena");
    const int rc = body();
    stage("occupancy, stream, events");
    if (rc != RT_OK) {
        rth::wf2_destroy(w);
        return rc;
    }
    *out = w;
    return RT_OK;
}

void rth::wf2_destroy(Wf2* w) {
    if (!w) return;
    free_space(w);
    if (w->side) (void)hipStreamDestroy(w->side);
    for (int g = 0; g <= RT_WF_LEVELS; ++g)
        for (hipEvent_t e : {w->prepared[g], w->spawned[g], w->lit[g]})
            if (e) (void)hipEventDestroy(e);
    delete w;
}

const Wf2Accel* rth::wf2_bind(Wf2* w, Arena& arena, unsigned char* base) {
    if (!w) return nullptr;
    w->accel.nodes = reinterpret_cast<const Bvh4Node*>(base + w->off_nodes);
    w->accel.pre = w->has_pre ? reinterpret_cast<const LeafPre*>(base + w->off_pre) : nullptr;
    std::memcpy(arena.host.data() + w->off_accel, &w->accel, sizeof(Wf2Accel));  // the copy trace4 reads (rt_bvh4.h)
    return reinterpret_cast<const Wf2Accel*>(base + w->off_accel);
}

namespace {

// The launch sequence of one frame (or row-band shard) with the workspace `sp`, `batch_rows` rows at a time.
int run_pipeline(const RtScene* scene, const Wf2* w, const Wf2Space& sp, size_t batch_rows, const RtCameraDesc* cam, uint32_t depth,
                 const RowMap& rows_in, double* d_out, hipStream_t stream, RtStats* d_stats) {
    const SceneDev& S = scene->dev;
    const Wf2Accel A = w->accel;
    const bool cands_pass = A.has_cands && !RT_WF2_CANDS_INLINE;  // the direction candidates in a kernel of their own
    unsigned* status = scene->d_status;
#ifdef RT_HOST_EMU  // (tests/emu: every block costs host threads; the kernels stride over their items whatever the grid)
    const dim3 wide(4), threads(256), overflow_grid(4);
#else
    const dim3 wide(2048), threads(256), overflow_grid(4096);
#endif
    // grids of the grid-stride shading kernels (RT_HIP_WF2_{SPAWN,PREPARE,LIGHT}_BLOCKS: studies).  Measured, 4K / 8K frame of
    // the 10k scene: spawn at 256 / 512 / 1024 / 2048 / 4096 / 8192 / 16384 blocks 17.00 / 16.68 / 16.55 / 16.43 / 16.35 / 16.35 /
    // 16.36 ms, 8K 52.9 (512) / 52.0 (2048) / 51.45 (8192); then prepare at 1024 / 2048 / 8192: 16.36 / 16.35 / 16.26, light at 512 /
    // 1024 / 2048 / 8192: 16.54 / 16.43 / 16.35 / 16.26 (8K: 51.4 -> 51.2 either way)
    auto blocks_of = [](const char* name, unsigned dflt) {
        const char* e = std::getenv(name);
        const long v = e && *e ? std::atol(e) : 0;
        return dim3(v >= 1 && v <= 65536 ? (unsigned)v : dflt);
    };
#ifdef RT_HOST_EMU
    const dim3 spawn_grid = wide, prepare_grid = wide, light_grid = wide;
#else
    const dim3 spawn_grid = blocks_of("RT_HIP_WF2_SPAWN_BLOCKS", 8192u), prepare_grid = blocks_of("RT_HIP_WF2_PREPARE_BLOCKS", 8192u),
               light_grid = blocks_of("RT_HIP_WF2_LIGHT_BLOCKS", 8192u);
#endif
    auto go = [&](auto stats_tag) {
        constexpr bool ST = decltype(stats_tag)::value;
        dim3 tgrid[4] = {dim3((unsigned)(w->n_cus * w->trace_blocks_per_cu[ST][0])), dim3((unsigned)(w->n_cus * w->trace_blocks_per_cu[ST][1])),
                         dim3((unsigned)(w->n_cus * w->trace_blocks_per_cu[ST][2])), dim3((unsigned)(w->n_cus * w->trace_blocks_per_cu[ST][3]))};
        // The feelers' waves are persistent and at 128 VGPRs four of them fill a SIMD's register file: with every slot taken,
        // whatever the caller's stream launches next waits for feelers to run dry.  On two streams they take `side_blocks` of
        // the slots per CU and leave the rest to the critical path.
        if (w->side != nullptr && w->side_blocks > 0 && w->side_blocks < w->trace_blocks_per_cu[ST][2])
            tgrid[2] = dim3((unsigned)(w->n_cus * w->side_blocks));
        // children of a stretch of 2^shift consecutive hits share a shard (picked by a hash of the stretch): large enough to
        // keep neighbours together, small enough that every shard gets MANY stretches -- a shard's slice of the queue holds
        // 1.5 x its fair share, and with 8 stretches per shard (a 1/8 share of a 4K frame under the old rule) the hash left
        // some shards with twice theirs: their hits overflowed into wf2_overflow_kernel, 5.8 ms against 2.5 for the share
        unsigned stretch_shift = 6;
        while (stretch_shift < RT_WF_CHUNK_SHIFT && ((size_t)sp.ctx_cap >> (stretch_shift + 1)) >= 64u * RT_WF_SHARDS) ++stretch_shift;
        for (uint32_t row0 = 0; row0 < rows_in.n_local_rows; row0 += (uint32_t)batch_rows) {
            RowMap rows = rows_in;
            rows.row0 = row0;
            rows.batch_rows = (uint32_t)std::min<size_t>(batch_rows, rows_in.n_local_rows - row0);
            (void)hipMemsetAsync(sp.ws.shards, 0, RT_WF_SHARDS * sizeof(WfShard), stream);
            (void)hipMemsetAsync(sp.cnt, 0, sizeof(Wf2Counters), stream);
            // Generation g works in buffer set g & 1 (contexts, feeler answers, overflow list).  The caller's stream carries
            // what the next generation waits for -- closest hits, contexts, containers walks, children, the queue made
            // dense --, the scene's second stream the generation's feelers, its lighting and its overflow subtrees.
            const bool two = w->side != nullptr;
            hipStream_t side = two ? w->side : stream;
            unsigned last = 0;
            for (unsigned g = 0; g <= depth; ++g) {
                last = g;
                Wf2Space spg = sp;
                if (g & 1u) spg.ctx_d = sp.alt_ctx_d, spg.ctx_u = sp.alt_ctx_u, spg.occ = sp.alt_occ, spg.overflow = sp.alt_overflow;
                if (two && g >= 2u) (void)hipStreamWaitEvent(stream, w->lit[g - 2u], 0);  // this set's last readers are through
                if (g == 0) {
                    if (cands_pass) hipLaunchKernelGGL((wf2_cands_kernel<ST, WF2_PRIMARY>), wide, threads, 0, stream, S, spg, *cam, rows, g, d_stats, status);
                    if (w->packet & 1u)
                        hipLaunchKernelGGL((wf2_packet_kernel<ST, WF2_PRIMARY>), dim3((unsigned)(w->n_cus * w->packet_blocks_per_cu[ST][0])), dim3(64), 0, stream, S, A, spg, *cam, rows, g, d_stats, status);
                    else
                        hipLaunchKernelGGL((wf2_trace_kernel<ST, WF2_PRIMARY>), tgrid[0], dim3(64), 0, stream, S, A, spg, *cam, rows, g, d_stats, status);
                    hipLaunchKernelGGL((wf2_prepare_kernel<ST, true>), prepare_grid, threads, 0, stream, S, A.pre, spg, *cam, rows, (int)depth, g, d_out, d_stats, status);
                } else {
                    if (cands_pass) hipLaunchKernelGGL((wf2_cands_kernel<ST, WF2_QUEUE>), wide, threads, 0, stream, S, spg, *cam, rows, g, d_stats, status);
                    hipLaunchKernelGGL((wf2_trace_kernel<ST, WF2_QUEUE>), tgrid[1], dim3(64), 0, stream, S, A, spg, *cam, rows, g, d_stats, status);
                    hipLaunchKernelGGL((wf2_prepare_kernel<ST, false>), prepare_grid, threads, 0, stream, S, A.pre, spg, *cam, rows, (int)depth, g, d_out, d_stats, status);
                }
                const bool more = g < depth && (S.flags & RT_SCENE_RECURSIVE);  // only hits with recursion budget left can have children
                if (two) {
                    (void)hipEventRecord(w->prepared[g], stream);
                    (void)hipStreamWaitEvent(side, w->prepared[g], 0);
                }
                if (cands_pass) hipLaunchKernelGGL((wf2_cands_kernel<ST, WF2_SHADOW>), wide, threads, 0, side, S, spg, *cam, rows, g, d_stats, status);
                if (g == 0u && (w->packet & 2u))
                    hipLaunchKernelGGL((wf2_packet_kernel<ST, WF2_SHADOW>), dim3((unsigned)(w->n_cus * w->packet_blocks_per_cu[ST][1])), dim3(64), 0, side, S, A, spg, *cam, rows, g, d_stats, status);
                else
                    hipLaunchKernelGGL((wf2_trace_kernel<ST, WF2_SHADOW>), tgrid[2], dim3(64), 0, side, S, A, spg, *cam, rows, g, d_stats, status);
                if (more) {
                    hipLaunchKernelGGL((wf2_trace_kernel<ST, WF2_CONTAINERS>), tgrid[3], dim3(64), 0, stream, S, A, spg, *cam, rows, g, d_stats, status);
                    hipLaunchKernelGGL((wf2_spawn_kernel<ST>), spawn_grid, threads, 0, stream, S, spg, *cam, rows, g, stretch_shift, w->inlane_below, d_stats, status);
                    if (two) {
                        (void)hipEventRecord(w->spawned[g], stream);
                        (void)hipStreamWaitEvent(side, w->spawned[g], 0);
                    }
                }
                hipLaunchKernelGGL((wf2_light_kernel<ST>), light_grid, threads, 0, side, S, spg, *cam, rows, g, d_out, d_stats, status);
                if (more) hipLaunchKernelGGL((wf2_overflow_kernel<ST>), overflow_grid, dim3(64), 0, side, S, spg, g, d_out, d_stats, status);
                if (two) (void)hipEventRecord(w->lit[g], side);
                if (!more) break;
                hipLaunchKernelGGL(wf2_scan_kernel, dim3(1), dim3(RT_WF_SHARDS), 0, stream, spg, g);
                hipLaunchKernelGGL(wf2_compact_kernel, dim3(RT_WF_SHARDS), dim3(RT_WF2_COMPACT_THREADS), 0, stream, spg, g, A.extent[0], A.extent[1], A.extent[2]);
            }
            if (two) {  // every colour is where the folding below (and the next batch, and the caller) expects it
                (void)hipStreamWaitEvent(stream, w->lit[last], 0);
                if (last >= 1u) (void)hipStreamWaitEvent(stream, w->lit[last - 1u], 0);
            }
            if (S.flags & RT_SCENE_RECURSIVE)
                for (unsigned k = depth; k-- > 0;) rth::launch_combine(sp.ws, k, d_out, stream);
        }
    };
    if (d_stats)
        go(std::true_type{});
    else
        go(std::false_type{});
    HIP_TRY(hipGetLastError());
#ifdef RT_WF2_PROFILE
    if (std::getenv("RT_HIP_WF2_PROFILE")) {  // (the last batch's counters: the workspace is reset per batch)
        Wf2Counters c;
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(&c, sp.cnt, sizeof c, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "wf2 counters: rays per generation");
        for (unsigned g = 0; g <= depth; ++g) std::fprintf(stderr, " %u", c.n_rays[g]);
        std::fprintf(stderr, " | hits whose children found no room");
        for (unsigned g = 0; g <= depth; ++g) std::fprintf(stderr, " %u", c.n_overflow[g]);
        std::fprintf(stderr, "\n");
        static const char* modes[4] = {"primary", "queue", "shadow", "containers"};
        for (int m = 0; m < 4; ++m) {
            const double tot = (double)(c.prof[m][0] + c.prof[m][1] + c.prof[m][2] + c.prof[m][3]);
            if (tot <= 0.0) continue;
            std::fprintf(stderr, "wf2 profile %-10s refill %.3f walk %.3f leaf %.3f finish %.3f | lanes in walk steps %.3f, lanes holding a leaf %.3f | Gcycles %.3f\n",
                         modes[m], c.prof[m][0] / tot, c.prof[m][1] / tot, c.prof[m][2] / tot, c.prof[m][3] / tot,
                         c.prof[m][5] ? (double)c.prof[m][4] / c.prof[m][5] : 0.0, c.prof[m][7] ? (double)c.prof[m][6] / c.prof[m][7] : 0.0, tot / 1e9);
        }
    }
#endif
    return RT_OK;
}

// items per row of pixels: tiles are 8 x 8 pixels, 64 items
size_t items_per_row(const RtCameraDesc* cam) { return (size_t)((cam->hsize + 7u) / 8u) * 8u; }

}  // namespace

int rth::wf2_render(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, const RowMap& rows_in, double* d_out,
                    hipStream_t stream, RtStats* d_stats) {
    Wf2* w = scene->wf2;
    const uint32_t n_lights = scene->dev.n_lights;
    // rows per batch: whole 8-row tiles, as many as the workspace budget holds
    const size_t row_items = items_per_row(cam);
    size_t batch_rows = (rows_in.n_local_rows + 7u) / 8u * 8u;
    if (!scene->wf_q_items && !scene->wf_node_items) {
        while (batch_rows > 8 && plan_for(batch_rows * row_items, 0, 0, n_lights).bytes > budget_bytes()) {
            const size_t n_batches = (rows_in.n_local_rows + batch_rows - 1) / batch_rows + 1;
            batch_rows = ((rows_in.n_local_rows + n_batches - 1) / n_batches + 7u) / 8u * 8u;
        }
    }
    if (batch_rows * row_items >= 0x40000000ull) return RT_ERR_INVALID_ARG;
    const int rc = ensure_space(w, batch_rows * row_items, scene->wf_q_items, scene->wf_node_items, n_lights);
    if (rc != RT_OK) return rc;
    return run_pipeline(scene, w, w->sp, batch_rows, cam, depth, rows_in, d_out, stream, d_stats);
}

// ---- the entry point that allocates nothing (rt_hip_render_rows_noalloc) -------------------------------------------
// Its workspace is this array: each GPU the library runs on holds its own copy, at its own address, until the process
// ends.  In host code the array's NAME is only a handle for hipGetSymbolAddress; everything below uses the per-device
// address that call returned (kernel arguments, hipMemsetAsync).
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
}  // namespace

int rth::scratch_init() {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return RT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev]) return RT_OK;
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));  // the current device's copy
    if (!ptr) return RT_ERR_HIP;
    g_scratch_ptr[dev] = static_cast<unsigned char*>(ptr);
    return RT_OK;
}

int rth::wf2_render_noalloc(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, const RowMap& rows_in, double* d_out,
                            hipStream_t stream, RtStats* d_stats) {
    const int rc = rth::scratch_init();  // (the scene's device is current: the caller's guard)
    if (rc != RT_OK) return rc;
    unsigned char* base = g_scratch_ptr[scene->device];
    const uint32_t n_lights = scene->dev.n_lights;
    const size_t row_items = items_per_row(cam);
    // the tallest batch of whole 8-row tiles whose workspace fits the array
    Wf2Space sp;
    size_t lo = 0, hi = (rows_in.n_local_rows + 7u) / 8u;  // in tiles of 8 rows: lo fits (0 = nothing yet), hi + 1 does not
    auto fits = [&](size_t tiles) {
        const size_t items = tiles * 8u * row_items;
        if (items >= 0x40000000ull) return false;
        Wf2Space probe;
        return carve_space(probe, base, RT_NOALLOC_SCRATCH_BYTES, plan_for(items, 0, 0, n_lights), n_lights);
    };
    if (!fits(1)) return RT_ERR_NOALLOC_SIZE;
    lo = 1;
    while (lo < hi) {
        const size_t mid = (lo + hi + 1) / 2;
        if (fits(mid)) lo = mid; else hi = mid - 1;
    }
    const size_t batch_rows = lo * 8u;
    (void)carve_space(sp, base, RT_NOALLOC_SCRATCH_BYTES, plan_for(batch_rows * row_items, 0, 0, n_lights), n_lights);
    return run_pipeline(scene, scene->wf2, sp, batch_rows, cam, depth, rows_in, d_out, stream, d_stats);
}

const char* rth::wf2_describe() {
    return "kernel pipeline per generation of rays: wf2_trace_kernel (4-wide f32 hierarchy, persistent waves) for colour rays, "
           "shadow feelers and containers walks + wf2_prepare / wf2_spawn / wf2_light / wf2_overflow / wf2_compact + wf_combine_kernel";
}
