// rt_hip.hip -- host side and C ABI of include/rt_hip.h: scene validation and upload, kernel selection and launch.
// The tracing kernels are in rt_kernels.h (instantiated by rt_inst.hip); the small bookkeeping kernels are here.
// There is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "rt_host_internal.h"

using namespace rtd;
using namespace rtk;

namespace {

__global__ void wf_set_camera_kernel(RtCameraDesc* dst, const RtCameraDesc cam) {
    if (threadIdx.x == 0) *dst = cam;
}

// generation g is traced: its nodes end here, generation g + 1's begin
__global__ void wf_mark_kernel(WfSpace ws, unsigned level) {
    WfShard& sh = ws.shards[threadIdx.x];
    const unsigned long long end = (unsigned long long)sh.node_begin[level] + sh.nodes_asked[level];
    sh.node_begin[level + 1u] = end < ws.node_cap ? (unsigned)end : ws.node_cap;
    for (unsigned o = 0; o < RT_WF_BINS; ++o) sh.items[level & 1u][o] = 0u;  // read by this generation; the one after next counts here
}

// shade_hit's last line for every node of generation g (world.rs:76-82), children first
__global__ __launch_bounds__(256) void wf_combine_kernel(WfSpace ws, unsigned level, double* __restrict__ out) {
    const unsigned shard = blockIdx.x % RT_WF_SHARDS;
    const WfShard& sh = ws.shards[shard];
    const unsigned begin = sh.node_begin[level], end = sh.node_begin[level + 1u];
    const WfNode* __restrict__ nodes = ws.nodes + (size_t)shard * ws.node_cap;
    for (unsigned i = begin + (blockIdx.x / RT_WF_SHARDS) * 256u + threadIdx.x; i < end; i += (gridDim.x / RT_WF_SHARDS) * 256u) {
        const WfNode n = nodes[i];
        if (n.flags & WF_DEAD) continue;
        const V3 reflected = (n.flags & WF_HAS_REFLECT) ? n.child[0] * n.reflective : v3(0.0, 0.0, 0.0);
        const V3 refracted = (n.flags & WF_HAS_REFRACT) ? n.child[1] * n.transparency : v3(0.0, 0.0, 0.0);
        wf_deliver(ws.nodes, out, n.parent, combine(n.surface, reflected, refracted, (n.flags & WF_USE_SCHLICK) != 0u, n.reflectance));
    }
}

// Canvas::scale_to_ppm_data (canvas.rs:85-90)
__global__ __launch_bounds__(256) void quantize_kernel(const double* __restrict__ rgb, unsigned long long n,
                                                       uint8_t* __restrict__ out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double v = rgb[i] * 255.0;
        uint8_t q = 0;
        if (v == v) {
            v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
            q = (uint8_t)round(v);  // half away from zero, like f64::round
        }
        out[i] = q;
    }
}

// ================================================================ host ===
thread_local std::string g_hip_error;
std::mutex g_mutex;
bool g_initialised = false;
int g_device = -1;

}  // namespace
int rth::hip_fail(hipError_t e, const char* what) {
    g_hip_error = std::string(what) + ": " + hipGetErrorString(e);
    return RT_ERR_HIP;
}
namespace {
using rth::hip_fail;

int ensure_device() {
    if (g_initialised) return RT_OK;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_hip_error = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no HIP device";
        return RT_ERR_NO_DEVICE;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RT_ERR_NO_DEVICE;
    g_device = dev;
    g_initialised = true;
    return RT_OK;
}

// A scene lives on the device that was current when it was created; every entry point that takes a scene runs
// on that device whatever the calling thread's current device is (hipSetDevice is per host thread), and puts
// the thread's device back when it returns.
struct DeviceGuard {
    int saved = -1;
    bool switched = false;
    int rc = RT_OK;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&saved) != hipSuccess) {
            rc = RT_ERR_NO_DEVICE;
            return;
        }
        if (saved != device) {
            const hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice(scene's device)");
            else switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(saved);
    }
};
#define RT_ON_DEVICE_OF(scene)                \
    DeviceGuard device_guard((scene)->device); \
    if (device_guard.rc != RT_OK) return device_guard.rc

template <class T>
int upload(const T* src, size_t n, T** dst) {
    *dst = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

unsigned max_hits_of_kind(unsigned kind) {
    switch (kind) {
        case RT_SPHERE: case RT_CUBE: return 2;
        case RT_PLANE: case RT_TRIANGLE: case RT_SMOOTH_TRIANGLE: return 1;
        default: return 4;
    }
}

// Everything the kernels assume about the tables is checked here, so that no index read
// from them can leave its table and every traversal loop advances strictly.
int validate(const RtSceneDesc& d, uint32_t* flags_out) {
    auto need = [](uint32_t n, const void* p) { return n == 0 || p != nullptr; };
    if (!need(d.n_shapes, d.shapes) || !need(d.n_conics, d.conics) || !need(d.n_triangles, d.triangles) ||
        !need(d.n_groups, d.groups) || !need(d.n_csgs, d.csgs) || !need(d.n_lhit, d.lhit) ||
        !need(d.n_materials, d.materials) || !need(d.n_patterns, d.patterns) || !need(d.n_lights, d.lights) ||
        !need(d.n_bvh_nodes, d.bvh_nodes))
        return RT_ERR_INVALID_ARG;
    if (d.n_shapes > 0x3fffffffu || d.n_bvh_nodes > 0x3fffffffu) return RT_ERR_INVALID_SCENE;
    uint32_t flags = 0;
    // patterns: children come after their parent (no cycles); arithmetic mixtures nest <= RT_PATTERN_STACK
    std::vector<int> pat_depth(d.n_patterns, 0);
    for (uint32_t k = d.n_patterns; k-- > 0;) {
        const RtPattern& p = d.patterns[k];
        if (p.kind > RT_PAT_TEST) return RT_ERR_INVALID_SCENE;
        int da = 0, db = 0;
        const bool unary = p.kind == RT_PAT_PERTURB;
        const bool binary = p.kind >= RT_PAT_BLEND && p.kind <= RT_PAT_CHECKER;
        if (unary || binary) {
            if (p.a <= (int32_t)k || p.a >= (int32_t)d.n_patterns) return RT_ERR_INVALID_SCENE;
            da = pat_depth[p.a];
        }
        if (binary) {
            if (p.b <= (int32_t)k || p.b >= (int32_t)d.n_patterns) return RT_ERR_INVALID_SCENE;
            db = pat_depth[p.b];
        }
        const bool arithmetic =
            p.kind == RT_PAT_BLEND || p.kind == RT_PAT_GRADIENT || p.kind == RT_PAT_RADIAL_GRADIENT;
        pat_depth[k] = (da > db ? da : db) + (arithmetic ? 1 : 0);
        if (pat_depth[k] > RT_PATTERN_STACK) return RT_ERR_NEST_TOO_DEEP;
    }
    for (uint32_t k = 0; k < d.n_materials; ++k)
        if (d.materials[k].pattern < -1 || d.materials[k].pattern >= (int32_t)d.n_patterns) return RT_ERR_INVALID_SCENE;
    // shapes: depth-first pre-order with consistent parent / depth / skip
    std::vector<uint32_t> open;  // enclosing compounds
    std::vector<uint32_t> max_hits(d.n_shapes, 0);
    for (uint32_t i = 0; i < d.n_shapes; ++i) {
        const RtShape& s = d.shapes[i];
        while (!open.empty() && d.shapes[open.back()].skip <= i) open.pop_back();
        const int32_t parent = open.empty() ? -1 : (int32_t)open.back();
        if (s.parent != parent || s.depth != open.size()) return RT_ERR_INVALID_SCENE;
        if (s.skip <= i || s.skip > d.n_shapes) return RT_ERR_INVALID_SCENE;
        if (parent >= 0 && s.skip > d.shapes[parent].skip) return RT_ERR_INVALID_SCENE;
        const unsigned kind = RT_KIND(s.kind_flags);
        if (kind > RT_CSG) return RT_ERR_INVALID_SCENE;
        if (kind <= RT_SMOOTH_TRIANGLE) {
            if (s.skip != i + 1 || s.material >= d.n_materials) return RT_ERR_INVALID_SCENE;
            if (s.depth > RT_MAX_NEST) return RT_ERR_NEST_TOO_DEEP;
            if ((kind == RT_CYLINDER || kind == RT_CONE) && s.param >= d.n_conics) return RT_ERR_INVALID_SCENE;
            if ((kind == RT_TRIANGLE || kind == RT_SMOOTH_TRIANGLE) && s.param >= d.n_triangles)
                return RT_ERR_INVALID_SCENE;
            if (!(s.kind_flags & RT_FLAG_HAS_SHADOW)) flags |= RT_SCENE_HAS_SHADOWLESS;
            max_hits[i] = max_hits_of_kind(kind);
        } else {
            if (s.depth >= RT_MAX_NEST) return RT_ERR_NEST_TOO_DEEP;
            if (kind == RT_GROUP) {
                if (s.param >= d.n_groups) return RT_ERR_INVALID_SCENE;
                const RtGroup& g = d.groups[s.param];
                if (g.bvh_root < -1 || g.bvh_root >= (int32_t)d.n_bvh_nodes) return RT_ERR_INVALID_SCENE;
                if ((g.n_children == 0) != (s.skip == i + 1)) return RT_ERR_INVALID_SCENE;
                if (g.bvh_root >= 0) {
                    // inside a CSG node only a group with no cylinder or cone below it may carry a tree: such a primitive can
                    // return a hit outside its box, which the direction index makes up for at the world level and for whole
                    // CSG nodes only
                    bool in_csg = false;
                    for (uint32_t a : open) in_csg = in_csg || RT_KIND(d.shapes[a].kind_flags) == RT_CSG;
                    if (in_csg)
                        for (uint32_t c = i + 1; c < s.skip && c < d.n_shapes; ++c) {
                            const unsigned ck = RT_KIND(d.shapes[c].kind_flags);
                            if (ck == RT_CYLINDER || ck == RT_CONE) return RT_ERR_INVALID_SCENE;
                        }
                }
            } else {
                if (s.param >= d.n_csgs) return RT_ERR_INVALID_SCENE;
                const RtCsg& c = d.csgs[s.param];
                const unsigned op = (s.kind_flags >> RT_FLAG_CSG_OP_SHIFT) & 3u;
                if (c.op > 2 || c.op != op) return RT_ERR_INVALID_SCENE;
                // exactly two children: left = i+1, right = left.skip, right.skip == skip
                if (i + 1 >= d.n_shapes || i + 1 >= s.skip) return RT_ERR_INVALID_SCENE;
                const uint32_t right = d.shapes[i + 1].skip;
                if (right <= i + 1 || right >= s.skip || d.shapes[right].skip != s.skip) return RT_ERR_INVALID_SCENE;
                if ((uint64_t)c.lhit_offset + (s.skip - i - 1) > d.n_lhit) return RT_ERR_INVALID_SCENE;
            }
            open.push_back(i);
        }
    }
    // a CSG subtree's list must fit the per-lane list whatever the ray does
    for (uint32_t i = d.n_shapes; i-- > 0;) {
        const RtShape& s = d.shapes[i];
        if (RT_KIND(s.kind_flags) > RT_SMOOTH_TRIANGLE) {
            uint64_t sum = 0;
            for (uint32_t c = i + 1; c < s.skip; c = d.shapes[c].skip) sum += max_hits[c];
            max_hits[i] = (uint32_t)(sum > 0x7fffffffu ? 0x7fffffffu : sum);
            if (RT_KIND(s.kind_flags) == RT_CSG && sum > RT_LIST_CAP) return RT_ERR_LIST_BOUND;
        }
    }
    // acceleration nodes: links only go forward (every walk terminates), leaves name real shapes
    for (uint32_t k = 0; k < d.n_bvh_nodes; ++k) {
        const RtBvhNode& nd = d.bvh_nodes[k];
        if (nd.miss != -1 && (nd.miss <= (int32_t)k || nd.miss >= (int32_t)d.n_bvh_nodes)) return RT_ERR_INVALID_SCENE;
        if (nd.shape < -1 || nd.shape >= (int32_t)d.n_shapes) return RT_ERR_INVALID_SCENE;
        if (nd.shape < 0 && k + 1 >= d.n_bvh_nodes) return RT_ERR_INVALID_SCENE;  // inner node needs a successor
    }
    if (d.world_bvh_root < -1 || d.world_bvh_root >= (int32_t)d.n_bvh_nodes) return RT_ERR_INVALID_SCENE;
    // Which nodes each walk can reach.  A tree belongs to ONE level -- the world, or one group -- and names only
    // shapes of that level: the traversal enters a level per leaf it descends into (`++level` on a stack of
    // RT_STACK entries), so a group's tree naming the group itself, an ancestor or a shape of another depth
    // would never terminate or run off that stack.  Trees of different levels share no node.
    std::vector<int32_t> owner(d.n_bvh_nodes, -1);  // -1 unreached, -2 the world's tree, else the owning group's shape index
    auto walk = [&](int32_t root, int32_t who, auto&& leaf_ok) -> bool {
        std::vector<uint32_t> todo{(uint32_t)root};
        while (!todo.empty()) {
            const uint32_t k = todo.back();
            todo.pop_back();
            if (owner[k] == who) continue;
            if (owner[k] != -1) return false;
            owner[k] = who;
            const RtBvhNode& nd = d.bvh_nodes[k];
            if (nd.miss >= 0) todo.push_back((uint32_t)nd.miss);
            if (nd.shape < 0)
                todo.push_back(k + 1);
            else if (!leaf_ok(d.shapes[nd.shape]))
                return false;
        }
        return true;
    };
    if (d.world_bvh_root >= 0) {
        // A world-level leaf is a top-level shape, or -- the global form of hierarchy -- a primitive or CSG node
        // that sits inside groups and outside any CSG node (the direction index's contract): the kernels then
        // reach it through its ancestors (RT_CLASS_NESTED_LEAVES) and never enter a group, so such a tree names
        // no group at all.
        bool has_group_leaf = false, ok = true;
        ok = walk(d.world_bvh_root, -2, [&](const RtShape& leaf) {
            if (RT_KIND(leaf.kind_flags) == RT_GROUP) has_group_leaf = true;
            if (leaf.parent < 0) return true;
            if (RT_KIND(leaf.kind_flags) == RT_GROUP) return false;
            for (int32_t a = leaf.parent; a >= 0; a = d.shapes[a].parent)
                if (RT_KIND(d.shapes[a].kind_flags) == RT_CSG) return false;
            flags |= RT_SCENE_NESTED_LEAVES;
            return true;
        });
        if (!ok || ((flags & RT_SCENE_NESTED_LEAVES) && has_group_leaf)) return RT_ERR_INVALID_SCENE;
    }
    for (uint32_t i = 0; i < d.n_shapes; ++i) {
        const RtShape& s = d.shapes[i];
        if (RT_KIND(s.kind_flags) != RT_GROUP || d.groups[s.param].bvh_root < 0) continue;
        if (flags & RT_SCENE_NESTED_LEAVES) return RT_ERR_INVALID_SCENE;  // the global form leaves no tree to the groups
        if (!walk(d.groups[s.param].bvh_root, (int32_t)i, [&](const RtShape& leaf) { return leaf.parent == (int32_t)i; }))
            return RT_ERR_INVALID_SCENE;
    }
    if (d.world_bvh_root < 0) {
        bool flat = true;
        for (uint32_t i = 0; i < d.n_shapes; ++i) flat = flat && RT_KIND(d.shapes[i].kind_flags) <= RT_SMOOTH_TRIANGLE;
        if (flat) flags |= RT_SCENE_FLAT;
    }
    // direction index: offsets monotone and inside the entry list, entries name candidates,
    // candidates name a conic and the shape to evaluate for it (itself or an enclosing CSG)
    // (dir_cell_start null: candidates only, the library builds the cells on the device)
    if (d.dir_res != 0) {
        if (d.dir_res > 2048 || !need(d.n_dir_cands, d.dir_cands)) return RT_ERR_INVALID_ARG;
        if (d.dir_cell_start) {
            if (!need(d.n_dir_entries, d.dir_entries)) return RT_ERR_INVALID_ARG;
            const size_t n_cells = (size_t)6 * d.dir_res * d.dir_res;
            if (d.dir_cell_start[0] != 0 || d.dir_cell_start[n_cells] != d.n_dir_entries) return RT_ERR_INVALID_SCENE;
            for (size_t c = 0; c < n_cells; ++c)
                if (d.dir_cell_start[c] > d.dir_cell_start[c + 1]) return RT_ERR_INVALID_SCENE;
            for (uint32_t k = 0; k < d.n_dir_entries; ++k)
                if (d.dir_entries[k] >= d.n_dir_cands) return RT_ERR_INVALID_SCENE;
        } else if (d.n_dir_entries != 0 || d.dir_entries) {
            return RT_ERR_INVALID_ARG;
        }
        for (uint32_t k = 0; k < d.n_dir_cands; ++k) {
            const RtDirCand& c = d.dir_cands[k];
            if (c.shape >= d.n_shapes || c.visit >= d.n_shapes) return RT_ERR_INVALID_SCENE;
            const unsigned ck = RT_KIND(d.shapes[c.shape].kind_flags), vk = RT_KIND(d.shapes[c.visit].kind_flags);
            if (ck != RT_CYLINDER && ck != RT_CONE) return RT_ERR_INVALID_SCENE;
            if (!(c.visit == c.shape || (vk == RT_CSG && c.visit < c.shape && c.shape < d.shapes[c.visit].skip)))
                return RT_ERR_INVALID_SCENE;
        }
    }
    *flags_out = flags;
    return RT_OK;
}

}  // namespace


namespace {

int check_camera(const RtCameraDesc* cam, uint32_t depth) {
    if (!cam || cam->hsize == 0 || cam->vsize == 0) return RT_ERR_INVALID_ARG;
    if (depth > RT_MAX_DEPTH) return RT_ERR_DEPTH;
    return RT_OK;
}

int status_to_error(unsigned flags) {
    if (flags & RT_STATUS_LIST_OVERFLOW) return RT_ERR_LIST_OVERFLOW;
    if (flags & RT_STATUS_NAN_T) return RT_ERR_NAN_T;
    return RT_OK;
}

int read_status(const RtScene* scene, hipStream_t stream) {
    unsigned flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, scene->d_status, sizeof flags, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (flags) HIP_TRY(hipMemsetAsync(scene->d_status, 0, sizeof(unsigned), stream));
    return status_to_error(flags);
}

#ifndef RT_WF_GRID
#define RT_WF_GRID 8192u  // blocks striding over a queue whose length only the device knows
#endif
#ifndef RT_WF_MIN_SHAPES
#define RT_WF_MIN_SHAPES 192u  // scenes below this stay with the one-launch kernel (see launch_render): measured crossover ~165 shapes
#endif
#ifndef RT_WF_QUEUE_PER_PIXEL
#define RT_WF_QUEUE_PER_PIXEL 2u  // queue slots per generation and node slots per frame, in units of the frame's pixels;
#define RT_WF_NODES_PER_PIXEL 3u  // past that a hit keeps its subtree to itself (slower, same result)
#endif

}  // namespace
void rth::launch_combine(const WfSpace& ws, unsigned level, double* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(wf_combine_kernel, dim3(RT_WF_GRID), dim3(256), 0, stream, ws, level, d_out);
}
namespace {

// RT_HIP_WAVEFRONT=0/1/2 in the environment overrides the policy for scenes left at "by policy" (experiments)
int wavefront_env() {
    static const int v = [] {
        const char* e = std::getenv("RT_HIP_WAVEFRONT");
        return e && *e ? std::atoi(e) : -1;
    }();
    return v;
}

int ensure_wavefront_space(const RtScene* scene, size_t pixels) {
    if (scene->wf.shards && scene->wf_pixels >= pixels) return RT_OK;
    if (scene->wf.shards) {  // grow: launches of one scene are ordered by the caller, the old space is idle by contract
        HIP_TRY(hipDeviceSynchronize());
        for (void* p : {(void*)scene->wf.queue[0], (void*)scene->wf.queue[1], (void*)scene->wf.nodes, (void*)scene->wf.shards})
            if (p) (void)hipFree(p);
        std::memset(&scene->wf, 0, sizeof scene->wf);
        scene->wf_pixels = 0;
    }
    // capacities are per shard; at least one wave's worth each, so that a tiny frame is not all refusals
    // Default capacities: 640 B per pixel, but never more than a fixed budget (24 GB, or half of the free device memory:
    // RT_HIP_WF_BUDGET_MB as for the kernel pipeline) -- a larger frame gets proportionally smaller queues and pool, and the
    // hits that find no room keep their subtrees to themselves (slower, same frame).
    size_t q_default = pixels * RT_WF_QUEUE_PER_PIXEL, node_default = pixels * RT_WF_NODES_PER_PIXEL;
    {
        const char* e = std::getenv("RT_HIP_WF_BUDGET_MB");
        const long mb = e && *e ? std::atol(e) : 0;
        size_t budget = (size_t)(mb > 0 ? mb : 24576) << 20;
        size_t free_b = 0, total_b = 0;
        if (mb <= 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 < budget) budget = free_b / 2;
        const size_t want = 2 * q_default * sizeof(WfItem) + node_default * sizeof(WfNode);
        if (want > budget && want > 0) {
            const double scale = (double)budget / (double)want;
            q_default = (size_t)((double)q_default * scale);
            node_default = (size_t)((double)node_default * scale);
        }
    }
    const size_t q_items = scene->wf_q_items ? scene->wf_q_items : q_default;
    const size_t node_items = scene->wf_node_items ? scene->wf_node_items : node_default;
    size_t q_cap = (q_items + RT_WF_SHARDS - 1) / RT_WF_SHARDS, node_cap = (node_items + RT_WF_SHARDS - 1) / RT_WF_SHARDS;
    if (!scene->wf_q_items && q_cap < 128) q_cap = 128;
    if (!scene->wf_node_items && node_cap < 64) node_cap = 64;
    q_cap = (q_cap + RT_WF_BINS - 1) / RT_WF_BINS * RT_WF_BINS;  // a slice is split evenly over the direction octants
    if (node_cap * RT_WF_SHARDS >= 0x40000000ull || q_cap * RT_WF_SHARDS >= 0x80000000ull) return RT_ERR_INVALID_ARG;
    WfSpace w{};
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.queue[0]), q_cap * RT_WF_SHARDS * sizeof(WfItem)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.queue[1]), q_cap * RT_WF_SHARDS * sizeof(WfItem)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.nodes), node_cap * RT_WF_SHARDS * sizeof(WfNode)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.shards), RT_WF_SHARDS * sizeof(WfShard) + sizeof(RtCameraDesc)));
        w.cam = reinterpret_cast<const RtCameraDesc*>(w.shards + RT_WF_SHARDS);
        return RT_OK;
    };
    const int rc = body();
    if (rc != RT_OK) {
        for (void* p : {(void*)w.queue[0], (void*)w.queue[1], (void*)w.nodes, (void*)w.shards})
            if (p) (void)hipFree(p);
        return rc;
    }
    w.q_cap = (unsigned)q_cap, w.node_cap = (unsigned)node_cap;
    scene->wf = w;
    scene->wf_pixels = pixels;
    return RT_OK;
}

// no_alloc: rt_hip_render_rows_noalloc -- the pipeline works in the library's fixed array, the older generation kernels
// (whose workspace is allocated) are not used
int launch_render(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, const RowMap& rows, double* d_out,
                  hipStream_t stream, RtStats* d_stats, bool no_alloc = false) {
    const dim3 block(256);
    const dim3 grid((cam->hsize + 15u) / 16u, (rows.n_local_rows + 15u) / 16u);
    if (grid.y > 65535u * 16u) return RT_ERR_INVALID_ARG;
    const bool flat = (scene->dev.flags & RT_SCENE_FLAT) != 0;
    const bool nested_leaves = (scene->dev.flags & RT_SCENE_NESTED_LEAVES) != 0;
    const bool staged = scene->dev.stage_bytes != 0;
    // the per-lane list area is read only by the containers walk and by shadow tests among shadowless
    // primitives: a scene with neither leaves the LDS to more resident blocks
    const bool opaque_scene = (scene->dev.flags & RT_SCENE_RECURSIVE) == 0;
    const bool needs_list = !opaque_scene || (scene->dev.flags & RT_SCENE_HAS_SHADOWLESS) != 0;
    auto lds_of = [&](unsigned threads) {
        return !staged ? 0u
                       : ((scene->dev.stage_bytes + 15u) & ~15u) +
                             (flat && needs_list ? RT_LDS_LIST_BYTES(threads, scene->dev.lds_list_cap) : 0u);
    };
    const unsigned lds = lds_of(256u);  // the generation kernels run 256-thread blocks
    // Large scenes whose materials can spawn reflected / refracted rays are traced one generation of rays per
    // launch (dense waves at every depth); the others in one launch, each lane keeping its pixel.  The
    // generations cost ~400 B of queue and node traffic per colour ray, which only pays where a ray is
    // expensive and the lanes of a wave drift apart: measured at 4096x2160, 10 000 primitives 135 -> 95 ms,
    // but cover.rs (30 primitives) 9.1 -> 15.7 ms and glass_bubble.rs 5.5 -> 10.2 ms.
    const size_t pixels = (size_t)cam->hsize * rows.n_local_rows;
    const bool can_wavefront = (scene->dev.flags & RT_SCENE_RECURSIVE) != 0 && depth > 0 && pixels < 0x20000000ull;
    const int mode = scene->wf_mode >= 0 ? scene->wf_mode : wavefront_env();
    const bool large = !flat && scene->dev.n_shapes >= RT_WF_MIN_SHAPES;
    // The kernel pipeline (rt_wf2.h) for the large scenes it covers: traversal in kernels of its own with persistent
    // waves over a 4-wide hierarchy, shading in between.  If its workspace cannot be had the one-launch kernel below
    // renders the same frame.
    if (scene->wf2 && pixels < 0x20000000ull && (mode == 2 || (mode < 0 && large && can_wavefront))) {
        if (no_alloc) return rth::wf2_render_noalloc(scene, cam, depth, rows, d_out, stream, d_stats);
        const int rc = rth::wf2_render(scene, cam, depth, rows, d_out, stream, d_stats);
        if (rc != RT_ERR_HIP) return rc;
        (void)hipGetLastError();
    }
    bool wavefront = !no_alloc && can_wavefront && (mode >= 1 || (mode < 0 && large));
    if (wavefront && ensure_wavefront_space(scene, pixels) != RT_OK) {  // no room for queues and nodes: same frame, one launch
        (void)hipGetLastError();
        wavefront = false;
    }
    if (wavefront) {
        const WfSpace ws = scene->wf;
        HIP_TRY(hipMemsetAsync(ws.shards, 0, RT_WF_SHARDS * sizeof(WfShard), stream));
        // (the camera record travels as a kernel argument: the caller's *cam may be gone when the stream gets here)
        hipLaunchKernelGGL(wf_set_camera_kernel, dim3(1), dim3(64), 0, stream, const_cast<RtCameraDesc*>(ws.cam), *cam);
        auto level = [&](auto primary, auto later, unsigned k) {
            if (k == 0)
                hipLaunchKernelGGL(primary, grid, block, lds, stream, scene->dev, *cam, (int)depth, rows, d_out, d_stats,
                                   scene->d_status, ws, k);
            else
                // (one-wave blocks where nothing is staged: a block's slot is free only when its slowest wave is done)
                hipLaunchKernelGGL(later, dim3(staged ? RT_WF_GRID : RT_WF_GRID * 4u), staged ? block : dim3(64), lds, stream,
                                   scene->dev, *cam, (int)depth, rows, d_out, d_stats, scene->d_status, ws, k);
        };
#define RT_WF_PAIR(ST, FL, SG) wf_level_kernel<ST, FL, SG, true>, wf_level_kernel<ST, FL, SG, false>
        for (unsigned k = 0; k <= depth; ++k) {
            if (d_stats)
                staged && flat ? level(RT_WF_PAIR(true, RT_CLASS_FLAT, true), k)
                : flat         ? level(RT_WF_PAIR(true, RT_CLASS_FLAT, false), k)
                : nested_leaves ? level(RT_WF_PAIR(true, RT_CLASS_NESTED_LEAVES, false), k)
                               : level(RT_WF_PAIR(true, RT_CLASS_GENERAL, false), k);
            else
                staged && flat ? level(RT_WF_PAIR(false, RT_CLASS_FLAT, true), k)
                : flat         ? level(RT_WF_PAIR(false, RT_CLASS_FLAT, false), k)
                : nested_leaves ? level(RT_WF_PAIR(false, RT_CLASS_NESTED_LEAVES, false), k)
                               : level(RT_WF_PAIR(false, RT_CLASS_GENERAL, false), k);
            hipLaunchKernelGGL(wf_mark_kernel, dim3(1), dim3(RT_WF_SHARDS), 0, stream, ws, k);
        }
        for (unsigned k = depth; k-- > 0;)  // generation `depth` has no budget left and so no nodes
            rth::launch_combine(ws, k, d_out, stream);
        HIP_TRY(hipGetLastError());
        return RT_OK;
    }
    const bool opaque = (scene->dev.flags & RT_SCENE_RECURSIVE) == 0;
    // Block size: the smallest of 64 / 128 / 256 threads whose LDS share still lets the CU hold the waves the
    // kernel's register budget allows (a block's slot is free only when its slowest wave is done: with one-wave
    // blocks cover.rs -13 %, teapot.rs -11 %, csg.rs -9 %, glass_bubble.rs -6 % against four-wave blocks).
    // DENSE (3 waves/SIMD) for staged flat scenes whenever some block size fits twelve waves' worth of LDS.
    constexpr unsigned kLdsPerCu = 160u * 1024u;
    auto fits = [&](unsigned threads, unsigned waves_per_simd) {
        return (size_t)lds_of(threads) * (waves_per_simd * 4u * 64u / threads) <= kLdsPerCu;
    };
    const bool dense = staged && flat && (fits(64u, 3u) || fits(128u, 3u) || fits(256u, 3u));
    const unsigned waves_per_simd = flat ? (staged && dense ? 3u : RT_WAVES_PER_SIMD_FLAT) : (opaque ? RT_WAVES_PER_SIMD_OPAQUE : RT_WAVES_PER_SIMD);
    unsigned threads = 256u;
    for (unsigned t : {64u, 128u})
        if (fits(t, waves_per_simd)) {
            threads = t;
            break;
        }
    const unsigned wx = threads >= 128u ? 2u : 1u, wy = (threads / 64u) / wx;
    const dim3 grid_r((cam->hsize + 8u * wx - 1u) / (8u * wx), (rows.n_local_rows + 8u * wy - 1u) / (8u * wy));
    if (grid_r.y > 65535u) return RT_ERR_INVALID_ARG;
    const unsigned lds_r = lds_of(threads);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid_r, dim3(threads), lds_r, stream, scene->dev, *cam, (int)depth,
                           rows, d_out, d_stats, scene->d_status);
    };
    auto pick = [&](auto stats_tag) {
        constexpr bool ST = decltype(stats_tag)::value;
        constexpr int G = RT_CLASS_GENERAL, N = RT_CLASS_NESTED_LEAVES, F = RT_CLASS_FLAT;
        if (!flat && nested_leaves && staged) return opaque ? go(render_kernel<ST, N, true, false>) : go(render_kernel<ST, N, true, true>);
        if (!flat && nested_leaves) return opaque ? go(render_kernel<ST, N, false, false>) : go(render_kernel<ST, N, false, true>);
        if (!flat && staged) return opaque ? go(render_kernel<ST, G, true, false>) : go(render_kernel<ST, G, true, true>);
        if (!flat) return opaque ? go(render_kernel<ST, G, false, false>) : go(render_kernel<ST, G, false, true>);
        if (!staged) return opaque ? go(render_kernel<ST, F, false, false>) : go(render_kernel<ST, F, false, true>);
        if (opaque) return dense ? go(render_kernel<ST, F, true, false, true>) : go(render_kernel<ST, F, true, false>);
#ifndef RT_DENSE_RECURSIVE
#define RT_DENSE_RECURSIVE 1
#endif
        return dense && RT_DENSE_RECURSIVE ? go(render_kernel<ST, F, true, true, true>) : go(render_kernel<ST, F, true, true>);
    };
    if (d_stats)
        pick(std::true_type{});
    else
        pick(std::false_type{});
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // namespace

// =============================================================== C ABI ===
extern "C" {

int rt_hip_init(int device) {
    std::lock_guard<std::mutex> lock(g_mutex);
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_hip_error = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no HIP device";
        return RT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return RT_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    g_initialised = true;
    return rth::scratch_init();  // the address of this device's copy of the fixed workspace (rt_hip_render_rows_noalloc)
}

void rt_hip_shutdown(void) {
    rth::multi_shutdown();
    rth::hostmem_shutdown();
    std::lock_guard<std::mutex> lock(g_mutex);
    g_initialised = false;
    g_device = -1;
}

const char* rt_hip_strerror(int code) {
    switch (code) {
        case RT_OK: return "ok";
        case RT_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU path)";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_INVALID_SCENE: return "malformed scene tables";
        case RT_ERR_NEST_TOO_DEEP: return "compound or pattern nesting too deep";
        case RT_ERR_LIST_BOUND: return "a CSG subtree can yield more intersections than the per-lane list holds";
        case RT_ERR_LIST_OVERFLOW: return "per-lane intersection list overflowed";
        case RT_ERR_NAN_T: return "a NaN intersection distance was dropped (the reference panics: t is not NaN)";
        case RT_ERR_INVALID_ARG: return "invalid argument";
        case RT_ERR_DEPTH: return "recursion budget above RT_MAX_DEPTH";
        case RT_ERR_NOALLOC_SIZE: return "rt_hip_render_rows_noalloc: the frame is too wide for the library's fixed 128 MiB workspace (8 rows must fit)";
        default: return "unknown error";
    }
}

const char* rt_hip_last_hip_error(void) { return g_hip_error.c_str(); }

int rt_hip_scene_create(const RtSceneDesc* desc, RtScene** out) {
    if (!desc || !out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    const bool timing = std::getenv("RT_HIP_CREATE_TIMING") != nullptr;  // (studies: where a scene's creation spends its time)
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "scene_create: %-34s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    uint32_t flags = 0;
    const int v = validate(*desc, &flags);
    if (v != RT_OK) return v;
    stage("validate");
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        const int rc = ensure_device();
        if (rc != RT_OK) return rc;
    }
    RtScene* s = new RtScene();
    std::memset(&s->dev, 0, sizeof s->dev);
    s->d_status = nullptr;
    s->d_dev = nullptr;
    s->tables = nullptr;
    for (hipEvent_t& e : s->band_done) e = nullptr;
    if (hipGetDevice(&s->device) != hipSuccess) s->device = g_device;  // the calling thread's current device
    std::memset(&s->wf, 0, sizeof s->wf);
    s->wf_pixels = 0;
    s->d_frame = nullptr, s->d_frame8 = nullptr, s->frame_values = 0;
    s->wf_mode = -1;
    s->wf_q_items = s->wf_node_items = 0;
    s->wf2 = nullptr;
    for (uint32_t k = 0; k < desc->n_materials; ++k)  // world.rs:99,110: these are the only tests that let color_at recurse
        if (desc->materials[k].reflective != 0.0 || desc->materials[k].transparency != 0.0) flags |= RT_SCENE_RECURSIVE;
    // Every table, the LDS staging image, the pipeline's hierarchy, the scene record and the status word in ONE device
    // block, filled by one host-to-device copy (the block comes from the library's cache: the drop-in pattern of a
    // scene per frame does not reach the driver's allocator after its first frame).
    rth::Arena arena;
    // (one allocation for the lot: the tables, as much again at most for the LDS image, the 4-wide hierarchy and leaf records)
    arena.host.reserve(2 * (sizeof(RtShape) * desc->n_shapes + sizeof(RtConic) * desc->n_conics + sizeof(RtTriangle) * desc->n_triangles +
                            sizeof(RtGroup) * desc->n_groups + sizeof(RtCsg) * desc->n_csgs + desc->n_lhit + sizeof(RtMaterial) * desc->n_materials +
                            sizeof(RtPattern) * desc->n_patterns + sizeof(RtBvhNode) * desc->n_bvh_nodes) +
                       sizeof(RtDirCand) * desc->n_dir_cands + sizeof(uint32_t) * ((size_t)desc->n_dir_entries + 6u * desc->dir_res * desc->dir_res) +
                       (size_t)desc->n_shapes * 160u + 65536u);
    int rc = RT_OK;
    const size_t o_shapes = arena.add(desc->shapes, sizeof(RtShape) * desc->n_shapes);
    {  // RT_FLAG_IDENTITY: primitives whose inverse transform is the identity matrix bit for bit (rt_trace.h: identity_ray)
        static const double kIdentity[12] = {1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0};
        RtShape* mine = reinterpret_cast<RtShape*>(arena.host.data() + o_shapes);
        const bool off = std::getenv("RT_HIP_NO_IDENTITY") != nullptr;  // test knob: every primitive reads its matrix
        for (uint32_t i = 0; i < desc->n_shapes; ++i) {
            mine[i].kind_flags &= ~RT_FLAG_IDENTITY;
            if (!off && RT_KIND(mine[i].kind_flags) <= RT_SMOOTH_TRIANGLE && std::memcmp(mine[i].inv, kIdentity, sizeof kIdentity) == 0)
                mine[i].kind_flags |= RT_FLAG_IDENTITY;
        }
        // RT_FLAG_CSG_RIGHT_LHIT: from the l_hit table as handed in (whoever flattened the scene), so that the skip of the
        // right operand in list_eval_core holds for tables of any origin
        for (uint32_t i = 0; i < desc->n_shapes; ++i) {
            mine[i].kind_flags &= ~RT_FLAG_CSG_RIGHT_LHIT;
            if (RT_KIND(mine[i].kind_flags) != RT_CSG) continue;
            const RtCsg& c = desc->csgs[mine[i].param];
            const uint32_t right = mine[i + 1].skip;  // (validate: left = i + 1, right = left.skip, right.skip == skip)
            for (uint32_t p = right; p < mine[i].skip; ++p)
                if (desc->lhit[c.lhit_offset + (p - i - 1u)] != 0) mine[i].kind_flags |= RT_FLAG_CSG_RIGHT_LHIT;
        }
    }
    const size_t o_conics = arena.add(desc->conics, sizeof(RtConic) * desc->n_conics);
    const size_t o_tris = arena.add(desc->triangles, sizeof(RtTriangle) * desc->n_triangles);
    const size_t o_groups = arena.add(desc->groups, sizeof(RtGroup) * desc->n_groups);
    const size_t o_csgs = arena.add(desc->csgs, sizeof(RtCsg) * desc->n_csgs);
    const size_t o_lhit = arena.add(desc->lhit, desc->n_lhit);
    const size_t o_mats = arena.add(desc->materials, sizeof(RtMaterial) * desc->n_materials);
    const size_t o_pats = arena.add(desc->patterns, sizeof(RtPattern) * desc->n_patterns);
    const size_t o_lights = arena.add(desc->lights, sizeof(RtLight) * desc->n_lights);
    const size_t o_bvh = arena.add(desc->bvh_nodes, sizeof(RtBvhNode) * desc->n_bvh_nodes);
    size_t o_cands = 0, o_cells = 0, o_entries = 0;
    const bool has_dir = desc->dir_res != 0 && desc->n_dir_cands != 0;
    if (has_dir) {
        o_cands = arena.add(desc->dir_cands, sizeof(RtDirCand) * desc->n_dir_cands);
        if (desc->dir_cell_start) {
            o_cells = arena.add(desc->dir_cell_start, sizeof(uint32_t) * ((size_t)6 * desc->dir_res * desc->dir_res + 1));
            o_entries = arena.add(desc->dir_entries, sizeof(uint32_t) * desc->n_dir_entries);
        }
    }
    s->dev.n_dir_cands = has_dir ? desc->n_dir_cands : 0u;
    s->dev.dir_res = has_dir ? desc->dir_res : 0u;
    s->dev.n_shapes = desc->n_shapes;
    s->dev.n_lights = desc->n_lights;
    s->dev.world_bvh_root = desc->world_bvh_root;
    s->dev.flags = flags;
    s->dev.stage = nullptr;
    s->dev.stage_bytes = 0;
    size_t o_stage = 0;
    {  // pack the tables lanes index individually: small scenes trace and shade out of an LDS copy
        const bool is_flat = (flags & RT_SCENE_FLAT) != 0;
        std::vector<unsigned char> image;
        auto put = [&](const void* src, size_t bytes, uint32_t* off) {
            *off = (uint32_t)image.size();
            const unsigned char* b = static_cast<const unsigned char*>(src);
            if (bytes) image.insert(image.end(), b, b + bytes);
            while (image.size() % 16u) image.push_back(0);
        };
        uint64_t world_hits = 0;  // most entries World::intersections can return: the per-lane list must hold them
        for (uint32_t i = 0; i < desc->n_shapes; ++i) world_hits += max_hits_of_kind(RT_KIND(desc->shapes[i].kind_flags));
        size_t bytes = sizeof(RtShape) * desc->n_shapes + sizeof(RtConic) * desc->n_conics + sizeof(RtMaterial) * desc->n_materials +
                       sizeof(RtPattern) * desc->n_patterns + sizeof(RtLight) * desc->n_lights;
        if (!is_flat)
            bytes += sizeof(RtGroup) * desc->n_groups + sizeof(RtCsg) * desc->n_csgs + desc->n_lhit + sizeof(RtBvhNode) * desc->n_bvh_nodes +
                     sizeof(RtTriangle) * desc->n_triangles;
        bool may_fit = is_flat ? bytes <= RT_STAGE_MAX_BYTES && world_hits <= RT_LDS_LIST_CAP : bytes <= RT_STAGE_MAX_BYTES_COMPOUND;
#ifdef RT_HOST_EMU  // (tests/emu runs a block's threads one after the other: no cooperative staging there)
        may_fit = false;
#endif
        if (may_fit) {  // (large scenes do not pay for an image they will not use)
            {  // (the library's copy: it carries RT_FLAG_IDENTITY)
                const std::vector<unsigned char> mine(arena.host.begin() + (long)o_shapes,
                                                      arena.host.begin() + (long)(o_shapes + sizeof(RtShape) * desc->n_shapes));
                put(mine.data(), mine.size(), &s->dev.stage_off[0]);
            }
            put(desc->conics, sizeof(RtConic) * desc->n_conics, &s->dev.stage_off[1]);
            put(desc->materials, sizeof(RtMaterial) * desc->n_materials, &s->dev.stage_off[2]);
            put(desc->patterns, sizeof(RtPattern) * desc->n_patterns, &s->dev.stage_off[3]);
            put(desc->lights, sizeof(RtLight) * desc->n_lights, &s->dev.stage_off[4]);
            if (!is_flat) {  // what the traversal of compounds reads (FLAT kernels fetch triangles through the scalar cache)
                put(desc->groups, sizeof(RtGroup) * desc->n_groups, &s->dev.stage_off[5]);
                put(desc->csgs, sizeof(RtCsg) * desc->n_csgs, &s->dev.stage_off[6]);
                put(desc->lhit, desc->n_lhit, &s->dev.stage_off[7]);
                put(desc->bvh_nodes, sizeof(RtBvhNode) * desc->n_bvh_nodes, &s->dev.stage_off[8]);
                put(desc->triangles, sizeof(RtTriangle) * desc->n_triangles, &s->dev.stage_off[9]);
            }
            const bool fits = is_flat ? image.size() <= RT_STAGE_MAX_BYTES && world_hits <= RT_LDS_LIST_CAP
                                      : image.size() <= RT_STAGE_MAX_BYTES_COMPOUND;
            if (!image.empty() && fits) {
                s->dev.lds_list_cap = !is_flat ? 0u : (world_hits ? (uint32_t)world_hits : 1u);
                o_stage = arena.add(image.data(), image.size());
                s->dev.stage_bytes = (uint32_t)image.size();
            }
        }
    }
    stage("tables into the arena");
    rc = rth::wf2_create(*desc, flags, arena, &s->wf2);
    stage("4-wide hierarchy, leaf records");
    if (rc != RT_OK) {
        rt_hip_scene_destroy(s);
        return rc;
    }
    const size_t o_record = arena.reserve(sizeof(SceneDev));
    const size_t o_status = arena.reserve(sizeof(unsigned));
    {
        void* block = nullptr;
        rc = rth::block_alloc(&block, arena.host.size());
        if (rc != RT_OK) {
            rt_hip_scene_destroy(s);
            return rc;
        }
        s->tables = static_cast<unsigned char*>(block);
    }
    unsigned char* base = s->tables;
    auto at = [&](size_t off, auto* type_tag) { return reinterpret_cast<decltype(type_tag)>(base + off); };
    s->dev.shapes = at(o_shapes, (const RtShape*)nullptr), s->dev.conics = at(o_conics, (const RtConic*)nullptr);
    s->dev.tris = at(o_tris, (const RtTriangle*)nullptr), s->dev.groups = at(o_groups, (const RtGroup*)nullptr);
    s->dev.csgs = at(o_csgs, (const RtCsg*)nullptr), s->dev.lhit = at(o_lhit, (const uint8_t*)nullptr);
    s->dev.mats = at(o_mats, (const RtMaterial*)nullptr), s->dev.pats = at(o_pats, (const RtPattern*)nullptr);
    s->dev.lights = at(o_lights, (const RtLight*)nullptr), s->dev.bvh = at(o_bvh, (const RtBvhNode*)nullptr);
    s->dev.g_shapes = s->dev.shapes, s->dev.g_conics = s->dev.conics, s->dev.g_tris = s->dev.tris;
    s->dev.dir_cands = has_dir ? at(o_cands, (const RtDirCand*)nullptr) : nullptr;
    s->dev.dir_cell_start = has_dir && desc->dir_cell_start ? at(o_cells, (const uint32_t*)nullptr) : nullptr;
    s->dev.dir_entries = has_dir && desc->dir_cell_start ? at(o_entries, (const uint32_t*)nullptr) : nullptr;
    if (s->dev.stage_bytes) s->dev.stage = base + o_stage;
    s->d_dev = reinterpret_cast<SceneDev*>(base + o_record);
    s->d_status = reinterpret_cast<unsigned*>(base + o_status);
    // (scenes small enough to trace out of their LDS copy keep the skip-link walk over it: hexagon.rs 0.79 ms, 0.89 through
    // the 4-wide nodes in global memory)
    const rtd::Wf2Accel* accel4 = rth::wf2_bind(s->wf2, arena, base);
    s->dev.accel4 = s->dev.stage_bytes ? nullptr : accel4;
    const bool cells_here = has_dir && !desc->dir_cell_start;
    if (!cells_here) std::memcpy(arena.host.data() + o_record, &s->dev, sizeof(SceneDev));
    {
        const hipError_t e = hipMemcpy(base, arena.host.data(), arena.host.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(scene tables)");
    }
    stage("device block and upload");
    if (rc == RT_OK && cells_here) {  // candidates only: the cells are built here, on the device (rt_dirindex.hip)
        rth::DirIndex ix;
        rc = rth::dir_index_build(s->dev.dir_cands, desc->n_dir_cands, desc->dir_res, &ix);
        if (rc == RT_OK) {
            s->dev.dir_cell_start = ix.d_cell_start, s->dev.dir_entries = ix.d_entries, s->dev.dir_res = ix.res;
            s->allocations.push_back(ix.d_cell_start);
            s->allocations.push_back(ix.d_entries);
            const hipError_t e = hipMemcpy(s->d_dev, &s->dev, sizeof(SceneDev), hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(scene record)");
        }
    }
    if (rc != RT_OK) {
        rt_hip_scene_destroy(s);
        return rc;
    }
    stage("direction index");
    *out = s;
    return RT_OK;
}

void rt_hip_scene_destroy(RtScene* scene) {
    if (!scene) return;
    DeviceGuard device_guard(scene->device);
    (void)hipDeviceSynchronize();  // nothing of this scene's may still be running when its blocks are handed on
    for (void* p : scene->allocations) rth::block_free(p);
    rth::block_free(scene->tables);
    rth::block_free(scene->d_frame);
    for (hipEvent_t e : scene->band_done)
        if (e) (void)hipEventDestroy(e);
    for (void* p : {(void*)scene->wf.queue[0], (void*)scene->wf.queue[1], (void*)scene->wf.nodes, (void*)scene->wf.shards})
        if (p) (void)hipFree(p);
    rth::wf2_destroy(scene->wf2);
    delete scene;
}

int rt_hip_scene_set_wavefront(RtScene* scene, int mode, uint64_t queue_items, uint64_t node_items) {
    if (!scene || mode < -1 || mode > 2) return RT_ERR_INVALID_ARG;
    if (queue_items >= 0x80000000ull || node_items >= 0x40000000ull) return RT_ERR_INVALID_ARG;
    scene->wf_mode = mode;
    if (scene->wf_q_items != queue_items || scene->wf_node_items != node_items) {
        scene->wf_q_items = (size_t)queue_items, scene->wf_node_items = (size_t)node_items;
        scene->wf_pixels = 0;  // the next launch re-sizes the workspace
    }
    return RT_OK;
}

const char* rt_hip_scene_kernels(const RtScene* scene, uint32_t depth) {
    if (!scene) return "";
    // (the choices of launch_render, in its order)
    const bool flat = (scene->dev.flags & RT_SCENE_FLAT) != 0;
    const bool recursive = (scene->dev.flags & RT_SCENE_RECURSIVE) != 0 && depth > 0;
    const int mode = scene->wf_mode >= 0 ? scene->wf_mode : wavefront_env();
    const bool large = !flat && scene->dev.n_shapes >= RT_WF_MIN_SHAPES;
    if (scene->wf2 && (mode == 2 || (mode < 0 && large && recursive))) return rth::wf2_describe();
    if (recursive && (mode >= 1 || (mode < 0 && large)))
        return "one launch per generation of rays: wf_level_kernel (camera rays, then queue slices) + wf_mark_kernel + wf_combine_kernel";
    if (flat) return scene->dev.stage_bytes ? "render_kernel, flat scene class, tables and hit lists staged in LDS" : "render_kernel, flat scene class";
    if ((scene->dev.flags & RT_SCENE_NESTED_LEAVES) && scene->dev.accel4)
        return "render_kernel, 4-wide f32 hierarchy over nested leaves (trace4)";
    if (scene->dev.flags & RT_SCENE_NESTED_LEAVES)
        return scene->dev.stage_bytes ? "render_kernel, world-level hierarchy over nested leaves, tables staged in LDS"
                                      : "render_kernel, world-level hierarchy over nested leaves";
    return scene->dev.stage_bytes ? "render_kernel, general scene class (groups / CSG), tables staged in LDS" : "render_kernel, general scene class (groups / CSG)";
}

int rt_hip_scene_status(const RtScene* scene, void* hip_stream) {
    if (!scene) return RT_ERR_INVALID_ARG;
    RT_ON_DEVICE_OF(scene);
    return read_status(scene, static_cast<hipStream_t>(hip_stream));
}

int rt_hip_render_rows(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, uint32_t part, uint32_t n_parts,
                       uint32_t band_rows, uint32_t n_local_rows, double* d_out_rgb, void* hip_stream,
                       RtStats* d_stats) {
    if (!scene || !d_out_rgb || n_parts == 0 || part >= n_parts || band_rows == 0) return RT_ERR_INVALID_ARG;
    RT_ON_DEVICE_OF(scene);
    const int rc = check_camera(cam, depth);
    if (rc != RT_OK) return rc;
    if (n_local_rows == 0) return RT_OK;
    const RowMap rows{part, n_parts, band_rows, n_local_rows};
    return launch_render(scene, cam, depth, rows, d_out_rgb, static_cast<hipStream_t>(hip_stream), d_stats);
}

int rt_hip_render_rows_noalloc(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, uint32_t part, uint32_t n_parts,
                               uint32_t band_rows, uint32_t n_local_rows, double* d_out_rgb, void* hip_stream,
                               RtStats* d_stats) {
    if (!scene || !d_out_rgb || n_parts == 0 || part >= n_parts || band_rows == 0) return RT_ERR_INVALID_ARG;
    RT_ON_DEVICE_OF(scene);
    const int rc = check_camera(cam, depth);
    if (rc != RT_OK) return rc;
    if (n_local_rows == 0) return RT_OK;
    const RowMap rows{part, n_parts, band_rows, n_local_rows};
    return launch_render(scene, cam, depth, rows, d_out_rgb, static_cast<hipStream_t>(hip_stream), d_stats, true);
}

// Camera::render on a resident scene, the frame returned as f64 colours (out_rgb) and / or as the 8-bit
// samples Canvas::to_ppm prints (out_rgb8).  The device framebuffers belong to the scene and are reused.
static int scene_render_to_host(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, double* out_rgb,
                                uint8_t* out_rgb8, RtStats* stats) {
    if (!scene || (!out_rgb && !out_rgb8)) return RT_ERR_INVALID_ARG;
    RT_ON_DEVICE_OF(scene);
    const int crc0 = check_camera(cam, depth);
    if (crc0 != RT_OK) return crc0;
    const size_t n_values = (size_t)cam->hsize * cam->vsize * 3u;
    if (scene->frame_values < n_values) {
        rth::block_free(scene->d_frame);
        scene->d_frame = nullptr, scene->d_frame8 = nullptr, scene->frame_values = 0;
        void* block = nullptr;  // one block: the f64 frame, then its 8-bit form
        const int brc = rth::block_alloc(&block, n_values * sizeof(double) + n_values);
        if (brc != RT_OK) return brc;
        scene->d_frame = static_cast<double*>(block);
        scene->d_frame8 = reinterpret_cast<uint8_t*>(scene->d_frame + n_values);
        scene->frame_values = n_values;
    }
    double* d_out = scene->d_frame;
    uint8_t* d_out8 = scene->d_frame8;
    RtStats* d_stats = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    constexpr uint32_t kFrameBands = RtScene::kFrameBands;
    hipEvent_t* band_done = scene->band_done;
    for (uint32_t k = 0; k < kFrameBands; ++k)
        if (!band_done[k]) HIP_TRY(hipEventCreateWithFlags(&band_done[k], hipEventDisableTiming));
    auto body = [&]() -> int {
        if (stats) {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_stats), sizeof(RtStats)));
            HIP_TRY(hipMemset(d_stats, 0, sizeof(RtStats)));
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            HIP_TRY(hipEventRecord(e0, nullptr));
        }
        // A large f64 frame of a one-launch scene goes out in four bands of rows: band k crosses PCIe while band
        // k + 1 is traced (the kernel pipeline's frames stay whole: its batches are better large, and its frames take
        // longer than their copy anyway).
        const bool pipeline = scene->wf2 && (scene->dev.flags & RT_SCENE_RECURSIVE) && depth > 0 && scene->wf_mode != 0 && scene->wf_mode != 1;
        const uint32_t n_bands = out_rgb && !pipeline && n_values * sizeof(double) >= ((size_t)64 << 20) && cam->vsize >= 256u ? kFrameBands : 1u;
        const uint32_t band_rows = (cam->vsize + n_bands - 1u) / n_bands;
        for (uint32_t k = 0; k < n_bands; ++k) {
            const uint32_t r0 = k * band_rows, r1 = r0 + band_rows < cam->vsize ? r0 + band_rows : cam->vsize;
            if (r0 >= r1) break;
            const RowMap rows{k, n_bands, band_rows, r1 - r0};
            const int lrc = launch_render(scene, cam, depth, rows, d_out + (size_t)r0 * cam->hsize * 3u, nullptr, d_stats);
            if (lrc != RT_OK) return lrc;
            if (n_bands > 1u) HIP_TRY(hipEventRecord(band_done[k], nullptr));
        }
        if (stats) HIP_TRY(hipEventRecord(e1, nullptr));
        if (out_rgb8) {  // quantise on the device: an eighth of the bytes cross PCIe
            const int qrc = rt_hip_quantize(d_out, n_values, d_out8, nullptr);
            if (qrc != RT_OK) return qrc;
        }
        if (out_rgb) {
            for (uint32_t k = 0; k < n_bands; ++k) {
                const uint32_t r0 = k * band_rows, r1 = r0 + band_rows < cam->vsize ? r0 + band_rows : cam->vsize;
                if (r0 >= r1) break;
                const size_t off = (size_t)r0 * cam->hsize * 3u, n = (size_t)(r1 - r0) * cam->hsize * 3u;
                const int crc = rth::copy_to_host(out_rgb + off, d_out + off, n * sizeof(double), nullptr, n_bands > 1u ? band_done[k] : nullptr);
                if (crc != RT_OK) return crc;
            }
        }
        if (out_rgb8) {
            const int crc = rth::copy_to_host(out_rgb8, d_out8, n_values, nullptr);
            if (crc != RT_OK) return crc;
        }
        const int src = read_status(scene, nullptr);
        if (stats) {
            HIP_TRY(hipMemcpy(stats, d_stats, sizeof(RtStats), hipMemcpyDeviceToHost));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            stats->kernel_ms = ms;
            if (src != RT_OK) stats->status_flags |= 1ull << (unsigned)(-src);
        }
        return src;
    };
    const int rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_stats) (void)hipFree(d_stats);
    return rc;
}

static int render_to_host(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, double* out_rgb,
                          uint8_t* out_rgb8, RtStats* stats) {
    if (!out_rgb && !out_rgb8) return RT_ERR_INVALID_ARG;
    int rc = check_camera(cam, depth);
    if (rc != RT_OK) return rc;
    RtScene* scene = nullptr;
    rc = rt_hip_scene_create(desc, &scene);
    if (rc != RT_OK) return rc;
    rc = scene_render_to_host(scene, cam, depth, out_rgb, out_rgb8, stats);
    rt_hip_scene_destroy(scene);
    return rc;
}

int rt_hip_scene_render(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, double* out_rgb, RtStats* stats) {
    if (!out_rgb) return RT_ERR_INVALID_ARG;
    return scene_render_to_host(scene, cam, depth, out_rgb, nullptr, stats);
}

int rt_hip_scene_render_rgb8(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, uint8_t* out_rgb8,
                             RtStats* stats) {
    if (!out_rgb8) return RT_ERR_INVALID_ARG;
    return scene_render_to_host(scene, cam, depth, nullptr, out_rgb8, stats);
}

int rt_hip_render(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, double* out_rgb, RtStats* stats) {
    if (!out_rgb) return RT_ERR_INVALID_ARG;
    return render_to_host(desc, cam, depth, out_rgb, nullptr, stats);
}

int rt_hip_render_rgb8(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, uint8_t* out_rgb8,
                       RtStats* stats) {
    if (!out_rgb8) return RT_ERR_INVALID_ARG;
    return render_to_host(desc, cam, depth, nullptr, out_rgb8, stats);
}

int rt_hip_color_at(const RtScene* scene, const double* rays, uint32_t n, uint32_t depth, double* out_rgb) {
    if (!scene || (n && (!rays || !out_rgb))) return RT_ERR_INVALID_ARG;
    if (depth > RT_MAX_DEPTH) return RT_ERR_DEPTH;
    RT_ON_DEVICE_OF(scene);
    if (n == 0) return RT_OK;
    double *d_rays = nullptr, *d_out = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_rays), (size_t)n * 6u * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 3u * sizeof(double)));
        HIP_TRY(hipMemcpy(d_rays, rays, (size_t)n * 6u * sizeof(double), hipMemcpyHostToDevice));
        if (scene->dev.flags & RT_SCENE_FLAT)
            hipLaunchKernelGGL(color_at_kernel<RT_CLASS_FLAT>, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, scene->dev,
                               d_rays, n, (int)depth, d_out, scene->d_status);
        else if (scene->dev.flags & RT_SCENE_NESTED_LEAVES)
            hipLaunchKernelGGL(color_at_kernel<RT_CLASS_NESTED_LEAVES>, dim3((n + 255u) / 256u), dim3(256), 0, nullptr,
                               scene->dev, d_rays, n, (int)depth, d_out, scene->d_status);
        else
            hipLaunchKernelGGL(color_at_kernel<RT_CLASS_GENERAL>, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, scene->dev,
                               d_rays, n, (int)depth, d_out, scene->d_status);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out_rgb, d_out, (size_t)n * 3u * sizeof(double), hipMemcpyDeviceToHost));
        return read_status(scene, nullptr);
    };
    const int rc = body();
    if (d_rays) (void)hipFree(d_rays);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int rt_hip_shade_rays(const RtScene* scene, const RtLight* light, const double* rays, uint32_t n, double* out_rgb,
                      uint8_t* out_hit) {
    if (!scene || !light || (n && (!rays || !out_rgb))) return RT_ERR_INVALID_ARG;
    RT_ON_DEVICE_OF(scene);
    if (n == 0) return RT_OK;
    double *d_rays = nullptr, *d_out = nullptr;
    uint8_t* d_hit = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_rays), (size_t)n * 6u * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 3u * sizeof(double)));
        if (out_hit) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_hit), (size_t)n));
        HIP_TRY(hipMemcpy(d_rays, rays, (size_t)n * 6u * sizeof(double), hipMemcpyHostToDevice));
        const dim3 grid((n + 255u) / 256u), block(256);
        if (scene->dev.flags & RT_SCENE_FLAT)
            hipLaunchKernelGGL(shade_rays_kernel<RT_CLASS_FLAT>, grid, block, 0, nullptr, scene->dev, *light, d_rays, n, d_out,
                               d_hit, scene->d_status);
        else if (scene->dev.flags & RT_SCENE_NESTED_LEAVES)
            hipLaunchKernelGGL(shade_rays_kernel<RT_CLASS_NESTED_LEAVES>, grid, block, 0, nullptr, scene->dev, *light, d_rays,
                               n, d_out, d_hit, scene->d_status);
        else
            hipLaunchKernelGGL(shade_rays_kernel<RT_CLASS_GENERAL>, grid, block, 0, nullptr, scene->dev, *light, d_rays, n,
                               d_out, d_hit, scene->d_status);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out_rgb, d_out, (size_t)n * 3u * sizeof(double), hipMemcpyDeviceToHost));
        if (out_hit) HIP_TRY(hipMemcpy(out_hit, d_hit, (size_t)n, hipMemcpyDeviceToHost));
        return read_status(scene, nullptr);
    };
    const int rc = body();
    if (d_rays) (void)hipFree(d_rays);
    if (d_out) (void)hipFree(d_out);
    if (d_hit) (void)hipFree(d_hit);
    return rc;
}

int rt_hip_probe(const RtScene* scene, const double* rays, uint32_t n, double* out) {
    if (!scene || (n && (!rays || !out))) return RT_ERR_INVALID_ARG;
    RT_ON_DEVICE_OF(scene);
    if (n == 0) return RT_OK;
    double *d_rays = nullptr, *d_out = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_rays), (size_t)n * 6u * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 16u * sizeof(double)));
        HIP_TRY(hipMemcpy(d_rays, rays, (size_t)n * 6u * sizeof(double), hipMemcpyHostToDevice));
        if (scene->dev.flags & RT_SCENE_FLAT)
            hipLaunchKernelGGL(probe_kernel<RT_CLASS_FLAT>, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, scene->dev, d_rays,
                               n, d_out, scene->d_status);
        else if (scene->dev.flags & RT_SCENE_NESTED_LEAVES)
            hipLaunchKernelGGL(probe_kernel<RT_CLASS_NESTED_LEAVES>, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, scene->dev,
                               d_rays, n, d_out, scene->d_status);
        else
            hipLaunchKernelGGL(probe_kernel<RT_CLASS_GENERAL>, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, scene->dev,
                               d_rays, n, d_out, scene->d_status);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 16u * sizeof(double), hipMemcpyDeviceToHost));
        return read_status(scene, nullptr);
    };
    const int rc = body();
    if (d_rays) (void)hipFree(d_rays);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int rt_hip_quantize(const double* d_rgb, uint64_t n_values, uint8_t* d_out, void* hip_stream) {
    if (n_values && (!d_rgb || !d_out)) return RT_ERR_INVALID_ARG;
    if (n_values == 0) return RT_OK;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        const int rc = ensure_device();
        if (rc != RT_OK) return rc;
    }
    uint64_t blocks = (n_values + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), d_rgb,
                       (unsigned long long)n_values, d_out);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"
