// rt_multi.hip -- rt_hip_render_multi: one Camera::render split over the HIP devices of ONE process.
//
// The frame's rows are dealt to the devices in interleaved bands (the same RowMap the per-process shards of
// rt_hip_render_rows use: no pixel depends on another, camera.rs:54-62), every device holds a copy of the scene and
// traces its rows on a stream of its own, driven by a host thread of its own.  The only exchange is the gather of the
// finished shards to device 0: one ncclSend per peer and the matching ncclRecvs in one group on device 0 (RCCL over
// xGMI, loaded at run time from librccl.so), or -- when RCCL cannot be had -- hipMemcpyPeerAsync; device 0 scatters
// the bands into the frame (one strided device copy per peer) and the frame crosses PCIe once.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_host_internal.h"

namespace {

// ---- RCCL, resolved at run time so that the library loads on machines without it -------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& r = g_rccl;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        void* lib = r.lib;
        auto sym = [lib](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, name));
            return fn != nullptr;
        };
        r.ok = sym(r.CommInitAll, "ncclCommInitAll") && sym(r.CommDestroy, "ncclCommDestroy") && sym(r.GroupStart, "ncclGroupStart") &&
               sym(r.GroupEnd, "ncclGroupEnd") && sym(r.Send, "ncclSend") && sym(r.Recv, "ncclRecv") &&
               sym(r.GetErrorString, "ncclGetErrorString");
    });
    return g_rccl;
}

// ---- what the process keeps per device count: communicators, streams, shard buffers ---------------------------------
struct PerDevice {
    hipStream_t stream = nullptr;
    double* shard = nullptr;  // this device's rows (device 0: also the landing area of the peers' shards, then the frame)
    size_t shard_values = 0;
    ncclComm_t comm = nullptr;
    bool peer_ok = false;  // device 0 may read this device's memory (hipMemcpyPeerAsync path)
    // kept across calls: this device's copy of the scene (while the caller keeps handing in the same tables: `scene_key`
    // is a hash of their contents), a block for the counters and the two events that time the launch
    RtScene* scene = nullptr;
    uint64_t scene_key = 0;
    RtStats* d_stats = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
struct Multi {
    uint32_t n = 0;
    uint32_t n_physical = 1;  // logical device d runs on HIP device d % n_physical (more logical than physical: the test knob below)
    bool use_rccl = false;
    std::vector<PerDevice> dev;
    double* landing = nullptr;  // device 0: the peers' shards, contiguous, peer after peer
    size_t landing_values = 0;
    double* frame = nullptr;    // device 0
    size_t frame_values = 0;
};
std::mutex g_multi_mutex;  // one rt_hip_render_multi at a time
Multi g_multi;

void multi_release_locked() {
    Multi& m = g_multi;
    int saved = 0;
    (void)hipGetDevice(&saved);
    for (uint32_t d = 0; d < m.dev.size(); ++d) {
        if (hipSetDevice((int)(d % m.n_physical)) != hipSuccess) continue;
        PerDevice& p = m.dev[d];
        if (p.scene) rt_hip_scene_destroy(p.scene);
        if (p.d_stats) (void)hipFree(p.d_stats);
        if (p.e0) (void)hipEventDestroy(p.e0);
        if (p.e1) (void)hipEventDestroy(p.e1);
        if (p.comm && rccl().ok) (void)rccl().CommDestroy(p.comm);
        if (p.stream) (void)hipStreamDestroy(p.stream);
        rth::block_free(p.shard);
        if (d == 0) rth::block_free(m.landing), rth::block_free(m.frame);
    }
    (void)hipSetDevice(saved);
    m = Multi();
}

int fail_text(const std::string& text) {
    return rth::hip_fail(hipErrorUnknown, text.c_str());
}

int multi_prepare(uint32_t n, uint32_t n_physical) {
    Multi& m = g_multi;
    if (m.n == n && m.n_physical == n_physical) return RT_OK;
    multi_release_locked();
    m.dev.resize(n);
    m.n_physical = n_physical;
    const char* how = std::getenv("RT_HIP_MULTI_GATHER");  // "peer": hipMemcpyPeerAsync even when RCCL is there
    m.use_rccl = n > 1 && n <= n_physical && rccl().ok && !(how && std::strcmp(how, "peer") == 0);
    for (uint32_t d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice((int)(d % n_physical)));
        HIP_TRY(hipStreamCreateWithFlags(&m.dev[d].stream, hipStreamNonBlocking));
    }
    if (m.use_rccl) {
        std::vector<int> list(n);
        std::vector<ncclComm_t> comms(n, nullptr);
        for (uint32_t d = 0; d < n; ++d) list[d] = (int)d;
        const ncclResult_t r = rccl().CommInitAll(comms.data(), (int)n, list.data());
        if (r != ncclSuccess) {
            (void)hipGetLastError();
            m.use_rccl = false;  // (peer copies below)
        } else {
            for (uint32_t d = 0; d < n; ++d) m.dev[d].comm = comms[d];
        }
    }
    if (!m.use_rccl && n > 1) {
        HIP_TRY(hipSetDevice(0));
        for (uint32_t d = 1; d < n && d < n_physical; ++d) {
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, 0, (int)d));
            if (can) {
                const hipError_t e = hipDeviceEnablePeerAccess((int)d, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
                (void)hipGetLastError();
            }
            m.dev[d].peer_ok = can != 0;  // (hipMemcpyPeerAsync works without peer access too, through the host)
        }
    }
    m.n = n;
    return RT_OK;
}

int grow(double** p, size_t* have, size_t want) {
    if (*have >= want) return RT_OK;
    rth::block_free(*p);
    *p = nullptr, *have = 0;
    void* q = nullptr;
    const int rc = rth::block_alloc(&q, want * sizeof(double));
    if (rc != RT_OK) return rc;
    *p = static_cast<double*>(q), *have = want;
    return RT_OK;
}

// A 64-bit hash of everything a scene's tables say (counts and contents): a device keeps its copy of the scene while the
// caller keeps handing in the same tables (a fly-through, a re-render: the camera is not part of it).  8 bytes a step,
// ~1 ms for the 4 MB of a 10 000-primitive scene, against 6 ms per device to validate, lay out and upload them again.
uint64_t hash_bytes(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    for (; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}
uint64_t scene_key_of(const RtSceneDesc& d) {
    uint64_t h = 0xCBF29CE484222325ull;
    auto table = [&](uint32_t n, const void* p, size_t each) {
        h = hash_bytes(h, &n, sizeof n);
        if (n && p) h = hash_bytes(h, p, (size_t)n * each);
    };
    table(d.n_shapes, d.shapes, sizeof(RtShape)), table(d.n_conics, d.conics, sizeof(RtConic));
    table(d.n_triangles, d.triangles, sizeof(RtTriangle)), table(d.n_groups, d.groups, sizeof(RtGroup));
    table(d.n_csgs, d.csgs, sizeof(RtCsg)), table(d.n_lhit, d.lhit, 1), table(d.n_materials, d.materials, sizeof(RtMaterial));
    table(d.n_patterns, d.patterns, sizeof(RtPattern)), table(d.n_lights, d.lights, sizeof(RtLight));
    table(d.n_bvh_nodes, d.bvh_nodes, sizeof(RtBvhNode)), table(d.n_dir_cands, d.dir_cands, sizeof(RtDirCand));
    h = hash_bytes(h, &d.world_bvh_root, sizeof d.world_bvh_root), h = hash_bytes(h, &d.dir_res, sizeof d.dir_res);
    if (d.dir_cell_start) {
        table((uint32_t)((size_t)6 * d.dir_res * d.dir_res + 1), d.dir_cell_start, sizeof(uint32_t));
        table(d.n_dir_entries, d.dir_entries, sizeof(uint32_t));
    }
    return h | 1ull;  // (0 = no scene kept)
}

// Where the device threads of one call agree on how the call goes on: every thread arrives once with its own outcome so
// far, and all leave knowing whether EVERY device got that far.  The gather is a collective -- device 0 posts a receive
// per peer, every peer a send -- so a device that failed before it (out of memory, a refused launch) must keep ALL the
// others from entering it: a receive whose send never comes blocks device 0's stream for good.
struct Gate {
    std::mutex m;
    std::condition_variable cv;
    uint32_t n, arrived = 0;
    bool all_ok = true;
    explicit Gate(uint32_t n_) : n(n_) {}
    bool arrive(bool ok) {
        std::unique_lock<std::mutex> lock(m);
        all_ok = all_ok && ok;
        if (++arrived == n) cv.notify_all();
        else cv.wait(lock, [this] { return arrived == n; });
        return all_ok;
    }
};

// rows of part `part` when the frame's `vsize` rows are dealt in bands of `band_rows` to `n` parts (RowMap's layout)
uint32_t rows_of_part(uint32_t vsize, uint32_t band_rows, uint32_t n, uint32_t part) {
    const uint32_t n_bands = (vsize + band_rows - 1u) / band_rows;
    uint32_t rows = 0;
    for (uint32_t b = part; b < n_bands; b += n) {
        const uint32_t r0 = b * band_rows;
        rows += r0 + band_rows <= vsize ? band_rows : vsize - r0;
    }
    return rows;
}

}  // namespace

void rth::multi_shutdown() {
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    multi_release_locked();
}

int rth::rccl_available() { return rccl().ok ? 1 : 0; }

extern "C" const char* rt_hip_multi_transport(void) {
    const char* how = std::getenv("RT_HIP_MULTI_GATHER");
    return rccl().ok && !(how && std::strcmp(how, "peer") == 0) ? "rccl" : "peer";
}

extern "C" int rt_hip_render_multi(const RtSceneDesc* desc, const RtCameraDesc* cam, uint32_t depth, uint32_t n_devices,
                                   double* out_rgb, RtStats* stats) {
    if (!desc || !cam || !out_rgb || n_devices == 0) return RT_ERR_INVALID_ARG;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return RT_ERR_NO_DEVICE;
    // RT_HIP_MULTI_ALIAS=1 (tests on a machine with fewer devices): logical device d runs on HIP device d % the device
    // count -- the same shards, landing area and scatter, gathered by peer copies
    const char* alias = std::getenv("RT_HIP_MULTI_ALIAS");
    if (n_devices > (uint32_t)have && !(alias && *alias == '1' && n_devices <= 64u)) return RT_ERR_INVALID_ARG;
    const uint32_t n_physical = n_devices < (uint32_t)have ? n_devices : (uint32_t)have;
    if (cam->hsize == 0 || cam->vsize == 0) return RT_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_multi_mutex);
    int saved = 0;
    HIP_TRY(hipGetDevice(&saved));
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore{saved};
    int rc = multi_prepare(n_devices, n_physical);
    if (rc != RT_OK) return rc;
    Multi& m = g_multi;
    const uint32_t n = n_devices;
    const uint32_t W = cam->hsize, H = cam->vsize;
    const uint32_t band_rows = 8;  // whole 8-row tiles of the tracing kernels; fine bands even out sky and ground
    const size_t row_values = (size_t)W * 3u;
    std::vector<uint32_t> part_rows(n);
    size_t peers_values = 0;
    for (uint32_t d = 0; d < n; ++d) {
        part_rows[d] = rows_of_part(H, band_rows, n, d);
        if (d) peers_values += (size_t)part_rows[d] * row_values;
    }
    HIP_TRY(hipSetDevice(0));
    if ((rc = grow(&m.frame, &m.frame_values, (size_t)H * row_values)) != RT_OK) return rc;
    if (n > 1 && (rc = grow(&m.landing, &m.landing_values, peers_values)) != RT_OK) return rc;

    // ---- every device: scene, shard, trace, send --------------------------------------------------------------------
    std::vector<int> result(n, RT_OK);
    std::vector<std::string> hip_text(n);
    std::vector<RtStats> dev_stats(n);
    const uint64_t key = scene_key_of(*desc);
    Gate enqueued(n);
    // RT_HIP_MULTI_FAIL_DEVICE=d (test knob): logical device d reports RT_ERR_INVALID_ARG instead of launching
    const char* fail_env = std::getenv("RT_HIP_MULTI_FAIL_DEVICE");
    const long fail_device = fail_env && *fail_env ? std::atol(fail_env) : -1;
    auto device_work = [&](uint32_t d) {
        PerDevice& p = m.dev[d];
        bool launched = false;
        // up to the launch: whatever goes wrong here, the thread still arrives at the gate
        auto enqueue = [&]() -> int {
            HIP_TRY(hipSetDevice((int)(d % n_physical)));
            if ((long)d == fail_device) return RT_ERR_INVALID_ARG;
            if (p.scene && p.scene_key != key) {
                rt_hip_scene_destroy(p.scene);
                p.scene = nullptr, p.scene_key = 0;
            }
            if (!p.scene) {
                const int r = rt_hip_scene_create(desc, &p.scene);
                if (r != RT_OK) return r;
                p.scene_key = key;
            }
            int r = grow(&p.shard, &p.shard_values, (size_t)part_rows[d] * row_values);
            if (r != RT_OK) return r;
            if (stats) {
                if (!p.d_stats) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p.d_stats), sizeof(RtStats)));
                if (!p.e0) HIP_TRY(hipEventCreate(&p.e0));
                if (!p.e1) HIP_TRY(hipEventCreate(&p.e1));
                HIP_TRY(hipMemsetAsync(p.d_stats, 0, sizeof(RtStats), p.stream));
                HIP_TRY(hipEventRecord(p.e0, p.stream));
            }
            launched = true;
            r = rt_hip_render_rows(p.scene, cam, depth, d, n, band_rows, part_rows[d], p.shard, p.stream, stats ? p.d_stats : nullptr);
            if (r == RT_OK && stats) HIP_TRY(hipEventRecord(p.e1, p.stream));
            return r;
        };
        int r = enqueue();
        const bool everyone = enqueued.arrive(r == RT_OK);
        // the gather: peers send their shard, device 0 receives them all, peer after peer, in one group -- entered by all
        // devices or by none
        if (everyone && n > 1 && m.use_rccl) {
            const Rccl& R = rccl();
            ncclResult_t q = R.GroupStart();
            if (d != 0) {
                if (q == ncclSuccess) q = R.Send(p.shard, (size_t)part_rows[d] * row_values, ncclDouble, 0, p.comm, p.stream);
            } else {
                size_t at = 0;
                for (uint32_t peer = 1; peer < n && q == ncclSuccess; ++peer) {
                    const size_t count = (size_t)part_rows[peer] * row_values;
                    q = R.Recv(m.landing + at, count, ncclDouble, (int)peer, p.comm, p.stream);
                    at += count;
                }
            }
            const ncclResult_t qe = R.GroupEnd();
            if (q == ncclSuccess) q = qe;
            if (q != ncclSuccess) r = fail_text(std::string("RCCL gather: ") + R.GetErrorString(q));
        }
        if (launched) {  // (also when another device failed: nothing of this call may still be running when it returns)
            const hipError_t e = hipStreamSynchronize(p.stream);
            if (e != hipSuccess && r == RT_OK) r = rth::hip_fail(e, "hipStreamSynchronize(device stream)");
            if (r == RT_OK) r = rt_hip_scene_status(p.scene, p.stream);
        }
        if (stats && launched && (r == RT_OK || r == RT_ERR_LIST_OVERFLOW || r == RT_ERR_NAN_T)) {
            (void)hipMemcpy(&dev_stats[d], p.d_stats, sizeof(RtStats), hipMemcpyDeviceToHost);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) dev_stats[d].kernel_ms = ms;
        }
        result[d] = r;
        if (r == RT_ERR_HIP) hip_text[d] = rt_hip_last_hip_error();
    };
    {
        std::vector<std::thread> threads;
        for (uint32_t d = 1; d < n; ++d) threads.emplace_back(device_work, d);
        device_work(0);
        for (std::thread& t : threads) t.join();
    }
    rc = RT_OK;
    for (uint32_t d = 0; d < n && rc == RT_OK; ++d)
        if (result[d] != RT_OK) {
            rc = result[d];
            if (rc == RT_ERR_HIP) (void)rth::hip_fail(hipErrorUnknown, ("device " + std::to_string(d) + ": " + hip_text[d]).c_str());
        }

    // ---- device 0: peers' shards (when not sent), bands into the frame, frame to the caller ---------------------------
    auto finish = [&]() -> int {
        HIP_TRY(hipSetDevice(0));
        hipStream_t s0 = m.dev[0].stream;
        if (n > 1 && !m.use_rccl) {
            size_t at = 0;
            for (uint32_t peer = 1; peer < n; ++peer) {
                const size_t count = (size_t)part_rows[peer] * row_values;
                HIP_TRY(hipMemcpyPeerAsync(m.landing + at, 0, m.dev[peer].shard, (int)(peer % n_physical), count * sizeof(double), s0));
                at += count;
            }
        }
        // part d's k-th band of rows is the frame's band d + k n: a strided copy per part (and the ragged last band)
        const size_t band_bytes = (size_t)band_rows * row_values * sizeof(double);
        size_t at = 0;
        for (uint32_t d = 0; d < n; ++d) {
            const double* src = d == 0 ? m.dev[0].shard : m.landing + at;
            if (d) at += (size_t)part_rows[d] * row_values;
            const uint32_t whole = part_rows[d] / band_rows, ragged = part_rows[d] % band_rows;
            double* dst = m.frame + (size_t)d * band_rows * row_values;
            if (whole)
                HIP_TRY(hipMemcpy2DAsync(dst, band_bytes * n, src, band_bytes, band_bytes, whole, hipMemcpyDeviceToDevice, s0));
            if (ragged)
                HIP_TRY(hipMemcpyAsync(dst + (size_t)whole * n * band_rows * row_values, src + (size_t)whole * band_rows * row_values,
                                       (size_t)ragged * row_values * sizeof(double), hipMemcpyDeviceToDevice, s0));
        }
        return rth::copy_to_host(out_rgb, m.frame, (size_t)H * row_values * sizeof(double), s0);
    };
    if (rc == RT_OK || rc == RT_ERR_LIST_OVERFLOW || rc == RT_ERR_NAN_T) {
        const int frc = finish();
        if (frc != RT_OK) rc = frc;
    }
    if (stats) {  // counters add up over the devices; the kernel time is the slowest device's
        std::memset(stats, 0, sizeof *stats);
        static_assert(sizeof(RtStats) % sizeof(uint64_t) == 0, "RtStats is a row of 64-bit counters and one double");
        for (uint32_t d = 0; d < n; ++d) {
            const double ms = stats->kernel_ms > dev_stats[d].kernel_ms ? stats->kernel_ms : dev_stats[d].kernel_ms;
            const uint64_t flags = stats->status_flags | dev_stats[d].status_flags;
            uint64_t* acc = reinterpret_cast<uint64_t*>(stats);
            const uint64_t* add = reinterpret_cast<const uint64_t*>(&dev_stats[d]);
            for (size_t k = 0; k < sizeof(RtStats) / sizeof(uint64_t); ++k) acc[k] += add[k];
            stats->kernel_ms = ms;
            stats->status_flags = flags;
        }
    }
    return rc;
}
