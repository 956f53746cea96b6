// rt_host_internal.h -- what the host-side translation units of librt_hip.so share (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "rt_kernels.h"

namespace rth {
// records the text for rt_hip_last_hip_error (thread-local) and returns RT_ERR_HIP
int hip_fail(hipError_t e, const char* what);
}  // namespace rth
#define HIP_TRY(expr)                                           \
    do {                                                        \
        const hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return rth::hip_fail(e_, #expr); \
    } while (0)

namespace rth {
// The host image of a scene's device block: pieces at 256-byte offsets, uploaded in one copy.
struct Arena {
    std::vector<unsigned char> host;
    size_t reserve(size_t bytes) {
        const size_t off = (host.size() + 255u) & ~(size_t)255u;
        host.resize(off + (bytes ? bytes : 16u), 0);
        return off;
    }
    size_t add(const void* src, size_t bytes) {
        const size_t off = reserve(bytes);
        if (bytes) std::memcpy(host.data() + off, src, bytes);
        return off;
    }
};
}  // namespace rth

struct Wf2;  // rt_wf2.hip: the 4-wide hierarchy and workspace of the kernel pipeline for large scenes

struct RtScene {
    rtd::SceneDev dev;      // host copy of the table pointers
    rtd::SceneDev* d_dev;   // the same record in device memory: what the kernels read
    unsigned* d_status;
    int device;
    unsigned char* tables;           // the one block everything above points into (rth::block_alloc)
    std::vector<void*> allocations;  // what a scene holds beyond it (device-built direction cells): hipFree
    // wave-front workspace (queues, node pool, counters), sized by the largest frame rendered so far;
    // launches on one scene are not re-entrant, so it is simply reused
    mutable rtd::WfSpace wf;
    mutable size_t wf_pixels;
    mutable double* d_frame;           // rt_hip_scene_render*: device framebuffers, grown to the largest frame asked for
    mutable uint8_t* d_frame8;
    mutable size_t frame_values;
    static constexpr unsigned kFrameBands = 4;  // a large frame goes to the host in bands of rows, copy over trace
    mutable hipEvent_t band_done[kFrameBands];
    int wf_mode;                       // rt_hip_scene_set_wavefront: -1 by policy, 0 one launch, 1 a launch per generation, 2 the kernel pipeline
    size_t wf_q_items, wf_node_items;  // capacities asked for (0: the defaults, per pixel of the frame)
    Wf2* wf2;                          // non-null when the scene qualifies for the kernel pipeline
};

namespace rth {
// rt_wf2.hip.  wf2_create leaves *out null (and returns RT_OK) for scenes the pipeline does not cover.
// Its tables go into the scene's arena; wf2_bind gives it the device address of the block once that exists.
int wf2_create(const RtSceneDesc& desc, uint32_t scene_flags, Arena& arena, Wf2** out);
const rtd::Wf2Accel* wf2_bind(Wf2* w, Arena& arena, unsigned char* base);  // -> the accel record on the device (null: no pipeline)
void wf2_destroy(Wf2* w);
// One frame (or row-band shard) through the pipeline, asynchronously on `stream`.  RT_ERR_HIP when the workspace
// cannot be had: the caller then falls back to the one-launch kernel (same frame).
int wf2_render(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, const rtk::RowMap& rows, double* d_out,
               hipStream_t stream, RtStats* d_stats);
const char* wf2_describe();
// The same frame with the workspace in the library's fixed device array (no allocation): rows are rendered in batches
// that fit it.  RT_ERR_NOALLOC_SIZE when not even one 8-row tile of this width fits.
#define RT_NOALLOC_SCRATCH_BYTES (128u << 20)
int wf2_render_noalloc(const RtScene* scene, const RtCameraDesc* cam, uint32_t depth, const rtk::RowMap& rows, double* d_out,
                       hipStream_t stream, RtStats* d_stats);
// hipGetSymbolAddress of that array for the CURRENT device, once per device (rt_hip_init, or the first call that needs it)
int scratch_init();
// rt_hostmem.hip: device blocks kept across calls (current device; block_free parks the block for the next
// block_alloc of a similar size, up to 1 GB / 16 blocks per process), the staged device -> pageable-host copy (ordered
// after the work `producer` has been given, or after the event `after`; blocking), and the release of both (rt_hip_shutdown).
int block_alloc(void** out, size_t bytes);
void block_free(void* p);
int copy_to_host(void* dst, const void* d_src, size_t bytes, hipStream_t producer, hipEvent_t after = nullptr);
void hostmem_shutdown();
// rt_multi.hip: what rt_hip_render_multi keeps (communicators, streams, buffers) released; whether librccl.so resolved
void multi_shutdown();
int rccl_available();
// rt_dirindex.hip: the cell table of the direction index built on the current device from the candidates' quadratic
// forms (blocking; default stream).  res_hint: the resolution the caller's tables were meant for; the library picks
// its own from the number of candidates (RT_HIP_DIR_RES in the environment overrides).
struct DirIndex {
    uint32_t res, n_entries;
    uint32_t *d_cell_start, *d_entries;
};
int dir_index_build(const RtDirCand* d_cands, uint32_t n_cands, uint32_t res_hint, DirIndex* out);
void dir_index_free(DirIndex* ix);
// rt_hip.hip: shade_hit's last line for the nodes of one generation (used by both wave-front forms)
void launch_combine(const rtd::WfSpace& ws, unsigned level, double* d_out, hipStream_t stream);
}  // namespace rth
