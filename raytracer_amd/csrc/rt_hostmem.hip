// rt_hostmem.hip -- what the library keeps across calls so that the drop-in pattern "one Camera::render per call"
// (rt_hip_render: tables in, frame out) pays for allocation and set-up once per process, not once per frame:
//   * a small cache of device blocks (scene tables, framebuffers) per device,
//   * the pinned staging buffer, copy stream, events and host copy threads of the device -> pageable-host path.
// Nothing here is part of the C ABI.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "rt_host_internal.h"

namespace {

// ---- device block cache ------------------------------------------------------------------------------------------
struct Block {
    void* p;
    size_t bytes;
    int device;
};
std::mutex g_block_mutex;
std::vector<Block> g_free_blocks;  // oldest first
std::vector<Block> g_live_blocks;
size_t g_cached_bytes = 0;
// per process; past that the oldest idle blocks go back to the driver.  Sized for the device: twice the kernel pipeline's
// largest workspace (rt_wf2.hip, RT_WF2_BUDGET_MB = 24 GiB of the 288) -- a block of more than half the limit is never kept.
constexpr size_t kMaxCachedBytes = (size_t)48 << 30;
constexpr size_t kMaxCachedBlocks = 32;

void evict_locked(size_t keep_bytes, size_t keep_blocks) {
    while (!g_free_blocks.empty() && (g_cached_bytes > keep_bytes || g_free_blocks.size() > keep_blocks)) {
        const Block b = g_free_blocks.front();
        g_free_blocks.erase(g_free_blocks.begin());
        g_cached_bytes -= b.bytes;
        (void)hipFree(b.p);
    }
}

// ---- host copy threads --------------------------------------------------------------------------------------------
// Persistent threads that move a range from the staging buffer to the caller's buffer chunk by chunk, each thread its
// slice of every chunk, as the calling thread announces the chunks' arrival (a fresh std::thread per slice costs
// ~50 us each, a condition-variable round trip per chunk ~30 us: here there is one wake-up per copy).
class CopyPool {
public:
    explicit CopyPool(unsigned n) {
        for (unsigned t = 0; t < n; ++t) threads_.emplace_back([this, t] { run(t); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread& th : threads_) th.join();
    }
    unsigned size() const { return (unsigned)threads_.size(); }
    // One caller at a time (the copier's mutex).  begin(): the threads start waiting for chunks; arrived(k): chunks
    // [0, k) are in the staging buffer; finish(): blocks until every announced byte is copied (call arrived(n_chunks)
    // or abandon() first).
    // `ring` chunks of staging: chunk c lies at src + (c % ring) * chunk
    void begin(void* dst, const void* src, size_t bytes, size_t chunk, size_t ring) {
        dst_ = static_cast<char*>(dst), src_ = static_cast<const char*>(src), bytes_ = bytes, chunk_ = chunk, ring_ = ring;
        for (Progress& p : progress_) p.chunks.store(0, std::memory_order_relaxed);
        ready_.store(0, std::memory_order_relaxed);
        abandoned_.store(false, std::memory_order_relaxed);
        if (size() == 0) return;
        {
            std::lock_guard<std::mutex> lock(m_);
            pending_ = size();
            ++generation_;
        }
        wake_.notify_all();
    }
    void arrived(size_t k) {
        if (size() == 0) {  // no threads on this machine: the caller copies
            for (size_t c = ready_.load(std::memory_order_relaxed); c < k; ++c) copy_slice(c, 0, 1);
        }
        ready_.store(k, std::memory_order_release);
    }
    // blocks until every thread has moved chunk c on (its slot of the staging ring may be written again)
    void wait_consumed(size_t c) {
        for (unsigned t = 0; t < size(); ++t) {
            unsigned spins = 0;
            while (progress_[t].chunks.load(std::memory_order_acquire) <= c && !abandoned_.load(std::memory_order_acquire)) nap(spins);
        }
    }
    void abandon() { abandoned_.store(true, std::memory_order_release); }
    void finish() {
        if (size() == 0) return;
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [this] { return pending_ == 0; });
    }

private:
    // A short spin (a chunk's DMA takes ~0.3 ms: most waits are longer), then sleeps: eight threads spinning through every
    // DMA of every large copy burn the cores the caller's other threads could use.
    static void nap(unsigned& spins) {
        if (++spins <= 256u) return;
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 4096u ? 20 : 200));
    }
    void copy_slice(size_t c, unsigned t, unsigned n) {
        const size_t off = c * chunk_, len = bytes_ - off < chunk_ ? bytes_ - off : chunk_;
        const size_t a = len * t / n / 64 * 64, b = t + 1 == n ? len : len * (t + 1) / n / 64 * 64;
        std::memcpy(dst_ + off + a, src_ + (c % ring_) * chunk_ + a, b - a);
    }
    void run(unsigned t) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            const size_t n_chunks = (bytes_ + chunk_ - 1) / chunk_;
            for (size_t c = 0; c < n_chunks; ++c) {
                unsigned spins = 0;
                while (ready_.load(std::memory_order_acquire) <= c && !abandoned_.load(std::memory_order_acquire)) nap(spins);
                if (ready_.load(std::memory_order_acquire) <= c) break;  // abandoned
                copy_slice(c, t, size());
                progress_[t].chunks.store(c + 1, std::memory_order_release);
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    bool stop_ = false;
    unsigned long generation_ = 0;
    unsigned pending_ = 0;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0, chunk_ = 1, ring_ = 1;
    struct alignas(64) Progress {
        std::atomic<size_t> chunks{0};  // chunks this thread has moved on
    };
    Progress progress_[8];  // (at most eight threads: pool())
    std::atomic<size_t> ready_{0};
    std::atomic<bool> abandoned_{false};
};

// ---- staged device -> pageable host copies ------------------------------------------------------------------------
constexpr size_t kChunk = 16u << 20;
constexpr size_t kRing = 8;  // chunks of pinned staging per device: 128 MB whatever the frame (an 8K f64 frame is 800 MB)
constexpr int kMaxDevices = 64;
struct Copier {
    std::mutex m;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;  // "the producer's stream has got this far"
    std::vector<hipEvent_t> done;  // one per chunk in flight
    void* staging = nullptr;
    size_t staging_bytes = 0;
};
Copier g_copier[kMaxDevices];
std::mutex g_pool_mutex, g_pool_use_mutex;
CopyPool* g_pool = nullptr;

CopyPool* pool() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (!g_pool) {
        unsigned n = std::thread::hardware_concurrency();
        if (const char* e = std::getenv("OMP_NUM_THREADS")) {  // the job's share of the machine, where one is stated
            const long v = std::atol(e);
            if (v > 0 && (unsigned long)v < n) n = (unsigned)v;
        }
        n = n < 2 ? 0 : (n > 8 ? 8 : n);
        g_pool = new CopyPool(n);
    }
    return g_pool;
}

}  // namespace

int rth::block_alloc(void** out, size_t bytes) {
    *out = nullptr;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (bytes == 0) bytes = 256;
    {
        std::lock_guard<std::mutex> lock(g_block_mutex);
        size_t best = g_free_blocks.size();
        for (size_t k = 0; k < g_free_blocks.size(); ++k) {
            const Block& b = g_free_blocks[k];
            if (b.device != dev || b.bytes < bytes || b.bytes > 2 * bytes + (1u << 20)) continue;
            if (best == g_free_blocks.size() || b.bytes < g_free_blocks[best].bytes) best = k;
        }
        if (best != g_free_blocks.size()) {
            const Block b = g_free_blocks[best];
            g_free_blocks.erase(g_free_blocks.begin() + (long)best);
            g_cached_bytes -= b.bytes;
            g_live_blocks.push_back(b);
            *out = b.p;
            return RT_OK;
        }
    }
    const size_t rounded = (bytes + 65535u) & ~(size_t)65535u;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, rounded);
    if (e != hipSuccess) {  // give the cache back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(g_block_mutex);
            evict_locked(0, 0);
        }
        e = hipMalloc(&p, rounded);
    }
    if (e != hipSuccess) return rth::hip_fail(e, "hipMalloc(device block)");
    std::lock_guard<std::mutex> lock(g_block_mutex);
    g_live_blocks.push_back(Block{p, rounded, dev});
    *out = p;
    return RT_OK;
}

void rth::block_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_block_mutex);
    for (size_t k = 0; k < g_live_blocks.size(); ++k) {
        if (g_live_blocks[k].p != p) continue;
        const Block b = g_live_blocks[k];
        g_live_blocks.erase(g_live_blocks.begin() + (long)k);
        if (b.bytes > kMaxCachedBytes / 2) {
            (void)hipFree(b.p);
        } else {
            g_free_blocks.push_back(b);
            g_cached_bytes += b.bytes;
            evict_locked(kMaxCachedBytes, kMaxCachedBlocks);
        }
        return;
    }
    (void)hipFree(p);  // not one of ours
}

void rth::hostmem_shutdown() {
    {
        std::lock_guard<std::mutex> lock(g_block_mutex);
        evict_locked(0, 0);
    }
    for (Copier& c : g_copier) {
        std::lock_guard<std::mutex> lock(c.m);
        if (c.staging) (void)hipHostFree(c.staging);
        for (hipEvent_t e : c.done) (void)hipEventDestroy(e);
        if (c.ready) (void)hipEventDestroy(c.ready);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        c.staging = nullptr, c.staging_bytes = 0, c.done.clear(), c.ready = nullptr, c.stream = nullptr;
    }
    CopyPool* idle = nullptr;
    {
        std::lock_guard<std::mutex> use(g_pool_use_mutex);  // (no copy is using the threads)
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        idle = g_pool;
        g_pool = nullptr;
    }
    delete idle;  // joins the copy threads (the next large copy starts new ones)
}

// Device -> the caller's (pageable) buffer.  A plain hipMemcpy into pageable memory runs at ~9 GB/s; here the DMA goes
// into the device's pinned staging buffer in 16 MB chunks (~55 GB/s) on the copier's own stream, ordered after what
// `producer` has been given so far (or, when `after` is given, after that event alone), while the host threads move finished chunks on to the caller's buffer.
int rth::copy_to_host(void* dst, const void* d_src, size_t bytes, hipStream_t producer, hipEvent_t after) {
    if (bytes == 0) return RT_OK;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (bytes < (4u << 20) || dev < 0 || dev >= kMaxDevices) {  // small: not worth the machinery
        if (after) HIP_TRY(hipEventSynchronize(after));
        else HIP_TRY(hipStreamSynchronize(producer));
        HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
        return RT_OK;
    }
    Copier& c = g_copier[dev];
    std::lock_guard<std::mutex> lock(c.m);
    if (!c.stream) HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    if (!c.ready) HIP_TRY(hipEventCreateWithFlags(&c.ready, hipEventDisableTiming));
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    const size_t ring = n_chunks < kRing ? n_chunks : kRing;
    if (c.staging_bytes < ring * kChunk) {
        if (c.staging) (void)hipHostFree(c.staging);
        c.staging = nullptr, c.staging_bytes = 0;
        if (hipHostMalloc(&c.staging, ring * kChunk, hipHostMallocDefault) != hipSuccess) {  // no pinned memory to be had
            (void)hipGetLastError();
            c.staging = nullptr;
            if (after) HIP_TRY(hipEventSynchronize(after));
            else HIP_TRY(hipStreamSynchronize(producer));
            HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
            return RT_OK;
        }
        c.staging_bytes = ring * kChunk;
    }
    while (c.done.size() < ring) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.done.push_back(e);
    }
    if (after) {
        HIP_TRY(hipStreamWaitEvent(c.stream, after, 0));
    } else {
        HIP_TRY(hipEventRecord(c.ready, producer));
        HIP_TRY(hipStreamWaitEvent(c.stream, c.ready, 0));
    }
    int rc = RT_OK;
    // chunk k goes through slot k % ring of the staging buffer: its DMA is issued once the host threads have moved chunk
    // k - ring on, so `ring - 1` DMAs are queued while the threads work on one chunk
    auto issue = [&](size_t k) {
        const size_t off = k * kChunk, len = bytes - off < kChunk ? bytes - off : kChunk;
        if (hipMemcpyAsync(static_cast<char*>(c.staging) + (k % ring) * kChunk, static_cast<const char*>(d_src) + off, len,
                           hipMemcpyDeviceToHost, c.stream) != hipSuccess ||
            hipEventRecord(c.done[k % ring], c.stream) != hipSuccess)
            rc = rth::hip_fail(hipGetLastError(), "staged device-to-host copy");
    };
    for (size_t k = 0; k < ring && rc == RT_OK; ++k) issue(k);
    if (rc == RT_OK) {
        CopyPool* workers = pool();
        std::lock_guard<std::mutex> pool_lock(g_pool_use_mutex);  // (copiers of different devices share the threads)
        workers->begin(dst, c.staging, bytes, kChunk, ring);
        for (size_t k = 0; k < n_chunks; ++k) {
            if (rc != RT_OK || hipEventSynchronize(c.done[k % ring]) != hipSuccess) {
                if (rc == RT_OK) rc = rth::hip_fail(hipGetLastError(), "hipEventSynchronize");
                workers->abandon();
                break;
            }
            workers->arrived(k + 1);
            if (k + ring < n_chunks) {
                workers->wait_consumed(k);
                issue(k + ring);
            }
        }
        workers->finish();
    }
    (void)hipStreamSynchronize(c.stream);
    return rc;
}
