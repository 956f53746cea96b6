// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets the kernel library's sources (raytracer_amd/csrc/*.hip, *.h) be compiled
// as plain C++ for the host, so that the SAME kernel source can be run on the CPU under -fsanitize=address,undefined
// (tests/emu/Makefile, tests/emu/emu_main.cpp): every out-of-bounds index into a per-lane array, a queue, a node pool
// or a table block shows up as a sanitizer report instead of a GPU memory fault.  Only tests/ builds or runs it; the
// product library is built by hipcc from the real header and has no CPU path.
//
// Two execution models, chosen per launch (hip_emu::lockstep):
//  * sequential (default): a launch runs its blocks one after the other and, inside a block, its threads one after the
//    other, each to completion.  Every thread is a wave of its own in which only its own lane is active:
//    ballot(x) = x << lane, __shfl(v, own lane) = v.  A legal GPU schedule for kernels whose lanes do not rely on each
//    other's stores inside one launch -- render_kernel / wf_level_kernel / color_at_kernel / probe_kernel of scenes that
//    are not staged in LDS (lanes of these kernels leave loops one by one with collectives inside, which the other model
//    cannot follow).
//  * lockstep: blocks one after the other, but the threads of a block run as real host threads and every wave collective
//    (ballot, __shfl*, readfirstlane, wave barrier) is a rendezvous of the wave's 64 lanes, __syncthreads one of the
//    block's threads.  For kernels written so that ALL lanes of a wave reach every collective together (the kernel
//    pipeline of rt_wf2.h: its loops are wave-uniform, lanes diverge only between collectives).  A collective reached by
//    a subset of the wave is reported after a time-out instead of hanging.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define RT_HOST_EMU 1
#define __host__
#define __device__ inline
#define __global__
#define __forceinline__ __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct __attribute__((aligned(8))) float2 { float x, y; };

// ---- wave intrinsics ------------------------------------------------------------------------------------------------
void hip_emu_collective_timeout(int what, unsigned arrived, unsigned n);  // (the driver: reports and aborts)
namespace hip_emu {
enum { kBallot, kShfl, kShflUp, kShflDown, kFirstLane, kWaveBarrier, kSyncThreads };
inline bool lockstep = false;  // the model of the launches that follow (see above)
// A rendezvous of `n` threads with a 64-bit payload per thread; payloads are double-buffered by generation, so a fast thread
// writing the next collective's payload cannot disturb a slow one still reading this one's.
struct Meet {
    std::mutex m;
    std::condition_variable cv;
    unsigned n = 64, arrived = 0;
    unsigned long gen = 0;
    unsigned long long vals[2][256];
    // returns the buffer holding every thread's payload of this collective
    const unsigned long long* exchange(unsigned me, unsigned long long payload, int what) {
        std::unique_lock<std::mutex> lock(m);
        const unsigned long my_gen = gen;
        unsigned long long* buf = vals[my_gen & 1ul];
        buf[me] = payload;
        if (++arrived == n) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else if (!cv.wait_for(lock, std::chrono::seconds(120), [&] { return gen != my_gen; })) {
            hip_emu_collective_timeout(what, arrived, n);
        }
        return buf;
    }
};
inline thread_local Meet* wave = nullptr;   // null: the sequential model
inline thread_local Meet* block = nullptr;
inline unsigned lane() { return threadIdx.x & 63u; }
inline unsigned long long ballot(bool x) {
    if (!wave) return x ? 1ull << lane() : 0ull;
    const unsigned long long* v = wave->exchange(lane(), x ? 1ull : 0ull, kBallot);
    unsigned long long r = 0;
    for (unsigned l = 0; l < 64u; ++l) r |= v[l] << l;
    return r;
}
template <class T> inline unsigned long long pack(T v) { unsigned long long b = 0; static_assert(sizeof(T) <= 8, "payload"); std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T unpack(unsigned long long b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
template <class T> inline T shfl(T v, int src) {
    if (!wave) {
        if ((unsigned)src != lane()) std::abort();  // a read from another lane: not in the sequential model
        return v;
    }
    return unpack<T>(wave->exchange(lane(), pack(v), kShfl)[(unsigned)src & 63u]);
}
template <class T> inline T shfl_up(T v, unsigned off) {
    if (!wave) return off == 0 ? v : T(0);
    const unsigned long long* b = wave->exchange(lane(), pack(v), kShflUp);
    return lane() >= off ? unpack<T>(b[lane() - off]) : v;
}
template <class T> inline T shfl_down(T v, unsigned off) {
    if (!wave) return off == 0 ? v : T(0);  // (the other lanes hold nothing)
    const unsigned long long* b = wave->exchange(lane(), pack(v), kShflDown);
    return lane() + off < 64u ? unpack<T>(b[lane() + off]) : v;
}
template <class T> inline T shfl_xor(T v, unsigned mask) {
    if (!wave) { if (mask != 0u) std::abort(); return v; }
    return unpack<T>(wave->exchange(lane(), pack(v), kShfl)[(lane() ^ mask) & 63u]);
}
template <class T> inline T first_lane(T v) { return wave ? unpack<T>(wave->exchange(lane(), pack(v), kFirstLane)[0]) : v; }
inline void wave_barrier() { if (wave) (void)wave->exchange(lane(), 0ull, kWaveBarrier); }
inline void block_barrier() { if (block) (void)block->exchange(threadIdx.x, 0ull, kSyncThreads); }
}  // namespace hip_emu
#define __builtin_amdgcn_ballot_w64(x) hip_emu::ballot(x)
#define __builtin_amdgcn_readfirstlane(x) hip_emu::first_lane(x)
#define __builtin_amdgcn_readlane(x, l) hip_emu::shfl((x), (l))
#define __builtin_amdgcn_wave_barrier() hip_emu::wave_barrier()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(x) std::this_thread::sleep_for(std::chrono::microseconds(20))
#define __builtin_amdgcn_s_memtime() 0ull
inline unsigned __lane_id() { return hip_emu::lane(); }
inline void __syncthreads() { hip_emu::block_barrier(); }
template <class T> inline T __shfl(T v, int src, int = 64) { return hip_emu::shfl(v, src); }
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return hip_emu::shfl_xor(v, (unsigned)mask); }
template <class T> inline T __shfl_down(T v, unsigned off, int = 64) { return hip_emu::shfl_down(v, off); }
template <class T> inline T __shfl_up(T v, unsigned off, int = 64) { return hip_emu::shfl_up(v, off); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }

// ---- conversions ----------------------------------------------------------------------------------------------------
inline long long __double_as_longlong(double x) { long long r; std::memcpy(&r, &x, 8); return r; }
inline double __longlong_as_double(long long x) { double r; std::memcpy(&r, &x, 8); return r; }
inline float __uint_as_float(unsigned x) { float r; std::memcpy(&r, &x, 4); return r; }
inline unsigned __float_as_uint(float x) { unsigned r; std::memcpy(&r, &x, 4); return r; }
inline float __double2float_ru(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, INFINITY); return f; }
inline float __double2float_rd(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -INFINITY); return f; }

// ---- runtime API: "device memory" is the heap (so the sanitizer sees every block), streams and events are nothing ------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999, hipErrorPeerAccessAlreadyEnabled = 704, hipErrorOutOfMemory = 2 };
typedef struct hip_emu_stream* hipStream_t;
typedef struct hip_emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
inline const char* hipGetErrorString(hipError_t) { return "hip_emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255u) & ~(size_t)255u); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)4 << 30, *t = (size_t)8 << 30; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }  // two "compute units": small persistent grids
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
inline hipError_t hipGetSymbolAddress(void** p, const void* symbol) { *p = const_cast<void*>(symbol); return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* out, K, int, size_t) { *out = 1; return hipSuccess; }

// ---- launches ---------------------------------------------------------------------------------------------------------
namespace hip_emu {
// 256 host threads that run the threads of one block at a time (lockstep model)
struct Pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable start, done;
    unsigned long job = 0;
    unsigned n_active = 0, remaining = 0;
    bool stop = false;
    std::function<void(unsigned)> fn;
    Pool() {
        for (unsigned t = 0; t < 256u; ++t) threads.emplace_back([this, t] { run(t); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lock(m);
            stop = true;
        }
        start.notify_all();
        for (std::thread& th : threads) th.join();
    }
    void run(unsigned t) {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            start.wait(lock, [&] { return stop || job != seen; });
            if (stop) return;
            seen = job;
            if (t >= n_active) continue;
            lock.unlock();
            fn(t);
            lock.lock();
            if (--remaining == 0) done.notify_one();
        }
    }
    void run_block(unsigned n, std::function<void(unsigned)> f) {
        std::unique_lock<std::mutex> lock(m);
        fn = std::move(f), n_active = remaining = n, ++job;
        start.notify_all();
        done.wait(lock, [&] { return remaining == 0; });
    }
};
inline Pool& pool() {
    static Pool p;
    return p;
}
}  // namespace hip_emu

template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*lds*/, hipStream_t /*stream*/, A... args) {
    const bool lock = hip_emu::lockstep && block.x % 64u == 0u && block.x <= 256u && block.y == 1u && block.z == 1u;
    if (lock) {
        static hip_emu::Meet waves[4], whole;
        for (unsigned bz = 0; bz < grid.z; ++bz)
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned bx = 0; bx < grid.x; ++bx) {
                    whole.n = block.x, whole.arrived = 0;
                    for (hip_emu::Meet& w : waves) w.n = 64, w.arrived = 0;
                    hip_emu::pool().run_block(block.x, [&](unsigned t) {
                        gridDim = grid, blockDim = block, blockIdx = dim3(bx, by, bz), threadIdx = dim3(t);
                        hip_emu::wave = &waves[t >> 6], hip_emu::block = &whole;
                        kernel(args...);
                        hip_emu::wave = nullptr, hip_emu::block = nullptr;
                    });
                }
        return;
    }
    gridDim = grid, blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx)
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx) {
                            blockIdx = dim3(bx, by, bz), threadIdx = dim3(tx, ty, tz);
                            kernel(args...);
                        }
}
