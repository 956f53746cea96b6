// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets the kernel library's sources (raytracer_amd/csrc/*.hip, *.h) be compiled
// as plain C++ for the host, so that the SAME kernel source can be run on the CPU under -fsanitize=address,undefined
// (tests/emu/Makefile, tests/emu/emu_main.cpp): every out-of-bounds index into a per-lane array, a queue, a node pool
// or a table block shows up as a sanitizer report instead of a GPU memory fault.  Only tests/ builds or runs it; the
// product library is built by hipcc from the real header and has no CPU path.
//
// Execution model: a launch runs its blocks one after the other and, inside a block, its threads one after the
// other, each to completion.  Every thread is a wave of its own in which only its own lane is active:
// ballot(x) = x << lane, __shfl(v, own lane) = v, atomics are plain read-modify-writes.  That is a legal GPU schedule for
// kernels whose lanes do not rely on each other's stores inside one launch -- render_kernel / wf_level_kernel /
// color_at_kernel / probe_kernel of scenes that are not staged in LDS.  Kernels that stage data cooperatively
// (__syncthreads, a wave's records put together in LDS) are NOT covered by this model.
#pragma once
#include <algorithm>
#include <cmath>
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
inline dim3 threadIdx, blockIdx, blockDim, gridDim;
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct __attribute__((aligned(8))) float2 { float x, y; };

// ---- wave intrinsics: a wave with ONE active lane ------------------------------------------------------------------
namespace hip_emu {
inline unsigned lane() { return threadIdx.x & 63u; }
inline unsigned long long ballot(bool x) { return x ? 1ull << lane() : 0ull; }
}  // namespace hip_emu
#define __builtin_amdgcn_ballot_w64(x) hip_emu::ballot(x)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ull
inline unsigned __lane_id() { return hip_emu::lane(); }
inline void __syncthreads() {}
template <class T> inline T __shfl(T v, int src, int = 64) {
    if ((unsigned)src != hip_emu::lane()) std::abort();  // a read from another lane: not in this model
    return v;
}
template <class T> inline T __shfl_down(T v, unsigned off, int = 64) { return off == 0 ? v : T(0); }  // (the other lanes hold nothing)
template <class T> inline T __shfl_up(T v, unsigned off, int = 64) { return off == 0 ? v : T(0); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }

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

// ---- launches: block after block, thread after thread --------------------------------------------------------------
template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*lds*/, hipStream_t /*stream*/, A... args) {
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
