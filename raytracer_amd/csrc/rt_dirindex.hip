// rt_dirindex.hip -- the direction index (RtDirCand, include/rt_hip.h) built on the device.
//
// The index answers, for a ray direction, "which truncated conics may take the branch `a ~ 0` of
// primitive.rs:486-490" (their hits there are not confined to the conic's box, so the hierarchy's boxes must not
// decide).  It is a cube map of direction cells, each with the list of candidates whose quadratic form a(u) = u^T Q u
// can come within the candidate's threshold for SOME unit direction u of the cell.  Acceleration only: a list that is
// too long costs time, never a result; the test below must only never leave a candidate out.
//
// Build: one block per (candidate, cube face); its 256 threads take the 16 x 16 coarse tiles of the face and refine
// the ones a conservative bound lets through, level by level, down to cells (the region of a cone is a thin band, so
// almost every tile ends at its first test).  Two passes of the same walk -- count, (scan), fill -- then every cell's
// short list is sorted so that the table does not depend on the order the blocks ran in.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "rt_host_internal.h"

namespace {

struct Form {
    double q[6], thr, qnorm;
};

// Can |a(u)| <= thr hold for a unit u whose cube-map point lies in the square of `size` cells at (i0, j0)?
// With c the unit direction of the square's centre and r = |u - c|:  a(u) - a(c) = 2 (Qc).(u - c) + (u - c)^T Q (u - c),
// so |a(u) - a(c)| <= 2 |Qc| r + |Q|_F r^2; a square of half-width h in (s, t) lies within chord distance h sqrt(2)
// of its centre (normalising points outside the unit ball onto it never lengthens a distance).  Written for the
// unnormalised centre p = (+-1, s, t), n2 = |p|^2, c = p / |p|, and multiplied through by n2 (no division):
//     |a(p)| <= n2 (thr + k |Q|_F reach^2) + 2 k reach sqrt(|Qp|^2 n2),   k = 1.000001,
// the square root bounded from above in f32 (faces +x -x +y -y +z -z as in cand_begin, rt_trace.h).
__device__ bool may_hold(const Form& f, uint32_t res, int face, uint32_t i0, uint32_t j0, uint32_t size) {
    const double inv_res = 1.0 / (double)res;  // (a power of two: exact)
    const double h = (double)size * inv_res;
    const double sc = -1.0 + (2.0 * (double)i0 + (double)size) * inv_res;
    const double tc = -1.0 + (2.0 * (double)j0 + (double)size) * inv_res;
    const double sign = (face & 1) ? -1.0 : 1.0;
    double x, y, z;
    if (face < 2) x = sign, y = sc, z = tc;
    else if (face < 4) x = sc, y = sign, z = tc;
    else x = sc, y = tc, z = sign;
    const double n2 = 1.0 + sc * sc + tc * tc;
    const double gx = f.q[0] * x + f.q[3] * y + f.q[4] * z;
    const double gy = f.q[3] * x + f.q[1] * y + f.q[5] * z;
    const double gz = f.q[4] * x + f.q[5] * y + f.q[2] * z;
    const double a = gx * x + gy * y + gz * z;
    const double g2n2 = (gx * gx + gy * gy + gz * gz) * n2;
    const double root_up = (double)sqrtf(__double2float_ru(g2n2)) * (1.0 + 0x1p-20) + 1e-18;  // (inf when out of f32 range: kept)
    const double reach = h * 1.4142135623730951 * 1.000001 + 1e-9;
    const double bound = n2 * (f.thr + 1.000001 * f.qnorm * reach * reach) + 2.000002 * reach * root_up;
    return !(fabs(a) > bound);  // (a NaN form is kept: every cell lists it)
}

// The walk of one (candidate, face): tiles that may hold the region are refined 4 x 4 (2 x 2 at the last step of an odd
// power) down to cells, level by level, the block's threads sharing every level's tests.  A level's passing tiles
// wait in an LDS list; lists are bounded, so a level is worked off in chunks whose children are certain to fit the
// next list, depth first (a form that holds everywhere still comes out right, just slowly).
// FILL false: cell_count[cell] += 1 per (cell, candidate) pair.  FILL true: the pair's candidate goes to
// entries[cursor[cell]++].
constexpr uint32_t kListCap = 2048, kLevels = 4;
template <bool FILL>
__global__ __launch_bounds__(256) void raster_kernel(const RtDirCand* __restrict__ cands, uint32_t n_cands, uint32_t res,
                                                     uint32_t* __restrict__ cell_count, uint32_t* __restrict__ entries) {
    __shared__ uint32_t list[kLevels][kListCap];
    __shared__ uint32_t s_n[kLevels + 1], s_r[kLevels + 1];
    const uint32_t cand = blockIdx.x / 6u;
    const int face = (int)(blockIdx.x % 6u);
    if (cand >= n_cands) return;
    const uint32_t tid = threadIdx.x;
    Form f;
    {
        const RtDirCand& c = cands[cand];
        for (int k = 0; k < 6; ++k) f.q[k] = c.q[k];
        f.thr = c.thr;
        f.qnorm = sqrt(f.q[0] * f.q[0] + f.q[1] * f.q[1] + f.q[2] * f.q[2] +
                       2.0 * (f.q[3] * f.q[3] + f.q[4] * f.q[4] + f.q[5] * f.q[5]));
    }
    auto emit = [&](uint32_t i, uint32_t j) {
        const uint32_t cell = ((uint32_t)face * res + j) * res + i;
        const uint32_t at = atomicAdd(&cell_count[cell], 1u);
        if (FILL) entries[at] = cand;
    };
    uint32_t sz[kLevels + 1];  // tile size of list k; the children of the last list are cells
    sz[0] = res / 16u;         // res is a power of two in [16, 2048]
    for (uint32_t k = 0; k < kLevels; ++k) sz[k + 1] = sz[k] >= 4u ? sz[k] / 4u : 1u;
    if (tid <= kLevels) s_n[tid] = 0u, s_r[tid] = 0u;
    __syncthreads();
    {
        const uint32_t ci = (tid & 15u) * sz[0], cj = (tid >> 4) * sz[0];
        if (may_hold(f, res, face, ci, cj, sz[0])) {
            if (sz[0] == 1u) emit(ci, cj);
            else list[0][atomicAdd(&s_n[0], 1u)] = ci | (cj << 12);
        }
    }
    __syncthreads();
    if (sz[0] == 1u) return;
    uint32_t l = 0;
    for (;;) {
        const uint32_t have = s_n[l], rd = s_r[l];  // (the same values in every thread: written before a barrier)
        if (rd >= have) {
            if (l == 0u) break;
            --l;
            continue;
        }
        const uint32_t szn = sz[l + 1], b = sz[l] / szn, branch = b * b;
        const uint32_t room = kListCap / branch;
        const uint32_t chunk = have - rd < room ? have - rd : room;
        __syncthreads();  // every thread has read the cursors
        if (tid == 0u) {
            s_r[l] = rd + chunk;
            if (szn > 1u) s_n[l + 1] = 0u, s_r[l + 1] = 0u;
        }
        __syncthreads();
        for (uint32_t p = tid; p < chunk * branch; p += 256u) {
            const uint32_t e = list[l][rd + p / branch], c = p % branch;
            const uint32_t i = (e & 0xfffu) + (c % b) * szn, j = (e >> 12) + (c / b) * szn;
            if (!may_hold(f, res, face, i, j, szn)) continue;
            if (szn == 1u) emit(i, j);
            else list[l + 1][atomicAdd(&s_n[l + 1], 1u)] = i | (j << 12);
        }
        __syncthreads();
        if (szn > 1u) ++l;
    }
}

// Every cell's list in ascending candidate order (lists are short; the order the blocks filled them in is not fixed).
__global__ __launch_bounds__(256) void sort_cells_kernel(const uint32_t* __restrict__ cell_start, uint32_t n_cells,
                                                         uint32_t* __restrict__ entries) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const uint32_t b = cell_start[cell], e = cell_start[cell + 1];
    for (uint32_t k = b + 1; k < e; ++k) {
        const uint32_t v = entries[k];
        uint32_t at = k;
        while (at > b && entries[at - 1] > v) entries[at] = entries[at - 1], --at;
        entries[at] = v;
    }
}

uint32_t pick_resolution(uint32_t n_cands, uint32_t hint) {
    // the band of a cone crosses ~res cells per nappe whatever the resolution, so entries grow with res while a
    // cell's list shrinks with it: more candidates, finer cells (measured on the 10 000-primitive scene)
    static const long env = [] {
        const char* e = std::getenv("RT_HIP_DIR_RES");
        return e && *e ? std::atol(e) : 0L;
    }();
    uint32_t res = env > 0 ? (uint32_t)env : (n_cands >= 1024u ? 1024u : (n_cands >= 256u ? 512u : 256u));
    if (env <= 0 && hint > res) res = hint;
    uint32_t p = 16;
    while (p * 2u <= res && p < 2048u) p *= 2u;
    return p;
}

}  // namespace

void rth::dir_index_free(DirIndex* ix) {
    if (ix->d_cell_start) rth::block_free(ix->d_cell_start);
    if (ix->d_entries) rth::block_free(ix->d_entries);
    ix->d_cell_start = ix->d_entries = nullptr;
    ix->res = ix->n_entries = 0;
}

int rth::dir_index_build(const RtDirCand* d_cands, uint32_t n_cands, uint32_t res_hint, DirIndex* out) {
    out->d_cell_start = out->d_entries = nullptr;
    out->res = out->n_entries = 0;
    if (n_cands == 0 || n_cands > 0x7fffffffu / 6u) return n_cands ? RT_ERR_INVALID_ARG : RT_OK;
    constexpr uint64_t kMaxEntries = 1ull << 27;  // 512 MB of entries: past that (forms that hold everywhere) coarser cells
    uint32_t res = pick_resolution(n_cands, res_hint);
    uint32_t* d_cursor = nullptr;
    void* d_temp = nullptr;
    int rc = RT_OK;
    auto fail = [&](hipError_t e, const char* what) { rc = rth::hip_fail(e, what); };
    for (;;) {
        const size_t n_cells = (size_t)6 * res * res;
        hipError_t e = hipSuccess;
        if ((rc = rth::block_alloc(reinterpret_cast<void**>(&out->d_cell_start), (n_cells + 1) * sizeof(uint32_t))) != RT_OK) break;
        if ((e = hipMemsetAsync(out->d_cell_start, 0, (n_cells + 1) * sizeof(uint32_t), nullptr)) != hipSuccess) { fail(e, "hipMemsetAsync"); break; }
        raster_kernel<false><<<n_cands * 6u, 256, 0, nullptr>>>(d_cands, n_cands, res, out->d_cell_start, nullptr);
        // 64-bit total first: the 32-bit offsets are only meaningful when it fits
        unsigned long long* d_total = nullptr;
        size_t temp_bytes = 0;
        if ((rc = rth::block_alloc(reinterpret_cast<void**>(&d_total), sizeof(unsigned long long))) != RT_OK) break;
        // (sum of 32-bit counts into a 64-bit total: a transform iterator keeps the reduction from wrapping)
        hipcub::TransformInputIterator<unsigned long long, hipcub::CastOp<unsigned long long>, uint32_t*> wide(out->d_cell_start, hipcub::CastOp<unsigned long long>());
        size_t scan_bytes = 0;
        if ((e = hipcub::DeviceReduce::Sum(nullptr, temp_bytes, wide, d_total, (int)n_cells, nullptr)) != hipSuccess ||
            (e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, out->d_cell_start, out->d_cell_start, (int)(n_cells + 1), nullptr)) != hipSuccess) {
            rth::block_free(d_total);
            fail(e, "hipcub (sizes)");
            break;
        }
        if (scan_bytes > temp_bytes) temp_bytes = scan_bytes;
        if ((rc = rth::block_alloc(&d_temp, temp_bytes ? temp_bytes : 16)) != RT_OK) { rth::block_free(d_total); break; }
        size_t tb = temp_bytes;
        unsigned long long total = 0;
        if ((e = hipcub::DeviceReduce::Sum(d_temp, tb, wide, d_total, (int)n_cells, nullptr)) == hipSuccess)
            e = hipMemcpy(&total, d_total, sizeof total, hipMemcpyDeviceToHost);
        rth::block_free(d_total);
        if (e != hipSuccess) { fail(e, "hipMemcpy(direction entries total)"); break; }
        if (total > kMaxEntries && res > 16u) {  // start over with cells twice as wide
            rth::block_free(d_temp), d_temp = nullptr;
            rth::block_free(out->d_cell_start), out->d_cell_start = nullptr;
            res /= 2u;
            continue;
        }
        if (total >= 0x80000000ull) { rc = RT_ERR_INVALID_SCENE; break; }
        tb = temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(d_temp, tb, out->d_cell_start, out->d_cell_start, (int)(n_cells + 1), nullptr)) != hipSuccess) { fail(e, "hipcub::DeviceScan"); break; }
        if ((rc = rth::block_alloc(reinterpret_cast<void**>(&out->d_entries), (total ? total : 1) * sizeof(uint32_t))) != RT_OK) break;
        if ((rc = rth::block_alloc(reinterpret_cast<void**>(&d_cursor), n_cells * sizeof(uint32_t))) != RT_OK) break;
        if ((e = hipMemcpyAsync(d_cursor, out->d_cell_start, n_cells * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr)) != hipSuccess) { fail(e, "hipMemcpyAsync"); break; }
        raster_kernel<true><<<n_cands * 6u, 256, 0, nullptr>>>(d_cands, n_cands, res, d_cursor, out->d_entries);
        sort_cells_kernel<<<(unsigned)((n_cells + 255) / 256), 256, 0, nullptr>>>(out->d_cell_start, (uint32_t)n_cells, out->d_entries);
        if ((e = hipGetLastError()) != hipSuccess) { fail(e, "direction index kernels"); break; }
        if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) { fail(e, "direction index kernels"); break; }
        out->res = res;
        out->n_entries = (uint32_t)total;
        break;
    }
    if (rc != RT_OK) (void)hipStreamSynchronize(nullptr);  // (the blocks below go back to a cache, not to the driver: nothing may still use them)
    if (d_cursor) rth::block_free(d_cursor);
    if (d_temp) rth::block_free(d_temp);
    if (rc != RT_OK) dir_index_free(out);
    return rc;
}
