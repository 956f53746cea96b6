// rt_device.h -- per-ray arithmetic of the render path as gfx950 device functions.
//
// Every function restates one reference function (file:line cited) on the flat
// record tables of include/rt_hip.h.  Arithmetic contract:
// binary64, one rounding per written operation, the reference's association
// order, true divisions; the translation unit is compiled with
// -ffp-contract=off so hipcc does not fuse a*b+c.  Branch decisions therefore
// match the reference's to the last bit; only pow()/hypot() (ocml vs libm) may
// differ by an ulp, and those feed colours.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

#define RT_DEV __device__ __forceinline__

namespace rtd {

constexpr double EPSILON = 1e-4;  // point.rs:69-71, vector.rs:103-105

struct V3 {
    double x, y, z;
};
RT_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_DEV V3 hadamard(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }  // color.rs:72-82
RT_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // vector.rs:29-31
RT_DEV V3 cross(V3 a, V3 b) {                                                    // vector.rs:33-39
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vector.rs:25-27
RT_DEV V3 normalize(V3 a) {                                                        // vector.rs:19-23
    const double m = magnitude(a);
    return v3(a.x / m, a.y / m, a.z / m);
}
RT_DEV V3 reflect(V3 v, V3 n) { return v - (n * 2.0) * dot(v, n); }  // vector.rs:41-43

struct Ray {
    V3 o, d;
};

// Matrix<4> * Point / * Vector on rows 0..2 (matrix.rs:286-308).  `m` points at 12 doubles.
RT_DEV V3 mul_point(const double* __restrict__ m, V3 p) {
    return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
RT_DEV V3 mul_vector(const double* __restrict__ m, V3 v) {
    return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// transpose(M) * Vector: column j of rows 0..2 (primitive.rs:185,189 with matrix.rs:25-35,298-308)
RT_DEV V3 mul_vector_transposed(const double* __restrict__ m, V3 v) {
    return v3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
              m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
RT_DEV Ray transform_ray(const double* __restrict__ m, const Ray& r) {  // ray.rs:19-24
    return Ray{mul_point(m, r.o), mul_vector(m, r.d)};
}

// approx 0.5.1: abs_diff_eq / ulps_eq on f64, as used by Intersection::eq
// (intersection.rs:102-107: eps 1e-5, 4 ulps).  Callers never pass NaN.
RT_DEV bool abs_diff_eq(double a, double b, double eps) { return fabs(a - b) <= eps; }
RT_DEV bool t_eq(double a, double b) {
    if (fabs(a - b) <= 1e-5) return true;
    const unsigned long long ia = (unsigned long long)__double_as_longlong(a);
    const unsigned long long ib = (unsigned long long)__double_as_longlong(b);
    if ((ia ^ ib) >> 63) return false;  // signum differs
    return (ia <= ib ? ib - ia : ia - ib) <= 4ull;
}

// primitive.rs:380-392
RT_DEV void check_axis(double origin, double direction, double mn, double mx, double& t_min, double& t_max) {
    const double a = (mn - origin) / direction;
    const double b = (mx - origin) / direction;
    if (a > b) {
        t_min = b;
        t_max = a;
    } else {
        t_min = a;
        t_max = b;
    }
}
// Bounds::intersects (bounds.rs:53-77).  f64::max/min ignore NaN == fmax/fmin.
RT_DEV bool box_intersects(const double* __restrict__ bmin, const double* __restrict__ bmax, const Ray& r) {
    double x0, x1, y0, y1, z0, z1;
    check_axis(r.o.x, r.d.x, bmin[0], bmax[0], x0, x1);
    check_axis(r.o.y, r.d.y, bmin[1], bmax[1], y0, y1);
    check_axis(r.o.z, r.d.z, bmin[2], bmax[2], z0, z1);
    const double t_min = fmax(fmax(x0, y0), z0);
    const double t_max = fmin(fmin(x1, y1), z1);
    return t_min <= t_max;
}

// ---- per-kind local intersection: raw t values in the reference's push order ----
// Each returns the number of values written to ts[] (<= 4).

RT_DEV int sphere_intersect(const Ray& r, double* ts) {  // primitive.rs:269-286
    const double a = dot(r.d, r.d);
    const double b = 2.0 * dot(r.o, r.d);
    const double c = dot(r.o, r.o) - 1.0;
    const double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return 0;
    const double s = sqrt(discriminant);
    ts[0] = (-b - s) / (2.0 * a);
    ts[1] = (-b + s) / (2.0 * a);
    return 2;
}
RT_DEV int plane_intersect(const Ray& r, double* ts) {  // primitive.rs:288-294
    if (fabs(r.d.y) < EPSILON) return 0;
    ts[0] = -r.o.y / r.d.y;
    return 1;
}
RT_DEV int cube_intersect(const Ray& r, double* ts) {  // primitive.rs:296-312
    double x0, x1, y0, y1, z0, z1;
    check_axis(r.o.x, r.d.x, -1.0, 1.0, x0, x1);
    check_axis(r.o.y, r.d.y, -1.0, 1.0, y0, y1);
    check_axis(r.o.z, r.d.z, -1.0, 1.0, z0, z1);
    const double t_min = fmax(fmax(x0, y0), z0);
    const double t_max = fmin(fmin(x1, y1), z1);
    if (t_min > t_max) return 0;
    ts[0] = t_min;
    ts[1] = t_max;
    return 2;
}
RT_DEV bool check_cap(const Ray& r, double t, double radius) {  // primitive.rs:528-533
    const double x = r.o.x + t * r.d.x;
    const double z = r.o.z + t * r.d.z;
    return (x * x + z * z) <= radius * radius;
}
// Conic::intersect + intersect_caps (primitive.rs:481-526); cone => cap radius |y| (primitive.rs:328)
RT_DEV int conic_intersect(const Ray& r, double a, double b, double c, double cmin, double cmax, bool closed,
                           bool cone, double* ts) {
    int n = 0;
    const double discriminant = b * b - 4.0 * a * c;
    if (closed && !abs_diff_eq(r.d.y, 0.0, EPSILON)) {
        double t = (cmin - r.o.y) / r.d.y;
        if (check_cap(r, t, cone ? fabs(cmin) : 1.0)) ts[n++] = t;
        t = (cmax - r.o.y) / r.d.y;
        if (check_cap(r, t, cone ? fabs(cmax) : 1.0)) ts[n++] = t;
    }
    if (discriminant >= 0.0) {
        if (abs_diff_eq(a, 0.0, EPSILON)) {
            if (!abs_diff_eq(b, 0.0, EPSILON)) ts[n++] = -c / (2.0 * b);
        } else {
            const double s = sqrt(discriminant);
            const double t0 = (-b - s) / (2.0 * a);
            const double t1 = (-b + s) / (2.0 * a);
            const double y0 = r.o.y + t0 * r.d.y;
            if (cmin < y0 && y0 < cmax) ts[n++] = t0;
            const double y1 = r.o.y + t1 * r.d.y;
            if (cmin < y1 && y1 < cmax) ts[n++] = t1;
        }
    }
    return n;
}
RT_DEV int cylinder_intersect(const Ray& r, double cmin, double cmax, bool closed, double* ts) {  // :314-320
    const double a = r.d.x * r.d.x + r.d.z * r.d.z;
    const double b = 2.0 * r.o.x * r.d.x + 2.0 * r.o.z * r.d.z;
    const double c = r.o.x * r.o.x + r.o.z * r.o.z - 1.0;
    return conic_intersect(r, a, b, c, cmin, cmax, closed, false, ts);
}
RT_DEV int cone_intersect(const Ray& r, double cmin, double cmax, bool closed, double* ts) {  // :322-329
    const double a = r.d.x * r.d.x - r.d.y * r.d.y + r.d.z * r.d.z;
    const double b = 2.0 * r.o.x * r.d.x - 2.0 * r.o.y * r.d.y + 2.0 * r.o.z * r.d.z;
    const double c = r.o.x * r.o.x - r.o.y * r.o.y + r.o.z * r.o.z;
    return conic_intersect(r, a, b, c, cmin, cmax, closed, true, ts);
}
// Triangular::intersect_uv (primitive.rs:416-440): everything is computed before any test.
RT_DEV int triangle_intersect(const RtTriangle* __restrict__ tri, const Ray& r, double* ts, double& u, double& v) {
    const V3 p1 = v3(tri->p1[0], tri->p1[1], tri->p1[2]);
    const V3 e1 = v3(tri->e1[0], tri->e1[1], tri->e1[2]);
    const V3 e2 = v3(tri->e2[0], tri->e2[1], tri->e2[2]);
    const V3 dir_cross_e2 = cross(r.d, e2);
    const double det = dot(e1, dir_cross_e2);
    const double f = 1.0 / det;
    const V3 p1_to_origin = r.o - p1;
    u = f * dot(p1_to_origin, dir_cross_e2);
    const V3 origin_cross_e1 = cross(p1_to_origin, e1);
    v = f * dot(r.d, origin_cross_e1);
    const bool u_in = (0.0 <= u) && (u <= 1.0);
    if (abs_diff_eq(fabs(det), 0.0, EPSILON) || !u_in || v < 0.0 || (u + v) > 1.0) return 0;
    ts[0] = f * dot(e2, origin_cross_e1);
    return 1;
}

// `.collect::<BTreeSet<_>>()` on one primitive's hits (primitive.rs:227-230, Q2):
// sort ascending, then drop element i when it is ~equal to element i+1.
// NaN values (the reference would panic: intersection.rs:119) are dropped first and
// reported through `nan_seen`.  Returns the new count.
RT_DEV int collect_ts(double* ts, int n, bool& nan_seen) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const double t = ts[i];
        if (t != t) {
            nan_seen = true;
            continue;
        }
        int j = m++;
        while (j > 0 && ts[j - 1] > t) {  // insertion sort, stable
            ts[j] = ts[j - 1];
            --j;
        }
        ts[j] = t;
    }
    int k = 0;
    for (int i = 0; i < m; ++i) {
        if (i + 1 < m && t_eq(ts[i], ts[i + 1])) continue;
        ts[k++] = ts[i];
    }
    return k;
}

// ---- local normals (primitive.rs:331-375, 469-479) ---------------------------
RT_DEV V3 conic_normal(V3 p, double y, double cmin, double cmax) {
    const double dist = p.x * p.x + p.z * p.z;
    if (dist < 1.0 && p.y >= cmax - EPSILON) return v3(0.0, 1.0, 0.0);
    if (dist < 1.0 && p.y <= cmin + EPSILON) return v3(0.0, -1.0, 0.0);
    return v3(p.x, y, p.z);
}
RT_DEV V3 cube_normal(V3 p) {
    const double max_c = fmax(fmax(fabs(p.x), fabs(p.y)), fabs(p.z));
    if (abs_diff_eq(max_c, fabs(p.x), EPSILON)) return v3(p.x, 0.0, 0.0);
    if (abs_diff_eq(max_c, fabs(p.y), EPSILON)) return v3(0.0, p.y, 0.0);
    return v3(0.0, 0.0, p.z);
}

// ---- Perturb noise: noise::SuperSimplex (crate `noise` 0.7.0, seed 0 = SuperSimplex::new(), pattern.rs:107-124) ------
// The crate is a dependency outside the reference tree; oracle/rt_oracle.cpp restates its published algorithm and derives
// this permutation table (PermutationTable::new(0): rand_xorshift 0.2.0 + rand 0.7.3's shuffle), and the reference's own
// render pins both: every Perturb pixel of images/png/pattern.png comes out identical.  tests/test_host_logic.py compares
// the table below with the oracle's derivation; the arithmetic is the oracle's, operation for operation.
//   point into simplex space: p' = -(p + (x + y + z) * -2/3); two cubic lattices, the second moved by 512.5; of the eight
//   corners of the cell around p' on each lattice those with attn = 0.75 - |d|^2 > 0 contribute attn^4 * dot(gradient, d);
//   hash = perm[perm[perm[x & 255] ^ (y & 255)] ^ (z & 255)]; gradients: 12 cube edges repeated twice, then 8 corners;
//   the sum times 1 / 0.0867664001655369.
__device__ const uint8_t SS_PERM[256] = {
    181, 206, 3, 127, 58, 121, 169, 47, 62, 123, 5, 142, 99, 140, 178, 171, 114, 60, 194, 65, 135, 190,
    85, 125, 208, 234, 39, 9, 182, 82, 92, 132, 212, 95, 160, 31, 84, 75, 90, 52, 108, 6, 7, 204,
    163, 19, 172, 122, 168, 63, 205, 111, 224, 166, 21, 59, 203, 88, 116, 104, 68, 13, 26, 170, 102, 220,
    159, 72, 77, 120, 196, 67, 185, 66, 27, 79, 144, 118, 219, 55, 139, 91, 183, 146, 113, 202, 207, 241,
    89, 29, 83, 130, 71, 10, 46, 87, 152, 211, 106, 145, 201, 51, 109, 232, 81, 134, 186, 78, 74, 137,
    124, 115, 184, 32, 165, 188, 57, 16, 175, 192, 128, 174, 180, 119, 117, 228, 221, 40, 110, 149, 96, 80,
    210, 94, 200, 86, 34, 97, 100, 156, 73, 155, 197, 45, 42, 44, 153, 191, 214, 173, 126, 187, 24, 223,
    30, 38, 193, 199, 101, 225, 8, 231, 176, 25, 148, 37, 33, 136, 12, 76, 198, 98, 105, 93, 69, 53,
    23, 150, 56, 107, 151, 162, 141, 36, 189, 143, 50, 48, 195, 112, 131, 138, 147, 209, 226, 215, 227, 61,
    222, 218, 177, 103, 167, 129, 64, 35, 239, 154, 161, 179, 213, 41, 22, 217, 216, 229, 54, 70, 20, 233,
    11, 164, 158, 18, 133, 157, 28, 43, 243, 230, 14, 49, 240, 235, 236, 4, 237, 238, 244, 17, 1, 242,
    2, 15, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254, 255, 0};
// Rust `f64 as isize`: saturating, NaN -> 0
RT_DEV long long f64_as_i64(double v) {
    if (v != v) return 0;
    if (v >= 9223372036854775807.0) return 9223372036854775807ll;
    if (v <= -9223372036854775808.0) return (-9223372036854775807ll - 1);
    return (long long)v;
}
__device__ __noinline__ double perturb_noise(double x, double y, double z) {
    const double off = ((x + y) + z) * (-2.0 / 3.0);
    const double sp[3] = {-(x + off), -(y + off), -(z + off)};
    double value = 0.0;
#pragma unroll
    for (int lattice = 0; lattice < 2; ++lattice) {
        double rel[3];
        unsigned long long bi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double p = lattice == 0 ? sp[a] : sp[a] + 512.5;
            const double base = floor(p);
            bi[a] = (unsigned long long)f64_as_i64(base);
            rel[a] = p - base;
        }
#pragma unroll
        for (unsigned c = 0; c < 8u; ++c) {
            const unsigned ox = c & 1u, oy = (c >> 1) & 1u, oz = (c >> 2) & 1u;
            const double dx = rel[0] - (double)ox, dy = rel[1] - (double)oy, dz = rel[2] - (double)oz;
            const double attn = 0.75 - ((dx * dx + dy * dy) + dz * dz);
            if (attn > 0.0) {
                const unsigned hx = SS_PERM[(unsigned)((bi[0] + ox) & 0xffu)];
                const unsigned hy = SS_PERM[hx ^ (unsigned)((bi[1] + oy) & 0xffu)];
                const unsigned hz = SS_PERM[hy ^ (unsigned)((bi[2] + oz) & 0xffu)];
                // gradient::get3(index % 32): "12 edges repeated twice then 8 corners"
                const double diag = 0.70710678118654752440, diag2 = 0.5773502691896258;
                const unsigned k = hz % 32u;
                double gx, gy, gz;
                if (k < 24u) {
                    const unsigned e = k % 12u, pair = e >> 2, sg = e & 3u;  // pair 0: (x, y), 1: (x, z), 2: (y, z)
                    const double a = (sg & 1u) ? -diag : diag, b = (sg & 2u) ? -diag : diag;
                    gx = pair == 2u ? 0.0 : a;
                    gy = pair == 0u ? b : (pair == 2u ? a : 0.0);
                    gz = pair == 0u ? 0.0 : b;
                } else {
                    const unsigned cc = k - 24u;
                    gx = (cc & 1u) ? -diag2 : diag2, gy = (cc & 2u) ? -diag2 : diag2, gz = (cc & 4u) ? -diag2 : diag2;
                }
                const double a2 = attn * attn;
                value += (a2 * a2) * ((gx * dx + gy * dy) + gz * dz);
            }
        }
    }
    return value * (1.0 / 0.0867664001655369);
}

// Rust `f64 as i32`: saturating, NaN -> 0 (pattern.rs:159,183,193)
RT_DEV int f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

// Inlined: as a real call the evaluator made every value live across it (the ray, the whole shading
// context) go through the calling convention's few callee-saved registers, i.e. scratch memory, once
// per shaded hit -- ~100 B per pixel of HBM writes in the 3 waves/SIMD variants (pattern.rs -14 %,
// glass_bubble.rs -20 % inlined).  RT_CALL_PATTERN restores the call for comparison.
#ifdef RT_CALL_PATTERN
#define RT_PATTERN_FN __device__ __noinline__
#else
#define RT_PATTERN_FN __device__ __forceinline__
#endif
#define RT_PATTERN_STACK 8  // nesting depth of blend/gradient/radial-gradient nodes

// Pattern::pattern_at (pattern.rs:87-91) with Kind::pattern_at / Mixture::* (pattern.rs:103-199)
// unrolled into a loop: selector mixtures (stripe/ring/checker) and perturb are
// tail calls; only the arithmetic mixtures need a frame.  The reference evaluates
// both children of a selector and discards one; patterns are pure, so the value
// is the same.
RT_PATTERN_FN V3 pattern_eval(const RtPattern* __restrict__ pats, int root, V3 object_point,
                                        unsigned& nodes_evaluated) {
    int f_node[RT_PATTERN_STACK];
    V3 f_pp[RT_PATTERN_STACK];
    V3 f_a[RT_PATTERN_STACK];
    bool f_have_a[RT_PATTERN_STACK];
    int sp = 0;
    int node = root;
    V3 pin = object_point;
    V3 val = v3(0.0, 0.0, 0.0);
    for (;;) {
        // descend until a leaf value is known
        for (;;) {
            const RtPattern* __restrict__ p = pats + node;
            const V3 pp = mul_point(p->inv, pin);
            const unsigned kind = p->kind;
            ++nodes_evaluated;
            if (kind == RT_PAT_SOLID) {
                val = v3(p->color[0], p->color[1], p->color[2]);
                break;
            }
            if (kind == RT_PAT_TEST) {
                val = pp;
                break;
            }
            if (kind == RT_PAT_PERTURB) {
                const double PERTURB_AMOUNT = 0.4;
                pin = v3(pp.x + perturb_noise(pp.x, pp.y, pp.z) * PERTURB_AMOUNT,
                         pp.y + perturb_noise(pp.x, pp.y, pp.z + 1.0) * PERTURB_AMOUNT,
                         pp.z + perturb_noise(pp.x, pp.y, pp.z + 2.0) * PERTURB_AMOUNT);
                node = p->a;
                continue;
            }
            if (kind == RT_PAT_STRIPE) {
                node = (f64_as_i32(floor(pp.x)) % 2 == 0) ? p->a : p->b;
                pin = pp;
                continue;
            }
            if (kind == RT_PAT_RING) {
                const double distance = hypot(pp.x, pp.z);
                node = (f64_as_i32(floor(distance)) % 2 == 0) ? p->a : p->b;
                pin = pp;
                continue;
            }
            if (kind == RT_PAT_CHECKER) {
                const double sum_floors = floor(pp.x) + floor(pp.y) + floor(pp.z);
                node = (f64_as_i32(sum_floors) % 2 == 0) ? p->a : p->b;
                pin = pp;
                continue;
            }
            // blend / gradient / radial gradient: both children are needed
            if (sp >= RT_PATTERN_STACK) {  // rejected at scene creation; defensive
                val = v3(0.0, 0.0, 0.0);
                break;
            }
            f_node[sp] = node;
            f_pp[sp] = pp;
            f_have_a[sp] = false;
            ++sp;
            node = p->a;
            pin = pp;
        }
        // ascend
        for (;;) {
            if (sp == 0) return val;
            const int top = sp - 1;
            const RtPattern* __restrict__ p = pats + f_node[top];
            if (!f_have_a[top]) {
                f_a[top] = val;
                f_have_a[top] = true;
                node = p->b;
                pin = f_pp[top];
                break;  // descend into b
            }
            const V3 a = f_a[top], b = val, pp = f_pp[top];
            if (p->kind == RT_PAT_BLEND) {
                val = (a + b) * 0.5;
            } else if (p->kind == RT_PAT_GRADIENT) {
                const V3 distance = b - a;
                const double fraction = pp.x - floor(pp.x);
                val = a + distance * fraction;
            } else {  // radial gradient
                const double distance = magnitude(pp);
                const double fraction = distance - floor(distance);
                val = a + (b - a) * fraction;
            }
            --sp;
        }
    }
}

}  // namespace rtd
