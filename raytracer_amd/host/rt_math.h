// Host-side value types of the drop-in API: Point, Vector, Color, Matrix4.
//
// These mirror the *public* types a user of the reference builds scenes with
// (reference: src/core/point.rs, src/core/vector.rs, src/core/matrix.rs,
// src/graphics/color.rs).  They are host-only: the device never builds a
// matrix, it only consumes inverses that were computed here.
//
// Arithmetic contract: plain IEEE-754 binary64, one
// rounding per written operation, no FMA contraction, no reassociation.  The
// file must be compiled with -ffp-contract=off.  Operation ORDER below follows
// the reference line by line because the last bit of an inverse matrix decides
// silhouette pixels (hexagon seam, Q13).
#pragma once
#include <cmath>
#include <cstddef>
#include <stdexcept>

namespace rt {

struct Vector;

// reference: src/core/point.rs:7-58
struct Point {
    double x = 0.0, y = 0.0, z = 0.0;
    Point() = default;
    Point(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};

// reference: src/core/vector.rs:7-92
struct Vector {
    double x = 0.0, y = 0.0, z = 0.0;
    Vector() = default;
    Vector(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}

    // vector.rs:25-27  powi(2) == x*x, summed left to right
    double magnitude() const { return std::sqrt(x * x + y * y + z * z); }
    // vector.rs:19-23  three true divisions
    Vector normalize() const {
        const double m = magnitude();
        return Vector(x / m, y / m, z / m);
    }
    // vector.rs:29-31
    double dot(const Vector& o) const { return x * o.x + y * o.y + z * o.z; }
    // vector.rs:33-39
    Vector cross(const Vector& o) const {
        return Vector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
    }
};

inline Vector operator-(const Point& a, const Point& b) { return Vector(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Point operator+(const Point& a, const Vector& b) { return Point(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Point operator-(const Point& a, const Vector& b) { return Point(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector operator+(const Vector& a, const Vector& b) { return Vector(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vector operator-(const Vector& a, const Vector& b) { return Vector(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector operator-(const Vector& a) { return Vector(-a.x, -a.y, -a.z); }
inline Vector operator*(const Vector& a, double s) { return Vector(a.x * s, a.y * s, a.z * s); }

// reference: src/graphics/color.rs:5-82
struct Color {
    double red = 0.0, green = 0.0, blue = 0.0;
    Color() = default;
    Color(double r, double g, double b) : red(r), green(g), blue(b) {}
    static Color BLACK() { return Color(0.0, 0.0, 0.0); }
    static Color WHITE() { return Color(1.0, 1.0, 1.0); }
};

// |a-b| <= eps, the `approx` crate's abs_diff_eq for floats.  inf-inf = NaN
// compares false, which is what makes infinite primitives unequal to
// themselves in CSG membership tests.
inline bool abs_diff_eq(double a, double b, double eps) { return std::fabs(a - b) <= eps; }
inline bool approx_eq(const Point& a, const Point& b) {
    return abs_diff_eq(a.x, b.x, 1e-4) && abs_diff_eq(a.y, b.y, 1e-4) && abs_diff_eq(a.z, b.z, 1e-4);
}
inline bool approx_eq(const Vector& a, const Vector& b) {
    return abs_diff_eq(a.x, b.x, 1e-4) && abs_diff_eq(a.y, b.y, 1e-4) && abs_diff_eq(a.z, b.z, 1e-4);
}
inline bool approx_eq(const Color& a, const Color& b) {
    return abs_diff_eq(a.red, b.red, 1e-4) && abs_diff_eq(a.green, b.green, 1e-4) &&
           abs_diff_eq(a.blue, b.blue, 1e-4);
}

// reference: src/core/matrix.rs:10 (`Matrix<4>`, row-major)
struct Matrix4 {
    double m[4][4];

    static Matrix4 zero() {
        Matrix4 r;
        for (auto& row : r.m)
            for (double& v : row) v = 0.0;
        return r;
    }
    // matrix.rs:15-23
    static Matrix4 identity() {
        Matrix4 r = zero();
        for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
        return r;
    }
    // matrix.rs:153-160
    static Matrix4 translation(double x, double y, double z) {
        Matrix4 r = identity();
        r.m[0][3] = x;
        r.m[1][3] = y;
        r.m[2][3] = z;
        return r;
    }
    // matrix.rs:162-169
    static Matrix4 scaling(double x, double y, double z) {
        Matrix4 r = identity();
        r.m[0][0] = x;
        r.m[1][1] = y;
        r.m[2][2] = z;
        return r;
    }
    // matrix.rs:171-181
    static Matrix4 rotation_x(double radians) {
        const double c = std::cos(radians), s = std::sin(radians);
        Matrix4 r = identity();
        r.m[1][1] = c;
        r.m[1][2] = -s;
        r.m[2][1] = s;
        r.m[2][2] = c;
        return r;
    }
    // matrix.rs:183-193
    static Matrix4 rotation_y(double radians) {
        const double c = std::cos(radians), s = std::sin(radians);
        Matrix4 r = identity();
        r.m[0][0] = c;
        r.m[0][2] = s;
        r.m[2][0] = -s;
        r.m[2][2] = c;
        return r;
    }
    // matrix.rs:195-205
    static Matrix4 rotation_z(double radians) {
        const double c = std::cos(radians), s = std::sin(radians);
        Matrix4 r = identity();
        r.m[0][0] = c;
        r.m[0][1] = -s;
        r.m[1][0] = s;
        r.m[1][1] = c;
        return r;
    }
    // matrix.rs:207-214
    static Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
        Matrix4 r = identity();
        r.m[0][1] = xy;
        r.m[0][2] = xz;
        r.m[1][0] = yx;
        r.m[1][2] = yz;
        r.m[2][0] = zx;
        r.m[2][1] = zy;
        return r;
    }
    // matrix.rs:216-230
    static Matrix4 view_transform(const Point& from, const Point& to, const Vector& up);

    // matrix.rs:25-35
    Matrix4 transpose() const {
        Matrix4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[i][j] = m[j][i];
        return r;
    }

    // matrix.rs:92-106.  Throws where the reference asserts (singular matrix).
    Matrix4 inverse() const;
    double determinant() const;

    bool approx_eq(const Matrix4& o) const {  // matrix.rs:247-266, eps 1e-5
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (!abs_diff_eq(m[i][j], o.m[i][j], 1e-5)) return false;
        return true;
    }
};

// matrix.rs:268-284: i-k-j loop, accumulating into a zeroed result.
inline Matrix4 operator*(const Matrix4& a, const Matrix4& b) {
    Matrix4 r = Matrix4::zero();
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) r.m[i][j] += a.m[i][k] * b.m[k][j];
    return r;
}
// matrix.rs:286-296: rows 0..2, column 3 added last.
inline Point operator*(const Matrix4& a, const Point& p) {
    return Point(a.m[0][0] * p.x + a.m[0][1] * p.y + a.m[0][2] * p.z + a.m[0][3],
                 a.m[1][0] * p.x + a.m[1][1] * p.y + a.m[1][2] * p.z + a.m[1][3],
                 a.m[2][0] * p.x + a.m[2][1] * p.y + a.m[2][2] * p.z + a.m[2][3]);
}
// matrix.rs:298-308: 3x3 block only.
inline Vector operator*(const Matrix4& a, const Vector& v) {
    return Vector(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
                  a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
                  a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}

namespace detail {
// 3x3 determinant by first-row cofactor expansion, accumulated from 0.0
// (matrix.rs:44-63); 2x2 minor is a*d - b*c (matrix.rs:38-42).
inline double det3(const double s[3][3]) {
    double det = 0.0;
    for (int col = 0; col < 3; ++col) {
        double sub[2][2];
        int j = 0;
        for (int c = 0; c < 3; ++c) {
            if (c == col) continue;
            sub[0][j] = s[1][c];
            sub[1][j] = s[2][c];
            ++j;
        }
        const double minor = sub[0][0] * sub[1][1] - sub[0][1] * sub[1][0];
        const double sign = (col % 2 == 0) ? 1.0 : -1.0;
        det += s[0][col] * (sign * minor);
    }
    return det;
}
// matrix.rs:118-151
inline double cofactor4(const Matrix4& a, int row, int col) {
    double s[3][3];
    int i = 0;
    for (int r = 0; r < 4; ++r) {
        if (r == row) continue;
        int j = 0;
        for (int c = 0; c < 4; ++c) {
            if (c == col) continue;
            s[i][j] = a.m[r][c];
            ++j;
        }
        ++i;
    }
    const double sign = ((row + col) % 2 == 0) ? 1.0 : -1.0;
    return sign * det3(s);
}
}  // namespace detail

inline double Matrix4::determinant() const {  // matrix.rs:108-116
    double det = 0.0;
    for (int col = 0; col < 4; ++col) det += m[0][col] * detail::cofactor4(*this, 0, col);
    return det;
}

inline Matrix4 Matrix4::inverse() const {
    const double det = determinant();
    if (det == 0.0) throw std::domain_error("matrix is not invertible");  // matrix.rs:94
    Matrix4 r = zero();
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) r.m[col][row] = detail::cofactor4(*this, row, col) / det;
    return r;
}

inline Matrix4 Matrix4::view_transform(const Point& from, const Point& to, const Vector& up) {
    const Vector forward = (to - from).normalize();
    const Vector upn = up.normalize();
    const Vector left = forward.cross(upn);
    const Vector true_up = left.cross(forward);
    Matrix4 o = identity();
    o.m[0][0] = left.x, o.m[0][1] = left.y, o.m[0][2] = left.z;
    o.m[1][0] = true_up.x, o.m[1][1] = true_up.y, o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x, o.m[2][1] = -forward.y, o.m[2][2] = -forward.z;
    return o * translation(-from.x, -from.y, -from.z);
}

}  // namespace rt
