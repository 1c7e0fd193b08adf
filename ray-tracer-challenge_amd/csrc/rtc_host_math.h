// rtc_host_math.h -- host-side 4x4 matrix arithmetic, bit-faithful to the reference so the device
// tables hold exactly the numbers the Rust program would have produced (SURVEY.md hard part 10).
// Compiled with -ffp-contract=off.  Citations: ray_tracer/src/matrix.rs, transformation.rs, tuple.rs.
#pragma once
#include <array>
#include <cmath>
#include <cstring>

namespace rtc {

struct Mat4 {
    double m[4][4];

    static Mat4 identity() {
        Mat4 r;
        std::memset(&r, 0, sizeof r);
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
        return r;
    }
    static Mat4 from16(const double *a) {
        Mat4 r;
        std::memcpy(&r.m[0][0], a, sizeof r.m);
        return r;
    }
    void to16(double *out) const { std::memcpy(out, &m[0][0], sizeof m); }
    bool is_identity() const {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                if (m[i][j] != (i == j ? 1.0 : 0.0)) return false;  // -0.0 == 0.0
        return true;
    }
};

// matrix.rs:302-326: ((a0*b0 + a1*b1) + a2*b2) + a3*b3
inline Mat4 mat_mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}

// matrix.rs:86-108, the closed form used for every 4x4 determinant
inline double mat_det(const Mat4 &s) {
    const double a = s.m[0][0], b = s.m[0][1], c = s.m[0][2], d = s.m[0][3];
    const double e = s.m[1][0], f = s.m[1][1], g = s.m[1][2], h = s.m[1][3];
    const double i = s.m[2][0], j = s.m[2][1], k = s.m[2][2], l = s.m[2][3];
    const double m = s.m[3][0], n = s.m[3][1], o = s.m[3][2], p = s.m[3][3];
    const double q1 = a * (f * (k * p - l * o) - g * (j * p - l * n) + h * (j * o - k * n));
    const double q2 = b * (e * (k * p - l * o) - g * (i * p - l * m) + h * (i * o - k * m));
    const double q3 = c * (e * (j * p - l * n) - f * (i * p - l * m) + h * (i * n - j * m));
    const double q4 = d * (e * (j * o - k * n) - f * (i * o - k * m) + g * (i * n - j * m));
    return q1 - q2 + q3 - q4;
}

// 3x3 minor of the 4x4 at (row, col), expanded along row 0 exactly like the generic branch of
// Matrix::determinant (matrix.rs:110-117) over Matrix::cofactor (matrix.rs:142-151):
// det starts at 0.0 and accumulates row0[j] * (+/-)(2x2 determinant).
inline double mat_minor3(const Mat4 &a, int row, int col) {
    double s[3][3];
    for (int i = 0, r = 0; i < 4; i++) {
        if (i == row) continue;
        for (int j = 0, c = 0; j < 4; j++) {
            if (j == col) continue;
            s[r][c++] = a.m[i][j];
        }
        r++;
    }
    const double m00 = s[1][1] * s[2][2] - s[1][2] * s[2][1];
    const double m01 = s[1][0] * s[2][2] - s[1][2] * s[2][0];
    const double m02 = s[1][0] * s[2][1] - s[1][1] * s[2][0];
    double det = 0.0;
    det += s[0][0] * m00;
    det += s[0][1] * -m01;
    det += s[0][2] * m02;
    return det;
}

// matrix.rs:153-179.  false <=> det == 0.0 (NonInvertibleMatrix)
inline bool mat_inverse(const Mat4 &a, Mat4 &out) {
    const double det = mat_det(a);
    if (!(det != 0.0)) return false;
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) {
            const double minor = mat_minor3(a, row, col);
            const double cof = ((row + col) & 1) ? -minor : minor;
            out.m[col][row] = cof / det;
        }
    return true;
}

struct Vec3 {
    double x, y, z;
};
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
// tuple.rs:96-108
inline Vec3 vnorm(Vec3 a) {
    const double mag = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / mag, a.y / mag, a.z / mag};
}
// tuple.rs:114-120
inline Vec3 vcross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// matrix.rs:328-346 / 348-365 (host copies, used for bounding boxes only)
inline Vec3 mat_point(const Mat4 &m, Vec3 p) {
    return {m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3],
            m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3],
            m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3]};
}

// matrix.rs:181-236
inline Mat4 mat_translation(double x, double y, double z) {
    Mat4 r = Mat4::identity();
    r.m[0][3] = x, r.m[1][3] = y, r.m[2][3] = z;
    return r;
}
inline Mat4 mat_scaling(double x, double y, double z) {
    Mat4 r = Mat4::identity();
    r.m[0][0] = x, r.m[1][1] = y, r.m[2][2] = z;
    return r;
}
inline Mat4 mat_rotation_x(double rad) {
    Mat4 r = Mat4::identity();
    const double s = std::sin(rad), c = std::cos(rad);
    r.m[1][1] = c, r.m[2][1] = s, r.m[1][2] = -s, r.m[2][2] = c;
    return r;
}
inline Mat4 mat_rotation_y(double rad) {
    Mat4 r = Mat4::identity();
    const double s = std::sin(rad), c = std::cos(rad);
    r.m[0][0] = c, r.m[2][0] = -s, r.m[0][2] = s, r.m[2][2] = c;
    return r;
}
inline Mat4 mat_rotation_z(double rad) {
    Mat4 r = Mat4::identity();
    const double s = std::sin(rad), c = std::cos(rad);
    r.m[0][0] = c, r.m[1][0] = s, r.m[0][1] = -s, r.m[1][1] = c;
    return r;
}
inline Mat4 mat_shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
    Mat4 r = Mat4::identity();
    r.m[0][1] = xy, r.m[0][2] = xz, r.m[1][0] = yx, r.m[1][2] = yz, r.m[2][0] = zx, r.m[2][1] = zy;
    return r;
}

// transformation.rs:3-15
inline Mat4 mat_view(Vec3 from, Vec3 to, Vec3 up) {
    const Vec3 forward = vnorm(to - from);
    const Vec3 upn = vnorm(up);
    const Vec3 left = vcross(forward, upn);
    const Vec3 true_up = vcross(left, forward);
    Mat4 o = Mat4::identity();
    o.m[0][0] = left.x, o.m[0][1] = left.y, o.m[0][2] = left.z;
    o.m[1][0] = true_up.x, o.m[1][1] = true_up.y, o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x, o.m[2][1] = -forward.y, o.m[2][2] = -forward.z;
    return mat_mul(o, mat_translation(-from.x, -from.y, -from.z));
}

}  // namespace rtc
