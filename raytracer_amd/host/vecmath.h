// Host-side float32 vector/matrix helpers used by the scene loader.
//
// The scene loader sits on the reference's side of the boundary, but its float results (camera
// basis, transform matrices, world-space mesh vertices) are inputs of the trace kernels, so every
// operation here keeps the operation ORDER of the reference's types (reference src/types.h:70-221):
// all arithmetic is binary32, sums are left to right, length() is the 3-argument std::hypot of
// libstdc++ (scaled form).  Build with -ffp-contract=off.
#pragma once

#include <cmath>
#include <cstdint>

namespace rta {

constexpr float kPi = 3.14159265358979323846f;   // reference src/types.h:19
constexpr float kInfinite = 3.402823466e+38f;    // FLT_MAX, reference src/types.h:20

struct Vec2 {
    float x = 0.0f, y = 0.0f;
};

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};

// libstdc++ __hypot3<float> (bits/cmath, C++17): a*sqrt((x/a)^2+(y/a)^2+(z/a)^2), a = max|.|
inline float hypot3(float x, float y, float z) {
    x = std::fabs(x);
    y = std::fabs(y);
    z = std::fabs(z);
    const float a = x < y ? (y < z ? z : y) : (x < z ? z : x);
    if (a == 0.0f) return 0.0f;
    const float qx = x / a, qy = y / a, qz = z / a;
    return a * std::sqrt(qx * qx + qy * qy + qz * qz);
}

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(Vec3 v) { return hypot3(v.x, v.y, v.z); }
inline Vec3 normalized(Vec3 v) {
    const float len = length(v);
    if (len == 0.0f) return {0.0f, 0.0f, 0.0f};
    return {v.x / len, v.y / len, v.z / len};
}
inline Vec3 reciprocal(Vec3 v) { return {1.0f / v.x, 1.0f / v.y, 1.0f / v.z}; }

// 3x4 affine matrix, row major, implied last row (0,0,0,1)
struct Mat34 {
    float m[3][4];

    static Mat34 identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}}}; }
    static Mat34 translation(Vec3 v) { return {{{1, 0, 0, v.x}, {0, 1, 0, v.y}, {0, 0, 1, v.z}}}; }
    static Mat34 scaling(Vec3 v) { return {{{v.x, 0, 0, 0}, {0, v.y, 0, 0}, {0, 0, v.z, 0}}}; }
    static Mat34 rotation_x(float rad) {
        const float s = sinf(rad), c = cosf(rad);
        return {{{1, 0, 0, 0}, {0, c, -s, 0}, {0, s, c, 0}}};
    }
    static Mat34 rotation_y(float rad) {
        const float s = sinf(rad), c = cosf(rad);
        return {{{c, 0, s, 0}, {0, 1, 0, 0}, {-s, 0, c, 0}}};
    }
    static Mat34 rotation_z(float rad) {
        const float s = sinf(rad), c = cosf(rad);
        return {{{c, -s, 0, 0}, {s, c, 0, 0}, {0, 0, 1, 0}}};
    }
    // camera basis: z = (eye-target)^, x = (up × z)^, y = (z × x)^; columns x,y,z,eye
    static Mat34 look_at(Vec3 eye, Vec3 target, Vec3 up) {
        const Vec3 z = normalized(eye - target);
        const Vec3 x = normalized(cross(up, z));
        const Vec3 y = normalized(cross(z, x));
        return {{{x.x, y.x, z.x, eye.x}, {x.y, y.y, z.y, eye.y}, {x.z, y.z, z.z, eye.z}}};
    }

    // point transform (adds the translation column)
    Vec3 apply(Vec3 p) const {
        return {p.x * m[0][0] + p.y * m[0][1] + p.z * m[0][2] + m[0][3],
                p.x * m[1][0] + p.y * m[1][1] + p.z * m[1][2] + m[1][3],
                p.x * m[2][0] + p.y * m[2][1] + p.z * m[2][2] + m[2][3]};
    }

    // matrix product; each entry is ((0 + a0*b0) + a1*b1) + a2*b2, translation column then gets
    // this->m[r][3] added (reference src/types.h:204-220)
    Mat34 operator*(const Mat34& rhs) const {
        Mat34 out;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 4; c++) {
                float s = 0.0f;
                for (int i = 0; i < 3; i++) s += m[r][i] * rhs.m[i][c];
                out.m[r][c] = s;
            }
            out.m[r][3] += m[r][3];
        }
        return out;
    }
};

}  // namespace rta
