// Small f32 vector helpers for the host-side scene pipeline.  Everything is computed in
// single precision, one rounding per operation (build with -ffp-contract=off), because the
// reference computes its host-side geometry in f32 (nalgebra Vector3<f32>).
#pragma once
#include <cmath>
#include <limits>

namespace v4ray {

constexpr float EPSILON = 0.001f;  // src/lib.rs:34

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit Vec3(const float *p) : x(p[0]), y(p[1]), z(p[2]) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    void store(float *p) const { p[0] = x; p[1] = y; p[2] = z; }
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float norm_squared(Vec3 a) { return dot(a, a); }
inline float norm(Vec3 a) { return std::sqrt(norm_squared(a)); }
inline Vec3 normalize(Vec3 a) { return a / norm(a); }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

// Row-major 3x3 matrix + translation: p' = m * p + t.
struct Affine {
    float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float t[3] = {0, 0, 0};
    Vec3 apply_point(Vec3 p) const {
        return {m[0] * p.x + m[1] * p.y + m[2] * p.z + t[0], m[3] * p.x + m[4] * p.y + m[5] * p.z + t[1],
                m[6] * p.x + m[7] * p.y + m[8] * p.z + t[2]};
    }
    Vec3 apply_vector(Vec3 p) const {
        return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[3] * p.x + m[4] * p.y + m[5] * p.z,
                m[6] * p.x + m[7] * p.y + m[8] * p.z};
    }
    float det() const {
        return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    }
    // Inverse of the affine map (what Projective3::inverse yields for an affine matrix,
    // src/hittable/transform.rs:18). Computed in double and rounded once.
    Affine inverse() const {
        double a[9];
        for (int i = 0; i < 9; ++i) a[i] = m[i];
        double d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
                   a[2] * (a[3] * a[7] - a[4] * a[6]);
        double id = 1.0 / d;
        double r[9] = {(a[4] * a[8] - a[5] * a[7]) * id, (a[2] * a[7] - a[1] * a[8]) * id,
                       (a[1] * a[5] - a[2] * a[4]) * id, (a[5] * a[6] - a[3] * a[8]) * id,
                       (a[0] * a[8] - a[2] * a[6]) * id, (a[2] * a[3] - a[0] * a[5]) * id,
                       (a[3] * a[7] - a[4] * a[6]) * id, (a[1] * a[6] - a[0] * a[7]) * id,
                       (a[0] * a[4] - a[1] * a[3]) * id};
        Affine out;
        for (int i = 0; i < 9; ++i) out.m[i] = (float)r[i];
        for (int i = 0; i < 3; ++i)
            out.t[i] = (float)(-(r[3 * i] * t[0] + r[3 * i + 1] * t[1] + r[3 * i + 2] * t[2]));
        return out;
    }
};

// Axis-aligned box, src/bvh/aabb.rs:9-66.
struct AABB {
    Vec3 min{std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(),
             std::numeric_limits<float>::infinity()};
    Vec3 max{-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
             -std::numeric_limits<float>::infinity()};
    static AABB empty() { return AABB(); }                       // aabb.rs:38-43
    static AABB with_bounds(Vec3 lo, Vec3 hi) { AABB b; b.min = lo; b.max = hi; return b; }
    bool is_empty() const { return min.x > max.x || min.y > max.y || min.z > max.z; }  // :44-46
    AABB join(const AABB &o) const { return with_bounds(vmin(min, o.min), vmax(max, o.max)); }  // :47-51
    AABB grow(Vec3 p) const { return with_bounds(vmin(min, p), vmax(max, p)); }                // :52-56
    Vec3 size() const { return max - min; }                      // :57-59
    Vec3 center() const { return min + size() * 0.5f; }          // :60-62
    float surface_area() const { return 2.0f * norm_squared(size()); }  // :63-65 (squared norm, as written there)
};

}  // namespace v4ray
