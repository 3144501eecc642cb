// Device-side vector helpers and the counter-based random number generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define V4R_DEV __device__ __forceinline__

namespace v4r {

constexpr float EPSILON = 0.001f;  // src/lib.rs:34
constexpr float PI = 3.14159265358979323846f;
constexpr float TWO_PI = 6.28318530717958647692f;
constexpr float FRAC_1_PI = 0.318309886183790671538f;
#define V4R_INF __builtin_huge_valf()

struct V3 {
    float x, y, z;
};
V4R_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
V4R_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V4R_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V4R_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
V4R_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
V4R_DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
V4R_DEV V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
V4R_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V4R_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V4R_DEV float norm2(V3 a) { return dot(a, a); }
V4R_DEV float norm(V3 a) { return sqrtf(norm2(a)); }
V4R_DEV V3 normalize(V3 a) { return a / norm(a); }
V4R_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// row-major 3x3 (9 consecutive floats) times vector
V4R_DEV V3 mat_mul(const float *m, V3 v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z};
}

// ---- random numbers -------------------------------------------------------------------------------
// Every draw is a pure function of (seed, pass, pixel, segment, dimension): a path key is derived
// from (seed, pass, pixel), a segment key from (path key, segment index), and a draw hashes
// (segment key, dimension).  A job split over any number of GPUs therefore consumes identical
// streams.  Replaces the per-lane ThreadRng draws of src/random.rs:6-16.
V4R_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
V4R_DEV void path_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t pass, uint32_t pixel, uint32_t &k0, uint32_t &k1) {
    k0 = mix32(pixel + mix32(pass + mix32(seed_lo)));
    k1 = mix32(pass ^ mix32(pixel ^ mix32(seed_hi ^ 0x9E3779B9u)));
}
V4R_DEV uint32_t segment_key(uint32_t k0, uint32_t k1, uint32_t segment) { return mix32(k0 + segment * 0x9E3779B9u) ^ k1; }
V4R_DEV uint32_t draw_u32(uint32_t seg_key, uint32_t dim) { return mix32(seg_key + dim * 0x85EBCA6Bu); }
V4R_DEV float draw(uint32_t seg_key, uint32_t dim) { return (float)(draw_u32(seg_key, dim) >> 8) * (1.0f / 16777216.0f); }

enum {
    DIM_PIXEL_X = 0,
    DIM_PIXEL_Y = 1,
    DIM_LENS_R = 2,
    DIM_LENS_THETA = 3,
    DIM_TIME = 4,
    DIM_SCATTER = 8,       // 8..13: material / pdf sampling at the end of the segment
    DIM_PACKET_SLOT = 14,  // packet-statistics rule of the light pdf: is the ray in its shape's last, partial packet
    DIM_PACKET_ANY = 16,   // 16..31, + (light index & 15): does a packet mate hit that light
    DIM_MEDIUM = 32   // 32..63: constant-medium free-path draw, + (primitive index & 31)
};

}  // namespace v4r
