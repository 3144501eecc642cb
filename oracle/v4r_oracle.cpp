// v4r_oracle.cpp -- CPU restatement of the reference's render loop.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker for the HIP path: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, load or call it.  Nothing the product ships links it.
//
// It restates, one scalar ray at a time and in f32 with one rounding per operation (build with
// -ffp-contract=off), the arithmetic of the reference's Renderer::render -> ray_color path and
// everything that path calls; every function cites the reference file:line it follows
// (paths are relative to the reference tree's src/).
//
// Pinning status -- "parity unpinned": the reference ships no tests, golden vectors or fixtures,
// and it can be neither compiled (nightly Rust + un-vendored crates) nor imported (its extension
// module was never built) in the build environment.  The oracle is therefore pinned only by
// closed-form known-answer tests written from the reference's formulas (tests/test_oracle_kat.py).
// Third-party arithmetic the reference relies on (nalgebra Rotation3::face_towards, simba
// packed_simd comparisons/transcendentals, rand's gen_range) is restated from those crates'
// published definitions.
//
// Deliberate, documented departures:
//  * Random numbers: the reference draws from an OS-seeded ThreadRng (random.rs:6-16), which is
//    irreproducible.  Here every draw is a pure function of (seed, pass, pixel, segment,
//    dimension) -- the same counter-based generator the device uses -- so oracle and device
//    trace the same paths and can be compared per pixel.
//  * Packets: the reference processes 8 rays per SIMD packet.  Per-lane results do not depend
//    on the other lanes except in two places: HittableGroup::generate draws one light index per
//    packet (group.rs:91-95), and Samplable::value for rectangles and triangles
//    (aa_rect.rs:158-179, triangle.rs:110-131) gives a lane that misses the light the un-clipped
//    plane density when some lane of its packet hits it and 0 when none does.  Render mode 2
//    restates the packets themselves (render_pass_packets: rays regrouped by closest shape,
//    packets of 8 in first-hit order, renderer.rs:131-298).  The per-ray modes follow the scene's
//    flags, like the device: no flag = the packet-statistics rule (lights_value_stat: the packet
//    rule in expectation, from the number of rays of the pass and level that end on the shape),
//    V4R_FLAG_PLANE_LIGHT_PDF = the plane density always, V4R_FLAG_BOUNDED_LIGHT_PDF = 0 for a
//    direction that misses (the textbook density).  Only mode 2 is the reference's estimator
//    sample for sample; the packet-statistics rule has its mean (measured in the GPU tests).
//  * Recursion: ray_color's breadth-first recursion (renderer.rs:123-320) is run per path as a
//    loop carrying the product of the per-bounce coefficients; colour = emitted + coef * child
//    unrolls to the same sum of products.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "v4ray_hip.h"

namespace {

constexpr float EPSILON = 0.001f;  // lib.rs:34
constexpr float INF = std::numeric_limits<float>::infinity();
constexpr float PI = 3.14159265358979323846f;
constexpr float TWO_PI = 6.28318530717958647692f;
constexpr float FRAC_1_PI = 0.318309886183790671538f;

thread_local std::string g_error;

// ---- vectors ------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm2(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(norm2(a)); }
inline V3 normalize(V3 a) { return a / norm(a); }  // Unit::new_normalize
inline V3 mat_mul(const float *m, V3 v) {          // row-major 3x3
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z};
}

// ---- counter-based random numbers (replaces random.rs:6-16) ---------------------------------------
// mix: a 32-bit finaliser with full avalanche.  A path key is derived from (seed, pass, pixel);
// each path segment derives a segment key; a draw hashes (segment key, dimension).
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct PathKey {
    uint32_t k0, k1;
};
inline PathKey path_key(uint64_t seed, uint32_t pass, uint32_t pixel) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    PathKey k;
    k.k0 = mix32(pixel + mix32(pass + mix32(lo)));
    k.k1 = mix32(pass ^ mix32(pixel ^ mix32(hi ^ 0x9E3779B9u)));
    return k;
}
inline uint32_t segment_key(PathKey k, uint32_t segment) { return mix32(k.k0 + segment * 0x9E3779B9u) ^ k.k1; }
inline uint32_t draw_u32(uint32_t seg_key, uint32_t dim) { return mix32(seg_key + dim * 0x85EBCA6Bu); }
inline float draw(uint32_t seg_key, uint32_t dim) { return (float)(draw_u32(seg_key, dim) >> 8) * (1.0f / 16777216.0f); }
inline float draw_range(uint32_t seg_key, uint32_t dim, float lo, float hi) { return lo + (hi - lo) * draw(seg_key, dim); }

// dimension assignment within a segment
enum {
    DIM_PIXEL_X = 0,
    DIM_PIXEL_Y = 1,
    DIM_LENS_R = 2,
    DIM_LENS_THETA = 3,
    DIM_TIME = 4,
    DIM_SCATTER = 8,       // 8..13: material / pdf sampling at the end of the segment
    DIM_PACKET_SLOT = 14,  // packet-statistics rule: is the ray in its shape's last, partial packet
    DIM_PACKET_ANY = 16,   // 16..31: packet-statistics rule, + (light index & 15): does a packet mate hit that light (lights_value_stat)
    DIM_MEDIUM = 32   // 32..63: constant-medium free-path draw, + (primitive index & 31)
};

// ---- scene view ---------------------------------------------------------------------------------
struct Ray {
    V3 o, d;
};

struct Hit {  // HitRecord, hittable/mod.rs:24-32
    V3 p{0, 0, 0};
    V3 normal{0, 1, 0};
    float t = INF;
    float u = 0, v = 0;
    bool front_face = false;
    bool mask = false;
};

struct Counters {
    uint64_t segments = 0, nodes = 0, prims = 0, paths = 0, shaded = 0, texels = 0;
    void add(const Counters &o) {
        segments += o.segments;
        nodes += o.nodes;
        prims += o.prims;
        paths += o.paths;
        shaded += o.shaded;
        texels += o.texels;
    }
};

struct WideTree;
struct Ctx {
    const v4r_scene_desc *s;
    uint32_t seg_key = 0;  // key of the segment being traced / shaded
    uint32_t packet_light = 0xFFFFFFFFu;  // packet mode: the light index HittableGroup::generate drew for the whole packet
    uint32_t packet_n = 0;  // packet-statistics rule (mode 3): rays of this pass and level whose closest hit is the shape being shaded; 0 = rule off
    Counters cnt;
    const WideTree *wide = nullptr;  // mode 1 only
};

// HitRecord::face_normal, hittable/mod.rs:146-154 (is_simd_negative on packed f32 is `< 0`)
inline void face_normal(V3 dir, V3 outward, bool &front, V3 &normal) {
    front = dot(dir, outward) < 0.0f;
    normal = front ? outward : -outward;
}

// sphere_uv, hittable/sphere.rs:38-45
inline void sphere_uv(V3 p, float &u, float &v) {
    float theta = std::acos(-p.y);
    float phi = std::atan2(-p.z, p.x) + PI;
    u = phi / TWO_PI;
    v = theta * FRAC_1_PI;
}

// Sphere::hit, hittable/sphere.rs:60-93
Hit hit_sphere(const float *f, const Ray &r, float t_min, float t_max) {
    Hit h;
    V3 center = v3(f);
    float radius = f[3];
    V3 oc = r.o - center;
    float half_b = dot(oc, r.d);
    float c = norm2(oc) - radius * radius;
    float disc = half_b * half_b - c;
    if (disc < 0.0f) return h;
    float sqrt_d = std::sqrt(disc);
    float root1 = -half_b - sqrt_d;
    bool m1 = root1 >= t_min && root1 <= t_max;
    float root2 = -half_b + sqrt_d;
    bool m2 = root2 >= t_min && root2 <= t_max;
    if (!(m1 || m2)) return h;
    float t = m1 ? root1 : root2;
    V3 p = r.o + r.d * t;
    V3 outward = normalize(p - center);
    face_normal(r.d, outward, h.front_face, h.normal);
    sphere_uv(outward, h.u, h.v);
    h.p = p;
    h.t = t;
    h.mask = true;
    return h;
}

// axis triple of a rect type: a-axis, b-axis, k-axis  (aa_rect.rs:190-192)
inline void rect_axes(uint32_t type, int &ia, int &ib, int &ik) {
    switch (type & V4R_PRIM_TYPE_MASK) {
        case V4R_PRIM_XY_RECT: ia = 0; ib = 1; ik = 2; break;
        case V4R_PRIM_YZ_RECT: ia = 1; ib = 2; ik = 0; break;
        default: ia = 2; ib = 0; ik = 1; break;  // ZX
    }
}
inline V3 axis_vec(int axis, float s) { return axis == 0 ? v3(s, 0, 0) : (axis == 1 ? v3(0, s, 0) : v3(0, 0, s)); }

// $ty::hit, hittable/aa_rect.rs:114-144.  `fields_only` keeps computing t / normal after a failed
// range or bounds test, which is what a SIMD lane sees when another lane of its packet hits.
Hit hit_rect(const v4r_prim &pr, const Ray &r, float t_min, float t_max, bool fields_only = false) {
    Hit h;
    int ia, ib, ik;
    rect_axes(pr.type, ia, ib, ik);
    const float *f = pr.f;
    float t = (f[4] - r.o[ik]) / r.d[ik];
    bool mask = t >= t_min && t <= t_max;
    if (!mask && !fields_only) return h;
    float a = r.o[ia] + t * r.d[ia];
    float b = r.o[ib] + t * r.d[ib];
    mask = mask && a >= f[0] && a <= f[1] && b >= f[2] && b <= f[3];
    if (!mask && !fields_only) return h;
    h.u = (a - f[0]) / (f[1] - f[0]);
    h.v = (b - f[2]) / (f[3] - f[2]);
    V3 outward = axis_vec(ik, (pr.type & V4R_PRIM_POSITIVE) ? 1.0f : -1.0f);  // rect_norm!, aa_rect.rs:9-31
    face_normal(r.d, outward, h.front_face, h.normal);
    h.p = r.o + r.d * t;
    h.t = t;
    h.mask = mask;
    return h;
}

// Triangle::hit, hittable/triangle.rs:56-96
Hit hit_triangle(const v4r_prim &pr, const v4r_tri_shading &ts, const Ray &r, float t_min, float t_max,
                 bool fields_only = false) {
    Hit h;
    V3 v0 = v3(pr.f), e12 = v3(pr.f + 3), e13 = v3(pr.f + 6);
    V3 p_vec = cross(r.d, e13);
    float det = dot(e12, p_vec);
    bool mask = std::fabs(det) > 0.0f;
    if (!mask && !fields_only) return h;
    float inv_det = 1.0f / det;
    V3 t_vec = r.o - v0;
    float u = inv_det * dot(t_vec, p_vec);
    mask = mask && !(u < 0.0f) && u <= 1.0f;
    if (!mask && !fields_only) return h;
    V3 q_vec = cross(t_vec, e12);
    float v = inv_det * dot(r.d, q_vec);
    mask = mask && !(v < 0.0f) && (u + v) <= 1.0f;
    if (!mask && !fields_only) return h;
    float t = inv_det * dot(e13, q_vec);
    mask = mask && t >= t_min && t <= t_max;
    if (!mask && !fields_only) return h;
    float w = 1.0f - u - v;
    h.p = r.o + r.d * t;
    h.normal = normalize(v3(ts.n[0]) * w + v3(ts.n[1]) * u + v3(ts.n[2]) * v);  // not flipped toward the ray
    h.t = t;
    h.u = ts.uv[0][0] * w + ts.uv[1][0] * u + ts.uv[2][0] * v;
    h.v = ts.uv[0][1] * w + ts.uv[1][1] * u + ts.uv[2][1] * v;
    h.front_face = det > 0.0f;
    h.mask = mask;
    return h;
}

Hit hit_prim(Ctx &ctx, const v4r_prim &pr, uint32_t prim_index, const Ray &r, float t_min, float t_max);

// HittableGroup::hit, hittables/group.rs:58-67
Hit hit_group(Ctx &ctx, const v4r_prim *prims, uint32_t count, uint32_t prim_index, const Ray &r, float t_min,
              float t_max) {
    Hit best;
    float closest = t_max;
    for (uint32_t i = 0; i < count; ++i) {
        Hit h = hit_prim(ctx, prims[i], prim_index, r, t_min, closest);
        if (h.mask) {
            closest = h.t;
            best = h;
        }
    }
    return best;
}

// ConstantMedium::hit, hittable/constant_medium.rs:40-85
Hit hit_medium(Ctx &ctx, const v4r_prim &pr, uint32_t prim_index, const Ray &r, float t_min, float t_max) {
    const v4r_medium &m = ctx.s->media[pr.aux];
    const v4r_prim *boundary = ctx.s->medium_prims + m.first;
    Hit none;
    Hit h1 = hit_group(ctx, boundary, m.count, prim_index, r, -INF, INF);
    if (!h1.mask) return none;
    Hit h2 = hit_group(ctx, boundary, m.count, prim_index, r, h1.t + EPSILON, INF);
    if (!h2.mask) return none;
    float lo = std::fmax(h1.t, t_min);
    float hi = std::fmin(h2.t, t_max);
    if (!(lo < hi)) return none;
    lo = std::fmax(lo, 0.0f);
    float distance_inside = hi - lo;
    float u = draw(ctx.seg_key, DIM_MEDIUM + (prim_index & 31u));
    float hit_distance = m.neg_inv_density * std::log(u);
    if (!(hit_distance <= distance_inside)) return none;
    Hit h;
    h.t = h1.t + hit_distance;
    h.p = r.o + r.d * h.t;
    h.normal = v3(1, 0, 0);
    h.u = h.v = 0;
    h.front_face = true;
    h.mask = true;
    return h;
}

// TransformHittable::hit, hittable/transform.rs:64-95, around any primitive kind
Hit hit_prim(Ctx &ctx, const v4r_prim &pr, uint32_t prim_index, const Ray &r, float t_min, float t_max) {
    Ray lr = r;
    float nrm = 1.0f;
    const v4r_xform *x = nullptr;
    if (pr.xform != V4R_NO_XFORM) {
        x = &ctx.s->xforms[pr.xform];
        V3 dir = mat_mul(x->inv, r.d);
        nrm = norm(dir);
        lr.o = mat_mul(x->inv, r.o) + v3(x->inv_t);
        lr.d = dir / nrm;
        t_min = t_min * nrm;
        t_max = t_max * nrm;
    }
    Hit h;
    switch (pr.type & V4R_PRIM_TYPE_MASK) {
        case V4R_PRIM_SPHERE: h = hit_sphere(pr.f, lr, t_min, t_max); break;
        case V4R_PRIM_XY_RECT:
        case V4R_PRIM_YZ_RECT:
        case V4R_PRIM_ZX_RECT: h = hit_rect(pr, lr, t_min, t_max); break;
        case V4R_PRIM_TRIANGLE: h = hit_triangle(pr, ctx.s->tri_shading[pr.aux], lr, t_min, t_max); break;
        case V4R_PRIM_MEDIUM: h = hit_medium(ctx, pr, prim_index, lr, t_min, t_max); break;
        default: break;
    }
    if (x && h.mask) {
        h.p = mat_mul(x->fwd, h.p) + v3(x->fwd_t);
        h.normal = normalize(mat_mul(x->fwd, h.normal));
        h.t = h.t / nrm;
    }
    return h;
}

// ---- BVH traversal ------------------------------------------------------------------------------
// Ray::intersects_aabb, ray.rs:167-189 (divisions, running interval, strict tmin < tmax).
inline bool slab(const float *lo, const float *hi, const Ray &r, float t_min, float t_max, float &entry) {
    for (int a = 0; a < 3; ++a) {
        float min_val = (lo[a] - r.o[a]) / r.d[a];
        float max_val = (hi[a] - r.o[a]) / r.d[a];
        t_min = std::fmax(std::fmin(min_val, max_val), t_min);
        t_max = std::fmin(std::fmax(min_val, max_val), t_max);
        if (!(t_min < t_max)) return false;
    }
    entry = t_min;
    return true;
}

struct Closest {
    uint32_t prim = 0xFFFFFFFFu;
    Hit hit;
    uint32_t first = 0xFFFFFFFFu;  // lowest-index candidate shape the ray hits at all (orders renderer.rs:176's hit_ray_indices)
};

// closest-hit rule of ray_color, renderer.rs:149-183: strictly smaller t wins; shapes are visited in
// ascending index, so equal t keeps the lowest index.
inline void consider(Ctx &ctx, Closest &c, uint32_t prim_index, const Ray &r) {
    ctx.cnt.prims++;
    Hit h = hit_prim(ctx, ctx.s->prims[prim_index], prim_index, r, EPSILON, INF);
    if (!h.mask) return;
    if (prim_index < c.first) c.first = prim_index;
    if (c.hit.t > h.t || (c.hit.t == h.t && c.hit.mask && prim_index < c.prim)) {
        c.prim = prim_index;
        c.hit = h;
    }
}

// BVH::traverse, bvh/bvh.rs:182-239: every leaf whose ancestors' boxes the ray enters is a
// candidate; no near/far ordering, no interval shrinking.
void traverse_reference(Ctx &ctx, uint32_t node, const Ray &r, Closest &c) {
    const v4r_bvh_node &n = ctx.s->nodes[node];
    ctx.cnt.nodes++;
    float entry;
    if (slab(n.lo0, n.hi0, r, EPSILON, INF, entry)) {
        if (n.child0 & V4R_LEAF_BIT) consider(ctx, c, n.child0 & ~V4R_LEAF_BIT, r);
        else traverse_reference(ctx, n.child0, r, c);
    }
    if (slab(n.lo1, n.hi1, r, EPSILON, INF, entry)) {
        if (n.child1 & V4R_LEAF_BIT) consider(ctx, c, n.child1 & ~V4R_LEAF_BIT, r);
        else traverse_reference(ctx, n.child1, r, c);
    }
}

// The device's traversal, restated for the work counters (it returns the same closest hit as
// traverse_reference): a 4-wide tree made from the binary one -- a wide node takes the two children
// of a binary node and, while it has a free slot, replaces its interior child of largest surface
// area (first on ties) by that child's two children -- walked near to far with one stack that holds
// interior nodes and leaf primitives alike; an entry is skipped once its box entry distance exceeds
// the closest hit.  Entry distances are carried with their two lowest mantissa bits replaced by the
// child slot while the four children of a node are ordered, and cleared afterwards.
struct WideNode {
    float lo[4][3], hi[4][3];
    uint32_t child[4];  // interior wide index, V4R_LEAF_BIT | primitive, or 0xFFFFFFFF for an empty slot
    bool never_cull[4]; // the child's subtree holds a constant medium, whose hit may lie in front of its box entry
};
struct WideTree {
    std::vector<WideNode> nodes;
};

WideTree build_wide(const v4r_scene_desc *s) {
    WideTree w;
    if (s->n_nodes == 0) return w;
    struct Item {
        float lo[3], hi[3];
        uint32_t child;
    };
    auto split = [&](uint32_t node, Item &a, Item &b) {
        const v4r_bvh_node &n = s->nodes[node];
        for (int k = 0; k < 3; ++k) {
            a.lo[k] = n.lo0[k];
            a.hi[k] = n.hi0[k];
            b.lo[k] = n.lo1[k];
            b.hi[k] = n.hi1[k];
        }
        a.child = n.child0;
        b.child = n.child1;
    };
    // subtrees of the binary tree that hold a constant medium
    std::vector<char> holds(s->n_nodes, 0);
    auto is_medium = [&](uint32_t word) { return (s->prims[word & ~V4R_LEAF_BIT].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM; };
    if (s->n_media) {
        std::vector<std::pair<uint32_t, bool>> walk{{0u, false}};
        while (!walk.empty()) {
            auto [node, done] = walk.back();
            walk.pop_back();
            const v4r_bvh_node &n = s->nodes[node];
            if (!done) {
                walk.push_back({node, true});
                if (!(n.child0 & V4R_LEAF_BIT)) walk.push_back({n.child0, false});
                if (!(n.child1 & V4R_LEAF_BIT)) walk.push_back({n.child1, false});
            } else {
                bool a = (n.child0 & V4R_LEAF_BIT) ? is_medium(n.child0) : holds[n.child0] != 0;
                bool b = (n.child1 & V4R_LEAF_BIT) ? is_medium(n.child1) : holds[n.child1] != 0;
                holds[node] = a || b;
            }
        }
    }
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};  // (binary node, wide node)
    w.nodes.resize(1);
    while (!todo.empty()) {
        auto [binary, wide] = todo.back();
        todo.pop_back();
        Item items[4];
        int n = 2;
        split(binary, items[0], items[1]);
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n; ++k) {
                if (items[k].child & V4R_LEAF_BIT) continue;
                float dx = items[k].hi[0] - items[k].lo[0], dy = items[k].hi[1] - items[k].lo[1], dz = items[k].hi[2] - items[k].lo[2];
                float area = 2.0f * (dx * dy + dy * dz + dz * dx);
                if (!(area == area)) area = INF;
                if (pick < 0 || area > best) {
                    pick = k;
                    best = area;
                }
            }
            if (pick < 0) break;
            uint32_t node = items[pick].child;
            split(node, items[pick], items[n]);
            ++n;
        }
        WideNode out{};
        for (int k = 0; k < 4; ++k) {
            if (k >= n) {
                out.child[k] = 0xFFFFFFFFu;
                continue;
            }
            for (int a = 0; a < 3; ++a) {
                out.lo[k][a] = items[k].lo[a];
                out.hi[k][a] = items[k].hi[a];
            }
            out.never_cull[k] = s->n_media && ((items[k].child & V4R_LEAF_BIT) ? is_medium(items[k].child) : holds[items[k].child] != 0);
            if (items[k].child & V4R_LEAF_BIT) {
                out.child[k] = items[k].child;
            } else {
                out.child[k] = (uint32_t)w.nodes.size();
                w.nodes.emplace_back();
                todo.push_back({items[k].child, out.child[k]});
            }
        }
        w.nodes[wide] = out;
    }
    return w;
}

void traverse_wide(Ctx &ctx, const WideTree &tree, const Ray &r, Closest &c) {
    struct Entry {
        uint32_t child;
        float entry;
    };
    std::vector<Entry> stack;
    auto bits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    auto from_bits = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    Entry cur{0u, 0.0f};
    while (true) {
        if (cur.child & V4R_LEAF_BIT) {
            if (cur.entry <= c.hit.t) consider(ctx, c, cur.child & ~V4R_LEAF_BIT, r);
        } else {
            const WideNode &n = tree.nodes[cur.child];
            ctx.cnt.nodes++;
            uint32_t key[4];
            for (int k = 0; k < 4; ++k) {
                float e = 0.0f;
                bool hit = n.child[k] != 0xFFFFFFFFu && slab(n.lo[k], n.hi[k], r, EPSILON, INF, e) && (e <= c.hit.t || n.never_cull[k]);
                key[k] = hit ? ((bits(e) & ~3u) | (uint32_t)k) : 0xFFFFFFFFu;
            }
            std::sort(key, key + 4);
            auto entry_of = [&](uint32_t k) { return n.never_cull[k & 3u] ? -INF : from_bits(k & ~3u); };
            for (int k = 3; k >= 1; --k)
                if (key[k] != 0xFFFFFFFFu) stack.push_back({n.child[key[k] & 3u], entry_of(key[k])});
            if (key[0] != 0xFFFFFFFFu) {
                cur = {n.child[key[0] & 3u], entry_of(key[0])};
                continue;
            }
        }
        bool found = false;
        while (!stack.empty()) {
            Entry e = stack.back();
            stack.pop_back();
            if (e.entry <= c.hit.t) {
                cur = e;
                found = true;
                break;
            }
        }
        if (!found) return;
    }
}

// ---- the device's own tree, walked with the device's arithmetic (trace mode 4) --------------------------------------
// For the work counters of the tree that is actually timed: the device library builds a tree of its own from the
// scene's leaf boxes (4-wide, or 8-wide with the children's boxes on an 8-bit grid; leaves may hold several
// primitives) and hands it out through v4r_tree_read (include/v4ray_hip.h documents the node formats).  The walk
// below restates the kernels' node step one ray at a time: same operations in the same order on the same node words,
// so it visits the nodes and tests the primitives the device does; the closest hit is the reference's (consider()).
struct DeviceTree {
    v4r_tree_info info{};
    std::vector<uint32_t> nodes;        // 32 words per node
    std::vector<uint32_t> scene_index;  // the scene's index of the device's primitive k
    std::vector<uint32_t> media_direct; // constant media every ray tests before the tree (device numbering)
};
DeviceTree g_device_tree;

inline float as_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t as_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

void traverse_device_tree(Ctx &ctx, const DeviceTree &T, const Ray &r, Closest &c) {
    for (uint32_t k : T.media_direct) consider(ctx, c, T.scene_index[k], r);
    struct Entry {
        uint32_t child;
        float entry;
    };
    std::vector<Entry> stack;
    const bool wide8 = T.info.width == 8;
    float inv[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    if (wide8)
        for (float &v : inv) v = std::fmin(std::fmax(v, -1e18f), 1e18f);
    const float o[3] = {r.o.x, r.o.y, r.o.z};
    const bool masks = T.info.media_in_tree != 0;
    uint32_t cur = 0;
    while (true) {
        if (cur & V4R_LEAF_BIT) {
            const uint32_t first = cur & ((1u << 28) - 1u), count = (cur >> 28 & 7u) + 1u;
            for (uint32_t k = 0; k < count; ++k) consider(ctx, c, T.scene_index[first + k], r);
        } else {
            const uint32_t *w = T.nodes.data() + 32 * (size_t)cur;
            ctx.cnt.nodes++;
            uint32_t key[8], child[8], never_cull = 0;
            const int width = wide8 ? 8 : 4;
            if (wide8) {
                const uint32_t bits = w[3];
                never_cull = masks ? bits >> 24 & 255u : 0u;
                float a[3], b[3];
                for (int k = 0; k < 3; ++k) {
                    a[k] = (as_float(w[k]) - o[k]) * inv[k];
                    b[k] = as_float((bits >> (8 * k) & 255u) << 23) * inv[k];
                }
                for (int ch = 0; ch < 8; ++ch) {
                    child[ch] = w[4 + ch];
                    // words 12-17: lower x, y, z planes (two words each: children 0-3, 4-7), words 18-23: the upper planes
                    auto byte = [&](int plane) { return (float)(w[12 + 2 * plane + (ch >> 2)] >> (8 * (ch & 3)) & 255u); };
                    float entry = EPSILON, leave = INF;
                    float near_d[3], far_d[3];
                    for (int k = 0; k < 3; ++k) {
                        const bool down = inv[k] < 0.0f;  // the ray meets the upper plane first
                        near_d[k] = std::fma(byte(down ? 3 + k : k), b[k], a[k]);
                        far_d[k] = std::fma(byte(down ? k : 3 + k), b[k], a[k]);
                    }
                    entry = std::fmax(std::fmax(near_d[0], near_d[1]), std::fmax(near_d[2], EPSILON));
                    leave = std::fmin(std::fmin(far_d[0], far_d[1]), far_d[2]);
                    const bool wanted = entry <= c.hit.t || (never_cull >> ch & 1u);
                    key[ch] = (entry < leave && wanted && child[ch] != 0xFFFFFFFFu) ? ((as_bits(entry) & ~7u) | (uint32_t)ch) : 0xFFFFFFFFu;
                }
            } else {
                never_cull = masks ? w[28] : 0u;
                for (int ch = 0; ch < 4; ++ch) {
                    child[ch] = w[24 + ch];
                    float lo_d[3], hi_d[3];
                    for (int k = 0; k < 3; ++k) {
                        lo_d[k] = (as_float(w[4 * k + ch]) - o[k]) * inv[k];
                        hi_d[k] = (as_float(w[12 + 4 * k + ch]) - o[k]) * inv[k];
                    }
                    const float entry = std::fmax(std::fmax(std::fmin(lo_d[0], hi_d[0]), std::fmin(lo_d[1], hi_d[1])), std::fmax(std::fmin(lo_d[2], hi_d[2]), EPSILON));
                    const float leave = std::fmin(std::fmin(std::fmax(lo_d[0], hi_d[0]), std::fmax(lo_d[1], hi_d[1])), std::fmax(lo_d[2], hi_d[2]));
                    const bool wanted = entry <= c.hit.t || (never_cull >> ch & 1u);
                    key[ch] = (entry < leave && wanted) ? ((as_bits(entry) & ~3u) | (uint32_t)ch) : 0xFFFFFFFFu;
                }
            }
            std::sort(key, key + width);
            const uint32_t slot_mask = (uint32_t)width - 1u;
            auto entry_of = [&](uint32_t k) { return (never_cull >> (k & slot_mask) & 1u) ? -INF : as_float(k & ~slot_mask); };
            for (int k = width - 1; k >= 1; --k)
                if (key[k] != 0xFFFFFFFFu) stack.push_back({child[key[k] & slot_mask], entry_of(key[k])});
            if (key[0] != 0xFFFFFFFFu) {
                cur = child[key[0] & slot_mask];
                continue;
            }
        }
        bool found = false;
        while (!stack.empty()) {
            Entry e = stack.back();
            stack.pop_back();
            if (e.entry <= c.hit.t) {
                cur = e.child;
                found = true;
                break;
            }
        }
        if (!found) return;
    }
}

Closest trace(Ctx &ctx, const Ray &r, int mode) {
    Closest c;
    ctx.cnt.segments++;
    const v4r_scene_desc *s = ctx.s;
    if (s->n_prims == 0) return c;
    if (s->n_nodes == 0) {  // a single shape: the root is a leaf, tested without any box (bvh.rs:218-224)
        consider(ctx, c, 0, r);
        return c;
    }
    if (mode == 4 && g_device_tree.info.n_nodes != 0) traverse_device_tree(ctx, g_device_tree, r, c);
    else if (mode == 1) traverse_wide(ctx, *ctx.wide, r, c);
    else traverse_reference(ctx, 0, r, c);
    return c;
}

// ---- textures -----------------------------------------------------------------------------------
// Rust `x as isize` for f32: truncate toward zero, saturate, NaN -> 0
inline int64_t f32_to_isize(float x) {
    if (!(x == x)) return 0;
    if (x >= 9.2233720368547758e18f) return INT64_MAX;
    if (x <= -9.2233720368547758e18f) return INT64_MIN;
    return (int64_t)x;
}
inline uint32_t f32_to_u32(float x) {  // Rust `x as u32`
    if (!(x == x) || x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// Perlin::noise + perlin_interp, texture/noise.rs:41-89,109-140
float perlin_noise(const v4r_perlin &pt, V3 p) {
    float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int64_t i = f32_to_isize(fx), j = f32_to_isize(fy), k = f32_to_isize(fz);
    float uu = u * u * (3.0f - u * 2.0f);
    float vv = v * v * (3.0f - v * 2.0f);
    float ww = w * w * (3.0f - w * 2.0f);
    float accum = 0.0f;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                // wrapping add then rem_euclid(256) == low 8 bits of the two's-complement sum
                uint32_t ix = (uint32_t)((uint64_t)i + (uint64_t)di) & 255u;
                uint32_t iy = (uint32_t)((uint64_t)j + (uint64_t)dj) & 255u;
                uint32_t iz = (uint32_t)((uint64_t)k + (uint64_t)dk) & 255u;
                const float *c = pt.vec[pt.perm_x[ix] ^ pt.perm_y[iy] ^ pt.perm_z[iz]];
                V3 weight = v3(u - (float)di, v - (float)dj, w - (float)dk);
                float term = (di ? uu : 1.0f - uu) * (dj ? vv : 1.0f - vv) * (dk ? ww : 1.0f - ww) * dot(v3(c), weight);
                accum += term;
            }
    return accum;
}

// Perlin::turb, texture/noise.rs:91-107
float perlin_turb(const v4r_perlin &pt, V3 p, uint32_t depth) {
    float accum = 0.0f, weight = 1.0f;
    for (uint32_t d = 0; d < depth; ++d) {
        accum += weight * perlin_noise(pt, p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return std::fabs(accum);
}

V3 texture_value(Ctx &ctx, uint32_t tex, float u, float v, V3 p) {
    const v4r_scene_desc *s = ctx.s;
    for (int guard = 0; guard < 64; ++guard) {
        const v4r_texture &t = s->textures[tex];
        switch (t.kind) {
            case V4R_TEX_SOLID:  // texture/solid_color.rs:25-27
                return v3(t.color);
            case V4R_TEX_CHECKER: {  // texture/checker.rs:31-38 (is_simd_positive is `> 0`)
                float sines = std::sin(t.s * p.x) * std::sin(t.s * p.y) * std::sin(t.s * p.z);
                tex = sines > 0.0f ? t.b : t.a;  // even : odd
                continue;
            }
            case V4R_TEX_IMAGE: {  // texture/image.rs:25-73
                const v4r_image &img = s->images[t.a];
                float cu = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
                float cv = 1.0f - (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
                uint32_t i = f32_to_u32((float)img.width * cu);
                if (!(i < img.width)) i = img.width - 1;
                uint32_t j = f32_to_u32((float)img.height * cv);
                if (!(j < img.height)) j = img.height - 1;
                const uint8_t *px = s->texels + img.offset + ((size_t)j * img.width + i) * 3;
                ctx.cnt.texels++;
                return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
            }
            case V4R_TEX_NOISE: {  // texture/noise.rs:165-170
                float g = perlin_turb(s->perlin[t.a], p * t.s, t.b);
                return v3(g, g, g);
            }
            default: return v3(0, 0, 0);
        }
    }
    return v3(0, 0, 0);
}

// ---- pdfs ---------------------------------------------------------------------------------------
// Rotation3::face_towards(dir, up) (nalgebra 0.27): columns x = normalize(up x z), y = normalize(z x x'),
// z = normalize(dir).
struct Rot {
    V3 x, y, z;
    V3 apply(V3 v) const { return x * v.x + y * v.y + z * v.z; }
};
inline Rot face_towards(V3 dir, V3 up) {
    Rot r;
    r.z = normalize(dir);
    r.x = normalize(cross(up, r.z));
    r.y = normalize(cross(r.z, r.x));
    return r;
}
// `up` selection shared by CosinePdf::new (pdf/cosine.rs:16-25) and Sphere::generate (sphere.rs:134-140)
inline V3 pick_up(float x_component) { return std::fabs(x_component) > 0.9f ? v3(0, 1, 0) : v3(1, 0, 0); }

// Matrix3 * Vector3 as nalgebra evaluates it: column-major accumulation x*col0 + y*col1 + z*col2
inline V3 rot_apply(const Rot &r, V3 v) { return r.apply(v); }

// CosinePdf::value, pdf/cosine.rs:32-37
inline float cosine_value(V3 normal, V3 dir) {
    float c = dot(dir, normal);
    return c > 0.0f ? c * FRAC_1_PI : 0.0f;
}
// CosinePdf::generate, pdf/cosine.rs:38-47
inline V3 cosine_generate(V3 normal, float r1, float r2) {
    Rot rot = face_towards(normal, pick_up(normal.x));
    float z = std::sqrt(1.0f - r2);
    float phi = TWO_PI * r1;
    float sq = std::sqrt(r2);
    float x = std::cos(phi) * sq;
    float y = std::sin(phi) * sq;
    return rot_apply(rot, v3(x, y, z));
}

// random_to_sphere, random.rs:27-39
inline V3 random_to_sphere(float radius, float distance_squared, float u_phi, float r2) {
    float phi = TWO_PI * u_phi;
    float z = 1.0f + r2 * (std::sqrt(1.0f - radius * radius / distance_squared) - 1.0f);
    float xy = std::sqrt(1.0f - z * z);
    return v3(std::cos(phi) * xy, std::sin(phi) * xy, z);
}

// Samplable::value of one light primitive (object space), for direction `dir` from `origin`.
float light_value_object(Ctx &ctx, const v4r_prim &pr, V3 origin, V3 dir) {
    Ray r{origin, dir};
    bool bounded = ctx.s->flags & V4R_FLAG_BOUNDED_LIGHT_PDF;
    switch (pr.type & V4R_PRIM_TYPE_MASK) {
        case V4R_PRIM_SPHERE: {  // hittable/sphere.rs:100-131
            Hit h = hit_sphere(pr.f, r, EPSILON, INF);
            if (!h.mask) return 0.0f;
            float radius = pr.f[3];
            float cos_theta_max = std::sqrt(1.0f - radius * radius / norm2(v3(pr.f) - origin));
            float solid_angle = TWO_PI * (1.0f - cos_theta_max);
            return solid_angle > 0.0f ? 1.0f / solid_angle : INF;
        }
        case V4R_PRIM_XY_RECT:
        case V4R_PRIM_YZ_RECT:
        case V4R_PRIM_ZX_RECT: {  // hittable/aa_rect.rs:151-180
            Hit h = hit_rect(pr, r, EPSILON, INF, !bounded);
            if (bounded && !h.mask) return 0.0f;
            float area = (pr.f[1] - pr.f[0]) * (pr.f[3] - pr.f[2]);
            float distance_squared = h.t * h.t;
            float cosine = std::fabs(dot(dir, h.normal));
            return cosine > 0.0f ? distance_squared / (cosine * area) : INF;
        }
        case V4R_PRIM_TRIANGLE: {  // hittable/triangle.rs:103-132
            Hit h = hit_triangle(pr, ctx.s->tri_shading[pr.aux], r, EPSILON, INF, !bounded);
            if (bounded && !h.mask) return 0.0f;
            float area = norm(cross(v3(pr.f + 3), v3(pr.f + 6))) * 0.5f;
            float distance_squared = h.t * h.t;
            float cosine = std::fabs(dot(dir, h.normal));
            return cosine > 0.0f ? distance_squared / (cosine * area) : INF;
        }
        default: return 0.0f;  // Samplable default, hittable/mod.rs:186-194
    }
}

// Samplable::generate of one light primitive (object space) from `origin`; u0,u1 are the two draws
// in the order the reference makes them.
V3 light_generate_object(const v4r_prim &pr, V3 origin, float u0, float u1) {
    switch (pr.type & V4R_PRIM_TYPE_MASK) {
        case V4R_PRIM_SPHERE: {  // hittable/sphere.rs:132-144
            V3 direction = v3(pr.f) - origin;
            Rot rot = face_towards(direction, pick_up(normalize(direction).x));
            return normalize(rot_apply(rot, random_to_sphere(pr.f[3], norm2(direction), u0, u1)));
        }
        case V4R_PRIM_XY_RECT: {  // rect_random!, aa_rect.rs:55-61: a first, then b
            float a = pr.f[0] + (pr.f[1] - pr.f[0]) * u0, b = pr.f[2] + (pr.f[3] - pr.f[2]) * u1;
            return normalize(v3(a, b, pr.f[4]) - origin);
        }
        case V4R_PRIM_YZ_RECT: {  // aa_rect.rs:62-68
            float a = pr.f[0] + (pr.f[1] - pr.f[0]) * u0, b = pr.f[2] + (pr.f[3] - pr.f[2]) * u1;
            return normalize(v3(pr.f[4], a, b) - origin);
        }
        case V4R_PRIM_ZX_RECT: {  // aa_rect.rs:69-75: the x coordinate (b range) is drawn first
            float b = pr.f[2] + (pr.f[3] - pr.f[2]) * u0, a = pr.f[0] + (pr.f[1] - pr.f[0]) * u1;
            return normalize(v3(b, pr.f[4], a) - origin);
        }
        case V4R_PRIM_TRIANGLE: {  // hittable/triangle.rs:134-146
            float lo = EPSILON, hi = 1.0f - EPSILON;
            float x = lo + (hi - lo) * u0, y = lo + (hi - lo) * u1;
            if (x + y > 1.0f) {
                x = 1.0f - EPSILON - x;
                y = 1.0f - EPSILON - y;
            }
            V3 point = v3(pr.f) + v3(pr.f + 3) * x + v3(pr.f + 6) * y;
            return normalize(point - origin);
        }
        default: return v3(1, 0, 0);  // hittable/mod.rs:195-197
    }
}

// TransformHittable::{value, generate}, hittable/transform.rs:103-124 (no Jacobian, as written there)
float light_value(Ctx &ctx, uint32_t prim_index, V3 origin, V3 dir) {
    const v4r_prim &pr = ctx.s->prims[prim_index];
    if (pr.xform != V4R_NO_XFORM) {
        const v4r_xform &x = ctx.s->xforms[pr.xform];
        origin = mat_mul(x.inv, origin) + v3(x.inv_t);
        dir = normalize(mat_mul(x.inv, dir));
    }
    return light_value_object(ctx, pr, origin, dir);
}
V3 light_generate(Ctx &ctx, uint32_t prim_index, V3 origin, float u0, float u1) {
    const v4r_prim &pr = ctx.s->prims[prim_index];
    if (pr.xform != V4R_NO_XFORM) {
        const v4r_xform &x = ctx.s->xforms[pr.xform];
        origin = mat_mul(x.inv, origin) + v3(x.inv_t);
        return normalize(mat_mul(x.fwd, light_generate_object(pr, origin, u0, u1)));
    }
    return light_generate_object(pr, origin, u0, u1);
}

// The two ingredients of Samplable::value for one lane of a SIMD packet (aa_rect.rs:151-180, triangle.rs:103-132,
// sphere.rs:100-131, through TransformHittable::value): does this lane's direction hit the light, and the density
// the lane is given when the hit record of its packet is not all-miss.  For rectangles and triangles that density is
// computed from the un-clipped plane hit (t may be negative); a sphere masks with its own hit.
float light_probe(Ctx &ctx, uint32_t prim_index, V3 origin, V3 dir, bool &hit, bool &per_lane) {
    const v4r_prim &pr = ctx.s->prims[prim_index];
    if (pr.xform != V4R_NO_XFORM) {
        const v4r_xform &x = ctx.s->xforms[pr.xform];
        origin = mat_mul(x.inv, origin) + v3(x.inv_t);
        dir = normalize(mat_mul(x.inv, dir));
    }
    Ray r{origin, dir};
    per_lane = false;
    hit = false;
    switch (pr.type & V4R_PRIM_TYPE_MASK) {
        case V4R_PRIM_SPHERE: {
            per_lane = true;
            Hit h = hit_sphere(pr.f, r, EPSILON, INF);
            hit = h.mask;
            if (!h.mask) return 0.0f;
            float radius = pr.f[3];
            float cos_theta_max = std::sqrt(1.0f - radius * radius / norm2(v3(pr.f) - origin));
            float solid_angle = TWO_PI * (1.0f - cos_theta_max);
            return solid_angle > 0.0f ? 1.0f / solid_angle : INF;
        }
        case V4R_PRIM_XY_RECT:
        case V4R_PRIM_YZ_RECT:
        case V4R_PRIM_ZX_RECT: {
            Hit h = hit_rect(pr, r, EPSILON, INF, true);
            hit = h.mask;
            float area = (pr.f[1] - pr.f[0]) * (pr.f[3] - pr.f[2]);
            float cosine = std::fabs(dot(dir, h.normal));
            return cosine > 0.0f ? (h.t * h.t) / (cosine * area) : INF;
        }
        case V4R_PRIM_TRIANGLE: {
            Hit h = hit_triangle(pr, ctx.s->tri_shading[pr.aux], r, EPSILON, INF, true);
            hit = h.mask;
            float area = norm(cross(v3(pr.f + 3), v3(pr.f + 6))) * 0.5f;
            float cosine = std::fabs(dot(dir, h.normal));
            return cosine > 0.0f ? (h.t * h.t) / (cosine * area) : INF;
        }
        default: per_lane = true; return 0.0f;
    }
}

// HittableGroup::value over the light list, hittables/group.rs:76-89
float lights_value(Ctx &ctx, V3 origin, V3 dir) {
    uint32_t n = ctx.s->n_lights;
    float weight = 1.0f / (float)n;
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; ++i) sum += weight * light_value(ctx, ctx.s->lights[i], origin, dir);
    return sum;
}

// The packet-statistics rule (mode 3; the device's default): what the reference's 8-lane packets do to the light pdf,
// in expectation, without forming the packets.  In the reference a lane whose direction misses a rectangle or triangle
// light still gets the un-clipped plane density when some lane of its packet hits that light, and 0 when none does
// (aa_rect.rs:158-179, triangle.rs:110-131).  A packet holds rays of one pass and level that hit the same shape, so:
//   * k, the size of the ray's packet: the shape receives `n` rays; they are cut into packets of 8, the last one
//     holding n mod 8; the ray sits in that last packet with probability (n mod 8) / n (one uniform draw);
//   * each of the k - 1 mates samples the light list half of the time (mixture.rs:27-32) -- the light the packet
//     drew, `li`, whose sampled direction hits it -- and a cosine direction otherwise, which hits light L with
//     probability h_L (its cosine-weighted solid angle, estimated from its centre, area and normal);
//   * so no mate hits L with probability (1 - (0.5 [L == li] + 0.5 h_L))^(k-1); one more uniform draw per light
//     decides, and the lane gets 0 or the plane density accordingly.  Sphere lights mask per lane (sphere.rs:107-130).
struct PacketRule {
    static constexpr uint32_t COUNTED_LEVELS = V4R_PACKET_COUNTED_LEVELS;  // levels whose per-shape ray counts are taken; deeper levels reuse the last
    static constexpr uint32_t MANY_RAYS = V4R_PACKET_MANY_RAYS;            // a shape with this many rays or more: every packet counts as full
};

// h_L of the rule above, for an untransformed rectangle or triangle light seen from surface point p with normal n.
float light_cosine_share(const v4r_prim &pr, V3 p, V3 n) {
    if (pr.xform != V4R_NO_XFORM) return 0.0f;
    V3 centre, ln;
    float area;
    uint32_t kind = pr.type & V4R_PRIM_TYPE_MASK;
    if (kind == V4R_PRIM_TRIANGLE) {
        V3 e12 = v3(pr.f + 3), e13 = v3(pr.f + 6), c = cross(e12, e13);
        centre = v3(pr.f) + (e12 + e13) * (1.0f / 3.0f);
        float len = norm(c);
        ln = c / len;
        area = len * 0.5f;
    } else if (kind >= V4R_PRIM_XY_RECT && kind <= V4R_PRIM_ZX_RECT) {
        int ia, ib, ik;
        rect_axes(pr.type, ia, ib, ik);
        float c[3];
        c[ia] = 0.5f * (pr.f[0] + pr.f[1]);
        c[ib] = 0.5f * (pr.f[2] + pr.f[3]);
        c[ik] = pr.f[4];
        centre = v3(c);
        ln = axis_vec(ik, 1.0f);
        area = (pr.f[1] - pr.f[0]) * (pr.f[3] - pr.f[2]);
    } else {
        return 0.0f;
    }
    V3 w = centre - p;
    float d2 = norm2(w), d = std::sqrt(d2);
    float cos_l = std::fabs(dot(w, ln)) / d, cos_p = dot(w, n) / d;
    if (!(cos_p > 0.0f)) return 0.0f;
    float h = area * cos_l * cos_p / (PI * d2);
    return h < 1.0f ? h : 1.0f;  // NaN (p at the centre) -> 1
}

// HittableGroup::value (group.rs:76-89) under the packet-statistics rule: `n` rays hit this shape, `li` is the light
// index the ray drew, `normal` the shading normal at `origin`.
float lights_value_stat(Ctx &ctx, V3 origin, V3 normal, V3 dir, uint32_t li, uint32_t n, uint32_t key) {
    const v4r_scene_desc *s = ctx.s;
    uint32_t k = 8;
    if (n < 8) {
        k = n < 1 ? 1 : n;
    } else if (n < PacketRule::MANY_RAYS) {
        uint32_t r = n % 8u;
        if (r != 0u && draw(key, DIM_PACKET_SLOT) * (float)n < (float)r) k = r;
    }
    const float weight = 1.0f / (float)s->n_lights;
    float sum = 0.0f;
    for (uint32_t l = 0; l < s->n_lights; ++l) {
        bool hit, per_lane;
        float density = light_probe(ctx, s->lights[l], origin, dir, hit, per_lane);
        if (!per_lane && !hit) {
            float q = 1.0f - (0.5f * (l == li ? 1.0f : 0.0f) + 0.5f * light_cosine_share(s->prims[s->lights[l]], origin, normal));
            float p_none = 1.0f;
            for (uint32_t m = 1; m < k; ++m) p_none = p_none * q;
            // one draw per light: dimension 16 + (l mod 16) of the segment's stream, for lights beyond the sixteenth of a key derived from it
            const uint32_t light_key = l < 16u ? key : mix32(key + (l >> 4) * 0x9E3779B9u);
            if (!(draw(light_key, DIM_PACKET_ANY + (l & 15u)) >= p_none)) density = 0.0f;
        }
        sum += weight * density;
    }
    return sum;
}

// ---- materials ----------------------------------------------------------------------------------
inline V3 reflect(V3 v, V3 n) { return v - n * (dot(v, n) * 2.0f); }  // material/mod.rs:47-52
inline V3 refract(V3 uv, V3 n, float etai_over_etat) {               // material/mod.rs:54-63
    float cos_theta = -dot(uv, n);
    V3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    V3 r_out_parallel = n * (-std::sqrt(1.0f - norm2(r_out_perp)));
    return r_out_perp + r_out_parallel;
}
inline float reflectance(float cosine, float ref_idx) {  // material/dielectric.rs:23-27
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    return r0 + (1.0f - r0) * (x2 * x2 * x);
}

struct Scatter {
    bool scattered = false;  // false = ScatterRecord::None
    V3 coef{0, 0, 0};
    V3 dir{0, 0, 0};
    V3 emitted{0, 0, 0};
    float pdf_v = 0.0f;  // packet mode only: the material pdf's value for `dir`, `coef` still being the bare attenuation
};

// Material::emitted + Material::scatter + the pdf branch of ray_color (renderer.rs:218-263)
Scatter shade(Ctx &ctx, uint32_t prim_index, const Ray &ray, const Hit &h) {
    const v4r_scene_desc *s = ctx.s;
    const v4r_material &m = s->materials[s->prim_material[prim_index]];
    uint32_t key = ctx.seg_key;
    Scatter out;
    ctx.cnt.shaded++;
    switch (m.kind) {
        case V4R_MAT_DIFFUSE_LIGHT: {  // material/diffuse_light.rs:19-22; scatter is the default None
            V3 c = texture_value(ctx, m.tex, h.u, h.v, h.p);
            out.emitted = h.front_face ? c : v3(0, 0, 0);
            return out;
        }
        case V4R_MAT_LAMBERTIAN: {  // material/lambertian.rs:36-46
            V3 attenuation = texture_value(ctx, m.tex, h.u, h.v, h.p);
            out.scattered = true;
            if (s->n_lights != 0) {  // renderer.rs:234-244
                // MixturePdf::generate, pdf/mixture.rs:27-32: both members generate, then select
                bool pick_light = draw(key, DIM_SCATTER + 0) < 0.5f;
                // HittableGroup::generate, hittables/group.rs:91-95 (one index per SIMD packet there: `packet_light`)
                uint32_t n = s->n_lights;
                uint32_t li = ctx.packet_light != 0xFFFFFFFFu ? ctx.packet_light : (uint32_t)(((uint64_t)draw_u32(key, DIM_SCATTER + 1) * n) >> 32);
                V3 d_light = light_generate(ctx, s->lights[li], h.p, draw(key, DIM_SCATTER + 2), draw(key, DIM_SCATTER + 3));
                V3 d_cos = cosine_generate(h.normal, draw(key, DIM_SCATTER + 4), draw(key, DIM_SCATTER + 5));
                V3 dir = pick_light ? d_light : d_cos;
                float pdf_v = cosine_value(h.normal, dir);
                out.dir = dir;
                if (ctx.packet_light != 0xFFFFFFFFu) {  // the packet finishes the coefficient: see shade_packet
                    out.coef = attenuation;
                    out.pdf_v = pdf_v;
                    return out;
                }
                float light_v = ctx.packet_n != 0 ? lights_value_stat(ctx, h.p, h.normal, dir, li, ctx.packet_n, key) : lights_value(ctx, h.p, dir);
                float mix_v = 0.5f * light_v + 0.5f * pdf_v;  // pdf/mixture.rs:22-25
                out.coef = attenuation * pdf_v / mix_v;
            } else {  // renderer.rs:245-255
                out.dir = cosine_generate(h.normal, draw(key, DIM_SCATTER + 4), draw(key, DIM_SCATTER + 5));
                out.coef = attenuation;
            }
            return out;
        }
        case V4R_MAT_METAL: {  // material/metal.rs:31-46, random_in_unit_sphere random.rs:53-65
            V3 reflected = reflect(ray.d, h.normal);
            float theta = TWO_PI * draw(key, DIM_SCATTER + 0);
            float cos_phi = -1.0f + 2.0f * draw(key, DIM_SCATTER + 1);
            float r = std::pow(draw(key, DIM_SCATTER + 2), 1.0f / 3.0f);
            float sin_phi = std::sqrt(1.0f - cos_phi * cos_phi);
            V3 ball = v3(r * sin_phi * std::cos(theta), r * sin_phi * std::sin(theta), r * cos_phi);
            out.scattered = true;
            out.dir = normalize(reflected + ball * m.fuzz);
            out.coef = v3(m.albedo);
            return out;
        }
        case V4R_MAT_DIELECTRIC: {  // material/dielectric.rs:33-60
            float ratio = h.front_face ? 1.0f / m.ir : m.ir;
            float cos_theta = -dot(ray.d, h.normal);
            float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
            bool cannot_refract = ratio * sin_theta > 1.0f;
            cannot_refract = cannot_refract || reflectance(cos_theta, ratio) > draw(key, DIM_SCATTER + 0);
            out.scattered = true;
            out.dir = cannot_refract ? reflect(ray.d, h.normal) : refract(ray.d, h.normal, ratio);
            out.coef = v3(1, 1, 1);
            return out;
        }
        case V4R_MAT_ISOTROPIC: {  // material/isotropic.rs:26-43, random_on_unit_sphere random.rs:41-51
            float z = -1.0f + 2.0f * draw(key, DIM_SCATTER + 0);
            float theta = TWO_PI * draw(key, DIM_SCATTER + 1);
            float xy = std::sqrt(1.0f - z * z);
            out.scattered = true;
            out.dir = v3(std::cos(theta) * xy, std::sin(theta) * xy, z);
            out.coef = texture_value(ctx, m.tex, h.u, h.v, h.p);
            return out;
        }
        default: return out;
    }
}

// ---- camera -------------------------------------------------------------------------------------
// Renderer::sample (renderer.rs:94-122) + Camera::get_ray (camera.rs:113-129) for pixel (row, col),
// row 0 = top of the image.
Ray camera_ray(const v4r_scene_desc *s, uint32_t row, uint32_t col, uint32_t key, bool deterministic) {
    const v4r_camera &c = s->camera;
    float width = (float)s->width, height = (float)s->height;
    float i = (float)col, j = (float)(s->height - 1 - row);
    float sx, sy;
    if (deterministic || !s->antialias) {
        sx = i / width;
        sy = j / height;
    } else {
        sx = draw_range(key, DIM_PIXEL_X, (i - 0.5f) / width, (i + 0.5f) / width);
        sy = draw_range(key, DIM_PIXEL_Y, (j - 0.5f) / height, (j + 0.5f) / height);
    }
    V3 offset = v3(0, 0, 0);
    if (!deterministic) {
        // random_in_unit_disk, random.rs:18-25
        float r = std::sqrt(draw(key, DIM_LENS_R));
        float theta = TWO_PI * draw(key, DIM_LENS_THETA);
        float rdx = r * std::cos(theta) * c.lens_radius, rdy = r * std::sin(theta) * c.lens_radius;
        offset = v3(c.u) * rdx + v3(c.v) * rdy;
    }
    V3 source = v3(c.origin) + offset;
    V3 target = v3(c.lower_left_corner) + v3(c.horizontal) * sx + v3(c.vertical) * sy;
    return Ray{source, normalize(target - source)};
}

// One camera path: the per-path unrolling of ray_color (renderer.rs:123-320).
V3 path_radiance(Ctx &ctx, uint64_t seed, uint32_t pass, uint32_t pixel, int mode) {
    const v4r_scene_desc *s = ctx.s;
    PathKey pk = path_key(seed, pass, pixel);
    ctx.cnt.paths++;
    ctx.seg_key = segment_key(pk, 0);
    Ray ray = camera_ray(s, pixel / s->width, pixel % s->width, ctx.seg_key, false);
    V3 thr = v3(1, 1, 1);
    for (uint32_t seg = 0; seg < s->max_depth; ++seg) {
        ctx.seg_key = segment_key(pk, seg);
        Closest c = trace(ctx, ray, mode);
        if (!c.hit.mask) return thr * v3(s->background);  // renderer.rs:184
        Scatter sc = shade(ctx, c.prim, ray, c.hit);
        if (!sc.scattered) return thr * sc.emitted;  // renderer.rs:218-230,261
        thr = thr * sc.coef;                         // renderer.rs:299-318
        ray = Ray{c.hit.p, sc.dir};
    }
    return thr * v3(s->environment);  // renderer.rs:128-130
}

// ---- packet-faithful pass (mode 2) ----------------------------------------------------------------
// ray_color as the reference runs it (renderer.rs:123-320): breadth first over the whole image, one recursion
// level at a time, with the reference's grouping of rays into 8-lane packets wherever the grouping changes a
// result.  Two things do: HittableGroup::generate draws ONE light index per packet (group.rs:91-95), and
// Samplable::value of a rectangle or triangle light returns 0 for every lane when no lane of the packet hits the
// light, the un-clipped plane density for every lane otherwise (aa_rect.rs:158-179, triangle.rs:110-131).
// The packets of the shading step are: the rays whose closest hit is shape s, for ascending s (BTreeMap,
// renderer.rs:189-204), in the order the rays first received a hit -- ascending (lowest-index candidate shape
// the ray hits, ray index), renderer.rs:152-183 -- cut into groups of LANES = 8 (renderer.rs:214-217); the
// rays of the next level are the scattered rays in that same order (renderer.rs:264-271).
// Per-ray arithmetic and random streams are those of path_radiance (mode 0): throughput carried as a product.
constexpr uint32_t LANES = 8;  // f32x8, src/main.rs:14

struct LevelRay {
    Ray ray;
    V3 thr;
    uint32_t pixel;
    PathKey pk;
};

// The scatter step of one packet of `n` <= LANES rays that hit shape `prim_index` (renderer.rs:214-272).
void shade_packet(Ctx &ctx, uint32_t prim_index, uint32_t seg, const LevelRay *const *rays, const Hit *hits, uint32_t n, Scatter *out) {
    const v4r_scene_desc *s = ctx.s;
    const v4r_material &m = s->materials[s->prim_material[prim_index]];
    const bool mixture = m.kind == V4R_MAT_LAMBERTIAN && s->n_lights != 0;
    if (mixture)  // one index for the packet; lane 0's stream stands in for the packet's draw
        ctx.packet_light = (uint32_t)(((uint64_t)draw_u32(segment_key(rays[0]->pk, seg), DIM_SCATTER + 1) * s->n_lights) >> 32);
    for (uint32_t k = 0; k < n; ++k) {
        ctx.seg_key = segment_key(rays[k]->pk, seg);
        out[k] = shade(ctx, prim_index, rays[k]->ray, hits[k]);
    }
    ctx.packet_light = 0xFFFFFFFFu;
    if (!mixture) return;
    // MixturePdf::value over the packet (mixture.rs:22-25) with HittableGroup::value (group.rs:76-89)
    float light_sum[LANES] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float weight = 1.0f / (float)s->n_lights;
    for (uint32_t l = 0; l < s->n_lights; ++l) {
        float density[LANES];
        bool any = false, per_lane = false;
        for (uint32_t k = 0; k < n; ++k) {
            bool hit;
            density[k] = light_probe(ctx, s->lights[l], hits[k].p, out[k].dir, hit, per_lane);
            any = any || hit;
        }
        for (uint32_t k = 0; k < n; ++k) light_sum[k] += weight * ((per_lane || any) ? density[k] : 0.0f);
    }
    for (uint32_t k = 0; k < n; ++k) {
        float mix_v = 0.5f * light_sum[k] + 0.5f * out[k].pdf_v;
        out[k].coef = out[k].coef * out[k].pdf_v / mix_v;
    }
}

template <class F>
void parallel_for(uint32_t n, int n_threads, F body) {  // body(begin, end, thread)
    std::atomic<uint32_t> next{0};
    const uint32_t chunk = 1024;
    auto worker = [&](int tid) {
        while (true) {
            uint32_t begin = next.fetch_add(chunk);
            if (begin >= n) break;
            body(begin, std::min(n, begin + chunk), tid);
        }
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto &t : threads) t.join();
}

// rule 2: the reference's packets; rule 3: the packet-statistics rule (per-ray shading, per-shape ray counts of the level)
// trace_mode: 0 = the reference's candidate-set traversal, 1 = the device's ordered traversal of `wide`, 4 = of the device's own tree (same hits).
void render_pass_packets(const v4r_scene_desc *s, uint64_t seed, uint32_t pass, float *accum, int n_threads, std::vector<Counters> &counters,
                         int rule, int trace_mode, const WideTree *wide) {
    const uint32_t npix = s->width * s->height;
    std::vector<uint32_t> shape_rays;  // rule 3: rays of this level per closest-hit shape
    if (rule == 3) shape_rays.assign(s->n_prims, 0u);
    std::vector<LevelRay> cur(npix), next;
    parallel_for(npix, n_threads, [&](uint32_t b, uint32_t e, int) {
        for (uint32_t pix = b; pix < e; ++pix) {
            PathKey pk = path_key(seed, pass, pix);
            cur[pix] = LevelRay{camera_ray(s, pix / s->width, pix % s->width, segment_key(pk, 0), false), v3(1, 1, 1), pix, pk};
        }
    });
    auto add = [&](uint32_t pixel, V3 c) {
        accum[3 * (size_t)pixel] += c.x;
        accum[3 * (size_t)pixel + 1] += c.y;
        accum[3 * (size_t)pixel + 2] += c.z;
    };
    counters[0].paths += npix;
    std::vector<Closest> found;
    struct Order {
        uint32_t shape, first, ray;
    };
    std::vector<Order> order;
    std::vector<uint32_t> packet_start;
    std::vector<Scatter> scattered;
    for (uint32_t seg = 0; seg < s->max_depth && !cur.empty(); ++seg) {
        const uint32_t n = (uint32_t)cur.size();
        found.assign(n, Closest{});
        parallel_for(n, n_threads, [&](uint32_t b, uint32_t e, int tid) {
            Ctx ctx{s};
            ctx.wide = wide;
            for (uint32_t i = b; i < e; ++i) {
                ctx.seg_key = segment_key(cur[i].pk, seg);
                found[i] = trace(ctx, cur[i].ray, trace_mode);
            }
            counters[(size_t)tid].add(ctx.cnt);
        });
        order.clear();
        for (uint32_t i = 0; i < n; ++i) {
            if (!found[i].hit.mask) add(cur[i].pixel, cur[i].thr * v3(s->background));  // renderer.rs:184
            else order.push_back({found[i].prim, found[i].first, i});
        }
        if (rule == 3) {
            if (seg < PacketRule::COUNTED_LEVELS) {  // deeper levels reuse the counts of the last counted one
                // a shape that MANY_RAYS rays of the pass ended on at some level counts as full from that level on
                for (uint32_t k = 0; k < s->n_prims; ++k) shape_rays[k] = shape_rays[k] >= PacketRule::MANY_RAYS ? PacketRule::MANY_RAYS : 0u;
                for (const Order &o : order)
                    if (shape_rays[o.shape] < PacketRule::MANY_RAYS) shape_rays[o.shape]++;
            }
            scattered.assign(order.size(), Scatter{});
            parallel_for((uint32_t)order.size(), n_threads, [&](uint32_t b, uint32_t e, int tid) {
                Ctx ctx{s};
                for (uint32_t q = b; q < e; ++q) {
                    const LevelRay &r = cur[order[q].ray];
                    ctx.seg_key = segment_key(r.pk, seg);
                    uint32_t n = shape_rays[order[q].shape];
                    ctx.packet_n = n < 1u ? 1u : n;
                    scattered[q] = shade(ctx, order[q].shape, r.ray, found[order[q].ray].hit);
                }
                counters[(size_t)tid].add(ctx.cnt);
            });
        }
        if (rule == 2) std::sort(order.begin(), order.end(), [](const Order &a, const Order &b) {
            if (a.shape != b.shape) return a.shape < b.shape;
            if (a.first != b.first) return a.first < b.first;
            return a.ray < b.ray;
        });
        packet_start.clear();
        for (size_t k = 0; rule == 2 && k < order.size();) {
            size_t end = k;
            while (end < order.size() && order[end].shape == order[k].shape) ++end;
            for (size_t q = k; q < end; q += LANES) packet_start.push_back((uint32_t)q);
            k = end;
        }
        if (rule == 2) scattered.assign(order.size(), Scatter{});
        parallel_for((uint32_t)packet_start.size(), n_threads, [&](uint32_t b, uint32_t e, int tid) {
            Ctx ctx{s};
            for (uint32_t p = b; p < e; ++p) {
                const uint32_t q = packet_start[p];
                uint32_t lanes = 0;
                const LevelRay *rays[LANES];
                Hit hits[LANES];
                while (lanes < LANES && q + lanes < order.size() && order[q + lanes].shape == order[q].shape) {
                    rays[lanes] = &cur[order[q + lanes].ray];
                    hits[lanes] = found[order[q + lanes].ray].hit;
                    ++lanes;
                }
                shade_packet(ctx, order[q].shape, seg, rays, hits, lanes, &scattered[q]);
            }
            counters[(size_t)tid].add(ctx.cnt);
        });
        next.clear();
        for (size_t k = 0; k < order.size(); ++k) {
            const LevelRay &r = cur[order[k].ray];
            const Scatter &sc = scattered[k];
            if (!sc.scattered) {  // renderer.rs:218-230,261
                add(r.pixel, r.thr * sc.emitted);
                continue;
            }
            V3 thr = r.thr * sc.coef;  // renderer.rs:299-318
            if (seg + 1 >= s->max_depth) add(r.pixel, thr * v3(s->environment));  // renderer.rs:128-130
            else next.push_back(LevelRay{Ray{found[order[k].ray].hit.p, sc.dir}, thr, r.pixel, r.pk});
        }
        cur.swap(next);
    }
}

int validate(const v4r_scene_desc *s) {
    if (!s) {
        g_error = "NULL scene";
        return V4R_E_INVALID;
    }
    if (s->abi_version != V4R_ABI_VERSION) {
        g_error = "ABI version mismatch";
        return V4R_E_INVALID;
    }
    if (s->width == 0 || s->height == 0) {
        g_error = "empty image";
        return V4R_E_INVALID;
    }
    return V4R_OK;
}

}  // namespace

extern "C" {

const char *v4o_last_error(void) { return g_error.c_str(); }

// Adds n_passes full-image passes into accum (H*W*3 floats, row 0 = top).  mode 0 = the
// reference's candidate-set traversal; 1 = the device's ordered traversal; both with the light pdf the scene's
// flags select, like the device (no flag: the packet-statistics rule, lights_value_stat);
// 2 = the reference's breadth-first pass with its 8-lane packets (render_pass_packets); 3 = the
// packet-statistics rule whatever the flags; 4 = as 1, but walking the tree set by v4o_set_device_tree.
// pixel_stride > 1 renders only pixels whose index is a multiple of it (for strided work counts;
// modes 0 and 1).
int v4o_render(const v4r_scene_desc *s, uint64_t seed, uint32_t first_pass, uint32_t pass_stride, uint32_t n_passes,
               float *accum, int n_threads, uint32_t pixel_stride, int mode, v4r_stats *stats) {
    if (int rc = validate(s)) return rc;
    if (!accum) {
        g_error = "NULL accumulator";
        return V4R_E_INVALID;
    }
    if (pixel_stride == 0) pixel_stride = 1;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    uint32_t npix = s->width * s->height;
    // main.rs:56-63 runs whole passes concurrently; here the image is split into chunks handed out
    // dynamically, and each pixel's passes are summed in ascending pass order (deterministic).
    std::atomic<uint32_t> next{0};
    const uint32_t chunk = 256;
    std::vector<Counters> counters((size_t)n_threads);
    auto report = [&]() {
        if (!stats) return;
        Counters total;
        for (auto &c : counters) total.add(c);
        stats->segments = total.segments;
        stats->nodes = total.nodes;
        stats->prims = total.prims;
        stats->paths = total.paths;
        stats->shaded = total.shaded;
        stats->texel_fetches = total.texels;
    };
    WideTree wide;
    if (mode == 1) wide = build_wide(s);
    // Scenes with lights and neither light-pdf flag follow the packet-statistics rule, which needs the per-shape ray
    // counts of whole passes: modes 0 and 1 then run the breadth-first pass too (a strided render, which only serves
    // work counts, keeps the per-path loop and the plane density: the paths are the same, only their weights differ).
    const bool packet_rule = s->n_lights != 0 && !(s->flags & (V4R_FLAG_BOUNDED_LIGHT_PDF | V4R_FLAG_PLANE_LIGHT_PDF));
    if (mode == 2 || mode == 3 || (packet_rule && pixel_stride == 1)) {  // whole passes, breadth first
        const int rule = mode == 2 ? 2 : 3, trace_mode = mode == 1 ? 1 : (mode == 4 ? 4 : 0);
        // Whole passes run concurrently, as the reference runs them (main.rs:56-63: one task per pass): `outer` passes at a
        // time, each on `inner` threads and into its own buffer; a pass adds exactly one term per pixel, so adding the
        // buffers in pass order gives the bits of a sequential loop.  `outer` is bounded by the passes asked for and by
        // the memory the per-pass ray lists take (about 250 bytes per pixel).
        const size_t budget = (size_t)24 << 30, per_pass = (size_t)npix * 250 + 1;
        const uint32_t most = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)n_threads, n_passes), budget / per_pass));
        const int inner = std::max(1, (n_threads + (int)most - 1) / (int)most);           // threads per pass
        const uint32_t outer = std::min<uint32_t>(most, (uint32_t)((n_threads + inner - 1) / inner));  // passes at a time
        if (outer > 1) {
            std::vector<std::vector<float>> buf(outer, std::vector<float>((size_t)npix * 3));
            std::vector<std::vector<Counters>> own(outer, std::vector<Counters>((size_t)inner));
            for (uint32_t k0 = 0; k0 < n_passes; k0 += outer) {
                const uint32_t group = std::min<uint32_t>(outer, n_passes - k0);
                std::vector<std::thread> threads;
                auto one = [&](uint32_t t) {
                    std::fill(buf[t].begin(), buf[t].end(), 0.0f);
                    render_pass_packets(s, seed, first_pass + (k0 + t) * pass_stride, buf[t].data(), inner, own[t], rule, trace_mode, &wide);
                };
                for (uint32_t t = 1; t < group; ++t) threads.emplace_back(one, t);
                one(0);
                for (auto &t : threads) t.join();
                for (uint32_t t = 0; t < group; ++t)
                    for (size_t q = 0; q < (size_t)npix * 3; ++q) accum[q] += buf[t][q];
            }
            for (uint32_t t = 0; t < outer; ++t)
                for (int k = 0; k < inner; ++k) counters[0].add(own[t][(size_t)k]);
        } else {
            for (uint32_t k = 0; k < n_passes; ++k)
                render_pass_packets(s, seed, first_pass + k * pass_stride, accum, n_threads, counters, rule, trace_mode, &wide);
        }
        report();
        return V4R_OK;
    }
    auto worker = [&](int tid) {
        Ctx ctx{s};
        ctx.wide = &wide;
        while (true) {
            uint32_t begin = next.fetch_add(chunk);
            if (begin >= npix) break;
            uint32_t end = begin + chunk < npix ? begin + chunk : npix;
            for (uint32_t pix = begin; pix < end; ++pix) {
                if (pix % pixel_stride) continue;
                for (uint32_t k = 0; k < n_passes; ++k) {
                    V3 c = path_radiance(ctx, seed, first_pass + k * pass_stride, pix, mode);
                    accum[3 * (size_t)pix] += c.x;
                    accum[3 * (size_t)pix + 1] += c.y;
                    accum[3 * (size_t)pix + 2] += c.z;
                }
            }
        }
        counters[(size_t)tid] = ctx.cnt;
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto &t : threads) t.join();
    report();
    return V4R_OK;
}

// The tree of a device renderer (v4r_tree_read) for trace mode 4; NULL info clears it.  Not thread safe against running renders.
int v4o_set_device_tree(const v4r_tree_info *info, const void *nodes, const uint32_t *scene_index, const uint32_t *media_direct) {
    g_device_tree = DeviceTree{};
    if (!info) return V4R_OK;
    if ((info->width != 4 && info->width != 8) || (info->n_nodes && !nodes) || (info->n_prims && !scene_index) || (info->n_media_direct && !media_direct)) {
        g_error = "malformed device tree";
        return V4R_E_INVALID;
    }
    g_device_tree.info = *info;
    g_device_tree.nodes.assign((const uint32_t *)nodes, (const uint32_t *)nodes + 32 * (size_t)info->n_nodes);
    g_device_tree.scene_index.assign(scene_index, scene_index + info->n_prims);
    if (info->n_media_direct) g_device_tree.media_direct.assign(media_direct, media_direct + info->n_media_direct);
    return V4R_OK;
}

// Deterministic primary rays (pixel centres, no lens sample): closest primitive and t per pixel.  Rows are handed
// out to every hardware thread (each ray is independent).
int v4o_trace_primary(const v4r_scene_desc *s, uint32_t *prim_id, float *t, int mode) {
    if (int rc = validate(s)) return rc;
    WideTree wide;
    if (mode == 1) wide = build_wide(s);
    const uint32_t npix = s->width * s->height;
    int n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    if (npix < 65536u) n_threads = 1;
    parallel_for(npix, n_threads, [&](uint32_t b, uint32_t e, int) {
        Ctx ctx{s};
        ctx.wide = &wide;
        for (uint32_t pix = b; pix < e; ++pix) {
            Ray r = camera_ray(s, pix / s->width, pix % s->width, 0, true);
            ctx.seg_key = 0;
            Closest c = trace(ctx, r, mode);
            if (prim_id) prim_id[pix] = c.prim;
            if (t) t[pix] = c.hit.t;
        }
    });
    return V4R_OK;
}

// Camera rays exactly as the device generates them for pass `pass` (origin, direction per pixel).
int v4o_camera_rays(const v4r_scene_desc *s, uint64_t seed, uint32_t pass, int deterministic, float *origin, float *dir) {
    if (int rc = validate(s)) return rc;
    uint32_t npix = s->width * s->height;
    for (uint32_t pix = 0; pix < npix; ++pix) {
        uint32_t key = segment_key(path_key(seed, pass, pix), 0);
        Ray r = camera_ray(s, pix / s->width, pix % s->width, key, deterministic != 0);
        if (origin) {
            origin[3 * pix] = r.o.x;
            origin[3 * pix + 1] = r.o.y;
            origin[3 * pix + 2] = r.o.z;
        }
        if (dir) {
            dir[3 * pix] = r.d.x;
            dir[3 * pix + 1] = r.d.y;
            dir[3 * pix + 2] = r.d.z;
        }
    }
    return V4R_OK;
}

// Closest hit of caller-supplied rays. rec (optional) receives 10 floats per ray:
// p.xyz, normal.xyz, u, v, front_face, hit mask.  Ray i uses segment key mix32(seed_lo + i) for media.
int v4o_trace_rays(const v4r_scene_desc *s, uint32_t n, const float *origin, const float *dir, uint64_t seed,
                   uint32_t *prim_id, float *t, float *rec, int mode) {
    if (int rc = validate(s)) return rc;
    Ctx ctx{s};
    WideTree wide;
    if (mode == 1) wide = build_wide(s);
    ctx.wide = &wide;
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{v3(origin + 3 * i), v3(dir + 3 * i)};
        ctx.seg_key = segment_key(path_key(seed, 0, i), 0);
        Closest c = trace(ctx, r, mode);
        if (prim_id) prim_id[i] = c.prim;
        if (t) t[i] = c.hit.t;
        if (rec) {
            float *o = rec + 10 * (size_t)i;
            o[0] = c.hit.p.x; o[1] = c.hit.p.y; o[2] = c.hit.p.z;
            o[3] = c.hit.normal.x; o[4] = c.hit.normal.y; o[5] = c.hit.normal.z;
            o[6] = c.hit.u; o[7] = c.hit.v;
            o[8] = c.hit.front_face ? 1.0f : 0.0f;
            o[9] = c.hit.mask ? 1.0f : 0.0f;
        }
    }
    return V4R_OK;
}

// Hittable::hit of primitive `prim_index` alone (no BVH) with an explicit interval.
int v4o_hit(const v4r_scene_desc *s, uint32_t prim_index, const float origin[3], const float dir[3], float t_min,
            float t_max, uint32_t seg_key, float rec[11]) {
    if (int rc = validate(s)) return rc;
    if (prim_index >= s->n_prims) {
        g_error = "primitive index out of range";
        return V4R_E_INVALID;
    }
    Ctx ctx{s};
    ctx.seg_key = seg_key;
    Hit h = hit_prim(ctx, s->prims[prim_index], prim_index, Ray{v3(origin), v3(dir)}, t_min, t_max);
    rec[0] = h.p.x; rec[1] = h.p.y; rec[2] = h.p.z;
    rec[3] = h.normal.x; rec[4] = h.normal.y; rec[5] = h.normal.z;
    rec[6] = h.t; rec[7] = h.u; rec[8] = h.v;
    rec[9] = h.front_face ? 1.0f : 0.0f;
    rec[10] = h.mask ? 1.0f : 0.0f;
    return V4R_OK;
}

int v4o_slab(const float lo[3], const float hi[3], const float origin[3], const float dir[3], float t_min, float t_max) {
    float e;
    return slab(lo, hi, Ray{v3(origin), v3(dir)}, t_min, t_max, e) ? 1 : 0;
}

int v4o_texture_value(const v4r_scene_desc *s, uint32_t tex, float u, float v, const float p[3], float rgb[3]) {
    if (int rc = validate(s)) return rc;
    if (tex >= s->n_textures) {
        g_error = "texture index out of range";
        return V4R_E_INVALID;
    }
    Ctx ctx{s};
    V3 c = texture_value(ctx, tex, u, v, v3(p));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return V4R_OK;
}

// Shade one hit: rec = the 11 floats v4o_hit returns.  out: scattered flag, coef.rgb, dir.xyz, emitted.rgb
int v4o_shade(const v4r_scene_desc *s, uint32_t prim_index, const float origin[3], const float dir[3], const float rec[11],
              uint32_t seg_key, float out[10]) {
    if (int rc = validate(s)) return rc;
    if (prim_index >= s->n_prims) {
        g_error = "primitive index out of range";
        return V4R_E_INVALID;
    }
    Ctx ctx{s};
    ctx.seg_key = seg_key;
    Hit h;
    h.p = v3(rec); h.normal = v3(rec + 3); h.t = rec[6]; h.u = rec[7]; h.v = rec[8];
    h.front_face = rec[9] != 0.0f; h.mask = rec[10] != 0.0f;
    Scatter sc = shade(ctx, prim_index, Ray{v3(origin), v3(dir)}, h);
    out[0] = sc.scattered ? 1.0f : 0.0f;
    out[1] = sc.coef.x; out[2] = sc.coef.y; out[3] = sc.coef.z;
    out[4] = sc.dir.x; out[5] = sc.dir.y; out[6] = sc.dir.z;
    out[7] = sc.emitted.x; out[8] = sc.emitted.y; out[9] = sc.emitted.z;
    return V4R_OK;
}

float v4o_cosine_value(const float normal[3], const float dir[3]) { return cosine_value(v3(normal), v3(dir)); }
void v4o_cosine_generate(const float normal[3], float r1, float r2, float out[3]) {
    V3 d = cosine_generate(v3(normal), r1, r2);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void v4o_onb(const float normal[3], float out9[9]) {  // columns x, y, z of the CosinePdf rotation
    Rot r = face_towards(v3(normal), pick_up(normal[0]));
    out9[0] = r.x.x; out9[1] = r.x.y; out9[2] = r.x.z;
    out9[3] = r.y.x; out9[4] = r.y.y; out9[5] = r.y.z;
    out9[6] = r.z.x; out9[7] = r.z.y; out9[8] = r.z.z;
}
int v4o_light_value(const v4r_scene_desc *s, uint32_t prim_index, const float origin[3], const float dir[3], float *out) {
    if (int rc = validate(s)) return rc;
    Ctx ctx{s};
    *out = light_value(ctx, prim_index, v3(origin), v3(dir));
    return V4R_OK;
}
// HittableGroup::value over the scene's lights under the packet-statistics rule, for a ray whose shape `n` rays of its pass
// and level ended on, `li` = the light index it drew, `seg_key` = the key of its segment's random stream.
int v4o_lights_value_stat(const v4r_scene_desc *s, const float origin[3], const float normal[3], const float dir[3], uint32_t li, uint32_t n,
                          uint32_t seg_key, float *out) {
    if (int rc = validate(s)) return rc;
    if (s->n_lights == 0 || li >= s->n_lights) {
        g_error = "no such light";
        return V4R_E_INVALID;
    }
    Ctx ctx{s};
    *out = lights_value_stat(ctx, v3(origin), v3(normal), v3(dir), li, n, seg_key);
    return V4R_OK;
}
// The share of cosine-distributed directions from surface point p (normal n) that the rule takes to hit light primitive
// `prim_index` (light_cosine_share).
int v4o_light_cosine_share(const v4r_scene_desc *s, uint32_t prim_index, const float p[3], const float n[3], float *out) {
    if (int rc = validate(s)) return rc;
    if (prim_index >= s->n_prims) {
        g_error = "primitive index out of range";
        return V4R_E_INVALID;
    }
    *out = light_cosine_share(s->prims[prim_index], v3(p), v3(n));
    return V4R_OK;
}
int v4o_light_generate(const v4r_scene_desc *s, uint32_t prim_index, const float origin[3], float u0, float u1, float out[3]) {
    if (int rc = validate(s)) return rc;
    Ctx ctx{s};
    V3 d = light_generate(ctx, prim_index, v3(origin), u0, u1);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
    return V4R_OK;
}
void v4o_reflect(const float v[3], const float n[3], float out[3]) {
    V3 r = reflect(v3(v), v3(n));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void v4o_refract(const float v[3], const float n[3], float ratio, float out[3]) {
    V3 r = refract(v3(v), v3(n), ratio);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float v4o_reflectance(float cosine, float ref_idx) { return reflectance(cosine, ref_idx); }

// the random-number definition, exposed so tests can pin device and oracle to the same streams
uint32_t v4o_rng_segment_key(uint64_t seed, uint32_t pass, uint32_t pixel, uint32_t segment) {
    return segment_key(path_key(seed, pass, pixel), segment);
}
float v4o_rng_draw(uint32_t seg_key, uint32_t dim) { return draw(seg_key, dim); }

}  // extern "C"
