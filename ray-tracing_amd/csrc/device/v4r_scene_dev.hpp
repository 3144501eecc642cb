// Device-side scene view and the per-ray device functions: slab test, primitive intersection,
// BVH traversal, hit reconstruction, textures, pdfs and materials.  Each function cites the
// reference lines (under the reference tree's src/) whose arithmetic it implements.
#pragma once
#include "v4r_math.hpp"
#include "v4ray_hip.h"

namespace v4r {

// Shading classes: rays are binned by the class of the material they hit so that each shading
// kernel runs one material's code on full waves.
enum {
    CLS_LIGHT = 0,          // DiffuseLight: emits, never scatters
    CLS_LAMBERT = 1,        // Lambertian with solid / checker / image textures
    CLS_LAMBERT_NOISE = 2,  // Lambertian whose texture tree contains Perlin noise (ALU heavy)
    CLS_METAL = 3,
    CLS_DIELECTRIC = 4,
    CLS_ISOTROPIC = 5,
    CLS_COUNT = 6
};

constexpr uint32_t NO_PRIM = 0xFFFFFFFFu;
// A leaf of the traversed tree is a block of 1 to LEAF_MAX_PRIMS primitives that are consecutive in the device's numbering:
// child word = V4R_LEAF_BIT | (count - 1) << 28 | first.  (0xFFFFFFFF stays "no entry": the last primitive index is never used.)
constexpr uint32_t LEAF_FIRST_MASK = (1u << 28) - 1u, LEAF_MAX_PRIMS = 8u;
constexpr uint32_t leaf_word(uint32_t first, uint32_t count) { return V4R_LEAF_BIT | (count - 1u) << 28 | first; }
constexpr uint32_t leaf_first(uint32_t word) { return word & LEAF_FIRST_MASK; }
constexpr uint32_t leaf_count(uint32_t word) { return (word >> 28 & 7u) + 1u; }
// Primitives per leaf that the kernels are built for.  1: as in the reference's tree (bvh.rs:10-12).  Larger blocks were
// measured and lose: an 8-wide node over eight one-primitive leaves culls most of them for the price of 1.6 primitive
// tests, where a block tests every one of its primitives (builds with -DV4R_LEAF_PRIMS=2|4|8 keep the experiment
// reproducible; V4R_LEAF_PRIMS in the environment then selects any size up to the built one).
#ifndef V4R_LEAF_PRIMS
#define V4R_LEAF_PRIMS 1
#endif
static_assert(V4R_LEAF_PRIMS >= 1 && V4R_LEAF_PRIMS <= 8, "a leaf word counts 1 to 8 primitives");
// prim_info word: material index (18 bits) | flags | shading class << 24
constexpr uint32_t PRIM_INFO_ALWAYS_FULL = 1u << 18;  // a Lambertian shape that so many camera rays end on that it is "full" (packet rule) in every pass
constexpr uint32_t SHADE_ALWAYS_FULL = 1u;            // the same, in the padding word of the shape's prim_shade record
constexpr uint32_t PRIM_INFO_SMALL = 1u << 19;        // the shape is too small to expect more than a few rays of a pass (k_trace's ray counters)
constexpr uint32_t PRIM_INFO_PLAIN_LIGHT = 1u << 23;  // one-colour DiffuseLight on an untransformed rectangle: k_trace ends the path itself
constexpr uint32_t PRIM_INFO_POSITIVE = 1u << 22;     // ... whose outward normal points along +axis
constexpr uint32_t PRIM_INFO_AXIS_SHIFT = 20;         // ... axis of that normal (2 bits)
constexpr size_t MAX_DIRECT_MEDIA = 4;  // each costs every ray about a sixth of an average traversal

struct SceneDev {
    const float4 *nodes4;        // 8 x float4 per 4-wide node (wide_step)
    const float4 *prims;         // 3 x float4 per primitive (v4r_prim)
    const uint4 *prim_shade;     // 2 x uint4 per primitive: its material's root texture record (textured materials) or material record
    const uint32_t *prim_info;   // material index | PRIM_INFO_* | shading class << 24
    const float *xforms;         // 24 floats per transform (v4r_xform)
    const float4 *tri;           // 4 x float4 per triangle (v4r_tri_shading)
    const uint4 *media;          // v4r_medium
    const float4 *medium_prims;  // 3 x float4 per boundary primitive
    const uint32_t *materials;   // 8 words per material (v4r_material)
    const uint32_t *textures;    // 8 words per texture (v4r_texture)
    const uint4 *images;         // {first texel, width, height, 0}
    const uint32_t *texels;      // RGBA8, one word per texel
    const float4 *perlin_vec;    // 256 x float4 per table
    const uint16_t *perlin_perm; // 768 entries per table: perm_x, perm_y, perm_z; entry i = perm[i] | perm[(i + 1) & 255] << 8
    const uint32_t *lights;      // primitive indices
    const uint32_t *scene_index; // per primitive: its index in the scene (the device numbers primitives in the order of its tree's leaves)
    uint32_t n_nodes, n_prims, n_lights, flags;
    const uint32_t *media_direct;  // primitives (constant media) tested before the tree instead of inside it
    uint32_t n_media_direct;
    uint32_t n_top_nodes;        // the first n_top_nodes wide nodes are the top levels of the tree (breadth first): k_trace keeps them in LDS
    uint32_t packet_rule;        // the light pdf follows the packet-statistics rule (lights_value): the scene has lights and neither light-pdf flag
    float bg[3], env[3];
    v4r_camera cam;
    uint32_t width, height, max_depth, antialias;
};

struct RayD {
    V3 o, d;
};

// Full hit record (HitRecord, hittable/mod.rs:24-32) as shading needs it.
struct HitD {
    V3 p, normal;
    float u, v;
    bool front;
};

struct Prim {
    float4 a, b, c;
    __device__ __forceinline__ uint32_t type() const { return __float_as_uint(c.y); }
    __device__ __forceinline__ uint32_t xform() const { return __float_as_uint(c.z); }
    __device__ __forceinline__ uint32_t aux() const { return __float_as_uint(c.w); }
};
V4R_DEV Prim load_prim(const float4 *prims, uint32_t i) {
    const float4 *p = prims + 3 * (size_t)i;
    return Prim{p[0], p[1], p[2]};
}

// ---- primitive intersection (distance only) -----------------------------------------------------
// Sphere::hit, hittable/sphere.rs:60-79
V4R_DEV bool sphere_t(const Prim &pr, V3 o, V3 d, float t_min, float t_max, float &t) {
    V3 oc = o - v3(pr.a.x, pr.a.y, pr.a.z);
    float half_b = dot(oc, d);
    float c = norm2(oc) - pr.a.w * pr.a.w;
    float disc = half_b * half_b - c;
    if (disc < 0.0f) return false;
    float sqrt_d = sqrtf(disc);
    float root1 = -half_b - sqrt_d;
    bool m1 = root1 >= t_min && root1 <= t_max;
    float root2 = -half_b + sqrt_d;
    bool m2 = root2 >= t_min && root2 <= t_max;
    t = m1 ? root1 : root2;
    return m1 || m2;
}

V4R_DEV void rect_axes(uint32_t type, int &ia, int &ib, int &ik) {
    uint32_t k = type & V4R_PRIM_TYPE_MASK;
    ia = k == V4R_PRIM_XY_RECT ? 0 : (k == V4R_PRIM_YZ_RECT ? 1 : 2);
    ib = k == V4R_PRIM_XY_RECT ? 1 : (k == V4R_PRIM_YZ_RECT ? 2 : 0);
    ik = k == V4R_PRIM_XY_RECT ? 2 : (k == V4R_PRIM_YZ_RECT ? 0 : 1);
}

// $ty::hit, hittable/aa_rect.rs:114-129
V4R_DEV bool rect_t(const Prim &pr, V3 o, V3 d, float t_min, float t_max, float &t) {
    int ia, ib, ik;
    rect_axes(pr.type(), ia, ib, ik);
    t = (pr.b.x - comp(o, ik)) / comp(d, ik);
    float a = comp(o, ia) + t * comp(d, ia);
    float b = comp(o, ib) + t * comp(d, ib);
    return t >= t_min && t <= t_max && a >= pr.a.x && a <= pr.a.y && b >= pr.a.z && b <= pr.a.w;
}

// Triangle::hit, hittable/triangle.rs:56-80
V4R_DEV bool tri_t(const Prim &pr, V3 o, V3 d, float t_min, float t_max, float &t, float &u, float &v, float &det) {
    V3 v0 = v3(pr.a.x, pr.a.y, pr.a.z), e12 = v3(pr.a.w, pr.b.x, pr.b.y), e13 = v3(pr.b.z, pr.b.w, pr.c.x);
    V3 p_vec = cross(d, e13);
    det = dot(e12, p_vec);
    float inv_det = 1.0f / det;
    V3 t_vec = o - v0;
    u = inv_det * dot(t_vec, p_vec);
    V3 q_vec = cross(t_vec, e12);
    v = inv_det * dot(d, q_vec);
    t = inv_det * dot(e13, q_vec);
    return fabsf(det) > 0.0f && !(u < 0.0f) && u <= 1.0f && !(v < 0.0f) && (u + v) <= 1.0f && t >= t_min && t <= t_max;
}

// TransformHittable::hit prologue, hittable/transform.rs:69-76
V4R_DEV void to_object(const float *x, V3 &o, V3 &d, float &nrm) {
    // the inverse map's twelve floats as three 16-byte loads (a transform record is 96 bytes in a 256-byte-aligned table)
    const float4 *q = reinterpret_cast<const float4 *>(x + 12);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2];
    const float inv[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
    V3 dir = mat_mul(inv, d);
    nrm = norm(dir);
    o = mat_mul(inv, o) + v3(inv[9], inv[10], inv[11]);
    d = dir / nrm;
}

struct TraceCounters {
    uint32_t nodes = 0, prims = 0;
    uint32_t wave_steps = 0, wave_leaves = 0;  // counted by every lane of the wave, whether it takes part or not
#ifdef V4R_PHASE_CLOCKS
    // A build for measurements only (-DV4R_PHASE_CLOCKS, with the work counters on): shader clock cycles a wave spends in
    // each part of k_trace, every part closed by a wait for its outstanding memory operations.
    unsigned long long clk_node = 0, clk_leaf = 0;
#endif
};
#ifdef V4R_PHASE_CLOCKS
V4R_DEV unsigned long long phase_clock() {
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the part that ends here has waited for its loads
    return __builtin_amdgcn_s_memtime();
}
#endif

// Closest hit so far during traversal.
struct Best {
    float t;
    uint32_t prim;
    float u, v;  // triangle barycentrics
};

// Distance to a sphere / rectangle / triangle record, through its own affine transform if it has one
// (TransformHittable::hit, transform.rs:64-95).  t is in the caller's space.
V4R_DEV bool solid_t(const SceneDev &S, const Prim &pr, V3 o, V3 d, float t_min, float t_max, float &t, float &u, float &v) {
    float nrm = 1.0f;
    uint32_t xf = pr.xform();
    if (xf != V4R_NO_XFORM) {
        to_object(S.xforms + 24 * (size_t)xf, o, d, nrm);
        t_min = t_min * nrm;
        t_max = t_max * nrm;
    }
    float det;
    bool hit;
    uint32_t kind = pr.type() & V4R_PRIM_TYPE_MASK;
    t = 0.0f;
    if (kind == V4R_PRIM_SPHERE) hit = sphere_t(pr, o, d, t_min, t_max, t);
    else if (kind == V4R_PRIM_TRIANGLE) hit = tri_t(pr, o, d, t_min, t_max, t, u, v, det);
    else if (kind <= V4R_PRIM_ZX_RECT) hit = rect_t(pr, o, d, t_min, t_max, t);
    else hit = false;
    if (xf != V4R_NO_XFORM) t = t / nrm;  // transform.rs:90
    return hit;
}

// HittableGroup::hit over the boundary of a medium (hittables/group.rs:58-67): the nearest boundary
// hit inside [t_min, t_max].
V4R_DEV bool boundary_t(const SceneDev &S, uint32_t first, uint32_t count, V3 o, V3 d, float t_min, float t_max, float &t) {
    bool any = false;
    float closest = t_max;
    for (uint32_t i = 0; i < count; ++i) {
        Prim bp = load_prim(S.medium_prims, first + i);
        float ti, u, v;
        if (solid_t(S, bp, o, d, t_min, closest, ti, u, v)) {
            closest = ti;
            any = true;
        }
    }
    t = closest;
    return any;
}

// ConstantMedium::hit, hittable/constant_medium.rs:40-85: entry and exit of the boundary along the
// whole line, an exponential free path inside; `key()` yields the segment key of the ray, asked for
// only when a medium is actually tested.
template <class KeyFn>
V4R_DEV bool medium_t(const SceneDev &S, const Prim &pr, uint32_t prim_index, V3 o, V3 d, float t_min, float t_max, float &t, KeyFn key) {
    uint4 m = S.media[pr.aux()];  // first, count, -1/density
    float t1, t2;
    if (!boundary_t(S, m.x, m.y, o, d, -V4R_INF, V4R_INF, t1)) return false;
    if (!boundary_t(S, m.x, m.y, o, d, t1 + EPSILON, V4R_INF, t2)) return false;
    float lo = fmaxf(t1, t_min), hi = fminf(t2, t_max);
    if (!(lo < hi)) return false;
    lo = fmaxf(lo, 0.0f);
    float distance_inside = hi - lo;
    float hit_distance = __uint_as_float(m.z) * logf(draw(key(), DIM_MEDIUM + (prim_index & 31u)));
    if (!(hit_distance <= distance_inside)) return false;
    t = t1 + hit_distance;
    return true;
}

// One primitive against the ray over [EPSILON, +inf) (renderer.rs:167) with the closest-hit rule
// of renderer.rs:174 (strictly smaller t wins; equal t keeps the lowest index).
// MEDIA = false compiles the constant-medium branch out (scenes without media): the branch holds a
// second ray in object space and a loop over boundary records, which costs ~40 registers.
template <bool MEDIA, class KeyFn>
V4R_DEV void consider(const SceneDev &S, uint32_t prim_index, V3 o, V3 d, Best &best, KeyFn key) {
    Prim pr = load_prim(S.prims, prim_index);
    float t, u = 0.0f, v = 0.0f;
    bool hit;
    if (MEDIA && (pr.type() & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM) {
        float nrm = 1.0f;
        uint32_t xf = pr.xform();
        if (xf != V4R_NO_XFORM) to_object(S.xforms + 24 * (size_t)xf, o, d, nrm);
        // (the free path is drawn from the dimension of the medium's index in the scene)
        hit = medium_t(S, pr, S.scene_index[prim_index], o, d, EPSILON * nrm, V4R_INF * nrm, t, key);
        if (xf != V4R_NO_XFORM) t = t / nrm;
    } else {
        hit = solid_t(S, pr, o, d, EPSILON, V4R_INF, t, u, v);
    }
    // equal distances: the shape of lower index in the scene wins, whatever the device's numbering (looked up only then)
    if (hit && (t < best.t || (t == best.t && best.prim != NO_PRIM && S.scene_index[prim_index] < S.scene_index[best.prim]))) {
        best.t = t;
        best.prim = prim_index;
        best.u = u;
        best.v = v;
    }
}

// Bounds checks of a checked build (-DV4R_DEBUG_CHECKS, lib/libv4ray_hip_checked.so): an index that leaves its array is
// not followed; its check's bit is raised in a device word that the host reads after the job and turns into an error.
// `V4R_CHECK(errors, code, condition)` is true when the access may go ahead; without the flag it is the constant true.
enum : int { CHECK_STACK_OVERFLOW = 0, CHECK_STACK_UNDERFLOW = 1, CHECK_QUEUE_INDEX = 2, CHECK_QUEUE_LENGTH = 3, CHECK_RAY_COUNTER = 4,
             CHECK_PATH_SLOT = 5, CHECK_NEXT_RAY = 6, CHECK_POOL_WAIT = 7 };
#ifdef V4R_DEBUG_CHECKS
#define V4R_CHECK(errors, code, condition) ((condition) ? true : (atomicOr((errors), 1ull << (code)), false))
#else
#define V4R_CHECK(errors, code, condition) true
#endif

// Stack of the traversal: the first entries of each lane live in LDS (conflict-free: entry e of thread t sits at
// [e][t]); deeper entries spill to a per-thread slab in HBM.  The LDS depth goes with the tree the kernel walks, because
// a CU's LDS is shared with the tree top (below): 10 entries beside the 9-node top of the 8-wide tree (six workgroups
// per CU), 8 beside the 85-node top of the 4-wide tree (five workgroups).  Depths of 8 to 12 trace equally fast.
#ifndef V4R_LDS_DEPTH
#define V4R_LDS_DEPTH 10
#endif
#ifndef V4R_LDS_DEPTH_4
#define V4R_LDS_DEPTH_4 8
#endif
constexpr int lds_depth(bool wide8) { return wide8 ? V4R_LDS_DEPTH : V4R_LDS_DEPTH_4; }
// The LDS part is addressed through an LDS-typed pointer: through a generic one the compiler merges
// the two halves of push/pop into flat loads and stores, which are slower and wait on both counters.
typedef __attribute__((address_space(3))) uint2 lds_uint2;
// Top of the tree in LDS (TOP_LEVELS levels, at most TOP_NODES nodes): every ray starts there, and an LDS read has
// a fraction of the latency of a vector-memory read that hits in L1 and does not occupy the texture addresser.
#ifndef V4R_TOP_LEVELS
#define V4R_TOP_LEVELS 4  // (three levels and 10-deep stacks: the spheres scene traces 2.4 % slower)
#endif
constexpr int TOP_LEVELS = V4R_TOP_LEVELS;  // of the 4-wide tree
constexpr int TOP_NODES = TOP_LEVELS == 4 ? 85 : (TOP_LEVELS == 3 ? 21 : (TOP_LEVELS == 2 ? 5 : 1));
#ifndef V4R_TOP_LEVELS_8
#define V4R_TOP_LEVELS_8 2
#endif
constexpr int TOP_LEVELS_8 = V4R_TOP_LEVELS_8;  // of the 8-wide tree
constexpr int TOP_NODES_8 = TOP_LEVELS_8 == 3 ? 73 : (TOP_LEVELS_8 == 2 ? 9 : 1);
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
constexpr uint32_t TRAV_NONE_WORD = 0xFFFFFFFFu;  // "no entry" (see Wide)
template <int LDS_DEPTH>
struct StackT {
    static constexpr int lds_entries = LDS_DEPTH;
    const lds_v4f *top = nullptr;  // 8 vectors per node, nodes 0 .. n_top-1; n_top = 0: no copy in LDS
    uint32_t n_top = 0;
    lds_uint2 *lds;    // this thread's column: lds[e * block_dim]
    uint32_t stride;   // block_dim
    uint2 *spill;      // this thread's column in the global slab: spill[e * spill_stride]
    size_t spill_stride;
    int sp;
#ifdef V4R_DEBUG_CHECKS
    unsigned long long *errors = nullptr;
    int spill_depth = 0;  // entries of this thread's column in the slab
#endif
    __device__ __forceinline__ void push(uint32_t node, float entry) {
        uint2 v = make_uint2(node, __float_as_uint(entry));
        if (sp < LDS_DEPTH) {
            lds_uint2 *p = lds + (uint32_t)sp * stride;
            p->x = v.x;
            p->y = v.y;
        } else {
#ifdef V4R_DEBUG_CHECKS
            if (!V4R_CHECK(errors, CHECK_STACK_OVERFLOW, sp - LDS_DEPTH < spill_depth)) return;
#endif
            spill[(size_t)(sp - LDS_DEPTH) * spill_stride] = v;
        }
        ++sp;
    }
    __device__ __forceinline__ uint2 pop() {
#ifdef V4R_DEBUG_CHECKS
        if (!V4R_CHECK(errors, CHECK_STACK_UNDERFLOW, sp > 0)) return make_uint2(TRAV_NONE_WORD, 0u);
#endif
        --sp;
        uint2 v;
        if (sp < LDS_DEPTH) {
            const lds_uint2 *p = lds + (uint32_t)sp * stride;
            v.x = p->x;
            v.y = p->y;
        } else {
            v = spill[(size_t)(sp - LDS_DEPTH) * spill_stride];
        }
        return v;
    }
};

#ifndef V4R_VOTE_NUM
#define V4R_VOTE_NUM 1
#define V4R_VOTE_DEN 2
#endif
constexpr uint32_t TRAV_NONE = TRAV_NONE_WORD;
#ifndef V4R_GIVE_UP_MIN_STEPS
#define V4R_GIVE_UP_MIN_STEPS 2u  // node steps a wave makes before it may give rays up (0 / 1 / 2 / 3: C4 -0.5 / +1 / +2 / 0 %, C3 -1 / +2 / +1.8 / 0 %)
#endif

// ---- traversal -------------------------------------------------------------------------------------
// BVH::traverse (bvh/bvh.rs:182-239) + closest-hit selection (renderer.rs:149-183), restructured for a
// wave of 64 rays over a 4-wide tree (built from the host's binary tree when the scene is uploaded: a
// node's children are its grandchildren where that fits): half as many dependent node fetches per
// ray as the binary tree, each one 128-byte line.  The closest hit is the one the reference's
// candidate-set traversal selects (ties go to the lowest primitive index, whatever the visiting order).
// Across the wave the kinds of work run in separate phases chosen by vote, so that the
// primitive-intersection code is not executed (mostly masked off) on every node step: a lane whose
// current entry is a primitive idles while the others keep expanding interior nodes; once at least
// V4R_VOTE_NUM/V4R_VOTE_DEN of the live lanes wait like that, their primitives are intersected together.  Every child that the ray enters -- interior node or leaf
// primitive -- goes through one stack ordered near to far, so a lane's state is just the entry it
// is working on: an interior node (expand it: wide_step) or a primitive (parked until the wave runs
// the leaf phase: wide_leaves).  Stack entries carry the box entry distance with its two lowest
// mantissa bits cleared (they hold the child slot while the four children are sorted); the value is
// only ever compared as `entry <= closest hit`, for which a slightly smaller entry is merely
// conservative.
//
// Node layout (8 float4): lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] child[4] {never-cull mask,-,-,-}; an
// empty slot has NaN planes (never entered) and child word TRAV_NONE.
struct Wide {
    uint32_t cur;      // interior node index, V4R_LEAF_BIT | primitive, or TRAV_NONE when finished / absent
    // (The box entry distance of the current entry is not kept: it was compared with the closest hit when the entry was
    // chosen -- in the node step or in wide_pop -- and a lane's closest hit only moves in its own leaf phase, after which
    // wide_pop chooses again; so a parked primitive never needs a second look at its distance.)
    __device__ __forceinline__ bool parked() const { return cur != TRAV_NONE && (cur & V4R_LEAF_BIT); }
    __device__ __forceinline__ bool stepping() const { return !(cur & V4R_LEAF_BIT); }
    __device__ __forceinline__ void start() {
        cur = 0u;
    }
    __device__ __forceinline__ void clear() {
        cur = TRAV_NONE;
    }
};

template <class Stack>
V4R_DEV void wide_pop(Stack &st, float best_t, Wide &W) {
    W.cur = TRAV_NONE;
    while (st.sp > 0) {
        uint2 e = st.pop();
        if (__uint_as_float(e.y) <= best_t) {
            W.cur = e.x;
            return;
        }
    }
}

// A constant medium breaks the rule that lets far subtrees be skipped: its "hit" lies at entry + free path
// along the whole line (constant_medium.rs:44-84), which may be in front of its box entry, even behind the
// origin.  Children whose subtree holds a medium (bit k of the node's eighth float4) are therefore never
// culled by distance: they are entered whenever the ray meets their box and carry entry distance -inf.
// Node step of the compressed 8-wide tree.  A node is one 128-byte line: {p.xyz, biased exponents x|y<<8|z<<16 | never-cull
// mask << 24}, eight child words, then the children's planes as bytes -- lower x, lower y, lower z, upper x, upper y, upper z,
// eight each: plane = p + byte * 2^exponent (lower planes were rounded down, upper ones up, so the decoded box contains the
// child's).  In ray space a plane's distance is (p - o) * inv + byte * (2^exponent * inv): one multiply-add per plane.  `inv`
// must be finite (see traverse).  Entered children are sorted near to far by keys `entry bits & ~7 | slot`.
template <bool COUNT, bool MEDIA, class Stack>
V4R_DEV void wide_step8(const SceneDev &S, V3 o, V3 inv, Stack &st, const Best &best, Wide &W, TraceCounters &cnt) {
    float4 head, kid_lo, kid_hi, qa, qb, qc;
    if (W.cur < st.n_top) {
        const lds_v4f *tp = st.top + 8u * W.cur;
        auto ld = [&](int k) {
            v4f v = tp[k];
            return make_float4(v.x, v.y, v.z, v.w);
        };
        head = ld(0), kid_lo = ld(1), kid_hi = ld(2), qa = ld(3), qb = ld(4), qc = ld(5);
    } else {
        const float4 *np = S.nodes4 + 8 * (size_t)W.cur;
        head = np[0], kid_lo = np[1], kid_hi = np[2], qa = np[3], qb = np[4], qc = np[5];
    }
    if (COUNT) cnt.nodes++;
    const uint32_t bits = __float_as_uint(head.w);
    const uint32_t never_cull = MEDIA ? bits >> 24 & 255u : 0u;  // children that hold a medium: entered when their box is met, at any distance
    const float ax = (head.x - o.x) * inv.x, ay = (head.y - o.y) * inv.y, az = (head.z - o.z) * inv.z;
    const float bx = __uint_as_float((bits & 255u) << 23) * inv.x, by = __uint_as_float((bits >> 8 & 255u) << 23) * inv.y,
                bz = __uint_as_float((bits >> 16 & 255u) << 23) * inv.z;
    // Along an axis the ray meets a box's lower plane first when it travels upwards, its upper plane first otherwise: byte
    // -> distance is monotonic (one rounding of byte * b + a, lower byte <= upper byte), so the nearer of the two distances
    // is the one of the plane picked by the sign of the reciprocal -- the same value min / max of the pair would give.
    // The words that hold the near and the far planes are picked once per node instead of a min and a max per child and axis.
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    const float nx0 = sx ? qb.z : qa.x, nx1 = sx ? qb.w : qa.y, fx0 = sx ? qa.x : qb.z, fx1 = sx ? qa.y : qb.w;
    const float ny0 = sy ? qc.x : qa.z, ny1 = sy ? qc.y : qa.w, fy0 = sy ? qa.z : qc.x, fy1 = sy ? qa.w : qc.y;
    const float nz0 = sz ? qc.z : qb.x, nz1 = sz ? qc.w : qb.y, fz0 = sz ? qb.x : qc.z, fz1 = sz ? qb.y : qc.w;
    uint32_t key[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // byte c of the eight that the two words (w0: children 0-3, w1: 4-7) hold, as a plane distance
        auto byte = [&](float w0, float w1) { return (float)(__float_as_uint(c < 4 ? w0 : w1) >> (8 * (c & 3)) & 255u); };
        // (the near and the far plane of an axis in one v_pk_fma_f32: 22 fewer vector instructions per step, no faster)
        auto plane = [&](float w0, float w1, float a, float b) { return __fmaf_rn(byte(w0, w1), b, a); };
        const float entry = fmaxf(fmaxf(plane(nx0, nx1, ax, bx), plane(ny0, ny1, ay, by)), fmaxf(plane(nz0, nz1, az, bz), EPSILON));
        const float leave = fminf(fminf(plane(fx0, fx1, ax, bx), plane(fy0, fy1, ay, by)), plane(fz0, fz1, az, bz));
        const bool wanted = entry <= best.t || (never_cull >> c & 1u);
        const float kid = c == 0 ? kid_lo.x : c == 1 ? kid_lo.y : c == 2 ? kid_lo.z : c == 3 ? kid_lo.w : c == 4 ? kid_hi.x : c == 5 ? kid_hi.y : c == 6 ? kid_hi.z : kid_hi.w;
        key[c] = (entry < leave && wanted && __float_as_uint(kid) != TRAV_NONE_WORD) ? ((__float_as_uint(entry) & ~7u) | (uint32_t)c) : 0xFFFFFFFFu;
    }
    // sort the eight keys ascending (Batcher's odd-even merge sort, 19 compare-exchanges); children not entered sort last
    auto cas = [&](int a, int b) {
        uint32_t lo = min(key[a], key[b]), hi = max(key[a], key[b]);
        key[a] = lo;
        key[b] = hi;
    };
    cas(0, 1), cas(2, 3), cas(4, 5), cas(6, 7);
    cas(0, 2), cas(1, 3), cas(4, 6), cas(5, 7);
    cas(1, 2), cas(5, 6);
    cas(0, 4), cas(1, 5), cas(2, 6), cas(3, 7);
    cas(2, 4), cas(3, 5);
    cas(1, 2), cas(3, 4), cas(5, 6);
    auto child_of = [&](uint32_t k) {  // a tree of selects on the slot's three bits (kept out of scratch: no indexed array)
        const float a0 = (k & 1u) ? kid_lo.y : kid_lo.x, a1 = (k & 1u) ? kid_lo.w : kid_lo.z, a2 = (k & 1u) ? kid_hi.y : kid_hi.x,
                    a3 = (k & 1u) ? kid_hi.w : kid_hi.z;
        const float b0 = (k & 2u) ? a1 : a0, b1 = (k & 2u) ? a3 : a2;
        return __float_as_uint((k & 4u) ? b1 : b0);
    };
    auto entry_of = [&](uint32_t k) { return (never_cull >> (k & 7u) & 1u) ? -V4R_INF : __uint_as_float(k & ~7u); };
    // farther children first, so that the nearest of them is popped first
#pragma unroll
    for (int j = 7; j >= 1; --j)
        if (key[j] != 0xFFFFFFFFu) st.push(child_of(key[j]), entry_of(key[j]));
    if (key[0] != 0xFFFFFFFFu) {
        W.cur = child_of(key[0]);
    } else {
        wide_pop(st, best.t, W);
    }
}
template <bool COUNT, bool MEDIA, class Stack>
V4R_DEV void wide_step(const SceneDev &S, V3 o, V3 inv, Stack &st, const Best &best, Wide &W, TraceCounters &cnt) {
    float4 lx, ly, lz, hx, hy, hz, kids;
    uint32_t never_cull = 0u;
    if (W.cur < st.n_top) {
        const lds_v4f *tp = st.top + 8u * W.cur;
        auto ld = [&](int k) {
            v4f v = tp[k];
            return make_float4(v.x, v.y, v.z, v.w);
        };
        lx = ld(0), ly = ld(1), lz = ld(2), hx = ld(3), hy = ld(4), hz = ld(5), kids = ld(6);
        if (MEDIA) never_cull = __float_as_uint(tp[7].x);
    } else {
        const float4 *np = S.nodes4 + 8 * (size_t)W.cur;
        lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5], kids = np[6];
        if (MEDIA) never_cull = __float_as_uint(np[7].x);
    }
    if (COUNT) cnt.nodes++;
    // Ray::intersects_aabb (ray.rs:167-189): the same running-interval test with strict tmin < tmax, the
    // per-axis divisions replaced by multiplications with a reciprocal computed once per ray; the interval
    // only ever shrinks, so testing it once after the third axis accepts exactly the same boxes.  Two
    // children per packed instruction (v_pk_add_f32 / v_pk_mul_f32), min/max as v_min3/v_max3.
    typedef float pair __attribute__((ext_vector_type(2)));
    uint32_t key[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // children (0,1) then (2,3), two per packed instruction
        pair ax0 = ((h ? pair{lx.z, lx.w} : pair{lx.x, lx.y}) - o.x) * inv.x, ax1 = ((h ? pair{hx.z, hx.w} : pair{hx.x, hx.y}) - o.x) * inv.x;
        pair ay0 = ((h ? pair{ly.z, ly.w} : pair{ly.x, ly.y}) - o.y) * inv.y, ay1 = ((h ? pair{hy.z, hy.w} : pair{hy.x, hy.y}) - o.y) * inv.y;
        pair az0 = ((h ? pair{lz.z, lz.w} : pair{lz.x, lz.y}) - o.z) * inv.z, az1 = ((h ? pair{hz.z, hz.w} : pair{hz.x, hz.y}) - o.z) * inv.z;
        float e0 = fmaxf(fmaxf(fminf(ax0.x, ax1.x), fminf(ay0.x, ay1.x)), fmaxf(fminf(az0.x, az1.x), EPSILON));
        float f0 = fminf(fminf(fmaxf(ax0.x, ax1.x), fmaxf(ay0.x, ay1.x)), fmaxf(az0.x, az1.x));
        float e1 = fmaxf(fmaxf(fminf(ax0.y, ax1.y), fminf(ay0.y, ay1.y)), fmaxf(fminf(az0.y, az1.y), EPSILON));
        float f1 = fminf(fminf(fmaxf(ax0.y, ax1.y), fmaxf(ay0.y, ay1.y)), fmaxf(az0.y, az1.y));
        // entry distances are >= EPSILON > 0, so their bit patterns order like the floats
        bool in0 = e0 <= best.t || (never_cull >> (2 * h) & 1u), in1 = e1 <= best.t || (never_cull >> (2 * h + 1) & 1u);
        key[2 * h] = (e0 < f0 && in0) ? ((__float_as_uint(e0) & ~3u) | (uint32_t)(2 * h)) : 0xFFFFFFFFu;
        key[2 * h + 1] = (e1 < f1 && in1) ? ((__float_as_uint(e1) & ~3u) | (uint32_t)(2 * h + 1)) : 0xFFFFFFFFu;
    }
    // sort the four keys ascending (5 compare-exchanges); missed children sort last
    auto cas = [&](int a, int b) {
        uint32_t lo = min(key[a], key[b]), hi = max(key[a], key[b]);
        key[a] = lo;
        key[b] = hi;
    };
    cas(0, 1);
    cas(2, 3);
    cas(0, 2);
    cas(1, 3);
    cas(1, 2);
    auto child_of = [&](uint32_t k) {
        uint32_t slot = k & 3u;
        float c = slot == 0u ? kids.x : (slot == 1u ? kids.y : (slot == 2u ? kids.z : kids.w));
        return __float_as_uint(c);
    };
    auto entry_of = [&](uint32_t k) { return (never_cull >> (k & 3u) & 1u) ? -V4R_INF : __uint_as_float(k & ~3u); };
    // farther children first, so that the nearest of them is popped first
    if (key[3] != 0xFFFFFFFFu) st.push(child_of(key[3]), entry_of(key[3]));
    if (key[2] != 0xFFFFFFFFu) st.push(child_of(key[2]), entry_of(key[2]));
    if (key[1] != 0xFFFFFFFFu) st.push(child_of(key[1]), entry_of(key[1]));
    if (key[0] != 0xFFFFFFFFu) {
        W.cur = child_of(key[0]);
    } else {
        wide_pop(st, best.t, W);
    }
}

// Intersect the parked leaf of every lane of the wave -- a block of consecutive primitives -- then pop what each of
// those lanes does next.
template <bool COUNT, bool MEDIA, class KeyFn, class Stack>
V4R_DEV void wide_leaves(const SceneDev &S, V3 o, V3 d, Stack &st, Best &best, Wide &W, TraceCounters &cnt, KeyFn key) {
    bool parked = W.parked();
    if (COUNT) cnt.wave_leaves++;
    if (parked) {
#if V4R_LEAF_PRIMS > 1
        // (the block is walked inside the leaf word itself -- first + 1, count - 1 -- so that the loop keeps no register of its own)
        while (true) {
            if (COUNT) cnt.prims++;
            consider<MEDIA>(S, leaf_first(W.cur), o, d, best, key);
            if ((W.cur >> 28 & 7u) == 0u) break;
            W.cur -= (1u << 28) - 1u;
        }
#else
        if (COUNT) cnt.prims++;
        consider<MEDIA>(S, leaf_first(W.cur), o, d, best, key);
#endif
        wide_pop(st, best.t, W);
    }
}

// Closest hit of one ray per lane.  Must be called by every lane of the wave; lanes with
// active == false only take part in the votes.  `best` comes in as the closest hit known so far
// (none: t = +inf).  When at most `give_up_lanes` lanes of the wave are still traversing (and the wave
// has made a few steps), those lanes give their rays up and the function returns true for them, with
// the closest hit found so far in `best`: the caller queues such rays for a second launch in which
// they sit next to each other instead of holding a nearly empty wave.  0 = never give up.
// MEDIA: how the scene's constant media are met -- MEDIA_NONE: the scene has none; MEDIA_IN_TREE: they are leaves of
// the tree (wide_step / consider carry the medium code and the never-cull rule); MEDIA_DIRECT: see below.
enum : int { MEDIA_NONE = 0, MEDIA_IN_TREE = 1, MEDIA_DIRECT = 2 };
template <bool COUNT, int MEDIA, bool WIDE8, class KeyFn, class Stack>
V4R_DEV bool traverse(const SceneDev &S, V3 o, V3 d, bool active, Stack &st, Best &best, TraceCounters &cnt, KeyFn key,
                      uint32_t give_up_lanes, bool resumed = false) {
    if (S.n_prims == 0) return false;
    // Scenes with a few constant media: the media are not in the tree; every ray tests them first (the closest-hit
    // rule does not depend on the order), and the traversal runs without the medium code and without subtrees that
    // must not be culled.  A ray that resumes (second trace launch) carries those hits in `best` already.
    if (MEDIA == MEDIA_DIRECT && !resumed) {
        for (uint32_t k = 0; k < S.n_media_direct; ++k)
            if (active) {
                if (COUNT) cnt.prims++;
                consider<true>(S, S.media_direct[k], o, d, best, key);
            }
    }
    if (S.n_nodes == 0) {  // single shape: tested without any box (bvh.rs:218-224)
        if (active) {
            if (COUNT) cnt.prims++;
            consider<MEDIA == MEDIA_IN_TREE>(S, 0, o, d, best, key);
        }
        return false;
    }
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    // the compressed node step adds byte * (scale * inv) to (p - o) * inv: both must stay finite.  A ray (anti)parallel to an
    // axis gets a huge finite reciprocal instead of an infinite one: its plane distances are +-huge with the right signs.
    if (WIDE8) inv = v3(fminf(fmaxf(inv.x, -1e18f), 1e18f), fminf(fmaxf(inv.y, -1e18f), 1e18f), fminf(fmaxf(inv.z, -1e18f), 1e18f));
    Wide W;
    W.clear();
    if (active) W.start();
    st.sp = 0;
    uint32_t steps = 0;
    while (true) {
        while (true) {
            unsigned long long m_step = __ballot(W.stepping());
            if (m_step == 0ull) break;
            unsigned long long m_parked = __ballot(W.parked());
            if (give_up_lanes != 0u && steps >= V4R_GIVE_UP_MIN_STEPS && (uint32_t)__popcll(m_step | m_parked) <= give_up_lanes) return W.cur != TRAV_NONE;
            if (m_parked != 0ull && __popcll(m_parked) * V4R_VOTE_DEN >= (__popcll(m_parked) + __popcll(m_step)) * V4R_VOTE_NUM) break;
            if (COUNT) cnt.wave_steps++;
            ++steps;
#ifdef V4R_PHASE_CLOCKS
            const unsigned long long t0 = COUNT ? phase_clock() : 0ull;
#endif
            if (W.stepping()) {
                if (WIDE8) wide_step8<COUNT, MEDIA == MEDIA_IN_TREE>(S, o, inv, st, best, W, cnt);
                else wide_step<COUNT, MEDIA == MEDIA_IN_TREE>(S, o, inv, st, best, W, cnt);
            }
#ifdef V4R_PHASE_CLOCKS
            if (COUNT) cnt.clk_node += phase_clock() - t0;
#endif
        }
#ifdef V4R_PHASE_CLOCKS
        const unsigned long long t1 = COUNT ? phase_clock() : 0ull;
#endif
        if (__ballot(W.parked()) != 0ull) wide_leaves<COUNT, MEDIA == MEDIA_IN_TREE>(S, o, d, st, best, W, cnt, key);
#ifdef V4R_PHASE_CLOCKS
        if (COUNT) cnt.clk_leaf += phase_clock() - t1;
#endif
        if (__ballot(W.cur != TRAV_NONE) == 0ull) break;
    }
    return false;
}

// ---- hit reconstruction -------------------------------------------------------------------------
// HitRecord::face_normal, hittable/mod.rs:146-154
V4R_DEV void face_normal(V3 dir, V3 outward, bool &front, V3 &normal) {
    front = dot(dir, outward) < 0.0f;
    normal = front ? outward : -outward;
}
// sphere_uv, hittable/sphere.rs:38-45
V4R_DEV void sphere_uv(V3 p, float &u, float &v) {
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PI;
    u = phi / TWO_PI;
    v = theta * FRAC_1_PI;
}

// Rebuild the full hit record of primitive `prim_index` at world distance t (and triangle
// barycentrics bu,bv) for the ray (o,d): the second half of each Hittable::hit
// (sphere.rs:80-92, aa_rect.rs:130-143, triangle.rs:81-94) and the epilogue of
// TransformHittable::hit (transform.rs:85-93).  uv is produced only when `want_uv`.
V4R_DEV HitD reconstruct(const SceneDev &S, uint32_t prim_index, V3 o, V3 d, float t, float bu, float bv, bool want_uv) {
    Prim pr = load_prim(S.prims, prim_index);
    uint32_t xf = pr.xform();
    const float *x = nullptr;
    float nrm = 1.0f;
    if (xf != V4R_NO_XFORM) {
        x = S.xforms + 24 * (size_t)xf;
        to_object(x, o, d, nrm);
        t = t * nrm;
    }
    HitD h;
    h.u = h.v = 0.0f;
    h.p = o + d * t;
    uint32_t kind = pr.type() & V4R_PRIM_TYPE_MASK;
    if (kind == V4R_PRIM_SPHERE) {
        V3 outward = normalize(h.p - v3(pr.a.x, pr.a.y, pr.a.z));
        face_normal(d, outward, h.front, h.normal);
        if (want_uv) sphere_uv(outward, h.u, h.v);
    } else if (kind == V4R_PRIM_TRIANGLE) {
        const float4 *tp = S.tri + 4 * (size_t)pr.aux();
        float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
        // n0 = q0.xyz, n1 = (q0.w,q1.x,q1.y), n2 = (q1.z,q1.w,q2.x); uv0 = (q2.y,q2.z), uv1 = (q2.w,q3.x), uv2 = (q3.y,q3.z)
        float w = 1.0f - bu - bv;
        h.normal = normalize(v3(q0.x, q0.y, q0.z) * w + v3(q0.w, q1.x, q1.y) * bu + v3(q1.z, q1.w, q2.x) * bv);
        h.u = q2.y * w + q2.w * bu + q3.y * bv;
        h.v = q2.z * w + q3.x * bu + q3.z * bv;
        V3 e12 = v3(pr.a.w, pr.b.x, pr.b.y), e13 = v3(pr.b.z, pr.b.w, pr.c.x);
        h.front = dot(e12, cross(d, e13)) > 0.0f;  // triangle.rs:92
    } else if (kind <= V4R_PRIM_ZX_RECT) {
        int ia, ib, ik;
        rect_axes(pr.type(), ia, ib, ik);
        float a = comp(o, ia) + t * comp(d, ia);
        float b = comp(o, ib) + t * comp(d, ib);
        h.u = (a - pr.a.x) / (pr.a.y - pr.a.x);
        h.v = (b - pr.a.z) / (pr.a.w - pr.a.z);
        float s = (pr.type() & V4R_PRIM_POSITIVE) ? 1.0f : -1.0f;
        V3 outward = v3(ik == 0 ? s : 0.0f, ik == 1 ? s : 0.0f, ik == 2 ? s : 0.0f);
        face_normal(d, outward, h.front, h.normal);
    } else {  // constant medium, constant_medium.rs:77-84
        h.normal = v3(1, 0, 0);
        h.front = true;
    }
    if (x) {
        h.p = mat_mul(x, h.p) + v3(x[9], x[10], x[11]);
        h.normal = normalize(mat_mul(x, h.normal));
    }
    return h;
}

// ---- textures -----------------------------------------------------------------------------------
V4R_DEV int f32_to_i32_sat(float x) {  // Rust `as isize`, narrowed: only the low 8 bits are used
    // Beyond the i32 range an f32 is a multiple of 256, so the low 8 bits of the reference's isize
    // are zero there; NaN converts to 0.
    if (!(fabsf(x) < 2147483648.0f)) return 0;
    return (int)x;
}
V4R_DEV uint32_t f32_to_u32_sat(float x) {  // Rust `as u32`
    if (!(x == x) || x <= 0.0f) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// Perlin::noise + perlin_interp, texture/noise.rs:41-89,109-140, in two halves: the lattice cell of a point (its
// fractional position and the six permutation bytes its corners hash from, read as three pairs), and the interpolated gradient noise of a
// cell.  Split so that Perlin::turb can ask for the next octave's permutation bytes before it waits for this
// octave's gradients: the two table look-ups of an octave depend on each other, the octaves do not.
struct PerlinCell {
    float u, v, w;
    uint32_t px0, px1, py0, py1, pz0, pz1;  // perm_x[i], perm_x[i + 1], perm_y[j], perm_y[j + 1], perm_z[k], perm_z[k + 1]
};
template <class PermPtr>
V4R_DEV PerlinCell perlin_cell(PermPtr perm, V3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    PerlinCell c;
    c.u = p.x - fx, c.v = p.y - fy, c.w = p.z - fz;
    uint32_t i = (uint32_t)f32_to_i32_sat(fx), j = (uint32_t)f32_to_i32_sat(fy), k = (uint32_t)f32_to_i32_sat(fz);
    const uint32_t px = perm[i & 255u], py = perm[256 + (j & 255u)], pz = perm[512 + (k & 255u)];  // each: this entry, the next << 8
    c.px0 = px & 255u, c.px1 = px >> 8;
    c.py0 = py & 255u, c.py1 = py >> 8;
    c.pz0 = pz & 255u, c.pz1 = pz >> 8;
    return c;
}
template <class VecPtr>
V4R_DEV float perlin_noise(VecPtr vec, const PerlinCell &cell) {
    const float u = cell.u, v = cell.v, w = cell.w;
    float uu = u * u * (3.0f - u * 2.0f);
    float vv = v * v * (3.0f - v * 2.0f);
    float ww = w * w * (3.0f - w * 2.0f);
    // The eight lattice corners in the reference's order (di, dj, dk), two at a time (dk = 0, 1) with packed f32
    // multiplies and adds: every product and sum is the one of the scalar form, rounded alike, and the eight terms
    // are added to `accum` in the same order.
    typedef float pair __attribute__((ext_vector_type(2)));
    const pair dw = {w, w - 1.0f}, wk = {1.0f - ww, ww};
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            uint32_t base = (di ? cell.px1 : cell.px0) ^ (dj ? cell.py1 : cell.py0);
            const auto c0 = vec[base ^ cell.pz0], c1 = vec[base ^ cell.pz1];
            float du = u - (float)di, dv = v - (float)dj;
            pair d = (pair{c0.x, c1.x} * du + pair{c0.y, c1.y} * dv) + pair{c0.z, c1.z} * dw;
            float wij = (di ? uu : 1.0f - uu) * (dj ? vv : 1.0f - vv);
            pair term = (wk * wij) * d;
            accum += term.x;
            accum += term.y;
        }
    return accum;
}

// Perlin::turb, texture/noise.rs:91-107; the next octave's permutation bytes are asked for before this octave's gradients are waited for
template <class VecPtr, class PermPtr>
V4R_DEV float perlin_turb(VecPtr vec, PermPtr perm, V3 q, uint32_t depth) {
    float accum = 0.0f, weight = 1.0f;
    PerlinCell cell = perlin_cell(perm, q);
    for (uint32_t dpt = 0; dpt < depth; ++dpt) {
        q = q * 2.0f;
        const PerlinCell next = perlin_cell(perm, q);  // (one cell past the last octave: three entries of a table that is always there)
        accum += weight * perlin_noise(vec, cell);
        weight *= 0.5f;
        cell = next;
    }
    return accum;
}
// Texture::value for the four texture kinds; `uvf` computes (u,v) on demand.
// NOISE = false compiles the Perlin branch out, for material classes whose texture trees hold no noise
// texture (the eight unrolled lattice terms cost ~25 registers).
// `root` is the 32-byte record of the texture to evaluate, already in registers (the shading kernels
// get it from the per-primitive table prim_shade, which saves the material -> texture pointer chase);
// children of checker textures are fetched from S.textures.
template <bool NOISE, class UV>
V4R_DEV V3 texture_value(const SceneDev &S, uint4 root0, uint4 root1, V3 p, UV uvf, uint32_t &texel_fetches) {
    uint32_t t[8] = {root0.x, root0.y, root0.z, root0.w, root1.x, root1.y, root1.z, root1.w};
    for (int guard = 0; guard < 64; ++guard) {
        uint32_t kind = t[0];
        if (kind == V4R_TEX_SOLID)  // texture/solid_color.rs:25-27
            return v3(__uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
        if (kind == V4R_TEX_CHECKER) {  // texture/checker.rs:31-38
            float dens = __uint_as_float(t[4]);
            float sines = sinf(dens * p.x) * sinf(dens * p.y) * sinf(dens * p.z);
            uint32_t tex = sines > 0.0f ? t[6] : t[5];  // even : odd
            const uint4 *child = reinterpret_cast<const uint4 *>(S.textures + 8 * (size_t)tex);
            uint4 c0 = child[0], c1 = child[1];
            t[0] = c0.x, t[1] = c0.y, t[2] = c0.z, t[3] = c0.w;
            t[4] = c1.x, t[5] = c1.y, t[6] = c1.z, t[7] = c1.w;
            continue;
        }
        if (kind == V4R_TEX_IMAGE) {  // texture/image.rs:25-73
            float u, v;
            uvf(u, v);
            uint4 img = S.images[t[5]];
            float cu = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
            float cv = 1.0f - (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
            uint32_t i = f32_to_u32_sat((float)img.y * cu);
            if (!(i < img.y)) i = img.y - 1;
            uint32_t j = f32_to_u32_sat((float)img.z * cv);
            if (!(j < img.z)) j = img.z - 1;
            uint32_t px = S.texels[(size_t)img.x + (size_t)j * img.y + i];
            texel_fetches++;
            return v3((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f);
        }
        if (NOISE && kind == V4R_TEX_NOISE) {  // texture/noise.rs:91-107,165-170
            float scale = __uint_as_float(t[4]);
            uint32_t depth = t[6];
            const float4 *vec = S.perlin_vec + 256 * (size_t)t[5];
            const uint16_t *perm = S.perlin_perm + 768 * (size_t)t[5];
            V3 q = p * scale;
            const float accum = perlin_turb(vec, perm, q, depth);
            float g = fabsf(accum);
            return v3(g, g, g);
        }
        break;
    }
    return v3(0, 0, 0);
}

// ---- pdfs ---------------------------------------------------------------------------------------
// Rotation3::face_towards(dir, up): columns x = normalize(up x z), y = normalize(z x x), z = normalize(dir)
struct Rot {
    V3 x, y, z;
};
V4R_DEV Rot face_towards(V3 dir, V3 up) {
    Rot r;
    r.z = normalize(dir);
    r.x = normalize(cross(up, r.z));
    r.y = normalize(cross(r.z, r.x));
    return r;
}
V4R_DEV V3 rot_apply(const Rot &r, V3 v) { return r.x * v.x + r.y * v.y + r.z * v.z; }
V4R_DEV V3 pick_up(float x_component) { return fabsf(x_component) > 0.9f ? v3(0, 1, 0) : v3(1, 0, 0); }

// CosinePdf::value / generate, pdf/cosine.rs:32-47
V4R_DEV float cosine_value(V3 normal, V3 dir) {
    float c = dot(dir, normal);
    return c > 0.0f ? c * FRAC_1_PI : 0.0f;
}
V4R_DEV V3 cosine_generate(V3 normal, float r1, float r2) {
    Rot rot = face_towards(normal, pick_up(normal.x));
    float z = sqrtf(1.0f - r2);
    float phi = TWO_PI * r1;
    float sq = sqrtf(r2);
    float s, c;
    sincosf(phi, &s, &c);
    return rot_apply(rot, v3(c * sq, s * sq, z));
}

// random_to_sphere, random.rs:27-39
V4R_DEV V3 random_to_sphere(float radius, float distance_squared, float u_phi, float r2) {
    float phi = TWO_PI * u_phi;
    float z = 1.0f + r2 * (sqrtf(1.0f - radius * radius / distance_squared) - 1.0f);
    float xy = sqrtf(1.0f - z * z);
    float s, c;
    sincosf(phi, &s, &c);
    return v3(c * xy, s * xy, z);
}

// Samplable::generate: sphere.rs:132-144, aa_rect.rs:182-185 (+ rect_random!, :54-76),
// triangle.rs:134-146, wrapped by TransformHittable::generate (transform.rs:118-124).
V4R_DEV V3 light_generate(const SceneDev &S, uint32_t prim_index, V3 origin, float u0, float u1) {
    Prim pr = load_prim(S.prims, prim_index);
    uint32_t xf = pr.xform();
    const float *x = nullptr;
    if (xf != V4R_NO_XFORM) {
        x = S.xforms + 24 * (size_t)xf;
        origin = mat_mul(x + 12, origin) + v3(x[21], x[22], x[23]);
    }
    V3 dir;
    uint32_t kind = pr.type() & V4R_PRIM_TYPE_MASK;
    if (kind == V4R_PRIM_SPHERE) {
        V3 direction = v3(pr.a.x, pr.a.y, pr.a.z) - origin;
        Rot rot = face_towards(direction, pick_up(normalize(direction).x));
        dir = normalize(rot_apply(rot, random_to_sphere(pr.a.w, norm2(direction), u0, u1)));
    } else if (kind == V4R_PRIM_TRIANGLE) {
        float lo = EPSILON, hi = 1.0f - EPSILON;
        float px = lo + (hi - lo) * u0, py = lo + (hi - lo) * u1;
        if (px + py > 1.0f) {
            px = 1.0f - EPSILON - px;
            py = 1.0f - EPSILON - py;
        }
        V3 v0 = v3(pr.a.x, pr.a.y, pr.a.z), e12 = v3(pr.a.w, pr.b.x, pr.b.y), e13 = v3(pr.b.z, pr.b.w, pr.c.x);
        dir = normalize(v0 + e12 * px + e13 * py - origin);
    } else if (kind == V4R_PRIM_XY_RECT) {
        float a = pr.a.x + (pr.a.y - pr.a.x) * u0, b = pr.a.z + (pr.a.w - pr.a.z) * u1;
        dir = normalize(v3(a, b, pr.b.x) - origin);
    } else if (kind == V4R_PRIM_YZ_RECT) {
        float a = pr.a.x + (pr.a.y - pr.a.x) * u0, b = pr.a.z + (pr.a.w - pr.a.z) * u1;
        dir = normalize(v3(pr.b.x, a, b) - origin);
    } else if (kind == V4R_PRIM_ZX_RECT) {  // the x coordinate (b range) is drawn first, aa_rect.rs:69-75
        float b = pr.a.z + (pr.a.w - pr.a.z) * u0, a = pr.a.x + (pr.a.y - pr.a.x) * u1;
        dir = normalize(v3(b, pr.b.x, a) - origin);
    } else {
        dir = v3(1, 0, 0);
    }
    if (x) dir = normalize(mat_mul(x, dir));
    return dir;
}

// Samplable::value of one light primitive (sphere.rs:100-131, aa_rect.rs:151-180, triangle.rs:103-132, wrapped by
// TransformHittable::value, transform.rs:103-116, no Jacobian), as its two ingredients for one lane of a SIMD packet:
// does the direction hit the light (`hit`), and the density the lane is given when the hit record of its packet is not all-miss -- for rectangles and triangles
// the un-clipped plane density (t may be negative); a sphere masks with its own hit (`per_lane`, sphere.rs:107-130).
V4R_DEV float light_probe(const SceneDev &S, const Prim &pr, V3 origin, V3 dir, bool &hit, bool &per_lane) {
    uint32_t xf = pr.xform();
    if (xf != V4R_NO_XFORM) {
        const float *inv = S.xforms + 24 * (size_t)xf + 12;
        origin = mat_mul(inv, origin) + v3(inv[9], inv[10], inv[11]);
        dir = normalize(mat_mul(inv, dir));
    }
    uint32_t kind = pr.type() & V4R_PRIM_TYPE_MASK;
    per_lane = false;
    hit = false;
    if (kind == V4R_PRIM_SPHERE) {
        per_lane = true;
        float t;
        hit = sphere_t(pr, origin, dir, EPSILON, V4R_INF, t);
        if (!hit) return 0.0f;
        float radius = pr.a.w;
        float cos_theta_max = sqrtf(1.0f - radius * radius / norm2(v3(pr.a.x, pr.a.y, pr.a.z) - origin));
        float solid_angle = TWO_PI * (1.0f - cos_theta_max);
        return solid_angle > 0.0f ? 1.0f / solid_angle : V4R_INF;
    }
    if (kind == V4R_PRIM_TRIANGLE) {
        float t, u, v, det;
        hit = tri_t(pr, origin, dir, EPSILON, V4R_INF, t, u, v, det);
        const float4 *tp = S.tri + 4 * (size_t)pr.aux();
        float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
        float w = 1.0f - u - v;
        V3 normal = normalize(v3(q0.x, q0.y, q0.z) * w + v3(q0.w, q1.x, q1.y) * u + v3(q1.z, q1.w, q2.x) * v);
        V3 e12 = v3(pr.a.w, pr.b.x, pr.b.y), e13 = v3(pr.b.z, pr.b.w, pr.c.x);
        float area = norm(cross(e12, e13)) * 0.5f;
        float cosine = fabsf(dot(dir, normal));
        return cosine > 0.0f ? (t * t) / (cosine * area) : V4R_INF;
    }
    if (kind <= V4R_PRIM_ZX_RECT) {
        float t;
        hit = rect_t(pr, origin, dir, EPSILON, V4R_INF, t);
        int ia, ib, ik;
        rect_axes(pr.type(), ia, ib, ik);
        float area = (pr.a.y - pr.a.x) * (pr.a.w - pr.a.z);
        float cosine = fabsf(comp(dir, ik));
        return cosine > 0.0f ? (t * t) / (cosine * area) : V4R_INF;
    }
    per_lane = true;
    return 0.0f;
}

// The share of a cosine-distributed direction from surface point p (normal n) that hits an untransformed rectangle or
// triangle light, estimated from the light's centre, area and normal; 0 for transformed lights and other kinds.
V4R_DEV float light_cosine_share(const Prim &pr, V3 p, V3 n) {
    if (pr.xform() != V4R_NO_XFORM) return 0.0f;
    V3 centre, ln;
    float area;
    uint32_t kind = pr.type() & V4R_PRIM_TYPE_MASK;
    if (kind == V4R_PRIM_TRIANGLE) {
        V3 e12 = v3(pr.a.w, pr.b.x, pr.b.y), e13 = v3(pr.b.z, pr.b.w, pr.c.x), c = cross(e12, e13);
        centre = v3(pr.a.x, pr.a.y, pr.a.z) + (e12 + e13) * (1.0f / 3.0f);
        float len = norm(c);
        ln = c / len;
        area = len * 0.5f;
    } else if (kind >= V4R_PRIM_XY_RECT && kind <= V4R_PRIM_ZX_RECT) {
        int ia, ib, ik;
        rect_axes(pr.type(), ia, ib, ik);
        float ca = 0.5f * (pr.a.x + pr.a.y), cb = 0.5f * (pr.a.z + pr.a.w), ck = pr.b.x;
        centre = v3(ia == 0 ? ca : (ib == 0 ? cb : ck), ia == 1 ? ca : (ib == 1 ? cb : ck), ia == 2 ? ca : (ib == 2 ? cb : ck));
        ln = v3(ik == 0 ? 1.0f : 0.0f, ik == 1 ? 1.0f : 0.0f, ik == 2 ? 1.0f : 0.0f);
        area = (pr.a.y - pr.a.x) * (pr.a.w - pr.a.z);
    } else {
        return 0.0f;
    }
    V3 w = centre - p;
    float d2 = norm2(w), d = sqrtf(d2);
    float cos_l = fabsf(dot(w, ln)) / d, cos_p = dot(w, n) / d;
    if (!(cos_p > 0.0f)) return 0.0f;
    float h = area * cos_l * cos_p / (PI * d2);
    return h < 1.0f ? h : 1.0f;
}

// HittableGroup::value over the light list (hittables/group.rs:76-89).  What a rectangle or triangle light gives a
// direction that misses it depends on the scene's rule:
//   V4R_FLAG_BOUNDED_LIGHT_PDF  0 (the textbook density);
//   V4R_FLAG_PLANE_LIGHT_PDF    the un-clipped plane density, always;
//   neither (S.packet_rule)     the packet-statistics rule: what the reference's 8-lane packets do to the pdf, in
// expectation, without forming the packets.  There a lane whose direction misses such a light gets the plane density
// when some lane of its packet hits the light and 0 when none does (aa_rect.rs:158-179, triangle.rs:110-131); a packet
// holds rays of one pass and level that hit the same shape (renderer.rs:189-217).  `n` rays hit this ray's shape: they
// are cut into packets of 8, the last holding n mod 8, and the ray sits in that one with probability (n mod 8) / n (one
// draw; from V4R_PACKET_MANY_RAYS rays on every packet counts as full); each of the k - 1 mates samples the light the
// packet drew (`li`, mixture.rs:27-32, group.rs:91-95) half of the time -- that direction hits it -- and a cosine
// direction otherwise, which hits light L with probability h_L (light_cosine_share); so no mate hits L with
// probability (1 - (0.5 [L == li] + 0.5 h_L))^(k-1), decided by one more draw per light.
V4R_DEV float lights_value(const SceneDev &S, V3 origin, V3 normal, V3 dir, uint32_t li, uint32_t n, uint32_t key) {
    const bool bounded = (S.flags & V4R_FLAG_BOUNDED_LIGHT_PDF) != 0u, stat = S.packet_rule != 0u;
    uint32_t k = 8u;
    if (n < 8u) {
        k = n < 1u ? 1u : n;
    } else if (n < V4R_PACKET_MANY_RAYS) {
        uint32_t r = n % 8u;
        if (r != 0u && draw(key, DIM_PACKET_SLOT) * (float)n < (float)r) k = r;
    }
    const float weight = 1.0f / (float)S.n_lights;
    float sum = 0.0f;
    for (uint32_t l = 0; l < S.n_lights; ++l) {
        Prim pr = load_prim(S.prims, S.lights[l]);
        bool hit, per_lane;
        float density = light_probe(S, pr, origin, dir, hit, per_lane);
        if (!per_lane && !hit) {
            if (bounded) {
                density = 0.0f;
            } else if (stat) {
                float q = 1.0f - (0.5f * (l == li ? 1.0f : 0.0f) + 0.5f * light_cosine_share(pr, origin, normal));
                float p_none = 1.0f;
                for (uint32_t m = 1u; m < k; ++m) p_none = p_none * q;
                // one draw per light: dimension 16 + (l mod 16) of the segment's stream, for lights beyond the sixteenth of a
                // key derived from it (lights l and l + 16 would otherwise share their draw)
                const uint32_t light_key = l < 16u ? key : mix32(key + (l >> 4) * 0x9E3779B9u);
                if (!(draw(light_key, DIM_PACKET_ANY + (l & 15u)) >= p_none)) density = 0.0f;
            }
        }
        sum += weight * density;
    }
    return sum;
}

// ---- materials ----------------------------------------------------------------------------------
V4R_DEV V3 reflect(V3 v, V3 n) { return v - n * (dot(v, n) * 2.0f); }  // material/mod.rs:47-52
V4R_DEV V3 refract(V3 uv, V3 n, float ratio) {                        // material/mod.rs:54-63
    float cos_theta = -dot(uv, n);
    V3 perp = (uv + n * cos_theta) * ratio;
    V3 parallel = n * (-sqrtf(1.0f - norm2(perp)));
    return perp + parallel;
}
V4R_DEV float reflectance(float cosine, float ref_idx) {  // material/dielectric.rs:23-27
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    return r0 + (1.0f - r0) * (x2 * x2 * x);
}

}  // namespace v4r
