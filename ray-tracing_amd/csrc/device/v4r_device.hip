// gfx950 wavefront path tracer: kernels and the C ABI of include/v4ray_hip.h.
//
// One batch = `P` full-image passes = P*W*H camera paths.  Per batch, on one of the renderer's streams:
//   for segment s = 0 .. max_depth-1:
//     k_trace(s)                   s = 0: camera rays generated in place (renderer.rs:94-122,
//                                  camera.rs:113-129); s > 0: rays read from the queue the shading
//                                  kernels filled.  4-wide BVH traversal + closest hit; a miss writes
//                                  the path's radiance (background); a hit is ranked within its
//                                  material class in LDS and its ray index is appended to the class
//                                  queue (one global atomic per class per 1024-ray pool)
//     k_shade<class>(s)            per-class material evaluation; the scattered ray goes to slot
//                                  class_base + i of the other ray queue (no atomics)
//   k_resolve                      per-pixel sum of the batch's path radiances into the accumulator
//
// Queue records are 16-byte columns (structure of float4 arrays), so a wave's access to consecutive
// queue entries is one contiguous 1 KiB transaction per column.  A path's radiance is written
// exactly once, when it terminates (miss, light, depth exhausted), into a per-path slot; k_resolve
// adds the slots of a pixel in ascending pass order, which keeps the image bit-reproducible.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "v4r_scene_dev.hpp"
#include "v4ray_hip.h"

using namespace v4r;

// Scratch of v4r_render_noalloc: the wavefront queues, segment counters and stack-spill slab of one small batch,
// in a file-scope device array instead of hipMalloc'd memory.  Every GPU the library runs on holds its own copy
// (at its own address) until the process ends; host code reaches it only through the per-device pointer that
// scratch_of_device() obtains once with hipGetSymbolAddress.
static __device__ __align__(256) unsigned char scratch[V4R_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int BLOCK = 256;
#ifndef V4R_TRACE_RAYS_PER_THREAD
#define V4R_TRACE_RAYS_PER_THREAD 4
#endif
constexpr int TRACE_RAYS_PER_THREAD = V4R_TRACE_RAYS_PER_THREAD;

// Per-segment queue counters, zeroed once per batch.  Every counter has a 64-byte line of its own: each pool of a trace launch
// ends with one atomic on every one of them, and atomics on one line are served one after the other.
struct alignas(64) CounterLine {
    uint32_t v;
    uint32_t pad[15];
};
struct SegCtr {
    CounterLine given_up;      // rays the first trace launch of the segment left for the second
    CounterLine trace_cursor;  // next pool of the ray queue that a workgroup of the first launch takes
    CounterLine second_cursor;  // trace_cursor of the second launch
    CounterLine hit_count[CLS_COUNT];
};
static_assert(sizeof(SegCtr) == 64 * (3 + CLS_COUNT), "one line per counter");

struct Queues {
    float4 *ray[2][3];          // [segment parity][column]: {o.xyz,d.x} {d.yz,slot,thr.r} {thr.gb,k0,k1}
    uint32_t *hit[CLS_COUNT];   // per shading class: the indices (into the ray queue of the segment) of the rays that hit it
    float4 *hitd;               // {t, prim, u, v} of ray i of the segment being traced
    uint32_t *given_up;         // indices of the rays whose wave gave them up (see traverse); traced by a second launch
    float4 *rad;                // per path slot: radiance of the finished path
    SegCtr *ctr;                // [max_depth + 1]
    uint2 *spill;               // traversal stack overflow, [entry][global thread]
    unsigned long long *stats;  // segments, nodes, prims, paths, shaded, texel fetches, wave node steps, wave leaf steps; [8]: failed V4R_CHECKs
    uint32_t *shape_rays;       // packet-statistics rule of the light pdf: [pass of the batch][primitive], the rays of the segment being
                                // traced whose closest hit is that primitive (Lambertian shapes only), counted up to V4R_PACKET_MANY_RAYS
    uint32_t spill_stride;
    uint32_t spill_depth;       // entries per thread in the spill slab
    uint32_t cap;               // path slots per batch
    uint32_t npix;              // pixels per pass of the batch being rendered (path slot = pass of the batch * npix + pixel)
    float inv_npix;             // 1 / npix (pass_of_slot)
    uint32_t flags_offset;      // shape_rays[flags_offset + i] != 0: counter i has reached V4R_PACKET_MANY_RAYS and is no longer incremented
};

V4R_DEV uint32_t lane_id() { return threadIdx.x & 63u; }

V4R_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct PathState {
    V3 o, d, thr;
    uint32_t slot, k0, k1;
};

V4R_DEV void store_ray(float4 *const col[3], uint32_t i, const PathState &p) {
    col[0][i] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
    col[1][i] = make_float4(p.d.y, p.d.z, __uint_as_float(p.slot), p.thr.x);
    col[2][i] = make_float4(p.thr.y, p.thr.z, __uint_as_float(p.k0), __uint_as_float(p.k1));
}
// The same records written past the caches' keep-list: what a shading launch writes -- the next segment's rays, the radiance of
// ended paths -- is not read again before the launches of the next segment (or the batch's resolve), by which time a cache line
// written now is long gone; written the plain way it would displace the scene tables that every ray of every pipeline re-reads.
V4R_DEV void stream_store(float4 *p, float4 v) { __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(p)); }
V4R_DEV void store_ray_streaming(float4 *const col[3], uint32_t i, const PathState &p) {
    stream_store(&col[0][i], make_float4(p.o.x, p.o.y, p.o.z, p.d.x));
    stream_store(&col[1][i], make_float4(p.d.y, p.d.z, __uint_as_float(p.slot), p.thr.x));
    stream_store(&col[2][i], make_float4(p.thr.y, p.thr.z, __uint_as_float(p.k0), __uint_as_float(p.k1)));
}
V4R_DEV PathState unpack_ray(float4 a, float4 b, float4 c) {
    PathState p;
    p.o = v3(a.x, a.y, a.z);
    p.d = v3(a.w, b.x, b.y);
    p.thr = v3(b.w, c.x, c.y);
    p.slot = __float_as_uint(b.z);
    p.k0 = __float_as_uint(c.z);
    p.k1 = __float_as_uint(c.w);
    return p;
}

// a / b for a < 2^31 and a quotient below 2^23, without the integer division (some 25 vector instructions): the float
// quotient with the reciprocal `inv_b` ~ 1 / b is off by at most one, which two comparisons repair.
V4R_DEV uint32_t div_small(uint32_t a, uint32_t b, float inv_b) {
    uint32_t q = (uint32_t)((float)a * inv_b);
    if ((unsigned long long)q * b > a) --q;
    else if ((unsigned long long)(q + 1u) * b <= a) ++q;
    return q;
}

// Renderer::sample (renderer.rs:94-122) + Camera::get_ray (camera.rs:113-129) for one pixel;
// row 0 is the top of the image (renderer.rs:101 iterates j from height-1 down).
V4R_DEV void camera_ray(const SceneDev &S, uint32_t pixel, uint32_t key, bool deterministic, V3 &o, V3 &d, float inv_width = 0.0f) {
    uint32_t row = inv_width != 0.0f ? div_small(pixel, S.width, inv_width) : pixel / S.width, col = pixel - row * S.width;
    float width = (float)S.width, height = (float)S.height;
    float i = (float)col, j = (float)(S.height - 1u - row);
    float sx, sy;
    if (deterministic || !S.antialias) {
        sx = i / width;
        sy = j / height;
    } else {
        float lo = (i - 0.5f) / width, hi = (i + 0.5f) / width;
        sx = lo + (hi - lo) * draw(key, DIM_PIXEL_X);
        lo = (j - 0.5f) / height;
        hi = (j + 0.5f) / height;
        sy = lo + (hi - lo) * draw(key, DIM_PIXEL_Y);
    }
    const v4r_camera &c = S.cam;
    V3 offset = v3(0, 0, 0);
    // (a pinhole camera -- lens_radius 0, every configured scene but the spheres -- skips the lens sample: its offset is
    // 0 * u + 0 * v, and origin + (+-0) is the origin)
    if (!deterministic && c.lens_radius != 0.0f) {  // random_in_unit_disk, random.rs:18-25
        float r = sqrtf(draw(key, DIM_LENS_R));
        float theta = TWO_PI * draw(key, DIM_LENS_THETA);
        float sn, cs;
        sincosf(theta, &sn, &cs);
        float rdx = r * cs * c.lens_radius, rdy = r * sn * c.lens_radius;
        offset = v3(c.u[0], c.u[1], c.u[2]) * rdx + v3(c.v[0], c.v[1], c.v[2]) * rdy;
    }
    V3 source = v3(c.origin[0], c.origin[1], c.origin[2]) + offset;
    V3 target = v3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]) +
                v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]) * sx + v3(c.vertical[0], c.vertical[1], c.vertical[2]) * sy;
    o = source;
    d = normalize(target - source);
}

// ---- kernels ------------------------------------------------------------------------------------
// Order of the camera rays in queue 0.  Neighbouring queue entries should be rays that do similar
// work, at the first segment and, because the queues keep rays roughly in order, at the later ones
// too: 64 consecutive entries (one wave of k_trace) hold a small tile of pixels times `1 << lp` passes
// of the batch, e.g. 2x2 pixels x 16 passes.  Pixels outside whole tiles (odd image sizes) and passes
// outside whole groups follow in plain order.  The map is a bijection, and since every random
// number is keyed by (pass, pixel) the image does not depend on it.
struct RayOrder {
    uint32_t lp;            // log2 passes per tile
    uint32_t tw_log, th_log;
    uint32_t tiles_x;       // whole tiles per image row
    uint32_t tiled_w, tiled_h;  // extent covered by whole tiles
    uint32_t npix_tiled;    // tiled_w * tiled_h
    uint32_t full_groups;   // groups of (1 << lp) passes in this batch
    float inv_group, inv_tiles_x, inv_width;  // reciprocals of npix << lp, tiles_x and the image width (div_small)
};
V4R_DEV void ray_order(const RayOrder &R, uint32_t width, uint32_t npix, uint32_t i, uint32_t &pass, uint32_t &pixel) {
    const uint32_t group = npix << R.lp;
    uint32_t g = div_small(i, group, R.inv_group);
    if (g >= R.full_groups) {
        uint32_t q = i - R.full_groups * group;
        uint32_t b = q / npix;
        pass = (R.full_groups << R.lp) + b;
        pixel = q - b * npix;
        return;
    }
    uint32_t r = i - g * group, x, y;
    if (r < (R.npix_tiled << R.lp)) {
        uint32_t tile = r >> 6, w = r & 63u, pix_log = 6u - R.lp;
        uint32_t pw = w & ((1u << pix_log) - 1u);
        pass = (g << R.lp) + (w >> pix_log);
        uint32_t ty = div_small(tile, R.tiles_x, R.inv_tiles_x), tx = tile - ty * R.tiles_x;
        x = (tx << R.tw_log) + (pw & ((1u << R.tw_log) - 1u));
        y = (ty << R.th_log) + (pw >> R.tw_log);
    } else {
        uint32_t q = r - (R.npix_tiled << R.lp), n_rest = npix - R.npix_tiled;
        uint32_t b = q / n_rest, qq = q - b * n_rest;
        pass = (g << R.lp) + b;
        uint32_t strip = width - R.tiled_w, n_right = strip * R.tiled_h;
        if (qq < n_right) {  // columns right of the tiles
            y = qq / strip;
            x = R.tiled_w + (qq - y * strip);
        } else {  // rows below the tiles
            qq -= n_right;
            uint32_t yy = qq / width;
            y = R.tiled_h + yy;
            x = qq - yy * width;
        }
    }
    pixel = y * width + x;
}

// slot / npix for a path slot (= pass of the batch * npix + pixel, < 2^31) without the integer division: the float
// quotient is off by at most one, which two comparisons repair.
V4R_DEV uint32_t pass_of_slot(uint32_t slot, uint32_t npix, float inv_npix) { return div_small(slot, npix, inv_npix); }

// Rays queued for segment `seg` >= 1: one ray per hit that the scattering classes of the previous
// segment shaded (lights never scatter, diffuse_light.rs).  Segment 0 traces one camera ray per path slot.
V4R_DEV uint32_t rays_in(const Queues &Q, uint32_t seg) {
    const SegCtr &prev = Q.ctr[seg - 1u];
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < CLS_COUNT; ++k)
        if (k != CLS_LIGHT) n += prev.hit_count[k].v;
    return n;
}
// Position of class `cls` inside the ray queue that the shading kernels of segment `seg` fill: the
// classes are laid out back to back in class order, so every shaded hit knows its slot without an atomic.
V4R_DEV uint32_t class_base(const Queues &Q, uint32_t seg, int cls) {
    const SegCtr &c = Q.ctr[seg];
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < CLS_COUNT; ++k)
        if (k != CLS_LIGHT && k < cls) n += c.hit_count[k].v;
    return n;
}

// Copies the top of the wide tree into the workgroup's LDS array and points the stack at it; the caller
// synchronises the workgroup before the first traversal.
template <class Stack>
V4R_DEV void stage_tree_top(const SceneDev &S, v4f *s_top, Stack &st) {
    for (uint32_t k = threadIdx.x; k < 8u * S.n_top_nodes; k += BLOCK) {
        float4 v = S.nodes4[k];
        s_top[k] = v4f{v.x, v.y, v.z, v.w};
    }
    st.top = (const lds_v4f *)s_top;
    st.n_top = S.n_top_nodes;
}

// BVH traversal + closest hit for the rays of one segment.  A workgroup takes a pool of RAYS_PER_THREAD*256 rays
// at a time (the one at its own grid position first, then pools from a shared cursor: one global
// atomic per pool) and hands it out to its waves in groups of 64 rays; a wave that is done with its group takes
// the next one, so no wave waits for another.  Within a
// wave the traversal alternates, by vote, between stepping through interior nodes and intersecting the
// leaves that lanes have parked.  A hit is ranked within its shading class in LDS; the wave that finishes a pool's
// last group reserves the pool's slots in each class queue with one global atomic per class and
// writes the rays' indices there.  (Letting a lane start its next ray while its neighbours are
// still inside the previous one -- rays handed out by an LDS ticket counter, the next ray kept
// preloaded in registers -- was measured: it removes 12 % of the wave-level node steps on the sphere
// scenes and 46 % on the bunny scene, but its extra votes and low-occupancy set-up code made all but
// the bunny scene 5-20 % slower, so a wave's 64 lanes start their rays together.)
#ifndef V4R_TRACE_MIN_WAVES
#define V4R_TRACE_MIN_WAVES 5  // 94-98 VGPRs unforced; the 8-wide tree's later segments run six waves (80 VGPRs, 28 B of scratch): TREE_8_WIDE_DENSE
#endif
#ifndef V4R_TRACE_WAVES_DENSE
#define V4R_TRACE_WAVES_DENSE 6
#endif
#ifndef V4R_TRACE_MIN_WAVES_PRIMARY
#define V4R_TRACE_MIN_WAVES_PRIMARY 5  // 109-113 VGPRs unforced (4 waves/SIMD); at 96 with 32-40 B of scratch: C4 +1-2 %, others flat
#endif
#ifndef V4R_TRACE_MIN_WAVES_DIRECT
#define V4R_TRACE_MIN_WAVES_DIRECT 5  // scenes with a few media tested before the tree
#endif
#ifndef V4R_TRACE_MIN_WAVES_MEDIA
#define V4R_TRACE_MIN_WAVES_MEDIA 4  // the kernels with the constant-medium code need 133-143 VGPRs unforced (3 waves/SIMD)
#endif
// Arguments of the first segment's launch, which generates the camera rays itself (PRIMARY).
struct PrimaryArgs {
    uint32_t seed_lo, seed_hi, first_pass, pass_stride, n_paths;
    uint32_t npix;          // pixels of the band of rows being rendered (the whole image for v4r_render)
    float inv_npix;         // 1 / npix, rounded down (pass_of_slot)
    uint32_t pixel_offset;  // index of the band's first pixel in the image
    RayOrder order;         // over the band
};
// A segment is traced by up to two launches.  In the first, a wave that is down to `give_up_lanes`
// traversing lanes gives those rays up (their indices go to Q.given_up, through the same pool-level
// ranking as the hits); the second launch (SECOND) traces the given-up rays of the whole
// segment packed next to each other, starting from the closest hit they had found, and never gives up.
// (SECOND is a template parameter because the extra index it carries costs the first launch an occupancy step.)
// TREE: which tree the scene keeps (TREE_4_WIDE, TREE_8_WIDE), or TREE_8_WIDE_DENSE: the 8-wide tree with the launches of
// the later segments held to 80 VGPRs (6 waves/SIMD; six workgroups' LDS fit a CU with the 8-wide tree's short top).  The
// sixth wave covers more waiting than its 28 B of scratch cost: the 1080p bunny scene +3.8 %, the 4K textured-spheres scene
// +2 %, the Cornell box +1 %; every scene on the 8-wide tree without media runs it (V4R_TRACE_WAVES=5|6 forces either).
// (At 40-70 B of scratch, before the node step shrank, only scenes far beyond the caches gained.  Seven waves -- 72 VGPRs,
// 64 B of scratch, 8-deep LDS stacks -- trace 1.6 % faster on the bunny scene and leave the job's rate where it is.)
enum : int { TREE_4_WIDE = 0, TREE_8_WIDE = 1, TREE_8_WIDE_DENSE = 2 };
template <bool COUNT, int RAYS_PER_THREAD, int MEDIA, bool PRIMARY, bool SECOND, int TREE>
__global__ __launch_bounds__(BLOCK, MEDIA == MEDIA_IN_TREE ? V4R_TRACE_MIN_WAVES_MEDIA : (PRIMARY ? V4R_TRACE_MIN_WAVES_PRIMARY : (MEDIA == MEDIA_DIRECT ? V4R_TRACE_MIN_WAVES_DIRECT : (TREE == TREE_8_WIDE_DENSE ? V4R_TRACE_WAVES_DENSE : V4R_TRACE_MIN_WAVES)))) void k_trace(SceneDev S, Queues Q, uint32_t seg, PrimaryArgs A, uint32_t give_up_lanes) {
    constexpr bool second = SECOND;
    constexpr bool WIDE8 = TREE != TREE_4_WIDE;
    __shared__ uint2 lds_stack[lds_depth(WIDE8) * BLOCK];
    __shared__ uint32_t s_cnt[2][CLS_COUNT + 1];  // per destination ([CLS_COUNT]: rays given up), one set per pool in use
    __shared__ uint32_t s_where[RAYS_PER_THREAD * BLOCK];  // as 2 x RAYS_PER_THREAD * BLOCK uint16: per ray of a pool, destination << 10 | rank
    __shared__ uint32_t s_pool[2][2], s_next[2], s_done[2], s_free[2], s_current, s_finished;
    __shared__ v4f s_top[8 * (WIDE8 ? TOP_NODES_8 : TOP_NODES)];
    StackT<lds_depth(WIDE8)> st;
    stage_tree_top(S, s_top, st);  // (the workgroup synchronises once below, before the first traversal)
    st.lds = (lds_uint2 *)lds_stack + threadIdx.x;
    st.stride = BLOCK;
    st.spill = Q.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    st.spill_stride = Q.spill_stride;
    st.sp = 0;
#ifdef V4R_DEBUG_CHECKS
    st.errors = Q.stats + 8;
    st.spill_depth = (int)Q.spill_depth;
#endif
    SegCtr *ctr = Q.ctr + seg;
    uint32_t n_in = second ? ctr->given_up.v : (PRIMARY ? A.n_paths : rays_in(Q, seg));
    if (!V4R_CHECK(Q.stats + 8, CHECK_QUEUE_LENGTH, n_in <= Q.cap)) n_in = 0;
    float4 *const *rq = Q.ray[seg & 1u];
    const uint32_t npix = A.npix;
    // packet-statistics rule of the light pdf: the first levels count the rays that end on each Lambertian shape
    const bool count_shape_rays = S.packet_rule != 0u && seg < V4R_PACKET_COUNTED_LEVELS;
    TraceCounters cnt;
    uint32_t n_traced = 0, n_lights = 0;
    constexpr uint32_t DEST_NONE = 0xFFu;
    // A short queue is spread over as many workgroups as possible: the pools then hold fewer rays per thread
    // (down to one), so the launch lasts about one ray's traversal instead of RAYS_PER_THREAD.
    const uint32_t per_thread = min((uint32_t)RAYS_PER_THREAD, max(1u, (n_in + gridDim.x * BLOCK - 1u) / (gridDim.x * BLOCK)));
    const uint32_t chunk = per_thread * BLOCK;
#ifdef V4R_PHASE_CLOCKS
    unsigned long long clk_fetch = 0, clk_end = 0, clk_chunk = 0, clk_traverse = 0, wave_rays = 0;
    const unsigned long long clk_start = COUNT ? phase_clock() : 0ull;
#endif
    // A workgroup takes the queue in pools of `chunk` rays and hands a pool out to its waves in groups of 64 rays through an LDS
    // word; no wave waits for another at a barrier.  The waves rank a pool's rays per destination with LDS atomics (s_cnt,
    // s_where); the wave that finishes the pool's last group adds the pool's counts to the segment's (one atomic per
    // destination) and writes the queue entries of all its rays, while the others are already at work on the next pool, which
    // the first wave to find the pool used up took from the cursor.  Two pools are in use at a time (the one being handed out
    // and the one whose last groups are still being traced) in two sets of LDS; a set is reused two pools later, after its
    // pool's entries are written.  s_next[set] packs the pool's sequence number, its number of groups and the next group to hand
    // out into one word, so that one LDS atomic tells a wave all three consistently, whichever pool lives in the set by then.
    static_assert(RAYS_PER_THREAD * BLOCK / 64 <= 31 && RAYS_PER_THREAD * BLOCK <= 1024, "a pool's groups fit 5 bits of s_next, a rank 10 bits of s_where");
    constexpr uint32_t WHERE_NONE = 0xFFFFu;  // s_where: destination << 10 | rank among the pool's rays with that destination
    constexpr uint32_t SPIN_LIMIT = 1u << 22;  // (a wave that waited this long for another gives the launch up: the checked build reports it)
    auto pack_next = [](uint32_t seq, uint32_t n_groups) { return (seq << 17) | (n_groups << 12); };
    auto pool_count = [&](uint32_t base) { return base < n_in ? min(chunk, n_in - base) : 0u; };
    if (threadIdx.x == 0) {
        const uint32_t count0 = pool_count(blockIdx.x * chunk);
        s_pool[0][0] = blockIdx.x * chunk;
        s_pool[0][1] = count0;
        s_next[0] = pack_next(0u, (count0 + 63u) / 64u);
        s_next[1] = pack_next(0x7FFFu, 0u);
        s_done[0] = s_done[1] = 0u;
        s_free[0] = 0u;
        s_free[1] = 1u;
        s_current = 0u;
        s_finished = count0 == 0u ? 1u : 0u;
    }
    if (threadIdx.x < 2 * (CLS_COUNT + 1)) (&s_cnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    // one LDS atomic for the wave, its result in every lane (and in a scalar register)
    auto wave_fetch_add = [](uint32_t *word, uint32_t v) {
        uint32_t r = 0u;
        if (lane_id() == 0) r = __hip_atomic_fetch_add(word, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    };
    auto peek = [](const uint32_t *word) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    while (true) {
#ifdef V4R_PHASE_CLOCKS
        unsigned long long tc = COUNT ? phase_clock() : 0ull;
#endif
        if (peek(&s_finished) != 0u) break;
        const uint32_t p = peek(&s_current) & 1u;
        const uint32_t word = wave_fetch_add(&s_next[p], 1u);
        const uint32_t seq = word >> 17, n_groups = (word >> 12) & 31u, g = word & 0xFFFu;
        if (g >= n_groups) {  // the pool in this set is used up
            bool stuck = false;
            if (g == n_groups) {  // ... and this wave is the first to find it so: it opens pool seq + 1 in the other set
                for (uint32_t spins = 0; peek(&s_free[p ^ 1u]) == 0u; ++spins) {  // (pool seq - 1 still has a group out)
                    if (spins >= SPIN_LIMIT) {
                        stuck = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!stuck && lane_id() == 0) {
                    const uint32_t nb = gridDim.x * chunk + atomicAdd(second ? &ctr->second_cursor.v : &ctr->trace_cursor.v, chunk);
                    const uint32_t nc = pool_count(nb);
                    if (nc == 0u) {
                        __hip_atomic_store(&s_finished, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        s_pool[p ^ 1u][0] = nb;
                        s_pool[p ^ 1u][1] = nc;
                        s_free[p ^ 1u] = 0u;
                        __hip_atomic_store(&s_next[p ^ 1u], pack_next((seq + 1u) & 0x7FFFu, (nc + 63u) / 64u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&s_current, (seq + 1u) & 0x7FFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            } else {  // another wave is opening the next pool
                for (uint32_t spins = 0; peek(&s_current) == seq && peek(&s_finished) == 0u; ++spins) {
                    if (spins >= SPIN_LIMIT) {
                        stuck = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (!V4R_CHECK(Q.stats + 8, CHECK_POOL_WAIT, !stuck) || stuck) break;
#ifdef V4R_PHASE_CLOCKS
            if (COUNT) clk_chunk += phase_clock() - tc;
#endif
            continue;
        }
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pool[p][0]), count = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pool[p][1]);
#ifdef V4R_PHASE_CLOCKS
        if (COUNT) clk_chunk += phase_clock() - tc;
#endif
#ifdef V4R_PHASE_CLOCKS
        const unsigned long long ta = COUNT ? phase_clock() : 0ull;
        wave_rays++;
#endif
        const uint32_t in_pool = g * 64u + lane_id();
        uint32_t pos = base + in_pool;
        bool active = in_pool < count;  // (base + count <= n_in)
        uint32_t i = active ? pos : n_in - 1u;  // index of the ray in the queue of this segment
        if (second) i = Q.given_up[i];
        const uint32_t ii = i;
        V3 o, d;
        uint32_t pass_in_batch = 0, pixel = 0;  // PRIMARY only
        uint32_t slot = 0;                       // later segments, while rays are counted per shape
        if (PRIMARY) {
            // Renderer::sample + Camera::get_ray for queue entry ii (see ray_order); the ray record is
            // written further down, and only if the ray hits something: a miss ends the path here
            ray_order(A.order, S.width, npix, ii, pass_in_batch, pixel);  // pixel: within the band
            uint32_t k0, k1;
            path_key(A.seed_lo, A.seed_hi, A.first_pass + pass_in_batch * A.pass_stride, pixel + A.pixel_offset, k0, k1);
            camera_ray(S, pixel + A.pixel_offset, segment_key(k0, k1, 0u), false, o, d, A.order.inv_width);
        } else {
            float4 a = rq[0][ii];
            o = v3(a.x, a.y, a.z);
            float2 b = *reinterpret_cast<const float2 *>(&rq[1][ii]);
            d = v3(a.w, b.x, b.y);
        }
        Best best;
        best.t = V4R_INF;
        best.prim = NO_PRIM;
        best.u = best.v = 0.0f;
        if (second) {  // the closest hit found before the ray was given up
            float4 h = Q.hitd[ii];
            best.t = h.x;
            best.prim = __float_as_uint(h.y);
            best.u = h.z;
            best.v = h.w;
        }
        // the random stream of the segment, needed only if the ray meets a constant medium
        auto key = [&]() {
            uint32_t k0, k1;
            if (PRIMARY) {
                path_key(A.seed_lo, A.seed_hi, A.first_pass + pass_in_batch * A.pass_stride, pixel + A.pixel_offset, k0, k1);
            } else {
                float4 c = rq[2][ii];
                k0 = __float_as_uint(c.z);
                k1 = __float_as_uint(c.w);
            }
            return segment_key(k0, k1, seg);
        };
#ifdef V4R_PHASE_CLOCKS
        const unsigned long long tb = COUNT ? phase_clock() : 0ull;
        if (COUNT) clk_fetch += tb - ta;
#endif
        bool gave_up = traverse<COUNT, MEDIA, WIDE8>(S, o, d, active, st, best, cnt, key, second ? 0u : give_up_lanes, second);
#ifdef V4R_PHASE_CLOCKS
        const unsigned long long td = COUNT ? phase_clock() : 0ull;
        if (COUNT) clk_traverse += td - tb;
#endif
        // (the ray's index as a value the compiler has not seen: the queue addresses below are then computed here, not
        // before the traversal and carried through it in registers the traversal needs)
        asm volatile("" : "+v"(i));
        // what becomes of the ray: the queue it goes to (a shading class, or CLS_COUNT = given up), or none (its path ended)
        uint32_t dest = DEST_NONE;
        // Everything the end of a ray may need goes out in one burst of loads, before any of it is looked at: the
        // class word of the primitive hit and the rest of the ray's record (path slot, throughput).  A wave holds
        // rays that end in every possible way, so it would otherwise wait for one trip to memory per way, one after the other.
        uint32_t info = 0u;
        float4 bb = make_float4(0, 0, 0, 0), cc = bb;
        if (active && !gave_up) {
            if (best.prim != NO_PRIM) info = S.prim_info[best.prim];
            if (!PRIMARY) bb = rq[1][i], cc = rq[2][i];
        }
        if (active && gave_up) {
            Q.hitd[i] = make_float4(best.t, __uint_as_float(best.prim), best.u, best.v);
            dest = (uint32_t)CLS_COUNT;
        } else if (active) {
            if (COUNT) n_traced++;
            if (best.prim != NO_PRIM) {
                // Packet-statistics rule of the light pdf: count the pass's rays per Lambertian shape they end on.  A
                // counter that reaches V4R_PACKET_MANY_RAYS raises its "full" flag and is left alone from then on,
                // at this level and the deeper ones of the batch (walls would otherwise serialise the chip on a
                // handful of addresses -- every launch anew, until each compute unit has seen the flag; from that
                // count on the rule treats every packet as full).  The flag is read and written with plain
                // accesses: a stale 0 only costs one more increment.  Shapes full in every pass (build()) skip all of it.
                const uint32_t cls = info >> 24;
                uint32_t *n_rays = Q.shape_rays;
                uint32_t full = 1u;
                if (count_shape_rays && !(info & PRIM_INFO_ALWAYS_FULL)) {  // (a shape full in every pass is not counted at all)
                    if (!PRIMARY) slot = __float_as_uint(bb.z);
                    uint32_t pass = PRIMARY ? pass_in_batch : pass_of_slot(slot, npix, A.inv_npix);
                    if (V4R_CHECK(Q.stats + 8, CHECK_RAY_COUNTER, (size_t)(pass + 1u) * S.n_prims <= Q.flags_offset)) {
                        n_rays += (size_t)pass * S.n_prims + best.prim;
                        // (a small shape's flag is never raised -- see below -- so it is not read: one trip to memory less
                        // at the end of every ray that ends on a mesh triangle)
                        full = (info & PRIM_INFO_SMALL) ? 0u : n_rays[Q.flags_offset];
                    }
                }
                if (info & PRIM_INFO_PLAIN_LIGHT) {
                    // DiffuseLight (material/diffuse_light.rs:19-22) of one colour on an untransformed rectangle: the
                    // path ends here with throughput x emitted; nothing is queued for a shading launch.  The face test
                    // is HitRecord::face_normal (hittable/mod.rs:146-154) against the rectangle's axis normal.
                    const uint4 colour = S.prim_shade[2 * (size_t)best.prim];
                    const float along = comp(d, (int)((info >> PRIM_INFO_AXIS_SHIFT) & 3u));
                    const bool front = ((info & PRIM_INFO_POSITIVE) ? along : -along) < 0.0f;
                    V3 e = front ? v3(__uint_as_float(colour.y), __uint_as_float(colour.z), __uint_as_float(colour.w)) : v3(0, 0, 0);
                    if (PRIMARY) {
                        Q.rad[pass_in_batch * npix + pixel] = make_float4(1.0f * e.x, 1.0f * e.y, 1.0f * e.z, 0.0f);
                    } else {
                        Q.rad[__float_as_uint(bb.z)] = make_float4(bb.w * e.x, cc.x * e.y, cc.y * e.z, 0.0f);
                    }
                    if (COUNT) n_lights++;
                } else {
                    Q.hitd[i] = make_float4(best.t, __uint_as_float(best.prim), best.u, best.v);
                    dest = cls;
                    if (count_shape_rays && (cls == (uint32_t)CLS_LAMBERT || cls == (uint32_t)CLS_LAMBERT_NOISE) && full == 0u) {
                        // A shape too small to expect many rays (a mesh triangle) is counted without waiting for
                        // the counter's value: it will not fill, and a few percent of the rays end on such shapes,
                        // so waiting would hold nearly every wave of the later segments for a trip to memory.
                        // (The size class is a guess from the shape's share of the scene's area; a wrong guess
                        // costs time -- contended or waited-for increments -- never a wrong count.)
                        if (info & PRIM_INFO_SMALL) atomicAdd(n_rays, 1u);
                        else if (atomicAdd(n_rays, 1u) + 1u >= V4R_PACKET_MANY_RAYS) n_rays[Q.flags_offset] = 1u;
                    }
                    if (PRIMARY) {
                        PathState p;
                        p.o = o;
                        p.d = d;
                        p.thr = v3(1, 1, 1);
                        p.slot = pass_in_batch * npix + pixel;
                        path_key(A.seed_lo, A.seed_hi, A.first_pass + pass_in_batch * A.pass_stride, pixel + A.pixel_offset, p.k0, p.k1);
                        store_ray(rq, i, p);
                    }
                }
            } else if (PRIMARY) {  // renderer.rs:184: a miss takes the background colour
                Q.rad[pass_in_batch * npix + pixel] = make_float4(S.bg[0], S.bg[1], S.bg[2], 0.0f);
            } else {
                Q.rad[__float_as_uint(bb.z)] = make_float4(bb.w * S.bg[0], cc.x * S.bg[1], cc.y * S.bg[2], 0.0f);
            }
        }
        // Rank of the ray among the pool's rays with the same destination: the lanes of the wave that share a
        // destination are found by ballot, one of them reserves their ranks with a single LDS atomic, and each
        // takes the rank of its position among them.
        uint32_t w = WHERE_NONE;
        for (unsigned long long todo = __ballot(dest != DEST_NONE); todo != 0ull;) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t which = (uint32_t)__shfl((int)dest, leader, 64);
            const unsigned long long same = __ballot(dest == which);
            uint32_t first = 0u;
            if ((int)lane_id() == leader) first = atomicAdd(&s_cnt[p][which], (uint32_t)__popcll(same));
            first = (uint32_t)__shfl((int)first, leader, 64);
            if (dest == which) w = (which << 10) | (first + (uint32_t)__popcll(same & ((1ull << lane_id()) - 1ull)));
            todo &= ~same;
        }
        reinterpret_cast<uint16_t *>(s_where)[p * (RAYS_PER_THREAD * BLOCK) + in_pool] = (uint16_t)w;
#ifdef V4R_PHASE_CLOCKS
        if (COUNT) clk_end += phase_clock() - td;
        tc = COUNT ? phase_clock() : 0ull;
#endif
        // (the atomic orders this wave's ranks and s_where entries before the count that another wave may act on)
        if (wave_fetch_add(&s_done[p], 1u) + 1u == n_groups) {
            // the pool's last group: its counts go to the segment's, its rays to their queues
            uint32_t first_of = 0u;
            if (lane_id() <= (uint32_t)CLS_COUNT) {
                const uint32_t n = s_cnt[p][lane_id()];
                uint32_t *total = lane_id() < (uint32_t)CLS_COUNT ? &ctr->hit_count[lane_id()].v : &ctr->given_up.v;
                first_of = n ? atomicAdd(total, n) : 0u;
                s_cnt[p][lane_id()] = 0u;
            }
            for (uint32_t it = 0; it < n_groups; ++it) {
                const uint32_t e = it * 64u + lane_id();
                const uint32_t w = e < count ? (uint32_t)reinterpret_cast<const uint16_t *>(s_where)[p * (RAYS_PER_THREAD * BLOCK) + e] : WHERE_NONE;
                const uint32_t k = (w >> 10) & 7u;
                const uint32_t idx = (uint32_t)__shfl((int)first_of, (int)k, 64) + (w & 0x3FFu);
                if (w != WHERE_NONE) {
                    const uint32_t i = second ? Q.given_up[base + e] : base + e;
                    if (V4R_CHECK(Q.stats + 8, CHECK_QUEUE_INDEX, idx < Q.cap && i < Q.cap)) (k < (uint32_t)CLS_COUNT ? Q.hit[k] : Q.given_up)[idx] = i;
                }
            }
            if (lane_id() == 0) {
                s_done[p] = 0u;
                __hip_atomic_store(&s_free[p], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#ifdef V4R_PHASE_CLOCKS
        if (COUNT) clk_chunk += phase_clock() - tc;
#endif
    }
#ifdef V4R_PHASE_CLOCKS
    if (COUNT && lane_id() == 0) {  // per kind of launch (camera rays / later segments / second launches): wave cycles by part
        unsigned long long *out = Q.stats + 16 + 8 * (SECOND ? 2 : (PRIMARY ? 0 : 1));
        atomicAdd(out + 0, clk_fetch);
        atomicAdd(out + 1, cnt.clk_node);
        atomicAdd(out + 2, cnt.clk_leaf);
        atomicAdd(out + 3, clk_traverse);
        atomicAdd(out + 4, clk_end);
        atomicAdd(out + 5, clk_chunk);
        atomicAdd(out + 6, phase_clock() - clk_start);
        atomicAdd(out + 7, wave_rays);
    }
#endif
    if (COUNT) {
        unsigned long long s = wave_sum(n_traced), n = wave_sum(cnt.nodes), q = wave_sum(cnt.prims), l = wave_sum(n_lights);
        if (lane_id() == 0) {
            atomicAdd(&Q.stats[0], s);
            atomicAdd(&Q.stats[1], n);
            atomicAdd(&Q.stats[2], q);
            atomicAdd(&Q.stats[4], l);  // lights shaded here count as shaded hits
            atomicAdd(&Q.stats[9], l);
            atomicAdd(&Q.stats[6], (unsigned long long)cnt.wave_steps);
            atomicAdd(&Q.stats[7], (unsigned long long)cnt.wave_leaves);
        }
        if (PRIMARY && !SECOND && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&Q.stats[3], (unsigned long long)A.n_paths);
    }
}

// Material::emitted + Material::scatter + the pdf branch of ray_color (renderer.rs:218-263) for one hit of
// material class CLS, then `colors += coef * child` carried forward as the path throughput
// (renderer.rs:299-318).  `dcol` is the hit as k_trace leaves it: distance, primitive, triangle barycentrics.
// Returns true with the path's next ray in `q`, or false with the radiance the path ends with in `rad`
// (it met a light, or its depth budget is spent).
// NOISE: some texture of the class's materials is (or contains) a Perlin noise texture.
template <int CLS, bool NOISE>
V4R_DEV bool shade_hit(const SceneDev &S, const Queues &Q, const PathState &p, float4 dcol, uint32_t seg, uint32_t &texel_fetches, PathState &q, float4 &rad) {
    const float t = dcol.x;
    const uint32_t prim = __float_as_uint(dcol.y);
    // the material's parameters (metal, dielectric) or the record of its texture (the textured kinds)
    const uint4 rec0 = S.prim_shade[2 * (size_t)prim], rec1 = S.prim_shade[2 * (size_t)prim + 1];
    const uint32_t key = segment_key(p.k0, p.k1, seg);
    // packet-statistics rule of the light pdf: the rays of the path's pass that ended on this shape at this level (k_trace
    // counted them); fetched first, beside the records above
    uint32_t n_rays = 0u;
    if ((CLS == CLS_LAMBERT || CLS == CLS_LAMBERT_NOISE) && S.packet_rule) {
        const uint32_t *counter = Q.shape_rays + (size_t)pass_of_slot(p.slot, Q.npix, Q.inv_npix) * S.n_prims + prim;
        // a shape known to fill up in every pass is not counted; a counter that filled up at a shallower level was left
        // standing (k_restart_counters)
        n_rays = (rec1.w & SHADE_ALWAYS_FULL) ? V4R_PACKET_MANY_RAYS : counter[0];
    }
    if (CLS == CLS_LIGHT) {  // material/diffuse_light.rs:19-22; scatter() is the default None
        HitD h = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, false);
        V3 e = texture_value<NOISE>(
            S, rec0, rec1, h.p,
            [&](float &u, float &v) {
                HitD hu = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, true);
                u = hu.u;
                v = hu.v;
            },
            texel_fetches);
        if (!h.front) e = v3(0, 0, 0);
        rad = make_float4(p.thr.x * e.x, p.thr.y * e.y, p.thr.z * e.z, 0.0f);
        return false;
    }

    V3 coef, dir;
    HitD h;
    if (CLS == CLS_LAMBERT || CLS == CLS_LAMBERT_NOISE) {  // material/lambertian.rs:36-46
        h = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, false);
        V3 attenuation = texture_value<NOISE>(
            S, rec0, rec1, h.p,
            [&](float &u, float &v) {
                HitD hu = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, true);
                u = hu.u;
                v = hu.v;
            },
            texel_fetches);
        V3 d_cos = cosine_generate(h.normal, draw(key, DIM_SCATTER + 4), draw(key, DIM_SCATTER + 5));
        if (S.n_lights != 0) {  // renderer.rs:234-244, pdf/mixture.rs:22-32, hittables/group.rs:91-95
            bool pick_light = draw(key, DIM_SCATTER + 0) < 0.5f;
            uint32_t li = (uint32_t)(((unsigned long long)draw_u32(key, DIM_SCATTER + 1) * S.n_lights) >> 32);
            V3 d_light = light_generate(S, S.lights[li], h.p, draw(key, DIM_SCATTER + 2), draw(key, DIM_SCATTER + 3));
            dir = pick_light ? d_light : d_cos;
            float pdf_v = cosine_value(h.normal, dir);
            float mix_v = 0.5f * lights_value(S, h.p, h.normal, dir, li, n_rays, key) + 0.5f * pdf_v;
            coef = attenuation * pdf_v / mix_v;
        } else {  // renderer.rs:245-255
            dir = d_cos;
            coef = attenuation;
        }
    } else if (CLS == CLS_METAL) {  // material/metal.rs:31-46, random_in_unit_sphere random.rs:53-65
        h = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, false);
        V3 reflected = reflect(p.d, h.normal);
        float fuzz = __uint_as_float(rec1.y);
        V3 ball = v3(0, 0, 0);
        if (fuzz != 0.0f) {
            float theta = TWO_PI * draw(key, DIM_SCATTER + 0);
            float cos_phi = -1.0f + 2.0f * draw(key, DIM_SCATTER + 1);
            float r = powf(draw(key, DIM_SCATTER + 2), 1.0f / 3.0f);
            float sin_phi = sqrtf(1.0f - cos_phi * cos_phi);
            float sn, cs;
            sincosf(theta, &sn, &cs);
            ball = v3(r * sin_phi * cs, r * sin_phi * sn, r * cos_phi);
        }
        dir = normalize(reflected + ball * fuzz);
        coef = v3(__uint_as_float(rec0.z), __uint_as_float(rec0.w), __uint_as_float(rec1.x));
    } else if (CLS == CLS_DIELECTRIC) {  // material/dielectric.rs:33-60
        h = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, false);
        float ir = __uint_as_float(rec1.z);
        float ratio = h.front ? 1.0f / ir : ir;
        float cos_theta = -dot(p.d, h.normal);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > 1.0f;
        cannot_refract = cannot_refract || reflectance(cos_theta, ratio) > draw(key, DIM_SCATTER + 0);
        dir = cannot_refract ? reflect(p.d, h.normal) : refract(p.d, h.normal, ratio);
        coef = v3(1, 1, 1);
    } else {  // CLS_ISOTROPIC: material/isotropic.rs:26-43, random_on_unit_sphere random.rs:41-51
        h = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, false);
        float z = -1.0f + 2.0f * draw(key, DIM_SCATTER + 0);
        float theta = TWO_PI * draw(key, DIM_SCATTER + 1);
        float xy = sqrtf(1.0f - z * z);
        float sn, cs;
        sincosf(theta, &sn, &cs);
        dir = v3(cs * xy, sn * xy, z);
        coef = texture_value<NOISE>(
            S, rec0, rec1, h.p,
            [&](float &u, float &v) {
                HitD hu = reconstruct(S, prim, p.o, p.d, t, dcol.z, dcol.w, true);
                u = hu.u;
                v = hu.v;
            },
            texel_fetches);
    }
    q.o = h.p;
    q.d = dir;
    q.thr = p.thr * coef;
    q.slot = p.slot;
    q.k0 = p.k0;
    q.k1 = p.k1;
    if (seg + 1u >= S.max_depth) {  // renderer.rs:128-130: the depth budget is spent, the child returns `environment`
        rad = make_float4(q.thr.x * S.env[0], q.thr.y * S.env[1], q.thr.z * S.env[2], 0.0f);
        return false;
    }
    return true;
}

// Shading of the hits of one material class of one segment.
#ifndef V4R_SHADE_MIN_WAVES_NOISE
#define V4R_SHADE_MIN_WAVES_NOISE 4  // the Lambertian kernels with the Perlin code need 129 VGPRs unforced
#endif
template <int CLS, bool COUNT, bool NOISE>
__global__ __launch_bounds__(BLOCK, NOISE ? V4R_SHADE_MIN_WAVES_NOISE : 1) void k_shade(SceneDev S, Queues Q, uint32_t seg) {
    SegCtr *ctr = Q.ctr + seg;
    uint32_t n_in = ctr->hit_count[CLS].v;
    if (!V4R_CHECK(Q.stats + 8, CHECK_QUEUE_LENGTH, n_in <= Q.cap)) n_in = 0;
    const uint32_t *hq = Q.hit[CLS];
    float4 *const *rq = Q.ray[seg & 1u];  // the rays k_trace just traced; the survivors go to the other queue
    float4 *const *out = Q.ray[(seg + 1u) & 1u];
    const uint32_t out_base = class_base(Q, seg, CLS);
    uint32_t n_shaded = 0, texel_fetches = 0;
    // shading cost is uniform per hit: waves walk the class queue in fixed 64-entry strides
    const uint32_t wave_stride = gridDim.x * (BLOCK / 64u) * 64u;
    for (uint32_t base = (blockIdx.x * (BLOCK / 64u) + (threadIdx.x >> 6)) * 64u; base < n_in; base += wave_stride) {
        uint32_t i = base + lane_id();
        bool valid = i < n_in;
        uint32_t ii = valid ? i : n_in - 1u;
        uint32_t src = hq[ii];
        if (!V4R_CHECK(Q.stats + 8, CHECK_QUEUE_INDEX, src < Q.cap)) src = 0;
        PathState p = unpack_ray(rq[0][src], rq[1][src], rq[2][src]);
        float4 dcol = Q.hitd[src];
        if (valid) n_shaded++;
        PathState q;
        float4 rad;
        uint32_t fetched = 0;
        bool goes_on = shade_hit<CLS, NOISE>(S, Q, p, dcol, seg, fetched, q, rad);
        if (!valid) continue;  // a lane past the end of the queue shaded a copy of the last hit
        texel_fetches += fetched;
        if (!V4R_CHECK(Q.stats + 8, CHECK_PATH_SLOT, p.slot < Q.cap) || !V4R_CHECK(Q.stats + 8, CHECK_NEXT_RAY, out_base + i < Q.cap)) continue;
        if (goes_on) store_ray_streaming(out, out_base + i, q);  // hit i of this class -> ray out_base + i of the next segment
        else stream_store(&Q.rad[p.slot], rad);
    }
    if (COUNT) {
        unsigned long long s = wave_sum(n_shaded), x = wave_sum(texel_fetches);
        if (lane_id() == 0) {
            atomicAdd(&Q.stats[4], s);
            atomicAdd(&Q.stats[5], x);
        }
    }
}

// The tail of a batch.  After a dozen segments a fraction of a percent of the paths is still alive, but the
// depth budget (renderer.rs:128) may run to 50: launching the trace and shading kernels for each of those
// segments costs more in launches than in work.  From segment `seg0` on, one launch finishes the paths: a lane takes
// a ray of the queue and follows its path -- closest hit, shading, next ray -- to its end, then takes another.  The hit and the
// shading are the functions the per-segment kernels call, with the same random streams, so the image does
// not depend on where the tail starts.
template <int CLS, bool NOISE>
V4R_DEV void tail_shade(const SceneDev &S, const Queues &Q, uint32_t cls, float4 dcol, uint32_t seg, PathState &p, bool &alive, uint32_t &n_shaded,
                        uint32_t &texel_fetches) {
    if (__ballot(cls == (uint32_t)CLS) == 0ull) return;
    if (cls == (uint32_t)CLS) {
        PathState q;
        float4 rad;
        n_shaded++;
        if (shade_hit<CLS, NOISE>(S, Q, p, dcol, seg, texel_fetches, q, rad)) {
            p = q;
        } else {
            Q.rad[p.slot] = rad;
            alive = false;
        }
    }
}
#ifndef V4R_TAIL_MIN_WAVES_NOISE
#define V4R_TAIL_MIN_WAVES_NOISE 3  // the Perlin code alone needs ~150 VGPRs; unforced the kernel has 172-196 (2 waves/SIMD)
#endif
template <bool COUNT, int MEDIA, bool NOISE, bool WIDE8>
__global__ __launch_bounds__(BLOCK, NOISE ? V4R_TAIL_MIN_WAVES_NOISE : 1) void k_tail(SceneDev S, Queues Q, uint32_t seg0) {
    __shared__ uint2 lds_stack[lds_depth(WIDE8) * BLOCK];
    __shared__ v4f s_top[8 * (WIDE8 ? TOP_NODES_8 : TOP_NODES)];
    StackT<lds_depth(WIDE8)> st;
    stage_tree_top(S, s_top, st);
    __syncthreads();
    st.lds = (lds_uint2 *)lds_stack + threadIdx.x;
    st.stride = BLOCK;
    st.spill = Q.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    st.spill_stride = Q.spill_stride;
    st.sp = 0;
#ifdef V4R_DEBUG_CHECKS
    st.errors = Q.stats + 8;
    st.spill_depth = (int)Q.spill_depth;
#endif
    const uint32_t n_in = rays_in(Q, seg0);
    float4 *const *rq = Q.ray[seg0 & 1u];
    TraceCounters cnt;
    uint32_t n_traced = 0, n_shaded = 0, texel_fetches = 0;
    // A wave keeps its lanes busy: whenever paths have ended, the free lanes take the next rays of the queue
    // (one atomic per wave and refill), so a wave is not held by its longest path while the queue has rays.
    uint32_t *cursor = &Q.ctr[seg0].trace_cursor.v;  // no k_trace launch uses the cursor of this segment
    bool alive = false, queue_empty = false;
    PathState p;
    p.o = p.d = p.thr = v3(0, 0, 0);
    p.slot = p.k0 = p.k1 = 0u;
    uint32_t seg = seg0;
    while (true) {
        const unsigned long long free_lanes = __ballot(!alive);
        if (!queue_empty && free_lanes != 0ull) {
            const uint32_t n_free = (uint32_t)__popcll(free_lanes);
            uint32_t first = 0;
            if (lane_id() == 0) first = atomicAdd(cursor, n_free);
            first = __shfl(first, 0, 64);
            const uint32_t i = first + (uint32_t)__popcll(free_lanes & ((1ull << lane_id()) - 1ull));
            if (!alive && i < n_in) {
                p = unpack_ray(rq[0][i], rq[1][i], rq[2][i]);
                seg = seg0;
                alive = true;
            }
            queue_empty = first + n_free >= n_in;
        }
        if (__ballot(alive) == 0ull) break;  // every path ends at the latest when shade_hit sees the last segment of the depth budget
        Best best;
        best.t = V4R_INF;
        best.prim = NO_PRIM;
        best.u = best.v = 0.0f;
        auto key = [&]() { return segment_key(p.k0, p.k1, seg); };
        traverse<COUNT, MEDIA, WIDE8>(S, p.o, p.d, alive, st, best, cnt, key, 0u);
        if (alive) {
            if (COUNT) n_traced++;
            if (best.prim == NO_PRIM) {  // renderer.rs:184: a miss takes the background colour
                Q.rad[p.slot] = make_float4(p.thr.x * S.bg[0], p.thr.y * S.bg[1], p.thr.z * S.bg[2], 0.0f);
                alive = false;
            }
        }
        const uint32_t cls = alive ? S.prim_info[best.prim] >> 24 : 0xFFu;
        const float4 dcol = make_float4(best.t, __uint_as_float(best.prim), best.u, best.v);
        tail_shade<CLS_LIGHT, NOISE>(S, Q, cls, dcol, seg, p, alive, n_shaded, texel_fetches);
        tail_shade<CLS_LAMBERT, false>(S, Q, cls, dcol, seg, p, alive, n_shaded, texel_fetches);
        tail_shade<CLS_LAMBERT_NOISE, NOISE>(S, Q, cls, dcol, seg, p, alive, n_shaded, texel_fetches);
        tail_shade<CLS_METAL, false>(S, Q, cls, dcol, seg, p, alive, n_shaded, texel_fetches);
        tail_shade<CLS_DIELECTRIC, false>(S, Q, cls, dcol, seg, p, alive, n_shaded, texel_fetches);
        tail_shade<CLS_ISOTROPIC, NOISE>(S, Q, cls, dcol, seg, p, alive, n_shaded, texel_fetches);
        ++seg;
    }
    if (COUNT) {
        unsigned long long s = wave_sum(n_traced), n = wave_sum(cnt.nodes), q = wave_sum(cnt.prims), h = wave_sum(n_shaded), x = wave_sum(texel_fetches);
        if (lane_id() == 0) {
            atomicAdd(&Q.stats[0], s);
            atomicAdd(&Q.stats[1], n);
            atomicAdd(&Q.stats[2], q);
            atomicAdd(&Q.stats[4], h);
            atomicAdd(&Q.stats[5], x);
            atomicAdd(&Q.stats[6], (unsigned long long)cnt.wave_steps);
            atomicAdd(&Q.stats[7], (unsigned long long)cnt.wave_leaves);
        }
    }
}

// Between two counted levels of a batch: the ray counters go back to zero, except those that have filled up (their shapes
// count as full at every deeper level, and the shading reads that from the standing count).
__global__ __launch_bounds__(BLOCK) void k_restart_counters(uint32_t *counters, size_t n) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride)
        if (counters[i] < V4R_PACKET_MANY_RAYS) counters[i] = 0u;
}

// max_depth = 0 (renderer.rs:128-130): ray_color returns `environment` before it traces anything.
__global__ __launch_bounds__(BLOCK) void k_fill_rad(Queues Q, uint32_t n, float r, float g, float b) {
    uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) Q.rad[i] = make_float4(r, g, b, 0.0f);
}

// RenderResult::add (renderer.rs:355-368) for a whole batch: accum[pixel] += sum over the batch's
// passes, in ascending pass order.
__global__ __launch_bounds__(BLOCK) void k_resolve(Queues Q, float *accum, uint32_t npix, uint32_t n_batch_passes) {
    uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t pix = blockIdx.x * BLOCK + threadIdx.x; pix < npix; pix += stride) {
        float r = accum[3 * (size_t)pix], g = accum[3 * (size_t)pix + 1], b = accum[3 * (size_t)pix + 2];
        for (uint32_t k = 0; k < n_batch_passes; ++k) {
            float4 v = Q.rad[(size_t)k * npix + pix];
            r += v.x;
            g += v.y;
            b += v.z;
        }
        accum[3 * (size_t)pix] = r;
        accum[3 * (size_t)pix + 1] = g;
        accum[3 * (size_t)pix + 2] = b;
    }
}

// RenderResult::get_raw, renderer.rs:378-402
__global__ __launch_bounds__(BLOCK) void k_tonemap(const float *accum, size_t n, float scale1, uint8_t *out) {
    size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        float v = sqrtf(accum[i] * scale1) * 256.0f;
        uint8_t o;
        if (v != v) o = 0;
        else o = (uint8_t)fminf(fmaxf(v, 0.5f), 255.5f);
        out[i] = o;
    }
}

// Closest hit of explicit rays (or of the deterministic primary rays when origin == nullptr).
// Ray i draws the free path of a constant medium from the stream of path (seed, pass 0, pixel i);
// the deterministic primary rays use key 0.
// COUNT: the traversal work is added to Q.stats ([1] node steps, [2] primitive tests, [6] wave steps, [7] wave leaf steps).
template <int MEDIA, bool WIDE8, bool COUNT = false>
__global__ __launch_bounds__(BLOCK) void k_trace_rays(SceneDev S, Queues Q, uint32_t n, const float *origin, const float *dir,
                                                      uint32_t seed_lo, uint32_t seed_hi, uint32_t *prim_out, float *t_out) {
    __shared__ uint2 lds_stack[lds_depth(WIDE8) * BLOCK];
    __shared__ v4f s_top[8 * (WIDE8 ? TOP_NODES_8 : TOP_NODES)];
    StackT<lds_depth(WIDE8)> st;
    stage_tree_top(S, s_top, st);
    __syncthreads();
    st.lds = (lds_uint2 *)lds_stack + threadIdx.x;
    st.stride = BLOCK;
    st.spill = Q.spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    st.spill_stride = Q.spill_stride;
    st.sp = 0;
#ifdef V4R_DEBUG_CHECKS
    st.errors = Q.stats + 8;
    st.spill_depth = (int)Q.spill_depth;
#endif
    uint32_t stride = gridDim.x * BLOCK;
    TraceCounters cnt;
    // the loop bound is wave-uniform: traverse() votes across the wave
    for (uint32_t base = blockIdx.x * BLOCK + (threadIdx.x & ~63u); base < n; base += stride) {
        uint32_t i = base + lane_id();
        bool valid = i < n;
        V3 o = v3(0, 0, 0), d = v3(1, 0, 0);
        if (valid) {
            if (origin) {
                o = v3(origin[3 * (size_t)i], origin[3 * (size_t)i + 1], origin[3 * (size_t)i + 2]);
                d = v3(dir[3 * (size_t)i], dir[3 * (size_t)i + 1], dir[3 * (size_t)i + 2]);
            } else {
                camera_ray(S, i, 0u, true, o, d);
            }
        }
        Best best;
        auto key = [&]() {
            if (!origin) return 0u;
            uint32_t k0, k1;
            path_key(seed_lo, seed_hi, 0u, i, k0, k1);
            return segment_key(k0, k1, 0u);
        };
        best.t = V4R_INF;
        best.prim = NO_PRIM;
        best.u = best.v = 0.0f;
        traverse<COUNT, MEDIA, WIDE8>(S, o, d, valid, st, best, cnt, key, 0u);
        if (valid) {
            prim_out[i] = best.prim;
            t_out[i] = best.t;
        }
    }
    if (COUNT) {
        unsigned long long n = wave_sum(cnt.nodes), q = wave_sum(cnt.prims);
        if (lane_id() == 0) {
            atomicAdd(&Q.stats[1], n);
            atomicAdd(&Q.stats[2], q);
            atomicAdd(&Q.stats[6], (unsigned long long)cnt.wave_steps);
            atomicAdd(&Q.stats[7], (unsigned long long)cnt.wave_leaves);
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------
thread_local std::string g_error;

int set_error(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return set_error(V4R_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

}  // namespace

// One batch pipeline: a stream with its own queues.  Consecutive batches go to different pipelines so
// that the long, nearly empty tail of one batch (a few paths bouncing up to max_depth times) overlaps
// the wide first segments of the next.
struct Pipe {
    hipStream_t stream = nullptr;
    Queues Q{};
    hipEvent_t done = nullptr;  // recorded after the pipeline's latest k_resolve
    bool used = false;
    std::vector<void *> allocations;  // the queues of Q, sized for Q.cap path slots
    size_t count_passes = 0;          // passes Q.shape_rays has rows for
    uint32_t grid_trace = 0, grid_tail = 0;  // workgroups of its traversal launches (the spill slab has one column per thread)
};

// One GPU's copy of `scratch` and what keeps its users apart (see scratch_of_device).
struct DeviceScratch {
    unsigned char *ptr = nullptr;
    hipEvent_t last_use = nullptr;  // recorded at the end of the latest v4r_render_noalloc on this GPU
    bool used = false;
    std::mutex mu;                  // held while a call queues its work: the order of the locks is the order on the device
};
struct v4r_renderer {
    std::recursive_mutex mu;  // host-side state of the handle: every entry point holds it for its whole call
    int device = 0;
    hipStream_t stream = nullptr;  // public stream = pipeline 0's
    SceneDev S{};
    std::vector<Pipe> pipes;
    uint32_t next_pipe = 0;
    hipEvent_t last_resolve = nullptr;  // `done` of the pipeline that ran the latest batch
    hipEvent_t fork = nullptr;
    unsigned long long *stats = nullptr;
    std::vector<void *> allocations;
    float *accum_own = nullptr, *accum = nullptr;
    uint8_t *tonemap_buf = nullptr;
    uint32_t npix = 0, n_passes = 0, batch_passes = 1;
    uint32_t grid_trace = 0, grid_shade = 0, grid_stream = 0, grid_tail = 0;
    uint32_t tail_segment = 0;  // first segment that k_tail finishes in one launch; 0 = none
    bool any_noise = false;
    bool counting = false;
    // HIP-event timing of the kernel launches (v4r_profile_*): 0 = off; 1 = a pair of events around the traversal
    // launches of every segment; 2 = every batch on pipeline 0 (so the kernels run one after the other and a
    // duration is exclusive) with pairs around the traversal, the shading and the resolve launches
    int profiling = 0;
    struct EventPair {
        hipEvent_t start, stop;
        int group;           // V4R_PROFILE_TRACE / _SHADE / _RESOLVE
        uint32_t launches;   // kernel launches between the two events
    };
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    double group_ms[3] = {0.0, 0.0, 0.0};
    uint64_t group_launches[3] = {0, 0, 0}, group_pairs[3] = {0, 0, 0};
    bool has_class[CLS_COUNT] = {false, false, false, false, false, false};
    int media_mode = MEDIA_NONE;  // selects the kernel instantiations (see traverse)
    bool class_noise[CLS_COUNT] = {false, false, false, false, false, false};  // a material of the class uses a noise texture
    uint32_t give_up_lanes = 0, give_up_segments = 0;  // see k_trace
    bool wide8 = false;          // the traversed tree is the compressed 8-wide one (wide_step8), else the 4-wide one
    bool dense = false;          // the later segments' trace launches run 6 waves/SIMD (TREE_8_WIDE_DENSE)
    uint32_t tile_passes_log2 = 4;  // passes per wave tile of the camera-ray order (see ray_order)
    size_t ctr_bytes = 0;
    uint32_t max_depth = 0, spill_depth = 0;
    std::vector<uint32_t> scene_index;     // the scene's index of the device's primitive k (build())
    uint32_t n_wide_nodes = 0;             // nodes of the tree the kernels traverse (S.nodes4)
    std::vector<uint32_t> media_direct;    // constant media tested before the tree (device numbering)
    unsigned char *scratch_ptr = nullptr;  // this device's copy of `scratch`
    DeviceScratch *device_scratch = nullptr;  // ... and the event that serialises its users
    Pipe noalloc;                          // queues carved out of it (Q.cap = 0: the scene does not fit), on the public stream
    std::string noalloc_why;               // what did not fit, when Q.cap = 0

    template <class T>
    int alloc(T **out, size_t count) {
        void *p = nullptr;
        size_t bytes = std::max<size_t>(count * sizeof(T), 256);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return set_error(V4R_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        allocations.push_back(p);
        *out = (T *)p;
        return V4R_OK;
    }
    template <class T>
    int upload(const T **out, const T *src, size_t count) {
        T *p = nullptr;
        if (int rc = alloc(&p, count)) return rc;
        if (count) {
            hipError_t e = hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
            if (e != hipSuccess) return set_error(V4R_E_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
        }
        *out = p;
        return V4R_OK;
    }
};

namespace {

// Does the texture tree rooted at `tex` contain Perlin noise?
bool has_noise(const v4r_scene_desc *d, uint32_t tex) {
    const v4r_texture &t = d->textures[tex];
    if (t.kind == V4R_TEX_NOISE) return true;
    if (t.kind == V4R_TEX_CHECKER) return has_noise(d, t.a) || has_noise(d, t.b);
    return false;
}

int validate_prim(const v4r_scene_desc *d, const v4r_prim &p, bool in_medium, const char *what) {
    uint32_t kind = p.type & V4R_PRIM_TYPE_MASK;
    if (kind > V4R_PRIM_MEDIUM) return set_error(V4R_E_INVALID, std::string(what) + ": unknown primitive type");
    if (p.xform != V4R_NO_XFORM && p.xform >= d->n_xforms) return set_error(V4R_E_INVALID, std::string(what) + ": transform index out of range");
    if (kind == V4R_PRIM_TRIANGLE && p.aux >= d->n_tri_shading) return set_error(V4R_E_INVALID, std::string(what) + ": triangle shading index out of range");
    if (kind == V4R_PRIM_MEDIUM) {
        if (in_medium) return set_error(V4R_E_INVALID, std::string(what) + ": nested constant medium");
        if (p.aux >= d->n_media) return set_error(V4R_E_INVALID, std::string(what) + ": medium index out of range");
    }
    return V4R_OK;
}

// Every index the kernels will follow is checked here, so a malformed table is an error code, not
// a GPU fault.
int validate(const v4r_scene_desc *d, uint32_t &depth_out) {
    if (!d) return set_error(V4R_E_INVALID, "NULL scene description");
    if (d->abi_version != V4R_ABI_VERSION) return set_error(V4R_E_INVALID, "scene description has a different ABI version");
    if (d->width == 0 || d->height == 0) return set_error(V4R_E_INVALID, "width and height must be positive");
    if ((uint64_t)d->width * d->height > (1ull << 28)) return set_error(V4R_E_LIMIT, "image larger than 2^28 pixels");
    if (d->n_prims > LEAF_FIRST_MASK) return set_error(V4R_E_LIMIT, "more than 2^28 - 1 primitives");
    auto need = [&](const void *p, uint32_t n, const char *name) {
        return (n != 0 && p == nullptr) ? set_error(V4R_E_INVALID, std::string(name) + " table is NULL") : V4R_OK;
    };
    if (int rc = need(d->nodes, d->n_nodes, "nodes")) return rc;
    if (int rc = need(d->prims, d->n_prims, "prims")) return rc;
    if (int rc = need(d->prim_material, d->n_prims, "prim_material")) return rc;
    if (int rc = need(d->xforms, d->n_xforms, "xforms")) return rc;
    if (int rc = need(d->tri_shading, d->n_tri_shading, "tri_shading")) return rc;
    if (int rc = need(d->media, d->n_media, "media")) return rc;
    if (int rc = need(d->medium_prims, d->n_medium_prims, "medium_prims")) return rc;
    if (int rc = need(d->materials, d->n_materials, "materials")) return rc;
    if (int rc = need(d->textures, d->n_textures, "textures")) return rc;
    if (int rc = need(d->images, d->n_images, "images")) return rc;
    if (int rc = need(d->perlin, d->n_perlin, "perlin")) return rc;
    if (int rc = need(d->lights, d->n_lights, "lights")) return rc;
    if (d->n_materials > 0x3FFFFu) return set_error(V4R_E_LIMIT, "too many materials");

    // BVH: n-1 interior nodes for n >= 2 primitives, every primitive reachable exactly once.
    if (d->n_prims <= 1) {
        if (d->n_nodes != 0) return set_error(V4R_E_INVALID, "a scene with fewer than two primitives has no interior BVH node");
    } else if (d->n_nodes != d->n_prims - 1) {
        return set_error(V4R_E_INVALID, "BVH must hold exactly n_prims - 1 interior nodes");
    }
    depth_out = 0;
    if (d->n_nodes) {
        std::vector<uint8_t> seen_node(d->n_nodes, 0), seen_prim(d->n_prims, 0);
        std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 1u}};
        seen_node[0] = 1;
        while (!stack.empty()) {
            auto [node, depth] = stack.back();
            stack.pop_back();
            depth_out = std::max(depth_out, depth);
            const v4r_bvh_node &n = d->nodes[node];
            for (uint32_t c : {n.child0, n.child1}) {
                if (c & V4R_LEAF_BIT) {
                    uint32_t p = c & ~V4R_LEAF_BIT;
                    if (p >= d->n_prims || seen_prim[p]) return set_error(V4R_E_INVALID, "BVH leaf index out of range or repeated");
                    seen_prim[p] = 1;
                } else {
                    if (c >= d->n_nodes || seen_node[c]) return set_error(V4R_E_INVALID, "BVH child index out of range or repeated");
                    seen_node[c] = 1;
                    stack.push_back({c, depth + 1});
                }
            }
        }
    }
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        if (int rc = validate_prim(d, d->prims[i], false, "prims")) return rc;
        if (d->prim_material[i] >= d->n_materials) return set_error(V4R_E_INVALID, "prim_material index out of range");
    }
    for (uint32_t i = 0; i < d->n_medium_prims; ++i)
        if (int rc = validate_prim(d, d->medium_prims[i], true, "medium_prims")) return rc;
    for (uint32_t i = 0; i < d->n_media; ++i) {
        const v4r_medium &m = d->media[i];
        if (m.count == 0 || (uint64_t)m.first + m.count > d->n_medium_prims) return set_error(V4R_E_INVALID, "medium boundary range out of bounds");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const v4r_texture &t = d->textures[i];
        switch (t.kind) {
            case V4R_TEX_SOLID: break;
            case V4R_TEX_CHECKER:
                // children must precede their parent: the table is then acyclic by construction
                if (t.a >= i || t.b >= i) return set_error(V4R_E_INVALID, "checker children must have lower texture indices");
                break;
            case V4R_TEX_IMAGE:
                if (t.a >= d->n_images) return set_error(V4R_E_INVALID, "image index out of range");
                break;
            case V4R_TEX_NOISE:
                if (t.a >= d->n_perlin) return set_error(V4R_E_INVALID, "perlin table index out of range");
                break;
            default: return set_error(V4R_E_INVALID, "unknown texture kind");
        }
    }
    for (uint32_t i = 0; i < d->n_images; ++i) {
        const v4r_image &im = d->images[i];
        if (im.width == 0 || im.height == 0 || im.offset + (uint64_t)im.width * im.height * 3 > d->texel_bytes || !d->texels)
            return set_error(V4R_E_INVALID, "image texels out of bounds");
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const v4r_material &m = d->materials[i];
        if (m.kind > V4R_MAT_ISOTROPIC) return set_error(V4R_E_INVALID, "unknown material kind");
        bool textured = m.kind == V4R_MAT_LAMBERTIAN || m.kind == V4R_MAT_DIFFUSE_LIGHT || m.kind == V4R_MAT_ISOTROPIC;
        if (textured && m.tex >= d->n_textures) return set_error(V4R_E_INVALID, "material texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        if (d->lights[i] >= d->n_prims) return set_error(V4R_E_INVALID, "light index out of range");
        if ((d->prims[d->lights[i]].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM)
            return set_error(V4R_E_INVALID, "a constant medium cannot be a light");
    }
    return V4R_OK;
}

uint32_t class_of(const v4r_scene_desc *d, const v4r_material &m) {
    switch (m.kind) {
        case V4R_MAT_DIFFUSE_LIGHT: return CLS_LIGHT;
        case V4R_MAT_LAMBERTIAN: return has_noise(d, m.tex) ? CLS_LAMBERT_NOISE : CLS_LAMBERT;
        case V4R_MAT_METAL: return CLS_METAL;
        case V4R_MAT_DIELECTRIC: return CLS_DIELECTRIC;
        default: return CLS_ISOTROPIC;
    }
}

// Surface area of a primitive in world space (0 for a constant medium): only used to guess which shapes few rays reach.
double surface_area(const v4r_scene_desc *d, const v4r_prim &pr) {
    double a = 0.0;
    switch (pr.type & V4R_PRIM_TYPE_MASK) {
        case V4R_PRIM_SPHERE: a = 4.0 * 3.14159265358979323846 * (double)pr.f[3] * (double)pr.f[3]; break;
        case V4R_PRIM_XY_RECT:
        case V4R_PRIM_YZ_RECT:
        case V4R_PRIM_ZX_RECT: a = ((double)pr.f[1] - (double)pr.f[0]) * ((double)pr.f[3] - (double)pr.f[2]); break;
        case V4R_PRIM_TRIANGLE: {
            const float *e = pr.f + 3;  // the two edge vectors
            double cx = (double)e[1] * e[5] - (double)e[2] * e[4], cy = (double)e[2] * e[3] - (double)e[0] * e[5], cz = (double)e[0] * e[4] - (double)e[1] * e[3];
            a = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
            break;
        }
        default: return 0.0;
    }
    if (pr.xform != V4R_NO_XFORM) {  // the area scale of an affine map, taken as the 2/3 power of its volume scale
        const float *m = d->xforms[pr.xform].fwd;
        double det = (double)m[0] * ((double)m[4] * m[8] - (double)m[5] * m[7]) - (double)m[1] * ((double)m[3] * m[8] - (double)m[5] * m[6]) +
                     (double)m[2] * ((double)m[3] * m[7] - (double)m[4] * m[6]);
        a *= std::cbrt(det * det);
    }
    return a == a && a > 0.0 ? a : 0.0;
}

template <class K>
uint32_t grid_for(K kernel, int n_cu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    per_cu = std::min(per_cu, 8);
    return (uint32_t)(n_cu * per_cu);
}

// A binary tree over the leaf boxes of the host's tree, split by the usual binned surface-area
// heuristic (16 bins per axis, all three axes, true surface area), in the node format of the ABI.
// The reference's builder (bvh.rs:84-141: 6 buckets, widest centroid axis only, squared-extent
// "area", one shape per leaf) stays the host's; the closest hit does not depend on the tree, and
// rays visit fewer nodes in this one.  V4R_FLAG_KEEP_HOST_BVH skips it.
// A leaf is a block of up to `leaf_max` primitives: the tree stops splitting a range that small (a constant medium is
// always a leaf of its own: its subtree must never be culled, see wide_step).  `order` receives the primitives in the
// order of the tree's leaves -- the device's own numbering of the primitives, under which a leaf's block is a run
// of consecutive records -- and a leaf's child word is leaf_word(first position in `order`, count).
void rebuild_binary(const v4r_scene_desc *d, bool without_media, uint32_t leaf_max, std::vector<v4r_bvh_node> &out, std::vector<uint32_t> &order) {
    struct Box {
        float lo[3], hi[3];
    };
    std::vector<Box> box(d->n_prims);
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        const v4r_bvh_node &n = d->nodes[i];
        if (n.child0 & V4R_LEAF_BIT) {
            Box &b = box[n.child0 & ~V4R_LEAF_BIT];
            std::memcpy(b.lo, n.lo0, 12);
            std::memcpy(b.hi, n.hi0, 12);
        }
        if (n.child1 & V4R_LEAF_BIT) {
            Box &b = box[n.child1 & ~V4R_LEAF_BIT];
            std::memcpy(b.lo, n.lo1, 12);
            std::memcpy(b.hi, n.hi1, 12);
        }
    }
    const float inf = std::numeric_limits<float>::infinity();
    auto grow = [](Box &a, const Box &b) {
        for (int k = 0; k < 3; ++k) {
            a.lo[k] = std::min(a.lo[k], b.lo[k]);
            a.hi[k] = std::max(a.hi[k], b.hi[k]);
        }
    };
    auto empty = [&]() {
        Box b;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = inf;
            b.hi[k] = -inf;
        }
        return b;
    };
    auto area = [](const Box &b) {
        float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto is_medium = [&](uint32_t p) { return (d->prims[p].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM; };
    order.clear();
    order.reserve(d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; ++i)
        if (!without_media || !is_medium(i)) order.push_back(i);
    // a range becomes one leaf when it is short enough and holds no constant medium (unless it is a single primitive)
    auto is_leaf = [&](uint32_t begin, uint32_t end) {
        if (end - begin == 1) return true;
        if (end - begin > leaf_max) return false;
        if (d->n_media != 0)
            for (uint32_t i = begin; i < end; ++i)
                if (is_medium(order[i])) return false;
        return true;
    };
    out.clear();
    out.reserve(d->n_nodes);
    struct Task {
        uint32_t begin, end, node;
        int slot;  // which child word of `node` receives this range (-1: the root)
    };
    std::vector<Task> todo{{0u, (uint32_t)order.size(), 0u, -1}};
    constexpr int BINS = 16;
    while (!todo.empty()) {
        Task t = todo.back();
        todo.pop_back();
        uint32_t n = t.end - t.begin;
        // a short range is a leaf word in its parent (is_leaf); the others become interior nodes
        uint32_t self = (uint32_t)out.size();
        out.emplace_back();
        if (t.slot >= 0) (t.slot == 0 ? out[t.node].child0 : out[t.node].child1) = self;
        Box cb = empty();
        for (uint32_t i = t.begin; i < t.end; ++i) {
            const Box &b = box[order[i]];
            for (int k = 0; k < 3; ++k) {
                float c = 0.5f * (b.lo[k] + b.hi[k]);
                cb.lo[k] = std::min(cb.lo[k], c);
                cb.hi[k] = std::max(cb.hi[k], c);
            }
        }
        int best_axis = -1, best_bin = 0;
        float best_cost = inf;
        for (int axis = 0; axis < 3; ++axis) {
            float ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.0f) || !(ext < inf)) continue;
            Box bb[BINS];
            uint32_t cnt[BINS] = {0};
            for (auto &b : bb) b = empty();
            float scale = (float)BINS / ext;
            for (uint32_t i = t.begin; i < t.end; ++i) {
                const Box &b = box[order[i]];
                int bin = std::min(BINS - 1, std::max(0, (int)((0.5f * (b.lo[axis] + b.hi[axis]) - cb.lo[axis]) * scale)));
                cnt[bin]++;
                grow(bb[bin], b);
            }
            float right_area[BINS];
            uint32_t right_cnt[BINS];
            Box acc = empty();
            uint32_t c = 0;
            for (int k = BINS - 1; k > 0; --k) {
                grow(acc, bb[k]);
                c += cnt[k];
                right_area[k] = area(acc);
                right_cnt[k] = c;
            }
            acc = empty();
            c = 0;
            for (int k = 0; k < BINS - 1; ++k) {
                grow(acc, bb[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                float cost = (float)c * area(acc) + (float)right_cnt[k + 1] * right_area[k + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = k;
                }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = (float)BINS / ext;
            auto in_left = [&](uint32_t p) {
                const Box &b = box[p];
                int bin = std::min(BINS - 1, std::max(0, (int)((0.5f * (b.lo[best_axis] + b.hi[best_axis]) - cb.lo[best_axis]) * scale)));
                return bin <= best_bin;
            };
            mid = (uint32_t)(std::partition(order.begin() + t.begin, order.begin() + t.end, in_left) - order.begin());
        } else {
            mid = t.begin + n / 2;  // coincident centroids: halve the list
        }
        Box l = empty(), r = empty();
        for (uint32_t i = t.begin; i < mid; ++i) grow(l, box[order[i]]);
        for (uint32_t i = mid; i < t.end; ++i) grow(r, box[order[i]]);
        v4r_bvh_node &nd = out[self];
        std::memset(&nd, 0, sizeof nd);
        std::memcpy(nd.lo0, l.lo, 12);
        std::memcpy(nd.hi0, l.hi, 12);
        std::memcpy(nd.lo1, r.lo, 12);
        std::memcpy(nd.hi1, r.hi, 12);
        if (is_leaf(t.begin, mid)) nd.child0 = leaf_word(t.begin, mid - t.begin);
        else todo.push_back({t.begin, mid, self, 0});
        if (is_leaf(mid, t.end)) nd.child1 = leaf_word(mid, t.end - mid);
        else todo.push_back({mid, t.end, self, 1});
    }
}

// The 4-wide tree of wide_step from the host's binary tree: a wide node starts with the two children
// of a binary node and, while it has a free slot, replaces its interior child of largest surface area
// by that child's two children.  Leaves stay single primitives, so the set of (ray, primitive) tests
// that can happen is the one of the binary tree.  Nodes are emitted parent before children, siblings
// next to each other.
// Returns the largest number of entries a traversal of the tree can have on its stack.
// WIDE = 8: the compressed 8-wide tree of wide_step8, built the same way; `encodable` turns false (and the tree must not be
// used) when a box is not finite.
template <int WIDE>
uint32_t collapse_to_wide(const v4r_scene_desc *d, const v4r_bvh_node *nodes, uint32_t n_nodes, std::vector<float4> &out, uint32_t &n_top,
                          bool &encodable) {
    out.clear();
    n_top = 0;
    encodable = true;
    if (n_nodes == 0) return 0;
    // which subtrees of the binary tree hold a constant medium (wide_step must not cull those by distance)
    std::vector<uint8_t> holds_medium(n_nodes, 0);
    if (d->n_media != 0) {
        auto leaf_is_medium = [&](uint32_t word) { return (d->prims[leaf_first(word)].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM; };
        std::vector<std::pair<uint32_t, bool>> walk{{0u, false}};  // (node, children done)
        while (!walk.empty()) {
            auto [node, done] = walk.back();
            walk.pop_back();
            const v4r_bvh_node &n = nodes[node];
            if (!done) {
                walk.push_back({node, true});
                if (!(n.child0 & V4R_LEAF_BIT)) walk.push_back({n.child0, false});
                if (!(n.child1 & V4R_LEAF_BIT)) walk.push_back({n.child1, false});
            } else {
                bool a = (n.child0 & V4R_LEAF_BIT) ? leaf_is_medium(n.child0) : holds_medium[n.child0] != 0;
                bool b = (n.child1 & V4R_LEAF_BIT) ? leaf_is_medium(n.child1) : holds_medium[n.child1] != 0;
                holds_medium[node] = a || b;
            }
        }
    }
    auto never_cull = [&](uint32_t word) {
        if (d->n_media == 0) return false;
        if (word & V4R_LEAF_BIT) return (d->prims[leaf_first(word)].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM;
        return holds_medium[word] != 0;
    };
    struct Item {
        float lo[3], hi[3];
        uint32_t child;  // word of the binary tree: interior index or V4R_LEAF_BIT | primitive
    };
    auto area = [](const Item &it) {
        float dx = it.hi[0] - it.lo[0], dy = it.hi[1] - it.lo[1], dz = it.hi[2] - it.lo[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    };
    auto children = [&](uint32_t node, Item &a, Item &b) {
        const v4r_bvh_node &n = nodes[node];
        std::memcpy(a.lo, n.lo0, 12);
        std::memcpy(a.hi, n.hi0, 12);
        a.child = n.child0;
        std::memcpy(b.lo, n.lo1, 12);
        std::memcpy(b.hi, n.hi1, 12);
        b.child = n.child1;
    };
    struct Task {
        uint32_t binary, wide, depth;
    };
    // Node order: the top TOP_LEVELS levels breadth first (they are the prefix that k_trace copies into LDS), the
    // rest depth first (a subtree stays together in memory).
    std::vector<Task> todo;          // depth first: taken from the back
    std::vector<Task> top{{0u, 0u, 1u}};  // breadth first: taken from the front, all before `todo`
    size_t top_next = 0;
    uint32_t depth = 0;
    std::vector<uint32_t> parent_of(1, 0u), fanout(1, 0u);  // per wide node (children are numbered after their parent)
    out.resize(8);
    const float inf = std::numeric_limits<float>::infinity();
    while (top_next < top.size() || !todo.empty()) {
        Task t;
        const bool from_top = top_next < top.size();
        if (from_top) {
            t = top[top_next++];
        } else {
            t = todo.back();
            todo.pop_back();
        }
        // the children of levels < TOP_LEVELS are numbered (and expanded) breadth first, before anything deeper
        std::vector<Task> &next = from_top && t.depth < (uint32_t)(WIDE == 8 ? TOP_LEVELS_8 : TOP_LEVELS) ? top : todo;
        depth = std::max(depth, t.depth);
        Item items[WIDE];
        int n = 2;
        children(t.binary, items[0], items[1]);
        // 8-wide: the children's boxes go on a 255-step grid over the node's box, so a child much smaller than the node
        // (the field of small spheres beside their 1000-unit ground sphere) is not opened here: its own children would each
        // be rounded out to a few coarse steps and every ray would enter all of them.  It becomes a node with a grid of its own.
        float node_extent = 0.0f;
        if (WIDE == 8) {
            const float min_steps = 16.0f;  // C2 forced onto the 8-wide tree: -30 % without the rule, -10 % with 8 to 24 steps
            for (int a = 0; a < 3; ++a)
                node_extent = std::max(node_extent, std::max(items[0].hi[a], items[1].hi[a]) - std::min(items[0].lo[a], items[1].lo[a]));
            node_extent = node_extent < inf ? node_extent * min_steps / 255.0f : 0.0f;
        }
        while (n < WIDE) {
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n; ++k)
                if (!(items[k].child & V4R_LEAF_BIT)) {
                    if (WIDE == 8 && std::max(items[k].hi[0] - items[k].lo[0], std::max(items[k].hi[1] - items[k].lo[1], items[k].hi[2] - items[k].lo[2])) < node_extent)
                        continue;
                    float a = area(items[k]);
                    if (!(a == a)) a = inf;  // a box with a NaN extent is expanded first
                    if (pick < 0 || a > best) {
                        pick = k;
                        best = a;
                    }
                }
            if (pick < 0) break;
            uint32_t node = items[pick].child;
            children(node, items[pick], items[n]);
            ++n;
        }
        uint32_t child[WIDE], mask = 0;
        for (int k = 0; k < n; ++k)
            if (never_cull(items[k].child)) mask |= 1u << k;
        for (int k = 0; k < WIDE; ++k) {
            if (k >= n) child[k] = 0xFFFFFFFFu;
            else if (items[k].child & V4R_LEAF_BIT) child[k] = items[k].child;
            else {
                child[k] = (uint32_t)(out.size() / 8);
                out.resize(out.size() + 8);
                next.push_back({items[k].child, child[k], t.depth + 1u});
                parent_of.resize(out.size() / 8, 0u);
                parent_of[child[k]] = t.wide;
            }
        }
        fanout.resize(out.size() / 8, 0u);
        fanout[t.wide] = (uint32_t)n;
        float4 *w = out.data() + 8 * (size_t)t.wide;
        if constexpr (WIDE == 4) {
        float lo[3][4], hi[3][4];
        for (int k = 0; k < 4; ++k)
            for (int a = 0; a < 3; ++a) {
                // NaN planes make every comparison of the slab test false (an inverted box would not: the test
                // orders the two planes of an axis itself)
                lo[a][k] = k < n ? items[k].lo[a] : std::numeric_limits<float>::quiet_NaN();
                hi[a][k] = k < n ? items[k].hi[a] : std::numeric_limits<float>::quiet_NaN();
            }
        for (int a = 0; a < 3; ++a) {
            w[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
            w[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
        }
        float c[4];
        std::memcpy(c, child, 16);
        w[6] = make_float4(c[0], c[1], c[2], c[3]);
        float mask_bits;
        std::memcpy(&mask_bits, &mask, 4);
        w[7] = make_float4(mask_bits, 0.0f, 0.0f, 0.0f);
        } else {
        // Compressed 8-wide node, one 128-byte line: origin p and one power-of-two scale per axis, the children's boxes as
        // 8-bit offsets from p in units of the scale (lower planes rounded down, upper planes up: the decoded box contains
        // the child's), eight child words.  A box that is not finite cannot be encoded: work[0] = inf then, and the tree must not be used.
        float p[3];
        uint32_t biased[3];
        uint8_t qlo[3][8], qhi[3][8];
        for (int k = 0; k < n; ++k)
            for (int a = 0; a < 3; ++a)
                if (!std::isfinite(items[k].lo[a]) || !std::isfinite(items[k].hi[a])) encodable = false;  // cannot be put on a grid
        if (!encodable) continue;  // (the tree is walked to the end, but no longer written)
        for (int a = 0; a < 3; ++a) {
            float lo_all = inf, hi_all = -inf;
            for (int k = 0; k < n; ++k) {
                lo_all = std::min(lo_all, items[k].lo[a]);
                hi_all = std::max(hi_all, items[k].hi[a]);
            }
            if (!(lo_all <= hi_all)) lo_all = hi_all = 0.0f;
            p[a] = lo_all;
            int e = -100;
            if (hi_all > lo_all) e = std::max(-100, (int)std::ceil(std::log2(((double)hi_all - (double)lo_all) / 255.0)));
            auto decoded = [&](int q, int ee) { return p[a] + (float)q * std::ldexp(1.0f, ee); };  // as the device computes it
            while (decoded(255, e) < hi_all) ++e;
            if (e > 24) encodable = false;  // 255 x 2^e x the largest reciprocal traverse() uses (1e18) must stay finite
            biased[a] = (uint32_t)(e + 127);
            const double scale = std::ldexp(1.0, e);
            for (int k = 0; k < 8; ++k) {
                qlo[a][k] = 255;
                qhi[a][k] = 0;
                if (k >= n) continue;
                int ql = (int)std::floor(((double)items[k].lo[a] - (double)p[a]) / scale), qh = (int)std::ceil(((double)items[k].hi[a] - (double)p[a]) / scale);
                ql = std::max(0, std::min(255, ql));
                qh = std::max(0, std::min(255, qh));
                while (ql > 0 && decoded(ql, e) > items[k].lo[a]) --ql;
                while (qh < 255 && decoded(qh, e) < items[k].hi[a]) ++qh;
                qlo[a][k] = (uint8_t)ql;
                qhi[a][k] = (uint8_t)qh;
            }
        }
        auto word = [](const uint8_t *b) { return (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24; };
        auto f = [](uint32_t u) {
            float v;
            std::memcpy(&v, &u, 4);
            return v;
        };
        w[0] = make_float4(p[0], p[1], p[2], f(biased[0] | biased[1] << 8 | biased[2] << 16 | mask << 24));
        w[1] = make_float4(f(child[0]), f(child[1]), f(child[2]), f(child[3]));
        w[2] = make_float4(f(child[4]), f(child[5]), f(child[6]), f(child[7]));
        w[3] = make_float4(f(word(qlo[0])), f(word(qlo[0] + 4)), f(word(qlo[1])), f(word(qlo[1] + 4)));
        w[4] = make_float4(f(word(qlo[2])), f(word(qlo[2] + 4)), f(word(qhi[0])), f(word(qhi[0] + 4)));
        w[5] = make_float4(f(word(qhi[1])), f(word(qhi[1] + 4)), f(word(qhi[2])), f(word(qhi[2] + 4)));
        w[6] = w[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    n_top = (uint32_t)top.size();  // one task per node of the top levels; they were numbered first
    // The deepest the traversal stack can get: a node step pushes every entered child but the nearest, so below node v the
    // stack holds at most (children of v - 1) + the most any of v's children can add.
    (void)depth;
    std::vector<uint32_t> below(fanout.size(), 0u);
    for (size_t v = fanout.size(); v-- > 0;) {
        below[v] += fanout[v] - 1u;
        if (v != 0) below[parent_of[v]] = std::max(below[parent_of[v]], below[v]);
    }
    return below[0];
}

// Under the packet-statistics rule a batch keeps one ray counter per pass and primitive: bound that table to 128 MiB.
uint32_t cap_counted_passes(const v4r_renderer *r, uint32_t passes) {
    if (!r->S.packet_rule || r->S.n_prims == 0) return passes;
    return std::max(1u, std::min<uint32_t>(passes, std::max(16u, (32u << 20) / r->S.n_prims)));
}

// Prices of the tree choice in build(), in node steps of the 4-wide tree per wave: a node step of the 8-wide tree, a leaf step.
// (Measured break-even of the 8-wide step: between 1.15 -- the spheres scene loses 5 % on the 8-wide tree -- and 1.38 -- the
// 4K textured-spheres scene gains 1 %.)
constexpr double WIDE8_STEP_COST = 1.35, WIDE_LEAF_COST = 1.0;

// Give `pipe` queues for at least `need` path slots (and ray counters for `n_passes` passes, see Queues::shape_rays).  Growing frees the old queues first (after the
// pipeline has drained), so a renderer's footprint follows the largest batch it has been asked for:
// 144 B per slot, i.e. 0.09 GB for one 800x800 pass and 27.6 GB for a 192 M batch.
int ensure_pipe(v4r_renderer *r, Pipe &pipe, size_t need, size_t n_passes) {
    Queues &Q = pipe.Q;
    if (!r->S.packet_rule) n_passes = 0;
    if (Q.cap >= need && pipe.count_passes >= n_passes && !pipe.allocations.empty()) return V4R_OK;
    need = std::max<size_t>(need, Q.cap);
    n_passes = std::max(n_passes, pipe.count_passes);
    HIP_TRY(hipStreamSynchronize(pipe.stream));
    for (void *p : pipe.allocations) (void)hipFree(p);
    pipe.allocations.clear();
    Q = Queues{};
    auto alloc = [&](auto **out, size_t count) -> int {
        void *p = nullptr;
        size_t bytes = std::max<size_t>(count * sizeof(**out), 256);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess)
            return set_error(V4R_E_HIP, std::string("hipMalloc of a ") + std::to_string(bytes >> 20) + " MiB queue: " + hipGetErrorString(e) +
                                            " (V4R_BATCH_PATHS / V4R_PIPES reduce the footprint)");
        pipe.allocations.push_back(p);
        *out = (std::remove_reference_t<decltype(**out)> *)p;
        return V4R_OK;
    };
    for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 3; ++c)
            if (int rc = alloc(&Q.ray[p][c], need)) return rc;
    for (int k = 0; k < CLS_COUNT; ++k)
        if (int rc = alloc(&Q.hit[k], r->has_class[k] ? need : 1)) return rc;
    if (int rc = alloc(&Q.hitd, need)) return rc;
    if (int rc = alloc(&Q.given_up, need)) return rc;
    if (int rc = alloc(&Q.rad, need)) return rc;
    if (int rc = alloc(&Q.ctr, (size_t)r->max_depth + 1)) return rc;
    // in stream order: a null-stream hipMemset is not ordered against the non-blocking streams and could land
    // in the middle of the pipeline's first batch
    HIP_TRY(hipMemsetAsync(Q.ctr, 0, r->ctr_bytes, pipe.stream));
    Q.stats = r->stats;
    Q.spill_stride = (size_t)r->grid_trace * BLOCK;
    Q.spill_depth = r->spill_depth;
    if (int rc = alloc(&Q.spill, Q.spill_stride * std::max(r->spill_depth, 1u))) return rc;
    if (int rc = alloc(&Q.shape_rays, 2 * n_passes * r->S.n_prims)) return rc;  // counters, then their "full" flags
    pipe.count_passes = n_passes;
    Q.cap = (uint32_t)need;
    pipe.grid_trace = r->grid_trace;
    pipe.grid_tail = r->grid_tail;
    return V4R_OK;
}

// One GPU's copy of `scratch`: its address (the array's name is a host-side handle, not a device address: only the
// pointer obtained here with hipGetSymbolAddress may be handed to kernels or memsets) and what keeps its users apart --
// every renderer on the GPU shares the array, so v4r_render_noalloc calls are queued one behind the other on the device:
// each call first makes its stream wait for the event the previous call recorded (whatever handle or thread made it).
// Created once per device and process, by the first v4r_renderer_create on that device; never destroyed.
int scratch_of_device(int device, DeviceScratch **out) {  // (the device must be current)
    static std::mutex mu;
    static std::vector<DeviceScratch *> known;
    std::lock_guard<std::mutex> hold(mu);
    if ((size_t)device >= known.size()) known.resize((size_t)device + 1, nullptr);
    if (!known[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return set_error(V4R_E_HIP, "hipGetSymbolAddress returned a NULL scratch address");
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DeviceScratch *ds = new DeviceScratch();
        ds->ptr = (unsigned char *)p;
        ds->last_use = e;
        known[device] = ds;
    }
    *out = known[device];
    return V4R_OK;
}

// Lay the queues of one batch pipeline out inside the scratch array: the stack-spill slab of a reduced traversal
// grid, the segment counters, then as many path slots as the rest holds.  Leaves Q.cap = 0 when the scene's tree is
// too deep or its depth budget too long for anything to fit.
constexpr uint32_t NOALLOC_COUNTED_PASSES = 16;
void carve_noalloc(v4r_renderer *r, int n_cu) {
    Pipe &pipe = r->noalloc;
    Queues &Q = pipe.Q;
    Q = Queues{};
    pipe.grid_trace = std::min<uint32_t>(r->grid_trace, (uint32_t)n_cu);
    pipe.grid_tail = std::min<uint32_t>(r->grid_tail, pipe.grid_trace);
    size_t off = 0;
    auto take = [&](auto **out, size_t count) {
        size_t bytes = (count * sizeof(**out) + 255) & ~(size_t)255;
        if (off + bytes > (size_t)V4R_NOALLOC_SCRATCH_BYTES) return false;
        *out = (std::remove_reference_t<decltype(**out)> *)(r->scratch_ptr + off);
        off += bytes;
        return true;
    };
    Q.spill_stride = (size_t)pipe.grid_trace * BLOCK;
    Q.spill_depth = r->spill_depth;
    auto mib = [](size_t bytes) { return std::to_string((bytes + (1u << 20) - 1) >> 20) + " MiB"; };
    const size_t spill_bytes = Q.spill_stride * std::max(r->spill_depth, 1u) * sizeof(uint2), ctr_bytes = ((size_t)r->max_depth + 1) * sizeof(SegCtr);
    if (!take(&Q.spill, Q.spill_stride * std::max(r->spill_depth, 1u))) {
        r->noalloc_why = "the traversal-stack spill slab of this scene's tree (" + mib(spill_bytes) + ") does not fit";
        return;
    }
    if (!take(&Q.ctr, (size_t)r->max_depth + 1)) {
        r->noalloc_why = "the segment counters of a depth budget of " + std::to_string(r->max_depth) + " (" + mib(ctr_bytes) + ") do not fit";
        return;
    }
    if (r->S.packet_rule) {  // ray counters of the packet-statistics rule, for batches of up to NOALLOC_COUNTED_PASSES passes
        if (!take(&Q.shape_rays, 2 * (size_t)NOALLOC_COUNTED_PASSES * r->S.n_prims)) {
            r->noalloc_why = "the per-shape ray counters of the light pdf (" + std::to_string(NOALLOC_COUNTED_PASSES) + " passes x " + std::to_string(r->S.n_prims) +
                             " primitives x 8 bytes = " + mib(2 * (size_t)NOALLOC_COUNTED_PASSES * r->S.n_prims * sizeof(uint32_t)) +
                             ") do not fit: the scene has too many primitives (V4R_FLAG_PLANE_LIGHT_PDF or V4R_FLAG_BOUNDED_LIGHT_PDF need no counters)";
            return;
        }
        pipe.count_passes = NOALLOC_COUNTED_PASSES;
    }
    // per path slot: two ray queues of three 16-byte columns, the hit record, the radiance slot, the given-up index
    // and one index per material class
    const size_t per_slot = 2 * 3 * sizeof(float4) + 2 * sizeof(float4) + (1 + CLS_COUNT) * sizeof(uint32_t);
    size_t cap = ((size_t)V4R_NOALLOC_SCRATCH_BYTES - off) / per_slot;
    cap = cap > 4096 ? (cap - 4096) & ~(size_t)63 : 0;  // room for the 256-byte alignment of each of the 15 arrays
    if (cap == 0) {
        r->noalloc_why = "the spill slab (" + mib(spill_bytes) + "), segment counters and ray counters leave no room for path slots";
        return;
    }
    bool ok = true;
    for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 3; ++c) ok = ok && take(&Q.ray[p][c], cap);
    for (int k = 0; k < CLS_COUNT; ++k) ok = ok && take(&Q.hit[k], cap);
    ok = ok && take(&Q.hitd, cap) && take(&Q.given_up, cap) && take(&Q.rad, cap);
    if (!ok) return;
    Q.stats = r->stats;
    Q.cap = (uint32_t)cap;
}

int build(v4r_renderer *r, const v4r_scene_desc *given, uint32_t bvh_depth) {
    (void)bvh_depth;  // the traversal stack is sized by the wide tree made below
    uint32_t wide_depth = 0, wide_depth8 = 0, n_top8 = 0, n_nodes8 = 0;  // deepest traversal stack of either tree
    const float4 *nodes8 = nullptr;  // the 8-wide tree, when the scene has one
    HIP_TRY(hipSetDevice(r->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, r->device));
    int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));

    SceneDev &S = r->S;
    // A few constant media are tested directly, before the tree, by the kernels without the medium code in their
    // traversal (see traverse): the tree is then built over the other primitives.
    size_t n_media_prims = 0;
    for (uint32_t i = 0; i < given->n_prims; ++i)
        if ((given->prims[i].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM) ++n_media_prims;
    const bool keep_host_tree = (given->flags & V4R_FLAG_KEEP_HOST_BVH) != 0 || given->n_nodes == 0;
    const bool direct = !keep_host_tree && n_media_prims != 0 && n_media_prims <= MAX_DIRECT_MEDIA && given->n_prims - n_media_prims >= 2;
    r->media_mode = direct ? MEDIA_DIRECT : (given->n_media != 0 ? MEDIA_IN_TREE : MEDIA_NONE);
    // The device's own numbering of the primitives: the order of the leaves of the tree it builds, so that the primitives
    // of a leaf -- and of neighbouring leaves -- are neighbours in every per-primitive table.  r->scene_index[k] is the
    // scene's index (Scene::hittables, scene.rs:20-23) of the device's primitive k: the closest-hit rule orders equal
    // distances by it (consider), and the entry points that return primitive indices translate through it.  The scene's
    // own tree (V4R_FLAG_KEEP_HOST_BVH, or a scene of one shape) keeps the scene's numbering.
    uint32_t leaf_max = V4R_LEAF_PRIMS;
    if (const char *env = std::getenv("V4R_LEAF_PRIMS")) leaf_max = (uint32_t)std::max(1, std::min((int)V4R_LEAF_PRIMS, std::atoi(env)));
    std::vector<v4r_bvh_node> better;
    std::vector<uint32_t> &scene_index = r->scene_index;
    if (!keep_host_tree) {
        rebuild_binary(given, direct, leaf_max, better, scene_index);
        if (direct)
            for (uint32_t i = 0; i < given->n_prims; ++i)
                if ((given->prims[i].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM) scene_index.push_back(i);  // outside the tree: numbered last
    } else {
        scene_index.resize(given->n_prims);
        for (uint32_t i = 0; i < given->n_prims; ++i) scene_index[i] = i;
    }
    std::vector<uint32_t> device_index(given->n_prims);
    for (uint32_t k = 0; k < given->n_prims; ++k) device_index[scene_index[k]] = k;
    std::vector<v4r_prim> prims_in_order(given->n_prims);
    std::vector<uint32_t> materials_in_order(given->n_prims), lights_renumbered(given->n_lights);
    for (uint32_t k = 0; k < given->n_prims; ++k) {
        prims_in_order[k] = given->prims[scene_index[k]];
        materials_in_order[k] = given->prim_material[scene_index[k]];
    }
    for (uint32_t l = 0; l < given->n_lights; ++l) lights_renumbered[l] = device_index[given->lights[l]];
    // from here on `d` is the scene under the device's numbering; `given` keeps the scene's tree
    v4r_scene_desc renumbered = *given;
    renumbered.prims = prims_in_order.data();
    renumbered.prim_material = materials_in_order.data();
    renumbered.lights = lights_renumbered.data();
    const v4r_scene_desc *d = &renumbered;
    if (int rc = r->upload(&S.scene_index, scene_index.data(), scene_index.size())) return rc;
    {
        std::vector<float4> wide;
        if (direct) {
            std::vector<uint32_t> media_list;
            for (uint32_t k = 0; k < d->n_prims; ++k)
                if ((d->prims[k].type & V4R_PRIM_TYPE_MASK) == V4R_PRIM_MEDIUM) media_list.push_back(k);
            if (int rc = r->upload(&S.media_direct, media_list.data(), media_list.size())) return rc;
            S.n_media_direct = (uint32_t)media_list.size();
            r->media_direct = media_list;
        }
        const v4r_bvh_node *binary = given->nodes;
        uint32_t n_binary = given->n_nodes;
        if (!keep_host_tree) {
            binary = better.data();
            n_binary = (uint32_t)better.size();
        }
        // Two shapes of the traversed tree: 4-wide with exact child boxes, and 8-wide with the child boxes on an 8-bit grid
        // (half as many dependent node fetches per ray, each step about 1.6 x the vector work, boxes a little larger).
        // Both are uploaded; pick_tree() below keeps the one that does less work for this scene's camera rays.
        std::vector<float4> wide8;
        bool encodable = true;
        wide_depth = collapse_to_wide<4>(d, binary, n_binary, wide, S.n_top_nodes, encodable);
        if (int rc = r->upload(&S.nodes4, wide.data(), wide.size())) return rc;
        r->n_wide_nodes = (uint32_t)(wide.size() / 8);
        if (!keep_host_tree && n_binary > 0) {
            wide_depth8 = collapse_to_wide<8>(d, binary, n_binary, wide8, n_top8, encodable);
            if (encodable) {
                if (int rc = r->upload(&nodes8, wide8.data(), wide8.size())) return rc;
                n_nodes8 = (uint32_t)(wide8.size() / 8);
            }
        }
    }
    if (int rc = r->upload(&S.prims, (const float4 *)d->prims, (size_t)d->n_prims * 3)) return rc;
    std::vector<uint32_t> info(d->n_prims);
    // shapes expected to receive fewer than 32 rays of a pass (their share of the scene's surface area x the pixels)
    std::vector<uint8_t> small(d->n_prims, 0);
    {
        std::vector<double> area(d->n_prims);
        double total = 0.0;
        for (uint32_t i = 0; i < d->n_prims; ++i) total += area[i] = surface_area(d, d->prims[i]);
        const double npix = (double)d->width * (double)d->height;
        for (uint32_t i = 0; i < d->n_prims; ++i) small[i] = total > 0.0 && npix * area[i] < 32.0 * total;
    }
    bool plain_lights = true;
    if (const char *env = std::getenv("V4R_PLAIN_LIGHTS")) plain_lights = std::atoi(env) != 0;
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const v4r_material &mat = d->materials[d->prim_material[i]];
        uint32_t cls = class_of(d, mat);
        info[i] = d->prim_material[i] | (cls << 24) | (small[i] ? PRIM_INFO_SMALL : 0u);
        // a one-colour light on an untransformed rectangle ends its paths inside k_trace (no queue, no shading launch)
        const v4r_prim &pr = d->prims[i];
        const uint32_t kind = pr.type & V4R_PRIM_TYPE_MASK;
        if (plain_lights && cls == CLS_LIGHT && d->textures[mat.tex].kind == V4R_TEX_SOLID && pr.xform == V4R_NO_XFORM &&
            (kind == V4R_PRIM_XY_RECT || kind == V4R_PRIM_YZ_RECT || kind == V4R_PRIM_ZX_RECT)) {
            const uint32_t axis = kind == V4R_PRIM_XY_RECT ? 2u : (kind == V4R_PRIM_YZ_RECT ? 0u : 1u);
            info[i] |= PRIM_INFO_PLAIN_LIGHT | (axis << PRIM_INFO_AXIS_SHIFT) | ((pr.type & V4R_PRIM_POSITIVE) ? PRIM_INFO_POSITIVE : 0u);
        } else {
            r->has_class[cls] = true;
        }
        bool textured = mat.kind == V4R_MAT_LAMBERTIAN || mat.kind == V4R_MAT_DIFFUSE_LIGHT || mat.kind == V4R_MAT_ISOTROPIC;
        if (textured && has_noise(d, mat.tex)) r->class_noise[cls] = true;
    }
    if (int rc = r->upload(&S.prim_info, info.data(), info.size())) return rc;
    std::vector<uint4> shade((size_t)d->n_prims * 2);
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const v4r_material &mat = d->materials[d->prim_material[i]];
        bool textured = mat.kind == V4R_MAT_LAMBERTIAN || mat.kind == V4R_MAT_DIFFUSE_LIGHT || mat.kind == V4R_MAT_ISOTROPIC;
        static_assert(sizeof(v4r_material) == 32 && sizeof(v4r_texture) == 32, "shade records are 32 bytes");
        std::memcpy(&shade[2 * (size_t)i], textured ? (const void *)&d->textures[mat.tex] : (const void *)&mat, 32);
        shade[2 * (size_t)i + 1].w = 0u;  // the records' padding word: SHADE_ALWAYS_FULL, set below
    }
    if (int rc = r->upload(&S.prim_shade, shade.data(), shade.size())) return rc;
    if (int rc = r->upload(&S.xforms, (const float *)d->xforms, (size_t)d->n_xforms * 24)) return rc;
    if (int rc = r->upload(&S.tri, (const float4 *)d->tri_shading, (size_t)d->n_tri_shading * 4)) return rc;
    if (int rc = r->upload(&S.media, (const uint4 *)d->media, (size_t)d->n_media)) return rc;
    if (int rc = r->upload(&S.medium_prims, (const float4 *)d->medium_prims, (size_t)d->n_medium_prims * 3)) return rc;
    if (int rc = r->upload(&S.materials, (const uint32_t *)d->materials, (size_t)d->n_materials * 8)) return rc;
    if (int rc = r->upload(&S.textures, (const uint32_t *)d->textures, (size_t)d->n_textures * 8)) return rc;
    // images: RGB8 -> one RGBA8 word per texel so a fetch is a single aligned load
    std::vector<uint4> images(d->n_images);
    std::vector<uint32_t> texels;
    for (uint32_t i = 0; i < d->n_images; ++i) {
        const v4r_image &im = d->images[i];
        images[i] = make_uint4((uint32_t)texels.size(), im.width, im.height, 0u);
        const uint8_t *src = d->texels + im.offset;
        size_t n = (size_t)im.width * im.height;
        if (texels.size() + n > 0xFFFFFFFFull) return set_error(V4R_E_LIMIT, "more than 2^32 texels");
        for (size_t k = 0; k < n; ++k) texels.push_back(src[3 * k] | (src[3 * k + 1] << 8) | ((uint32_t)src[3 * k + 2] << 16));
    }
    if (int rc = r->upload(&S.images, images.data(), images.size())) return rc;
    if (int rc = r->upload(&S.texels, texels.data(), texels.size())) return rc;
    std::vector<float4> pvec((size_t)d->n_perlin * 256);
    // perm_x, perm_y, perm_z of a table as 16-bit entries: entry i holds perm[i] in its low byte and perm[(i + 1) & 255] in
    // its high byte, the two values a lattice cell needs along that axis, so that a cell costs three loads, not six
    std::vector<uint16_t> pperm((size_t)d->n_perlin * 768);
    for (uint32_t i = 0; i < d->n_perlin; ++i) {
        for (int k = 0; k < 256; ++k) pvec[(size_t)i * 256 + k] = make_float4(d->perlin[i].vec[k][0], d->perlin[i].vec[k][1], d->perlin[i].vec[k][2], 0.0f);
        const uint8_t *axes[3] = {d->perlin[i].perm_x, d->perlin[i].perm_y, d->perlin[i].perm_z};
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 256; ++k) pperm[(size_t)i * 768 + a * 256 + k] = (uint16_t)(axes[a][k] | (axes[a][(k + 1) & 255] << 8));
    }
    if (int rc = r->upload(&S.perlin_vec, pvec.data(), pvec.size())) return rc;
    if (int rc = r->upload(&S.perlin_perm, pperm.data(), pperm.size())) return rc;
    if (int rc = r->upload(&S.lights, d->lights, (size_t)d->n_lights)) return rc;
    S.n_nodes = d->n_nodes;
    r->give_up_lanes = 12;
    r->give_up_segments = 8;
    if (const char *env = std::getenv("V4R_GIVE_UP_LANES")) r->give_up_lanes = (uint32_t)std::max(0, std::min(63, std::atoi(env)));
    if (const char *env = std::getenv("V4R_TILE_PASSES_LOG2")) r->tile_passes_log2 = (uint32_t)std::min(6, std::max(0, std::atoi(env)));
    if (const char *env = std::getenv("V4R_GIVE_UP_SEGMENTS")) r->give_up_segments = (uint32_t)std::max(0, std::atoi(env));
    S.n_prims = d->n_prims;
    S.n_lights = d->n_lights;
    S.flags = d->flags;
    S.packet_rule = d->n_lights != 0 && !(d->flags & (V4R_FLAG_BOUNDED_LIGHT_PDF | V4R_FLAG_PLANE_LIGHT_PDF));
    std::memcpy(S.bg, d->background, sizeof S.bg);
    std::memcpy(S.env, d->environment, sizeof S.env);
    S.cam = d->camera;
    S.width = d->width;
    S.height = d->height;
    S.max_depth = d->max_depth;
    S.antialias = d->antialias;

    r->npix = d->width * d->height;
    if (int rc = r->alloc(&r->stats, 64)) return rc;
    HIP_TRY(hipMemset(r->stats, 0, 512));
    // Which tree: the deterministic camera rays (one through each pixel centre) are traced through both with the work counted;
    // the 8-wide tree is kept when its node steps, priced at WIDE8_STEP_COST steps of the 4-wide tree, and its extra leaf
    // steps come to less.  The counts are exact, so the choice is a function of the scene (V4R_WIDE=4|8 forces one).
    if (nodes8) {
        const uint32_t grid = (uint32_t)n_cu * 4u;
        const uint32_t depth_max = std::max(wide_depth, wide_depth8) + 1u;
        constexpr uint32_t in_lds = (uint32_t)std::min(lds_depth(false), lds_depth(true));  // one slab for both trees' runs
        const size_t spill = depth_max > in_lds ? depth_max - in_lds : 0;
        // (freed on every way out of this block, the early error returns included)
        struct Temporaries {
            uint2 *slab = nullptr;
            uint32_t *d_prim = nullptr;
            float *d_t = nullptr;
            ~Temporaries() {
                (void)hipFree(slab);
                (void)hipFree(d_prim);
                (void)hipFree(d_t);
            }
        } tmp;
        HIP_TRY(hipMalloc((void **)&tmp.slab, std::max<size_t>(256, spill * grid * BLOCK * sizeof(uint2))));
        HIP_TRY(hipMalloc((void **)&tmp.d_prim, (size_t)r->npix * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&tmp.d_t, (size_t)r->npix * sizeof(float)));
        uint2 *slab = tmp.slab;
        uint32_t *d_prim = tmp.d_prim;
        float *d_t = tmp.d_t;
        Queues Q{};
        Q.spill = slab;
        Q.spill_stride = (size_t)grid * BLOCK;
        Q.spill_depth = (uint32_t)spill;
        Q.stats = r->stats;
        unsigned long long work[2][8];
        for (int w8 = 0; w8 < 2; ++w8) {
            SceneDev T = S;
            if (w8) T.nodes4 = nodes8, T.n_top_nodes = n_top8;
            auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, r->stream, T, Q, r->npix, (const float *)nullptr, (const float *)nullptr, 0u, 0u,
                                   d_prim, d_t);
            };
            switch (r->media_mode) {
                case MEDIA_IN_TREE: w8 ? launch(k_trace_rays<MEDIA_IN_TREE, true, true>) : launch(k_trace_rays<MEDIA_IN_TREE, false, true>); break;
                case MEDIA_DIRECT: w8 ? launch(k_trace_rays<MEDIA_DIRECT, true, true>) : launch(k_trace_rays<MEDIA_DIRECT, false, true>); break;
                default: w8 ? launch(k_trace_rays<MEDIA_NONE, true, true>) : launch(k_trace_rays<MEDIA_NONE, false, true>); break;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(r->stream));
            HIP_TRY(hipMemcpy(work[w8], r->stats, sizeof work[w8], hipMemcpyDeviceToHost));
            HIP_TRY(hipMemset(r->stats, 0, 128));
        }
        const double cost4 = (double)work[0][6] + WIDE_LEAF_COST * (double)work[0][7], cost8 = WIDE8_STEP_COST * (double)work[1][6] + WIDE_LEAF_COST * (double)work[1][7];
        r->wide8 = cost8 < cost4;
        if (const char *env = std::getenv("V4R_WIDE")) r->wide8 = std::atoi(env) == 8;
        if (std::getenv("V4R_VERBOSE"))
            std::fprintf(stderr, "v4r: camera rays: 4-wide %llu nodes %llu tests %llu wave steps %llu wave leaves; 8-wide %llu nodes %llu tests %llu wave steps %llu wave leaves -> %d-wide\n",
                         work[0][1], work[0][2], work[0][6], work[0][7], work[1][1], work[1][2], work[1][6], work[1][7], r->wide8 ? 8 : 4);
        if (r->wide8) {
            S.nodes4 = nodes8;
            S.n_top_nodes = n_top8;
            r->n_wide_nodes = n_nodes8;
            wide_depth = wide_depth8;
            // six waves per SIMD for the later segments (TREE_8_WIDE_DENSE): every scene without media
            r->dense = r->media_mode == MEDIA_NONE;
            if (const char *env = std::getenv("V4R_TRACE_WAVES")) r->dense = r->media_mode == MEDIA_NONE && std::atoi(env) == 6;
        }
    }

    // queues
    // Paths per batch.  A launch ends with the few waves that hold its longest rays, a stretch whose length does not depend on
    // the launch's size: measured on the bunny scene, the traversal launches of one 128 M-path batch take 17 % less time than
    // those of two 64 M-path batches, and with three pipelines overlapping a 512-pass 1080p job runs 3.9 % (128 M), 6.8 %
    // (192 M), 6.2 % (256 M) faster than with 64 M.  192 M path slots are 27.6 GB per pipeline, of 288 GB.
    size_t target = 192u << 20;
    if (const char *env = std::getenv("V4R_BATCH_PATHS")) {
        long long v = std::atoll(env);
        if (v > 0) target = (size_t)v;
    }
    r->batch_passes = (uint32_t)std::max<size_t>(1, std::min<size_t>(target / r->npix, 4096));
    if ((uint64_t)r->batch_passes * r->npix > 0x7FFFFFFFull) r->batch_passes = (uint32_t)(0x7FFFFFFFull / r->npix);
    r->batch_passes = cap_counted_passes(r, r->batch_passes);
    if (r->batch_passes >= 16u) r->batch_passes &= ~15u;  // whole groups of passes for the tiled order of the camera rays
    uint32_t n_pipes = 3;
    if (const char *env = std::getenv("V4R_PIPES")) n_pipes = (uint32_t)std::max(1, std::min(4, std::atoi(env)));
    auto with_media_mode = [&](auto f) {  // f(std::integral_constant<int, mode>)
        switch (r->media_mode) {
            case MEDIA_IN_TREE: return f(std::integral_constant<int, MEDIA_IN_TREE>{});
            case MEDIA_DIRECT: return f(std::integral_constant<int, MEDIA_DIRECT>{});
            default: return f(std::integral_constant<int, MEDIA_NONE>{});
        }
    };
    r->grid_trace = with_media_mode([&](auto m) {
        constexpr int M = decltype(m)::value;
        if (r->wide8)
            return std::max(std::max(r->dense && M == MEDIA_NONE ? grid_for(k_trace<false, TRACE_RAYS_PER_THREAD, MEDIA_NONE, false, false, TREE_8_WIDE_DENSE>, n_cu)
                                                                 : grid_for(k_trace<false, TRACE_RAYS_PER_THREAD, M, false, false, TREE_8_WIDE>, n_cu),
                                     grid_for(k_trace<false, TRACE_RAYS_PER_THREAD, M, true, false, TREE_8_WIDE>, n_cu)),
                            grid_for(k_trace_rays<M, true>, n_cu));
        return std::max(std::max(grid_for(k_trace<false, TRACE_RAYS_PER_THREAD, M, false, false, TREE_4_WIDE>, n_cu),
                                 grid_for(k_trace<false, TRACE_RAYS_PER_THREAD, M, true, false, TREE_4_WIDE>, n_cu)),
                        grid_for(k_trace_rays<M, false>, n_cu));
    });
    r->grid_shade = (uint32_t)n_cu * 8u;
    r->grid_stream = (uint32_t)n_cu * 8u;
    for (bool noise : r->class_noise) r->any_noise = r->any_noise || noise;
    {
        uint32_t g = with_media_mode([&](auto m) {
            constexpr int M = decltype(m)::value;
            if (r->wide8) return r->any_noise ? grid_for(k_tail<false, M, true, true>, n_cu) : grid_for(k_tail<false, M, false, true>, n_cu);
            return r->any_noise ? grid_for(k_tail<false, M, true, false>, n_cu) : grid_for(k_tail<false, M, false, false>, n_cu);
        });
        r->grid_tail = std::min(g, r->grid_trace);  // the stack spill slab has one column per thread of a trace launch
    }
    // measured: +2 % (spheres) to +5 % (bunny) with the tail from segment 12; the variants that carry the medium or the
    // Perlin code (3 waves/SIMD) only pay off later (C1 +1.5 % from 24, -3 % from 12)
    r->tail_segment = r->media_mode == MEDIA_IN_TREE || r->any_noise ? 24 : 12;
    if (const char *env = std::getenv("V4R_TAIL_SEGMENT")) r->tail_segment = (uint32_t)std::max(0, std::atoi(env));
    // the tail kernel does not count rays per shape: under the packet-statistics rule it starts after the counted levels
    if (S.packet_rule && r->tail_segment != 0) r->tail_segment = std::max(r->tail_segment, V4R_PACKET_COUNTED_LEVELS);
    // the traversal stack: lds_depth(tree) entries per thread in LDS, the rest of the tree's worst case in a slab in memory
    uint32_t stack_max = wide_depth + 1u;
    const uint32_t in_lds = (uint32_t)lds_depth(r->wide8);
    uint32_t spill_depth = stack_max > in_lds ? stack_max - in_lds : 0;
    r->ctr_bytes = sizeof(SegCtr) * ((size_t)d->max_depth + 1);
    HIP_TRY(hipEventCreateWithFlags(&r->fork, hipEventDisableTiming));
    r->pipes.resize(n_pipes);
    r->max_depth = d->max_depth;
    r->spill_depth = spill_depth;
    for (uint32_t pi = 0; pi < n_pipes; ++pi) {
        Pipe &pipe = r->pipes[pi];
        if (pi == 0) pipe.stream = r->stream;
        else HIP_TRY(hipStreamCreateWithFlags(&pipe.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&pipe.done, hipEventDisableTiming));
    }
    // Queues are sized by the jobs that arrive (ensure_pipe): pipeline 0 starts with room for one pass,
    // which also gives v4r_trace_rays its stack spill slab.
    if (int rc = ensure_pipe(r, r->pipes[0], r->npix, 1)) return rc;
    // v4r_render_noalloc: this device's scratch array, carved into one more pipeline that runs on the public stream
    if (int rc = scratch_of_device(r->device, &r->device_scratch)) return rc;
    r->scratch_ptr = r->device_scratch->ptr;
    r->noalloc.stream = r->stream;
    HIP_TRY(hipEventCreateWithFlags(&r->noalloc.done, hipEventDisableTiming));
    carve_noalloc(r, n_cu);

    // Packet-statistics rule: shapes that fill up at the first level of every pass.  The deterministic camera rays (one
    // through each pixel centre) are traced once; a Lambertian shape that at least 4 x V4R_PACKET_MANY_RAYS of them end on
    // receives V4R_PACKET_MANY_RAYS jittered camera rays in any pass, so at level 0 -- and, a full shape staying full, at
    // every deeper level -- it is full without being counted: no flag to read, no counter to contend for (walls, floors).
    if (S.packet_rule && d->n_prims != 0 && !std::getenv("V4R_NO_ALWAYS_FULL")) {
        uint32_t *d_prim = nullptr;
        float *d_t = nullptr;
        if (int rc = r->alloc(&d_prim, r->npix)) return rc;
        if (int rc = r->alloc(&d_t, r->npix)) return rc;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(r->grid_trace), dim3(BLOCK), 0, r->stream, r->S, r->pipes[0].Q, r->npix, (const float *)nullptr,
                               (const float *)nullptr, 0u, 0u, d_prim, d_t);
        };
        switch (r->media_mode) {
            case MEDIA_IN_TREE: r->wide8 ? launch(k_trace_rays<MEDIA_IN_TREE, true>) : launch(k_trace_rays<MEDIA_IN_TREE, false>); break;
            case MEDIA_DIRECT: r->wide8 ? launch(k_trace_rays<MEDIA_DIRECT, true>) : launch(k_trace_rays<MEDIA_DIRECT, false>); break;
            default: r->wide8 ? launch(k_trace_rays<MEDIA_NONE, true>) : launch(k_trace_rays<MEDIA_NONE, false>); break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(r->stream));
        std::vector<uint32_t> seen(r->npix);
        HIP_TRY(hipMemcpy(seen.data(), d_prim, (size_t)r->npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::vector<uint32_t> pixels(d->n_prims, 0u);
        for (uint32_t p : seen)
            if (p < d->n_prims) pixels[p]++;
        bool any = false;
        for (uint32_t i = 0; i < d->n_prims; ++i) {
            const uint32_t cls = info[i] >> 24;
            if (pixels[i] >= 4u * V4R_PACKET_MANY_RAYS && (cls == CLS_LAMBERT || cls == CLS_LAMBERT_NOISE)) {
                info[i] |= PRIM_INFO_ALWAYS_FULL;
                shade[2 * (size_t)i + 1].w = SHADE_ALWAYS_FULL;
                any = true;
            }
        }
        if (any) {
            HIP_TRY(hipMemcpy(const_cast<uint32_t *>(S.prim_info), info.data(), info.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(const_cast<uint4 *>(S.prim_shade), shade.data(), shade.size() * sizeof(uint4), hipMemcpyHostToDevice));
        }
    }

    if (int rc = r->alloc(&r->accum_own, (size_t)r->npix * 3)) return rc;
    HIP_TRY(hipMemset(r->accum_own, 0, (size_t)r->npix * 3 * sizeof(float)));
    r->accum = r->accum_own;
    if (int rc = r->alloc(&r->tonemap_buf, (size_t)r->npix * 3)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return V4R_OK;
}

// Instantiations of k_trace: work counters x media mode x (segment 0 generates its camera rays) x (second launch).
template <bool COUNT, int MEDIA, bool PRIMARY>
uint32_t launch_trace_as(v4r_renderer *r, Pipe &pipe, uint32_t seg, const PrimaryArgs &A) {
    dim3 grid(pipe.grid_trace), block(BLOCK);
    constexpr int R = TRACE_RAYS_PER_THREAD;
    // the first segments hold nearly all rays: there a wave gives its last few rays up to a second launch
    uint32_t give_up = seg < r->give_up_segments ? r->give_up_lanes : 0u;
    auto launch = [&](auto tree) {
        constexpr int TREE = decltype(tree)::value;
        hipLaunchKernelGGL((k_trace<COUNT, R, MEDIA, PRIMARY, false, TREE>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg, A, give_up);
        if (give_up) hipLaunchKernelGGL((k_trace<COUNT, R, MEDIA, PRIMARY, true, TREE>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg, A, 0u);
    };
    if (!r->wide8) launch(std::integral_constant<int, TREE_4_WIDE>{});
    else if (MEDIA == MEDIA_NONE && !PRIMARY && r->dense) launch(std::integral_constant<int, MEDIA == MEDIA_NONE && !PRIMARY ? TREE_8_WIDE_DENSE : TREE_8_WIDE>{});
    else launch(std::integral_constant<int, TREE_8_WIDE>{});
    return give_up ? 2u : 1u;
}
template <bool COUNT, int MEDIA>
uint32_t launch_trace_media(v4r_renderer *r, Pipe &pipe, uint32_t seg, const PrimaryArgs &A) {
    if (seg == 0) return launch_trace_as<COUNT, MEDIA, true>(r, pipe, seg, A);
    return launch_trace_as<COUNT, MEDIA, false>(r, pipe, seg, A);
}
template <bool COUNT>
uint32_t launch_trace_counting(v4r_renderer *r, Pipe &pipe, uint32_t seg, const PrimaryArgs &A) {
    switch (r->media_mode) {
        case MEDIA_IN_TREE: return launch_trace_media<COUNT, MEDIA_IN_TREE>(r, pipe, seg, A);
        case MEDIA_DIRECT: return launch_trace_media<COUNT, MEDIA_DIRECT>(r, pipe, seg, A);
        default: return launch_trace_media<COUNT, MEDIA_NONE>(r, pipe, seg, A);
    }
}
// Returns the number of kernels launched.
uint32_t launch_trace(v4r_renderer *r, Pipe &pipe, uint32_t seg, const PrimaryArgs &A) {
    return r->counting ? launch_trace_counting<true>(r, pipe, seg, A) : launch_trace_counting<false>(r, pipe, seg, A);
}

template <int CLS>
uint32_t launch_shade(v4r_renderer *r, Pipe &pipe, uint32_t seg) {
    if (!r->has_class[CLS]) return 0u;
    dim3 grid(r->grid_shade), block(BLOCK);
    // Lambertian materials are classed by noise already; metal and dielectric have no texture
    constexpr bool maybe_noise = CLS == CLS_LIGHT || CLS == CLS_ISOTROPIC;
    if (CLS == CLS_LAMBERT_NOISE || (maybe_noise && r->class_noise[CLS])) {
        if (r->counting) hipLaunchKernelGGL((k_shade<CLS, true, true>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
        else hipLaunchKernelGGL((k_shade<CLS, false, true>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
    } else {
        if (r->counting) hipLaunchKernelGGL((k_shade<CLS, true, false>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
        else hipLaunchKernelGGL((k_shade<CLS, false, false>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
    }
    return 1u;
}

template <bool COUNT, int MEDIA>
void launch_tail_as(v4r_renderer *r, Pipe &pipe, uint32_t seg) {
    dim3 grid(pipe.grid_tail), block(BLOCK);
    if (r->wide8) {
        if (r->any_noise) hipLaunchKernelGGL((k_tail<COUNT, MEDIA, true, true>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
        else hipLaunchKernelGGL((k_tail<COUNT, MEDIA, false, true>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
    } else {
        if (r->any_noise) hipLaunchKernelGGL((k_tail<COUNT, MEDIA, true, false>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
        else hipLaunchKernelGGL((k_tail<COUNT, MEDIA, false, false>), grid, block, 0, pipe.stream, r->S, pipe.Q, seg);
    }
}
template <bool COUNT>
void launch_tail_counting(v4r_renderer *r, Pipe &pipe, uint32_t seg) {
    switch (r->media_mode) {
        case MEDIA_IN_TREE: launch_tail_as<COUNT, MEDIA_IN_TREE>(r, pipe, seg); break;
        case MEDIA_DIRECT: launch_tail_as<COUNT, MEDIA_DIRECT>(r, pipe, seg); break;
        default: launch_tail_as<COUNT, MEDIA_NONE>(r, pipe, seg); break;
    }
}
void launch_tail(v4r_renderer *r, Pipe &pipe, uint32_t seg) {
    if (r->counting) launch_tail_counting<true>(r, pipe, seg);
    else launch_tail_counting<false>(r, pipe, seg);
}

int run_batch(v4r_renderer *r, Pipe &pipe, uint64_t seed, uint32_t first_pass, uint32_t pass_stride, uint32_t n_batch_passes, uint32_t row0,
              uint32_t n_rows, bool may_profile = true) {
    const uint32_t band_pix = n_rows * r->S.width, band_offset = row0 * r->S.width;
    uint32_t n_paths = n_batch_passes * band_pix;
    HIP_TRY(hipMemsetAsync(pipe.Q.ctr, 0, r->ctr_bytes, pipe.stream));
    RayOrder order;
    {
        uint32_t lp = 0;
        while (lp < r->tile_passes_log2 && (2u << lp) <= n_batch_passes) ++lp;
        order.lp = lp;
        order.tw_log = (6u - lp + 1u) / 2u;
        order.th_log = (6u - lp) / 2u;
        order.tiles_x = r->S.width >> order.tw_log;
        order.tiled_w = order.tiles_x << order.tw_log;
        order.tiled_h = (n_rows >> order.th_log) << order.th_log;
        order.npix_tiled = order.tiled_w * order.tiled_h;
        order.full_groups = n_batch_passes >> lp;
        order.inv_group = 1.0f / (float)((unsigned long long)band_pix << lp);
        order.inv_tiles_x = 1.0f / (float)std::max(1u, order.tiles_x);
        order.inv_width = 1.0f / (float)r->S.width;
    }
    PrimaryArgs primary{(uint32_t)seed, (uint32_t)(seed >> 32), first_pass, pass_stride, n_paths, band_pix, 1.0f / (float)band_pix, band_offset, order};
    pipe.Q.npix = band_pix;
    pipe.Q.inv_npix = 1.0f / (float)band_pix;
#ifdef V4R_DEBUG_CHECKS
    // proof that a failed check is reported: the kernels are told a smaller capacity than the queues have
    const uint32_t true_cap = pipe.Q.cap;
    if (const char *env = std::getenv("V4R_DEBUG_CAP")) pipe.Q.cap = (uint32_t)std::max(0, std::atoi(env));
    struct Restore {
        uint32_t &cap, value;
        ~Restore() { cap = value; }
    } restore{pipe.Q.cap, true_cap};
#endif
    pipe.Q.flags_offset = n_batch_passes * r->S.n_prims;
    if (r->S.packet_rule && n_batch_passes > pipe.count_passes) return set_error(V4R_E_INVALID, "internal: batch larger than the pipeline's ray counters");
    // segments 0 .. n_wavefront-1 run as one trace launch (or two) and one shading launch per class; one k_tail finishes the rest
    const uint32_t n_wavefront = r->tail_segment >= 1u && r->tail_segment < r->S.max_depth ? r->tail_segment : r->S.max_depth;
    // v4r_profile_*: a pair of HIP events on this pipeline's stream around a group of launches
    auto bracket_begin = [&](int group, v4r_renderer::EventPair *&pair) -> int {
        pair = nullptr;
        if (!may_profile || !r->profiling || (r->profiling == 1 && group != V4R_PROFILE_TRACE)) return V4R_OK;
        if (r->ev_used == r->ev_pool.size()) {
            v4r_renderer::EventPair e{};
            HIP_TRY(hipEventCreate(&e.start));
            HIP_TRY(hipEventCreate(&e.stop));
            r->ev_pool.push_back(e);
        }
        pair = &r->ev_pool[r->ev_used++];
        pair->group = group;
        pair->launches = 0;
        HIP_TRY(hipEventRecord(pair->start, pipe.stream));
        return V4R_OK;
    };
    auto bracket_end = [&](v4r_renderer::EventPair *pair, uint32_t launches) -> int {
        if (!pair) return V4R_OK;
        pair->launches = launches;
        HIP_TRY(hipEventRecord(pair->stop, pipe.stream));
        return V4R_OK;
    };
    for (uint32_t seg = 0; seg <= n_wavefront && seg < r->S.max_depth; ++seg) {
        const bool tail = seg == n_wavefront;
        v4r_renderer::EventPair *pair = nullptr;
        if (int rc = bracket_begin(V4R_PROFILE_TRACE, pair)) return rc;
        uint32_t launches = 1;
        // the counters start every counted level at zero (deeper levels reuse the counts of the last counted one); the "full"
        // flags behind them are cleared once per batch: a shape that filled up at one level counts as full at the deeper ones
        if (!tail && r->S.packet_rule && seg < V4R_PACKET_COUNTED_LEVELS)
        {
            const size_t n_counters = (size_t)n_batch_passes * r->S.n_prims;
            if (seg == 0) HIP_TRY(hipMemsetAsync(pipe.Q.shape_rays, 0, 2 * n_counters * sizeof(uint32_t), pipe.stream));
            else hipLaunchKernelGGL(k_restart_counters, dim3(r->grid_stream), dim3(BLOCK), 0, pipe.stream, pipe.Q.shape_rays, n_counters);
        }
        if (tail) launch_tail(r, pipe, seg);  // timed with the trace launches: nearly all of it is traversal
        else launches = launch_trace(r, pipe, seg, primary);
        if (int rc = bracket_end(pair, launches)) return rc;
        if (tail) break;
        if (int rc = bracket_begin(V4R_PROFILE_SHADE, pair)) return rc;
        launches = launch_shade<CLS_LIGHT>(r, pipe, seg);
        launches += launch_shade<CLS_LAMBERT>(r, pipe, seg);
        launches += launch_shade<CLS_LAMBERT_NOISE>(r, pipe, seg);
        launches += launch_shade<CLS_METAL>(r, pipe, seg);
        launches += launch_shade<CLS_DIELECTRIC>(r, pipe, seg);
        launches += launch_shade<CLS_ISOTROPIC>(r, pipe, seg);
        if (int rc = bracket_end(pair, launches)) return rc;
    }
    if (r->S.max_depth == 0)
        hipLaunchKernelGGL(k_fill_rad, dim3(r->grid_stream), dim3(BLOCK), 0, pipe.stream, pipe.Q, n_paths, r->S.env[0], r->S.env[1], r->S.env[2]);
    // accumulate in batch order: this batch's k_resolve runs after the previous batch's
    if (r->last_resolve && r->last_resolve != pipe.done) HIP_TRY(hipStreamWaitEvent(pipe.stream, r->last_resolve, 0));
    v4r_renderer::EventPair *resolve_pair = nullptr;
    if (int rc = bracket_begin(V4R_PROFILE_RESOLVE, resolve_pair)) return rc;
    hipLaunchKernelGGL(k_resolve, dim3(r->grid_stream), dim3(BLOCK), 0, pipe.stream, pipe.Q, r->accum + 3 * (size_t)band_offset, band_pix,
                       n_batch_passes);
    if (int rc = bracket_end(resolve_pair, 1u)) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(pipe.done, pipe.stream));
    r->last_resolve = pipe.done;
    pipe.used = true;
    return V4R_OK;
}

// Checked build: wait for the job and turn failed device-side bounds checks into an error.
int report_checks(v4r_renderer *r) {
#ifdef V4R_DEBUG_CHECKS
    // (a stream that is being captured into a graph cannot be waited for: the checks are then read by the next call outside a capture)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(r->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return V4R_OK;
    HIP_TRY(hipStreamSynchronize(r->stream));
    unsigned long long failed = 0;
    HIP_TRY(hipMemcpy(&failed, r->stats + 8, sizeof failed, hipMemcpyDeviceToHost));
    if (failed) {
        HIP_TRY(hipMemset(r->stats + 8, 0, sizeof failed));
        static const char *names[] = {"stack overflow", "stack underflow", "queue index", "queue length", "ray counter", "path slot", "next ray",
                                      "a wave waited too long for its workgroup's next pool of rays"};
        std::string what;
        for (int k = 0; k < 8; ++k)
            if (failed >> k & 1ull) what += std::string(what.empty() ? "" : ", ") + names[k];
        return set_error(V4R_E_HIP, "device bounds checks failed: " + what);
    }
#else
    (void)r;
#endif
    return V4R_OK;
}

// Work queued on the public stream so far happens before anything the other pipelines do next.
int fork_pipes(v4r_renderer *r) {
    if (r->pipes.size() < 2) return V4R_OK;
    HIP_TRY(hipEventRecord(r->fork, r->stream));
    for (size_t i = 1; i < r->pipes.size(); ++i) HIP_TRY(hipStreamWaitEvent(r->pipes[i].stream, r->fork, 0));
    return V4R_OK;
}
// Everything the pipelines have queued happens before whatever is queued on the public stream next.
int join_pipes(v4r_renderer *r) {
    // a pipeline's last queued operation is always its `done` record (after k_resolve)
    for (size_t i = 1; i < r->pipes.size(); ++i)
        if (r->pipes[i].used) HIP_TRY(hipStreamWaitEvent(r->stream, r->pipes[i].done, 0));
    return V4R_OK;
}

}  // namespace

extern "C" {

uint32_t v4r_abi_version(void) { return V4R_ABI_VERSION; }
const char *v4r_last_error(void) { return g_error.c_str(); }

int v4r_renderer_create(const v4r_scene_desc *desc, int device, v4r_renderer **out) {
    if (!out) return set_error(V4R_E_INVALID, "NULL output pointer");
    *out = nullptr;
    uint32_t depth = 0;
    if (int rc = validate(desc, depth)) return rc;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return set_error(V4R_E_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= count) return set_error(V4R_E_NO_DEVICE, "HIP device index out of range");
    v4r_renderer *r = new v4r_renderer();
    r->device = device;
    int rc = build(r, desc, depth);
    if (rc != V4R_OK) {
        std::string keep = g_error;
        v4r_renderer_destroy(r);
        g_error = keep;
        return rc;
    }
    *out = r;
    return V4R_OK;
}

void v4r_renderer_destroy(v4r_renderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    for (auto &p : r->pipes)
        if (p.stream) (void)hipStreamSynchronize(p.stream);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    for (size_t i = 1; i < r->pipes.size(); ++i)
        if (r->pipes[i].stream) (void)hipStreamDestroy(r->pipes[i].stream);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    for (auto &p : r->pipes)
        if (p.done) (void)hipEventDestroy(p.done);
    if (r->noalloc.done) (void)hipEventDestroy(r->noalloc.done);
    if (r->fork) (void)hipEventDestroy(r->fork);
    for (auto &p : r->pipes)
        for (void *q : p.allocations) (void)hipFree(q);
    for (void *p : r->allocations) (void)hipFree(p);
    for (auto &e : r->ev_pool) {
        (void)hipEventDestroy(e.start);
        (void)hipEventDestroy(e.stop);
    }
    delete r;
}

int v4r_render_rows(v4r_renderer *r, uint64_t seed, uint32_t first_pass, uint32_t pass_stride, uint32_t n_passes, uint32_t row0,
                    uint32_t n_rows) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    if (n_rows == 0 || row0 >= r->S.height || n_rows > r->S.height - row0) return set_error(V4R_E_INVALID, "row band outside the image");
    if (pass_stride == 0) pass_stride = 1;
    HIP_TRY(hipSetDevice(r->device));
    if (int rc = fork_pipes(r)) return rc;
    const uint32_t band_pix = n_rows * r->S.width;
    // passes per batch for this band: the same number of path slots as a full-image batch
    uint32_t batch = (uint32_t)std::min<uint64_t>((uint64_t)r->batch_passes * r->npix / band_pix, 4096u);
    batch = cap_counted_passes(r, batch);
    if (batch >= 16u) batch &= ~15u;
    batch = std::max(1u, batch);
    // A job of fewer batches than pipelines is cut into one piece per pipeline (multiples of 16 passes, the
    // camera-ray tile) so that the pieces overlap: measured +12 % for a 32-pass 1080p job and +6 % for 128 passes
    // of 800x800; jobs of many batches are better off with few large ones.
    // profile mode 2 keeps every batch on pipeline 0, so that kernels do not overlap and the event pairs time them alone
    const uint32_t n_pipes = r->profiling == 2 ? 1u : (uint32_t)r->pipes.size();
    if (r->profiling == 2) r->next_pipe = 0;
    if ((n_passes + batch - 1u) / batch < n_pipes) {
        uint32_t share = (n_passes + n_pipes - 1u) / n_pipes;
        if (share >= 16u && ((share + 15u) & ~15u) * (n_pipes - 1u) < n_passes) share = (share + 15u) & ~15u;  // whole tiles, if every pipeline still gets a piece
        batch = std::max(1u, std::min(batch, share));
    } else if (n_pipes > 1u) {
        // larger jobs: as many batches as a multiple of the pipelines, all of about the same size, so that the pipelines
        // finish together (4 batches on 3 pipelines would leave two of them idle for a quarter of the job)
        uint32_t n_batches = (n_passes + batch - 1u) / batch;
        n_batches = (n_batches + n_pipes - 1u) / n_pipes * n_pipes;
        const uint32_t even = (n_passes + n_batches - 1u) / n_batches;  // (a few passes outside whole 16-pass tiles cost less than uneven batches)
        batch = std::max(1u, std::min(batch, even));
    }
    uint32_t done = 0;
    while (done < n_passes) {
        uint32_t nb = std::min(batch, n_passes - done);
        Pipe &pipe = r->pipes[r->next_pipe];
        r->next_pipe = (r->next_pipe + 1) % n_pipes;
        if (int rc = ensure_pipe(r, pipe, (size_t)nb * band_pix, nb)) return rc;
        if (int rc = run_batch(r, pipe, seed, first_pass + done * pass_stride, pass_stride, nb, row0, n_rows)) return rc;
        done += nb;
    }
    if (int rc = join_pipes(r)) return rc;
    if (n_rows == r->S.height) r->n_passes += n_passes;
    return report_checks(r);
}

int v4r_render(v4r_renderer *r, uint64_t seed, uint32_t first_pass, uint32_t pass_stride, uint32_t n_passes) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    return v4r_render_rows(r, seed, first_pass, pass_stride, n_passes, 0, r->S.height);
}

int v4r_render_noalloc(v4r_renderer *r, uint64_t seed, uint32_t first_pass, uint32_t pass_stride, uint32_t n_passes) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    Pipe &pipe = r->noalloc;
    if (pipe.Q.cap == 0)
        return set_error(V4R_E_LIMIT, "v4r_render_noalloc: in the " + std::to_string(V4R_NOALLOC_SCRATCH_BYTES >> 20) + " MiB scratch, " +
                                          (r->noalloc_why.empty() ? std::string("no room is left for path slots") : r->noalloc_why));
    if (r->npix > pipe.Q.cap)
        return set_error(V4R_E_LIMIT, "v4r_render_noalloc: one pass of " + std::to_string(r->S.width) + " x " + std::to_string(r->S.height) + " = " +
                                          std::to_string(r->npix) + " paths exceeds the " + std::to_string(pipe.Q.cap) +
                                          " path slots of the scratch for this scene (use v4r_render)");
    if (pass_stride == 0) pass_stride = 1;
    HIP_TRY(hipSetDevice(r->device));
    uint32_t batch = std::min<uint32_t>(pipe.Q.cap / r->npix, 4096u);
    if (r->S.packet_rule) batch = std::min(batch, NOALLOC_COUNTED_PASSES);
    if (batch >= 16u) batch &= ~15u;
    // One pipeline on the public stream: batches run one after the other, ordered after whatever the handle queued
    // before (v4r_render ends by joining its pipelines into this stream) and before whatever it queues next -- and after
    // the latest v4r_render_noalloc of any other handle on this GPU, which used the same scratch.
    DeviceScratch &shared = *r->device_scratch;
    std::lock_guard<std::mutex> turn(shared.mu);
    if (shared.used) HIP_TRY(hipStreamWaitEvent(r->stream, shared.last_use, 0));
    int rc = V4R_OK;
    for (uint32_t done = 0; done < n_passes && rc == V4R_OK;) {
        uint32_t nb = std::min(batch, n_passes - done);
        rc = run_batch(r, pipe, seed, first_pass + done * pass_stride, pass_stride, nb, 0, r->S.height, false);
        done += nb;
    }
    // (recorded on the error path too: whatever was queued must finish before the next user of the scratch starts)
    if (hipEventRecord(shared.last_use, r->stream) == hipSuccess) shared.used = true;
    if (rc != V4R_OK) return rc;
    r->n_passes += n_passes;
    return report_checks(r);
}

int v4r_accum_bind(v4r_renderer *r, float *device_hwc) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->accum = device_hwc ? device_hwc : r->accum_own;
    return V4R_OK;
}

int v4r_reduce(v4r_renderer *r, void *rccl_comm, int root) {
    if (!r || !rccl_comm) return set_error(V4R_E_INVALID, "NULL argument");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    // ncclResult_t ncclReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t)
    using reduce_fn = int (*)(const void *, void *, size_t, int, int, int, void *, hipStream_t);
    using version_fn = int (*)(int *);
    struct Rccl {
        reduce_fn reduce = nullptr;
        int version = 0;  // ncclGetVersion's code; 0: not known
    };
    static const Rccl rccl = [] {
        Rccl f;
        void *lib = nullptr;
        void *sym = dlsym(RTLD_DEFAULT, "ncclReduce");  // the RCCL the caller's communicator came from, if it is loaded globally
        if (!sym && (lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL))) sym = dlsym(lib, "ncclReduce");
        f.reduce = (reduce_fn)sym;
        void *ver = lib ? dlsym(lib, "ncclGetVersion") : dlsym(RTLD_DEFAULT, "ncclGetVersion");
        if (ver && ((version_fn)ver)(&f.version) != 0) f.version = 0;
        return f;
    }();
    if (!rccl.reduce) return set_error(V4R_E_UNSUPPORTED, "ncclReduce not found: load RCCL (librccl.so.1) before calling v4r_reduce");
    // The two enumerators passed below are written out here (the library does not include rccl.h): ncclFloat32 = 7 and
    // ncclSum = 0 in every NCCL / RCCL 2.x.  A library that does not report a 2.x version is not called.
    if (rccl.version < 2000 || rccl.version >= 30000)
        return set_error(V4R_E_UNSUPPORTED, "v4r_reduce needs an RCCL that reports version 2.x through ncclGetVersion (got code " + std::to_string(rccl.version) + ")");
    const reduce_fn reduce = rccl.reduce;
    HIP_TRY(hipSetDevice(r->device));
    constexpr int nccl_float32 = 7, nccl_sum = 0;  // rccl.h: ncclFloat32, ncclSum
    int rc = reduce(r->accum, r->accum, (size_t)r->npix * 3, nccl_float32, nccl_sum, root, rccl_comm, r->stream);
    if (rc != 0) return set_error(V4R_E_HIP, "ncclReduce failed with ncclResult_t " + std::to_string(rc));
    return V4R_OK;
}

int v4r_accum_clear(v4r_renderer *r) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipMemsetAsync(r->accum, 0, (size_t)r->npix * 3 * sizeof(float), r->stream));
    r->n_passes = 0;
    return V4R_OK;
}

int v4r_accum_read(v4r_renderer *r, float *host_hwc, uint32_t *n_passes) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (host_hwc) HIP_TRY(hipMemcpy(host_hwc, r->accum, (size_t)r->npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (n_passes) *n_passes = r->n_passes;
    return V4R_OK;
}

int v4r_accum_write(v4r_renderer *r, const float *host_hwc, uint32_t n_passes) {
    if (!r || !host_hwc) return set_error(V4R_E_INVALID, "NULL argument");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpyAsync(r->accum, host_hwc, (size_t)r->npix * 3 * sizeof(float), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));  // the caller may reuse host_hwc; later passes are ordered after the copy
    r->n_passes = n_passes;
    return V4R_OK;
}

int v4r_accum_device(v4r_renderer *r, float **device_hwc, uint64_t *n_floats) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    if (device_hwc) *device_hwc = r->accum;
    if (n_floats) *n_floats = (uint64_t)r->npix * 3;
    return V4R_OK;
}

int v4r_tonemap_u8(v4r_renderer *r, uint32_t n_passes, uint8_t *host_rgb) {
    if (!r || !host_rgb) return set_error(V4R_E_INVALID, "NULL argument");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    if (n_passes == 0) return set_error(V4R_E_INVALID, "no passes to average");
    HIP_TRY(hipSetDevice(r->device));
    size_t n = (size_t)r->npix * 3;
    hipLaunchKernelGGL(k_tonemap, dim3(r->grid_stream), dim3(BLOCK), 0, r->stream, r->accum, n, 1.0f / (float)n_passes, r->tonemap_buf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(host_rgb, r->tonemap_buf, n, hipMemcpyDeviceToHost));
    return V4R_OK;
}

static int trace_common(v4r_renderer *r, uint32_t n, const float *host_origin, const float *host_dir, uint64_t seed, uint32_t *host_prim_id,
                        float *host_t) {
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    if (n == 0) return V4R_OK;
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    uint32_t *d_p = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_o);
        (void)hipFree(d_d);
        (void)hipFree(d_t);
        (void)hipFree(d_p);
    };
    hipError_t e = hipSuccess;
    if (host_origin) {
        e = hipMalloc((void **)&d_o, (size_t)n * 3 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&d_d, (size_t)n * 3 * sizeof(float));
        if (e == hipSuccess) e = hipMemcpyAsync(d_o, host_origin, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, r->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_d, host_dir, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, r->stream);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&d_t, (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&d_p, (size_t)n * sizeof(uint32_t));
    if (e == hipSuccess) {
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(r->grid_trace), dim3(BLOCK), 0, r->stream, r->S, r->pipes[0].Q, n, d_o, d_d, (uint32_t)seed,
                               (uint32_t)(seed >> 32), d_p, d_t);
        };
        switch (r->media_mode) {
            case MEDIA_IN_TREE: r->wide8 ? launch(k_trace_rays<MEDIA_IN_TREE, true>) : launch(k_trace_rays<MEDIA_IN_TREE, false>); break;
            case MEDIA_DIRECT: r->wide8 ? launch(k_trace_rays<MEDIA_DIRECT, true>) : launch(k_trace_rays<MEDIA_DIRECT, false>); break;
            default: r->wide8 ? launch(k_trace_rays<MEDIA_NONE, true>) : launch(k_trace_rays<MEDIA_NONE, false>); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    if (e == hipSuccess && host_prim_id) {
        e = hipMemcpy(host_prim_id, d_p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess)  // the device numbers the primitives in the order of its tree's leaves
            for (uint32_t i = 0; i < n; ++i)
                if (host_prim_id[i] < r->scene_index.size()) host_prim_id[i] = r->scene_index[host_prim_id[i]];
    }
    if (e == hipSuccess && host_t) e = hipMemcpy(host_t, d_t, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return set_error(V4R_E_HIP, std::string("trace: ") + hipGetErrorString(e));
    return V4R_OK;
}

int v4r_trace_primary(v4r_renderer *r, uint32_t *host_prim_id, float *host_t) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    return trace_common(r, r->npix, nullptr, nullptr, 0, host_prim_id, host_t);
}

int v4r_trace_rays(v4r_renderer *r, uint32_t n, const float *host_origin, const float *host_dir, uint64_t seed,
                   uint32_t *host_prim_id, float *host_t) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    if (n && (!host_origin || !host_dir)) return set_error(V4R_E_INVALID, "NULL ray arrays");
    return trace_common(r, n, host_origin, host_dir, seed, host_prim_id, host_t);
}

int v4r_tree_read(v4r_renderer *r, v4r_tree_info *info, void *nodes, uint32_t *scene_index, uint32_t *media_direct) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    if (info) {
        std::memset(info, 0, sizeof *info);
        info->width = r->wide8 ? 8u : 4u;
        info->n_nodes = r->S.n_nodes == 0 ? 0u : r->n_wide_nodes;
        info->n_top_nodes = r->S.n_top_nodes;
        info->n_prims = r->S.n_prims;
        info->n_media_direct = r->S.n_media_direct;
        info->media_in_tree = r->media_mode == MEDIA_IN_TREE ? 1u : 0u;
    }
    if (nodes && r->S.n_nodes != 0 && r->n_wide_nodes != 0) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        HIP_TRY(hipMemcpy(nodes, r->S.nodes4, (size_t)r->n_wide_nodes * 128, hipMemcpyDeviceToHost));
    }
    if (scene_index && !r->scene_index.empty()) std::memcpy(scene_index, r->scene_index.data(), r->scene_index.size() * sizeof(uint32_t));
    if (media_direct && !r->media_direct.empty()) std::memcpy(media_direct, r->media_direct.data(), r->media_direct.size() * sizeof(uint32_t));
    return V4R_OK;
}

int v4r_stats_enable(v4r_renderer *r, int on) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    r->counting = on != 0;
    return V4R_OK;
}

int v4r_stats_read(v4r_renderer *r, v4r_stats *out, int reset) {
    if (!r || !out) return set_error(V4R_E_INVALID, "NULL argument");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    unsigned long long h[10];
    HIP_TRY(hipMemcpy(h, r->stats, sizeof h, hipMemcpyDeviceToHost));
    out->segments = h[0];
    out->nodes = h[1];
    out->prims = h[2];
    out->paths = h[3];
    out->shaded = h[4];
    out->texel_fetches = h[5];
    out->wave_node_steps = h[6];
    out->wave_leaf_steps = h[7];
    out->ended_on_light = h[9];
#ifdef V4R_PHASE_CLOCKS
    {
        unsigned long long c[24];
        HIP_TRY(hipMemcpy(c, r->stats + 16, sizeof c, hipMemcpyDeviceToHost));
        static const char *kind[3] = {"camera rays", "later segments", "second launches"};
        for (int k = 0; k < 3; ++k)
            std::fprintf(stderr, "v4r phase clocks, %s: wave-rays %llu; cycles per wave-ray: fetch %.0f, node steps %.0f, leaf phases %.0f, traverse %.0f, "
                         "end of ray %.0f, chunk %.0f, wave lifetime %.0f\n", kind[k], c[8 * k + 7], (double)c[8 * k] / std::max(1ull, c[8 * k + 7]),
                         (double)c[8 * k + 1] / std::max(1ull, c[8 * k + 7]), (double)c[8 * k + 2] / std::max(1ull, c[8 * k + 7]),
                         (double)c[8 * k + 3] / std::max(1ull, c[8 * k + 7]), (double)c[8 * k + 4] / std::max(1ull, c[8 * k + 7]),
                         (double)c[8 * k + 5] / std::max(1ull, c[8 * k + 7]), (double)c[8 * k + 6] / std::max(1ull, c[8 * k + 7]));
        if (reset) HIP_TRY(hipMemset(r->stats + 16, 0, sizeof c));
    }
#endif
    if (reset) {
        HIP_TRY(hipMemsetAsync(r->stats, 0, 64, r->stream));  // ordered before whatever is rendered next
        HIP_TRY(hipMemsetAsync(r->stats + 9, 0, 8, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream));
    }
    return V4R_OK;
}

// Fold the recorded event pairs into the running totals (every pipeline must be idle).
static int profile_collect(v4r_renderer *r) {
    for (size_t i = 0; i < r->ev_used; ++i) {
        const auto &e = r->ev_pool[i];
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, e.start, e.stop));
        r->group_ms[e.group] += ms;
        r->group_launches[e.group] += e.launches;
        r->group_pairs[e.group]++;
    }
    r->ev_used = 0;
    return V4R_OK;
}

int v4r_profile_enable(v4r_renderer *r, int mode) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    if (mode < 0 || mode > 2) return set_error(V4R_E_INVALID, "profile mode must be 0, 1 or 2");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (int rc = profile_collect(r)) return rc;
    r->profiling = mode;
    return V4R_OK;
}

int v4r_profile_read(v4r_renderer *r, double *trace_ms, uint64_t *trace_launches, int reset) {
    double ms[3];
    uint64_t launches[3], pairs[3];
    if (int rc = v4r_profile_read_groups(r, ms, launches, pairs, reset)) return rc;
    if (trace_ms) *trace_ms = ms[V4R_PROFILE_TRACE];
    if (trace_launches) *trace_launches = pairs[V4R_PROFILE_TRACE];
    return V4R_OK;
}

int v4r_profile_read_groups(v4r_renderer *r, double ms[3], uint64_t launches[3], uint64_t pairs[3], int reset) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (int rc = profile_collect(r)) return rc;
    for (int g = 0; g < 3; ++g) {
        if (ms) ms[g] = r->group_ms[g];
        if (launches) launches[g] = r->group_launches[g];
        if (pairs) pairs[g] = r->group_pairs[g];
        if (reset) {
            r->group_ms[g] = 0.0;
            r->group_launches[g] = r->group_pairs[g] = 0;
        }
    }
    return V4R_OK;
}

int v4r_synchronize(v4r_renderer *r) {
    if (!r) return set_error(V4R_E_INVALID, "NULL renderer");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return V4R_OK;
}

int v4r_stream(v4r_renderer *r, void **hip_stream) {
    if (!r || !hip_stream) return set_error(V4R_E_INVALID, "NULL argument");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    *hip_stream = (void *)r->stream;
    return V4R_OK;
}

int v4r_pass_count(v4r_renderer *r, uint32_t *n_passes) {
    if (!r || !n_passes) return set_error(V4R_E_INVALID, "NULL argument");
    std::lock_guard<std::recursive_mutex> hold(r->mu);
    *n_passes = r->n_passes;
    return V4R_OK;
}

}  // extern "C"
