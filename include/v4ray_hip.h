/*
 * v4ray_hip.h -- C ABI of the gfx950 wavefront path tracer.
 *
 * This is the boundary a host (the reference's Rust crate, this repository's C++ host, or
 * Python/ctypes) binds.  The host flattens its scene into the plain SoA tables of
 * `v4r_scene_desc`; `v4r_renderer_create` copies them into HBM (the caller keeps ownership of
 * every pointer it passes) and the render entry points run hand-written HIP kernels on them.
 *
 * Every function returns 0 (V4R_OK) or a negative V4R_E_* code and never aborts the process;
 * `v4r_last_error()` returns a thread-local description of the last failure.  Calls on one
 * renderer handle may come from several threads: each entry point holds the handle's lock for its
 * whole call and queues its device work on the handle's HIP stream (a sequence such as clear +
 * render + read that must not interleave with another thread's is the caller's to bracket);
 * different handles are independent.
 *
 * Each declaration cites the reference interface (file:line under the reference tree) it replaces.
 */
#ifndef V4RAY_HIP_H
#define V4RAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V4R_ABI_VERSION 1u

/* ---- status codes ------------------------------------------------------------------------- */
#define V4R_OK 0
#define V4R_E_INVALID (-1)     /* bad argument / malformed scene table                         */
#define V4R_E_NO_DEVICE (-2)   /* no usable gfx950 device / HIP runtime failure at set-up       */
#define V4R_E_HIP (-3)         /* a HIP call failed (message in v4r_last_error)                 */
#define V4R_E_UNSUPPORTED (-4) /* scene uses a feature the device path does not implement       */
#define V4R_E_LIMIT (-5)       /* a size exceeds a documented limit (e.g. more than 2^28 - 1 primitives) */

/* ---- geometry tables ---------------------------------------------------------------------- */

/* Leaf marker in v4r_bvh_node::child*: MSB set => low 31 bits are a primitive index. */
#define V4R_LEAF_BIT 0x80000000u

/* One interior BVH node, 64 bytes, holding BOTH children's boxes (as src/bvh/bvh.rs:13-16 does).
 * Leaves of the reference tree (bvh.rs:10-12, one shape each) are folded into the child words. */
typedef struct v4r_bvh_node {
    float lo0[3], hi0[3]; /* AABB of child 0 (src/bvh/aabb.rs:9-12) */
    float lo1[3], hi1[3]; /* AABB of child 1 */
    uint32_t child0, child1;
    uint32_t pad[2];
} v4r_bvh_node;

/* Primitive kinds (src/hittable/{sphere,aa_rect,triangle,constant_medium}.rs). */
enum {
    V4R_PRIM_SPHERE = 0,  /* f = {cx,cy,cz,r}                                   sphere.rs:27-30   */
    V4R_PRIM_XY_RECT = 1, /* f = {x0,x1,y0,y1,k(z)}                             aa_rect.rs:190    */
    V4R_PRIM_YZ_RECT = 2, /* f = {y0,y1,z0,z1,k(x)}                             aa_rect.rs:191    */
    V4R_PRIM_ZX_RECT = 3, /* f = {z0,z1,x0,x1,k(y)}                             aa_rect.rs:192    */
    V4R_PRIM_TRIANGLE = 4,/* f = {v0.xyz, edge12.xyz, edge13.xyz}; aux -> tri_shading  triangle.rs:11-18 */
    V4R_PRIM_MEDIUM = 5   /* aux -> media[]                                     constant_medium.rs:11-14 */
};
#define V4R_PRIM_TYPE_MASK 0xFFu
#define V4R_PRIM_POSITIVE 0x100u /* rect `positive` flag (aa_rect.rs:87) */
#define V4R_NO_XFORM 0xFFFFFFFFu

/* 48-byte primitive record; prims[i] pairs with prim_material[i] exactly like
 * Scene::hittables[i] / Scene::materials[i] (src/scene.rs:20-23). */
typedef struct v4r_prim {
    float f[9];
    uint32_t type;  /* V4R_PRIM_* | flags */
    uint32_t xform; /* index into xforms[] or V4R_NO_XFORM (TransformHittable, transform.rs:8-14) */
    uint32_t aux;
} v4r_prim;

/* Affine wrapper, row-major 3x3 + translation, forward and inverse (transform.rs:8-14). */
typedef struct v4r_xform {
    float fwd[9], fwd_t[3];
    float inv[9], inv_t[3];
} v4r_xform;

/* Per-triangle shading data (triangle.rs:13-14): vertex normals and uvs. 64 bytes. */
typedef struct v4r_tri_shading {
    float n[3][3];
    float uv[3][2];
    float pad;
} v4r_tri_shading;

/* ConstantMedium (constant_medium.rs:11-22): boundary = medium_prims[first .. first+count)
 * (count > 1 is a HittableGroup boundary, hittables/group.rs:58-67). */
typedef struct v4r_medium {
    uint32_t first, count;
    float neg_inv_density;
    uint32_t pad;
} v4r_medium;

/* ---- appearance tables -------------------------------------------------------------------- */
enum {
    V4R_MAT_LAMBERTIAN = 0,   /* tex                 material/lambertian.rs:21-23 */
    V4R_MAT_METAL = 1,        /* albedo, fuzz        material/metal.rs:15-18      */
    V4R_MAT_DIELECTRIC = 2,   /* ir                  material/dielectric.rs:13-15 */
    V4R_MAT_DIFFUSE_LIGHT = 3,/* tex (emit)          material/diffuse_light.rs:8-10 */
    V4R_MAT_ISOTROPIC = 4     /* tex (albedo)        material/isotropic.rs:11-13  */
};
typedef struct v4r_material {
    uint32_t kind;
    uint32_t tex;
    float albedo[3];
    float fuzz;
    float ir;
    uint32_t pad;
} v4r_material;

enum {
    V4R_TEX_SOLID = 0,   /* color                          texture/solid_color.rs:11-13 */
    V4R_TEX_CHECKER = 1, /* a = odd tex, b = even tex, s = density   texture/checker.rs:15-19 */
    V4R_TEX_IMAGE = 2,   /* a = image index                texture/image.rs:9-11        */
    V4R_TEX_NOISE = 3    /* a = perlin table, b = depth, s = scale   texture/noise.rs:144-148 */
};
typedef struct v4r_texture {
    uint32_t kind;
    float color[3];
    float s;
    uint32_t a, b;
    uint32_t pad;
} v4r_texture;

/* 8-bit RGB image, rows top to bottom, 3 bytes per texel, at texels + offset. */
typedef struct v4r_image {
    uint64_t offset;
    uint32_t width, height;
} v4r_image;

/* One Perlin table set (texture/noise.rs:13-18). */
typedef struct v4r_perlin {
    float vec[256][3];
    uint8_t perm_x[256], perm_y[256], perm_z[256];
} v4r_perlin;

/* Camera state after Camera::new (src/camera.rs:65-75). */
typedef struct v4r_camera {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius, time0, time1;
} v4r_camera;

/* Scene flags. */
/* The pdf value of a rectangle / triangle light for a direction that misses it.  In the reference a SIMD lane gets
 * the un-clipped plane density when some lane of its 8-ray packet hits the light and 0 when none does
 * (aa_rect.rs:158-179, triangle.rs:110-131); packets hold rays of one pass and recursion level that hit the same
 * shape (renderer.rs:189-217).  Default (no flag): that rule in expectation -- the size of the ray's packet follows
 * from the number of rays of its pass and level that hit its shape (counted on the device for the first
 * V4R_PACKET_COUNTED_LEVELS levels), and "some packet mate hits the light" is drawn with the probability it has in
 * such a packet.  V4R_FLAG_PLANE_LIGHT_PDF: the plane density always (what a lane of a full packet nearly always
 * gets; a pure function of the ray, used for per-pixel comparisons).  V4R_FLAG_BOUNDED_LIGHT_PDF: 0 for every
 * direction that misses the light (the textbook density). */
#define V4R_FLAG_BOUNDED_LIGHT_PDF 0x1u
#define V4R_FLAG_PLANE_LIGHT_PDF 0x4u
#define V4R_PACKET_COUNTED_LEVELS 8u
#define V4R_PACKET_MANY_RAYS 256u

#define V4R_FLAG_KEEP_HOST_BVH 0x2u     /* traverse the topology of `nodes` as given (widened 2 -> 4);
                                           default off = the device library re-splits the same leaf
                                           boxes with a binned surface-area heuristic first, which
                                           changes how many nodes a ray visits, never what it hits */

/* Flattened scene + renderer parameters.  All pointers are host pointers, copied by
 * v4r_renderer_create; a table with n == 0 may be NULL. */
typedef struct v4r_scene_desc {
    uint32_t abi_version; /* V4R_ABI_VERSION */
    uint32_t flags;

    uint32_t n_nodes;
    uint32_t bvh_depth; /* max number of interior nodes on a root-to-leaf path (0 if n_nodes==0) */
    const v4r_bvh_node *nodes; /* node 0 is the root; n_prims == 1 => n_nodes == 0 */

    uint32_t n_prims;
    uint32_t n_xforms;
    const v4r_prim *prims;
    const uint32_t *prim_material; /* material index per primitive */
    const v4r_xform *xforms;

    uint32_t n_tri_shading;
    uint32_t n_media;
    const v4r_tri_shading *tri_shading;
    const v4r_medium *media;
    uint32_t n_medium_prims;
    uint32_t n_materials;
    const v4r_prim *medium_prims;
    const v4r_material *materials;

    uint32_t n_textures;
    uint32_t n_images;
    const v4r_texture *textures;
    const v4r_image *images;
    const uint8_t *texels;
    uint64_t texel_bytes;

    uint32_t n_perlin;
    uint32_t n_lights;
    const v4r_perlin *perlin;
    const uint32_t *lights; /* primitive indices of the `important` objects (scene.rs:52-61) */

    float background[3];  /* colour of any miss (renderer.rs:184)              */
    float environment[3]; /* colour returned when the depth budget hits 0 (renderer.rs:128-130) */
    v4r_camera camera;

    uint32_t width, height; /* RendererParam (renderer.rs:44-51) */
    uint32_t max_depth;     /* already defaulted (20 when absent, renderer.rs:331) */
    uint32_t antialias;     /* 0 = pixel centres (renderer.rs:103-104), else jittered */
} v4r_scene_desc;

typedef struct v4r_renderer v4r_renderer;

/* Work counters of the traversal stage, summed over every ray segment traced while counting
 * is enabled; they feed the algorithmic-bytes figure of the roofline report. */
typedef struct v4r_stats {
    uint64_t segments;     /* rays traced (one per path segment)         */
    uint64_t nodes;        /* 128-byte 4-wide BVH nodes fetched          */
    uint64_t prims;        /* 48-byte primitive records intersected      */
    uint64_t paths;        /* camera paths started                       */
    uint64_t shaded;       /* hit records shaded                         */
    uint64_t texel_fetches;/* image-texture texels fetched               */
    uint64_t wave_node_steps; /* times a 64-lane wave executed the node step: nodes / (64 * this) = lane utilisation */
    uint64_t wave_leaf_steps; /* times a wave executed the primitive test: prims / (64 * this) = lane utilisation  */
    uint64_t ended_on_light;  /* of `shaded`: hits on one-colour rectangle lights that the trace kernel finished itself */
} v4r_stats;

/* ---- entry points ------------------------------------------------------------------------- */

/* Build a renderer on HIP device `device`: validates and uploads the tables, allocates the
 * wavefront queues and a zeroed (H,W,3) fp32 accumulator.
 * Replaces Renderer::new (src/renderer.rs:84-93) -- BVH construction stays with the host. */
int v4r_renderer_create(const v4r_scene_desc *desc, int device, v4r_renderer **out);

/* Replaces dropping the Arc<Renderer> (src/main.rs:35). NULL is accepted. */
void v4r_renderer_destroy(v4r_renderer *r);

/* Render passes first_pass, first_pass+pass_stride, ... (n_passes of them), each a full-image
 * 1-sample-per-pixel pass, and add them into the device accumulator.  Asynchronous on the
 * renderer's stream.  Pass p uses random stream (seed, p, pixel, bounce, dimension), so a job
 * split over N ranks as first_pass = rank, pass_stride = N consumes the same streams as N = 1.
 * Replaces Renderer::render (src/renderer.rs:321-332) followed by RenderResult::add (:355-368). */
int v4r_render(v4r_renderer *r, uint64_t seed, uint32_t first_pass, uint32_t pass_stride,
               uint32_t n_passes);

/* The same for the band of image rows [row0, row0 + n_rows) only (row 0 = top): the tile split of a
 * multi-GPU job, where every GPU renders all passes of its own band and the accumulators (zero
 * outside the band) are summed.  Pixels draw the random streams of their position in the whole
 * image, so under a light pdf that is a function of the ray alone (V4R_FLAG_PLANE_LIGHT_PDF,
 * V4R_FLAG_BOUNDED_LIGHT_PDF, or a scene without lights) bands rendered separately add up to exactly
 * the image v4r_render produces.  Under the default packet-statistics rule the rays that end on a
 * shape are counted within the band, so a shape that straddles bands sees smaller counts than in a
 * whole pass: the bands then add up to another estimate of the same image (same mean, not the same
 * bits).  The pass count of the accumulator advances only when the band is the whole image.  No
 * counterpart in the reference, whose passes are always full images (src/renderer.rs:321-332). */
int v4r_render_rows(v4r_renderer *r, uint64_t seed, uint32_t first_pass, uint32_t pass_stride, uint32_t n_passes, uint32_t row0,
                    uint32_t n_rows);

/* v4r_render for callers that cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc,
 * code that must not disturb a HIP graph capture going on elsewhere in the process (recording this call itself into a
 * graph is not covered by a test: it records and waits for events of the handle).  It has no counterpart in the reference.  Same arguments, same passes, same
 * random streams and the same accumulator as v4r_render -- the image is bit-identical -- but the wavefront queues,
 * segment counters and stack-spill slab live in a V4R_NOALLOC_SCRATCH_BYTES array that is part of the library's
 * device image (every GPU the library runs on holds it until the process ends), and the call makes no hipMalloc,
 * hipFree or other allocating call, creates no event or stream, and launches on the renderer's public stream
 * only.  Limit: one pass (width x height paths) must fit the path slots the scratch has left after the scene's
 * stack-spill slab and segment counters -- about 300 000 for a shallow tree, about 100 000 for the 79 488-triangle
 * bunny; larger images, and scenes whose tree is too deep, get V4R_E_LIMIT with the numbers in v4r_last_error().
 * Passes run in batches of scratch_slots / (width x height).  The scratch is shared by every renderer on the same
 * GPU and keeps the previous call's contents (the call clears the counters it needs at zero).  Calls from several
 * handles or threads on one GPU do not race: each call's stream waits for the event the previous call on that GPU
 * recorded (the events are created with the first renderer on a GPU, not by this call), so they run one behind the
 * other on the device. */
#define V4R_NOALLOC_SCRATCH_BYTES (48u << 20)
int v4r_render_noalloc(v4r_renderer *r, uint64_t seed, uint32_t first_pass, uint32_t pass_stride,
                       uint32_t n_passes);

/* Use caller-owned device memory (H*W*3 floats, e.g. a framework tensor that a collective will
 * reduce) as the accumulator from now on; NULL restores the renderer's own buffer.  The buffer is
 * not cleared.  No counterpart in the reference (its accumulator is the RwLock'd Vec of
 * src/renderer.rs:336-340). */
int v4r_accum_bind(v4r_renderer *r, float *device_hwc);

/* Sum the accumulators of all ranks of an RCCL communicator onto rank `root`, in place, on the
 * renderer's stream (after the passes queued so far): one ncclReduce(sum, fp32, H*W*3) over xGMI, the
 * only exchange step of a multi-GPU job.  `rccl_comm` is the caller's ncclComm_t; the library does not
 * link RCCL: it uses the ncclReduce already loaded in the process, else librccl.so.1, and refuses (V4R_E_UNSUPPORTED) a
 * library whose ncclGetVersion does not report 2.x, the versions whose ncclFloat32 / ncclSum values it passes.  Replaces the
 * RwLock'd sum of RenderResult::add across the reference's concurrent passes (src/main.rs:56-63,
 * src/renderer.rs:355-368). */
int v4r_reduce(v4r_renderer *r, void *rccl_comm, int root);

/* Zero the accumulator and the pass count. Replaces RenderResult::new (src/renderer.rs:343-354). */
int v4r_accum_clear(v4r_renderer *r);

/* Synchronise, then copy the (H,W,3) fp32 radiance sums (row 0 = top) to host memory and report how
 * many passes they hold.  Either pointer may be NULL.  Replaces the Vec<f32> hand-off of
 * PyRenderer::py_render (src/renderer.rs:450-476). */
int v4r_accum_read(v4r_renderer *r, float *host_hwc, uint32_t *n_passes);

/* Replace the accumulator's contents and pass count with host data (H*W*3 fp32 sums, row 0 = top):
 * the restore half of a checkpoint, after which v4r_render(..., first_pass = n_passes, ...)
 * continues the job.  RenderResult is a plain (Vec<f32>, count) pair (src/renderer.rs:336-340), so
 * the reference's equivalent is constructing one from saved sums. */
int v4r_accum_write(v4r_renderer *r, const float *host_hwc, uint32_t n_passes);

/* Device address and length (floats) of the accumulator currently bound, for an RCCL reduce
 * issued by the caller after v4r_synchronize. */
int v4r_accum_device(v4r_renderer *r, float **device_hwc, uint64_t *n_floats);

/* 8-bit image of accumulator / n_passes: u8(clamp(256*sqrt(x), 0.5, 255.5)), NaN -> 0.
 * Replaces RenderResult::get_raw (src/renderer.rs:369-406). */
int v4r_tonemap_u8(v4r_renderer *r, uint32_t n_passes, uint8_t *host_rgb);

/* Parity gate: trace the W*H deterministic primary rays (pixel centres, no lens sample, time0)
 * and return the closest primitive index (0xFFFFFFFF = miss) and hit distance per pixel,
 * row 0 = top.  Maps to BVH::traverse (src/bvh/bvh.rs:182-239) + the closest-hit selection of
 * Renderer::ray_color (src/renderer.rs:149-183) for depth-0 rays. */
int v4r_trace_primary(v4r_renderer *r, uint32_t *host_prim_id, float *host_t);

/* Trace n caller-supplied rays (origins and unit directions as n x 3 floats) and
 * return closest primitive / t per ray. `seed` keys the stochastic constant-medium test.
 * Maps to BVH::traverse + Hittable::hit (src/hittable/mod.rs:180-182). */
int v4r_trace_rays(v4r_renderer *r, uint32_t n, const float *host_origin, const float *host_dir,
                   uint64_t seed, uint32_t *host_prim_id, float *host_t);

/* The tree the kernels traverse, for checkers and tools.  The device library does not walk `nodes` of the scene
 * description as given (unless V4R_FLAG_KEEP_HOST_BVH): it re-splits the same leaf boxes, numbers the primitives in the
 * order of its own tree's leaves and collapses the binary tree into 4-wide or 8-wide nodes of 128 bytes, whichever
 * does less work for the scene's camera rays.  v4r_tree_read copies that tree out: `nodes` receives n_nodes * 128 bytes,
 * `scene_index` n_prims words (scene_index[k] = index in the scene description of the primitive the device calls k),
 * `media_direct` n_media_direct words (device numbers of the constant media that every ray tests before the tree
 * instead of meeting them as leaves); any of the three may be NULL.
 *   Child word: an interior node's index; or V4R_LEAF_BIT | (count - 1) << 28 | first = the leaf holding the device's
 *   primitives first .. first + count - 1; or 0xFFFFFFFF = empty slot.
 *   4-wide node, 32 floats: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] child[4] {never-cull mask, 0, 0, 0}; an empty
 *   slot has NaN planes.
 *   8-wide node: {p.x, p.y, p.z, biased exponents ex | ey << 8 | ez << 16 | never-cull mask << 24}, child[8], then the
 *   children's planes as bytes, eight each (two words, child c in byte c & 3 of word c >> 2): lower x, lower y, lower z,
 *   upper x, upper y, upper z; plane = p + byte * 2^(e - 127).  The last 32 bytes are zero.
 *   Never-cull mask: children whose subtree holds a constant medium, entered whenever the ray meets their box.
 * No counterpart in the reference, whose BVH is private to its Renderer (src/bvh/bvh.rs:156-160, src/renderer.rs:75-82). */
typedef struct v4r_tree_info {
    uint32_t width;          /* children per node: 4 or 8 */
    uint32_t n_nodes;        /* 128-byte nodes; 0 for a scene of one primitive, which is tested without any box */
    uint32_t n_top_nodes;    /* the first n_top_nodes nodes are the top levels, breadth first (kept in LDS by the kernels) */
    uint32_t n_prims;
    uint32_t n_media_direct;
    uint32_t media_in_tree;  /* 1: constant media are leaves of the tree and the never-cull masks are in use */
    uint32_t pad[2];
} v4r_tree_info;
int v4r_tree_read(v4r_renderer *r, v4r_tree_info *info, void *nodes, uint32_t *scene_index, uint32_t *media_direct);

/* Enable/disable the traversal work counters (an instrumented kernel variant) and read them. */
int v4r_stats_enable(v4r_renderer *r, int on);
int v4r_stats_read(v4r_renderer *r, v4r_stats *out, int reset);

/* Device time of the kernels from pairs of HIP events recorded on the stream they are launched on.
 * mode 0: off.  mode 1: a pair around the traversal launches of every segment; batches keep running on all
 * pipelines, so launches of different streams share the GPU and a pair's duration is an upper bound.
 * mode 2: every batch runs on pipeline 0 -- kernels run one after the other, durations are exclusive --
 * with pairs around the traversal launches, the shading launches and the resolve launch of every segment
 * (slower than mode 0: nothing overlaps).  v4r_profile_read returns the traversal group: summed milliseconds
 * and number of event pairs; v4r_profile_read_groups returns, per group, milliseconds, kernel launches and
 * event pairs.  Both synchronise first. */
#define V4R_PROFILE_TRACE 0   /* k_trace (first and second launch of a segment) and k_tail */
#define V4R_PROFILE_SHADE 1   /* k_shade, all material classes of a segment             */
#define V4R_PROFILE_RESOLVE 2 /* k_resolve                                               */
int v4r_profile_enable(v4r_renderer *r, int mode);
int v4r_profile_read(v4r_renderer *r, double *trace_ms, uint64_t *trace_pairs, int reset);
int v4r_profile_read_groups(v4r_renderer *r, double ms[3], uint64_t launches[3], uint64_t pairs[3], int reset);

/* Block until every queued pass has finished. */
int v4r_synchronize(v4r_renderer *r);

/* The HIP stream (hipStream_t) the renderer launches on, so callers can order their own work
 * (events, collectives) against it. */
int v4r_stream(v4r_renderer *r, void **hip_stream);

/* Number of passes held by the accumulator (RenderResult.1, src/renderer.rs:366-367). */
int v4r_pass_count(v4r_renderer *r, uint32_t *n_passes);

/* Thread-local message for the last non-zero status. Replaces the reference's panics
 * (src/json.rs:351-353, src/main.rs:71). */
const char *v4r_last_error(void);

uint32_t v4r_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* V4RAY_HIP_H */
