/*
 * v4ray_host.h -- C API of the host-side scene pipeline (libv4ray_host.so).
 *
 * This is the part of the reference that stays on the host: the Scene container
 * (src/scene.rs:20-92), the shape expanders (src/hittables/), the JSON scene schema
 * (src/json.rs), BVH::build (src/bvh/bvh.rs:162) and Camera::new (src/camera.rs:78).  It ends in
 * v4rh_scene_flatten(), which yields the v4r_scene_desc that v4r_renderer_create() uploads.
 * Python binds it with ctypes; the C++ CLI links it directly.
 *
 * Functions that return int yield 0 / a non-negative id on success and a negative value on
 * failure, with the message available from v4rh_last_error().
 */
#ifndef V4RAY_HOST_H
#define V4RAY_HOST_H

#include "v4ray_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct v4rh_scene v4rh_scene; /* Scene + RendererParam + CameraParam */
typedef struct v4rh_shape v4rh_shape; /* the (hittable, samplable) list a json.rs Shape expands to */

/* CameraParam (src/camera.rs:16-32); bit i of `has` marks optional field i as present. */
#define V4RH_CAM_UP 0x1u
#define V4RH_CAM_ASPECT 0x2u
#define V4RH_CAM_APERTURE 0x4u
#define V4RH_CAM_FOCUS 0x8u
#define V4RH_CAM_TIME0 0x10u
#define V4RH_CAM_TIME1 0x20u
typedef struct v4rh_camera_param {
    float look_from[3], look_at[3];
    float vfov;
    float up[3];
    float aspect_ratio, aperture, focus_dist, time0, time1;
    uint32_t has;
} v4rh_camera_param;

/* RendererParam (src/renderer.rs:44-51); max_depth < 0 and antialias < 0 mean "absent". */
typedef struct v4rh_renderer_param {
    uint32_t width, height;
    int32_t max_depth;
    int32_t antialias;
} v4rh_renderer_param;

/* Scene::new (src/scene.rs:29-37). */
int v4rh_scene_new(const float background[3], const float environment[3], v4rh_scene **out);
void v4rh_scene_free(v4rh_scene *s);

/* serde_json::from_reader + build_scene (src/main.rs:31-34, src/json.rs:702-720): also fills the
 * renderer and camera parameters from the file. `perlin_seed` replaces thread_rng for noise tables. */
int v4rh_scene_load_json_file(const char *path, uint64_t perlin_seed, v4rh_scene **out);
int v4rh_scene_load_json_text(const char *text, const char *base_dir, uint64_t perlin_seed, v4rh_scene **out);

/* Textures (src/texture/) and materials (src/material/): return the table index. */
int v4rh_texture_solid_color(v4rh_scene *s, const float color[3]);
int v4rh_texture_checker(v4rh_scene *s, int odd, int even, float density);
int v4rh_texture_image(v4rh_scene *s, const uint8_t *rgb, uint32_t width, uint32_t height);
int v4rh_texture_image_file(v4rh_scene *s, const char *path);
int v4rh_texture_noise(v4rh_scene *s, float scale, uint32_t depth, uint64_t seed);
int v4rh_material_lambertian(v4rh_scene *s, int texture);
int v4rh_material_metal(v4rh_scene *s, const float albedo[3], float fuzz);
int v4rh_material_dielectric(v4rh_scene *s, float ir);
int v4rh_material_diffuse_light(v4rh_scene *s, int texture);
int v4rh_material_isotropic(v4rh_scene *s, int texture);

/* Shapes (src/hittable/, src/hittables/).  `rect_type` is V4R_PRIM_{XY,YZ,ZX}_RECT. */
v4rh_shape *v4rh_shape_sphere(const float center[3], float radius);
v4rh_shape *v4rh_shape_rect(uint32_t rect_type, float a0, float a1, float b0, float b1, float k, int positive);
v4rh_shape *v4rh_shape_triangle(const float vertices[9], const float *normals9, const float *uvs6);
v4rh_shape *v4rh_shape_cuboid(const float p0[3], const float p1[3]);
/* model < 0: first model; subdivide: number of 1:4 midpoint subdivisions applied first */
v4rh_shape *v4rh_shape_mesh_file(const char *path, int model, int subdivide);
v4rh_shape *v4rh_shape_constant_medium(const v4rh_shape *boundary, float density);
/* Wrap every hittable of the shape in a TransformHittable (src/json.rs:441-467); either may be NULL. */
int v4rh_shape_transform(v4rh_shape *shape, const float matrix9[9], const float translate[3]);
int v4rh_shape_count(const v4rh_shape *shape);
void v4rh_shape_free(v4rh_shape *shape);

/* Scene::add / Scene::add_important for every hittable of the shape (src/json.rs:685-699). */
int v4rh_scene_add(v4rh_scene *s, const v4rh_shape *shape, int material, int important);
int v4rh_scene_len(const v4rh_scene *s);

int v4rh_scene_set_renderer(v4rh_scene *s, const v4rh_renderer_param *p);
int v4rh_scene_get_renderer(const v4rh_scene *s, v4rh_renderer_param *p);
int v4rh_scene_set_camera(v4rh_scene *s, const v4rh_camera_param *p);
int v4rh_scene_get_camera(const v4rh_scene *s, v4rh_camera_param *p);
/* pretransform_triangles: bake transforms into non-light triangles; flags: V4R_FLAG_* */
int v4rh_scene_set_options(v4rh_scene *s, int pretransform_triangles, uint32_t flags);

/* Build the BVH (Renderer::new -> Scene::build_bvh, src/renderer.rs:86) and flatten. The
 * descriptor and every table it points to belong to the scene and stay valid until the next
 * flatten or v4rh_scene_free. */
int v4rh_scene_flatten(v4rh_scene *s, const v4r_scene_desc **out);

/* BVH::build on caller-supplied boxes (n x {min xyz, max xyz}), returning the reference-form
 * node array (leaves included, DFS pre-order) for inspection. `nodes` needs room for 2n-1. */
typedef struct v4rh_bvh_node {
    uint32_t parent_index, depth, leaf, shape_index;
    uint32_t children_index[2];
    float children_aabb[2][6];
} v4rh_bvh_node;
int v4rh_bvh_build(const float *boxes, uint32_t n, v4rh_bvh_node *nodes, uint32_t *n_nodes);

/* Bounding box of hittable `index` of the scene (Bounded::bounding_box), as {min, max}. */
int v4rh_scene_bounding_box(const v4rh_scene *s, uint32_t index, float out6[6]);

/* Decode a baseline JPEG / binary PPM file to 8-bit RGB; rgb may be NULL to query the size. */
int v4rh_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb, uint64_t capacity);

/* RenderResult::get_raw arithmetic on host data (src/renderer.rs:378-402): sums / n -> u8. */
int v4rh_tonemap_u8(const float *sums, uint64_t n_floats, uint32_t n_passes, uint8_t *out);
/* Write an RGB image whose row 0 is the top; the format follows the extension like `image::save` (src/main.rs:70-72):
 * .png, .ppm, .tga, otherwise a 24-bit BMP; extensions of formats this writer lacks (.jpg ...) are an error. */
int v4rh_write_image(const char *path, const uint8_t *rgb, uint32_t width, uint32_t height);

/* Accumulator checkpoint: a restartable image of RenderResult (src/renderer.rs:336-340: the fp32
 * sums and the pass count) plus what identifies the job (size, seed).  File = 48-byte header
 * {"V4RCKPT1", width, height, n_passes, reserved, seed, FNV-1a-64 of the payload, payload bytes}
 * followed by H*W*3 little-endian floats; written to "<path>.tmp" and renamed, so a crash while
 * saving leaves the previous checkpoint intact.  Load rejects a truncated or corrupted file; sums
 * may be NULL to read only the header. */
int v4rh_checkpoint_save(const char *path, const float *sums, uint32_t width, uint32_t height, uint32_t n_passes, uint64_t seed);
int v4rh_checkpoint_load(const char *path, uint32_t *width, uint32_t *height, uint32_t *n_passes, uint64_t *seed, float *sums,
                         uint64_t capacity_floats);

const char *v4rh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* V4RAY_HOST_H */
