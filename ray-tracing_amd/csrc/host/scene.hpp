// Host-side scene model: the part of the reference that stays on the host and feeds the device
// path.  It mirrors Scene (src/scene.rs:20-92), the shape expanders (src/hittables/), the
// bounding boxes (Bounded impls), BVH::build (src/bvh/bvh.rs:34-180) and Camera::new
// (src/camera.rs:78-112), and flattens everything into the tables of v4r_scene_desc.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "v4ray_hip.h"
#include "vecmath.hpp"

namespace v4ray {

// ---- appearance --------------------------------------------------------------------------------
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;  // rows top to bottom, 3 bytes per texel
};

struct PerlinTables {  // texture/noise.rs:13-18
    float vec[256][3];
    uint8_t perm_x[256], perm_y[256], perm_z[256];
    // Deterministic replacement for Perlin::new(thread_rng) (noise.rs:21-39): same construction
    // (256 points on the unit sphere, three Fisher-Yates permutations) from a PCG32 stream.
    static PerlinTables generate(uint64_t seed);
};

// ---- geometry ----------------------------------------------------------------------------------
// One leaf-level hittable: a primitive, optionally wrapped in a TransformHittable.
struct Hittable {
    uint32_t type = V4R_PRIM_SPHERE;  // V4R_PRIM_* (never MEDIUM here; see Medium)
    bool positive = true;             // rects
    float f[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // sphere {c,r}; rect {a0,a1,b0,b1,k}
    // triangle: vertices as given (edge12/edge13 are derived at flatten time, triangle.rs:30-31)
    Vec3 verts[3];
    // triangle shading data
    float n[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    bool has_xform = false;
    Affine xform;
    // ConstantMedium wrapper (constant_medium.rs): when is_medium, `boundary` holds the wrapped
    // hittable(s) and this record's own xform (if any) is the wrapper's transform.
    bool is_medium = false;
    float density = 0;
    std::vector<Hittable> boundary;
    bool samplable = true;  // media are not Samplable (json.rs:487-499)

    AABB bounding_box() const;  // Bounded::bounding_box of the wrapped object
};

Hittable make_sphere(Vec3 center, float radius);
Hittable make_rect(uint32_t type, float a0, float a1, float b0, float b1, float k, bool positive);
Hittable make_triangle(const Vec3 v[3], const Vec3 n[3], const float uv[3][2]);
std::vector<Hittable> make_cuboid(Vec3 p0, Vec3 p1);  // hittables/cuboid.rs:30-61
Hittable make_medium(std::vector<Hittable> boundary, float density);
// Wrap in a TransformHittable (hittable/transform.rs:16-31).
Hittable with_transform(Hittable h, const Affine &a);

// Triangulated OBJ model (what tobj::load_obj(triangulate=true) returns per model).
struct ObjMesh {
    std::string name;
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> indices, normal_indices, texcoord_indices;
};
std::vector<ObjMesh> load_obj(const std::string &path);
std::vector<ObjMesh> parse_obj(const std::string &text);
// hittables/obj.rs:30-104: one Triangle per face, averaged vertex normals when the file has none.
std::vector<Hittable> mesh_to_triangles(const ObjMesh &mesh);
// 1:4 midpoint subdivision of every face, `levels` times (used to synthesise larger meshes).
ObjMesh subdivide(const ObjMesh &mesh, int levels);

// ---- BVH ---------------------------------------------------------------------------------------
// Node array exactly as BVH::build produces it (DFS pre-order, leaves are nodes).
struct BVHNode {
    uint32_t parent_index = 0, depth = 0;
    bool leaf = true;
    uint32_t shape_index = 0;
    uint32_t children_index[2] = {0, 0};
    AABB children_aabb[2];
};
struct BVH {
    std::vector<BVHNode> nodes;
    static BVH build(const std::vector<AABB> &shapes);  // bvh.rs:162-180
    // Device form: interior nodes only, leaves folded into the child words.
    void flatten(std::vector<v4r_bvh_node> &out, uint32_t &depth) const;
};

// ---- camera / renderer parameters ----------------------------------------------------------------
struct CameraParam {  // camera.rs:16-32
    Vec3 look_from, look_at;
    float vfov = 40;
    bool has_up = false, has_aspect = false, has_aperture = false, has_focus = false, has_time0 = false,
         has_time1 = false;
    Vec3 up{0, 1, 0};
    float aspect_ratio = 1, aperture = 0, focus_dist = 1, time0 = 0, time1 = 0;
};
v4r_camera make_camera(const CameraParam &p, float default_aspect_ratio);  // camera.rs:78-112

struct RendererParam {  // renderer.rs:44-51
    uint32_t width = 0, height = 0;
    bool has_max_depth = false;
    uint32_t max_depth = 20;
    bool has_antialias = false;
    bool antialias = true;
};

// ---- scene -------------------------------------------------------------------------------------
class Scene {
  public:
    Scene(Vec3 background, Vec3 environment) : background_(background), environment_(environment) {}

    // texture / material registries (ids index the flattened tables)
    uint32_t add_solid_color(Vec3 color);
    uint32_t add_checker(uint32_t odd, uint32_t even, float density);
    uint32_t add_image(Image img);
    uint32_t add_noise(float scale, uint32_t depth, uint64_t seed);
    uint32_t add_lambertian(uint32_t tex);
    uint32_t add_metal(Vec3 albedo, float fuzz);
    uint32_t add_dielectric(float ir);
    uint32_t add_diffuse_light(uint32_t tex);
    uint32_t add_isotropic(uint32_t tex);

    void add(Hittable h, uint32_t material);            // scene.rs:38-41
    void add_important(Hittable h, uint32_t material);  // scene.rs:52-61
    size_t len() const { return hittables_.size(); }
    size_t n_textures() const { return textures_.size(); }
    size_t n_materials() const { return materials_.size(); }
    BVH build_bvh() const;  // scene.rs:62-70

    // Optional host-side optimisation: bake TransformHittable into triangles that are not
    // lights and whose transform preserves orientation, so the device intersects them in
    // world space (mathematically the same hit; see flatten()).
    void set_pretransform_triangles(bool on) { pretransform_ = on; }
    void set_flags(uint32_t flags) { flags_ = flags; }

    // Build the BVH and fill the flat tables. The returned reference stays valid until the next
    // flatten() or the scene's destruction.
    const v4r_scene_desc &flatten(const RendererParam &rp, const CameraParam &cp);

    const std::vector<Hittable> &hittables() const { return hittables_; }
    Vec3 background() const { return background_; }
    Vec3 environment() const { return environment_; }

  private:
    uint32_t push_prim(const Hittable &h, std::vector<v4r_prim> &dst, bool allow_bake);
    Vec3 background_, environment_;
    std::vector<Hittable> hittables_;
    std::vector<uint32_t> materials_of_;
    std::vector<uint32_t> lights_;
    std::vector<v4r_material> materials_;
    std::vector<v4r_texture> textures_;
    std::vector<Image> images_;
    std::vector<PerlinTables> perlin_;
    bool pretransform_ = false;
    uint32_t flags_ = 0;

    // flattened storage
    v4r_scene_desc desc_{};
    std::vector<v4r_bvh_node> f_nodes_;
    std::vector<v4r_prim> f_prims_, f_medium_prims_;
    std::vector<v4r_xform> f_xforms_;
    std::vector<v4r_tri_shading> f_tri_;
    std::vector<v4r_medium> f_media_;
    std::vector<v4r_image> f_images_;
    std::vector<uint8_t> f_texels_;
    std::vector<v4r_perlin> f_perlin_;
};

// ---- JSON scene files (src/json.rs) ------------------------------------------------------------
struct LoadedScene {
    std::unique_ptr<Scene> scene;
    RendererParam renderer;
    CameraParam camera;
};
// build_scene (json.rs:702-720) + the renderer/camera blocks of SceneParam (json.rs:235-250).
// `base_dir` resolves relative `file` fields (the reference resolves them against the CWD);
// `perlin_seed` seeds the noise tables (the reference uses thread_rng, json.rs:355-359).
LoadedScene load_scene_json(const std::string &json_text, const std::string &base_dir, uint64_t perlin_seed);
LoadedScene load_scene_file(const std::string &path, uint64_t perlin_seed);

// Baseline JPEG / PPM decode into 8-bit RGB.
Image load_image(const std::string &path);
Image decode_jpeg(const uint8_t *data, size_t size);

}  // namespace v4ray
