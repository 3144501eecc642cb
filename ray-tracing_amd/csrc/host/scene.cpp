// Host-side scene model: shapes, bounding boxes, BVH construction, camera set-up and
// flattening into the device tables.  Reference lines are cited per function.
#include "scene.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <stdexcept>

namespace v4ray {

// ---- PCG32 (XSH-RR 64/32) used wherever the reference drew from thread_rng on the host ---------
namespace {
struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed, uint64_t seq = 0xda3e39cb94b95bdbULL) {
        state = 0;
        inc = (seq << 1) | 1u;
        next();
        state += seed;
        next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    float uniform() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }      // [0,1)
    uint32_t below(uint32_t n) { return (uint32_t)(((uint64_t)next() * n) >> 32); }  // [0,n)
};
}  // namespace

PerlinTables PerlinTables::generate(uint64_t seed) {
    PerlinTables t;
    Pcg32 rng(seed);
    const float two_pi = 2.0f * 3.14159265358979323846f;
    for (int i = 0; i < 256; ++i) {
        // random_on_unit_sphere, src/random.rs:41-51
        float z = -1.0f + 2.0f * rng.uniform();
        float theta = two_pi * rng.uniform();
        float xy = std::sqrt(1.0f - z * z);
        t.vec[i][0] = std::cos(theta) * xy;
        t.vec[i][1] = std::sin(theta) * xy;
        t.vec[i][2] = z;
    }
    // perlin_generate_perm, src/texture/noise.rs:32-39
    auto perm = [&](uint8_t *p) {
        for (int i = 0; i < 256; ++i) p[i] = (uint8_t)i;
        for (int i = 255; i >= 1; --i) {
            uint32_t target = rng.below((uint32_t)i + 1);
            std::swap(p[target], p[i]);
        }
    };
    perm(t.perm_x);
    perm(t.perm_y);
    perm(t.perm_z);
    return t;
}

// ---- shapes ------------------------------------------------------------------------------------
Hittable make_sphere(Vec3 center, float radius) {
    Hittable h;
    h.type = V4R_PRIM_SPHERE;
    h.f[0] = center.x;
    h.f[1] = center.y;
    h.f[2] = center.z;
    h.f[3] = radius;
    return h;
}

Hittable make_rect(uint32_t type, float a0, float a1, float b0, float b1, float k, bool positive) {
    if (type != V4R_PRIM_XY_RECT && type != V4R_PRIM_YZ_RECT && type != V4R_PRIM_ZX_RECT)
        throw std::runtime_error("make_rect: not a rectangle type");
    Hittable h;
    h.type = type;
    h.positive = positive;
    h.f[0] = a0;
    h.f[1] = a1;
    h.f[2] = b0;
    h.f[3] = b1;
    h.f[4] = k;
    return h;
}

Hittable make_triangle(const Vec3 v[3], const Vec3 n[3], const float uv[3][2]) {
    Hittable h;
    h.type = V4R_PRIM_TRIANGLE;
    for (int i = 0; i < 3; ++i) {
        h.verts[i] = v[i];
        n[i].store(h.n[i]);
        h.uv[i][0] = uv[i][0];
        h.uv[i][1] = uv[i][1];
    }
    return h;
}

// hittables/cuboid.rs:30-61: six rectangles, in this order and with these orientations.
std::vector<Hittable> make_cuboid(Vec3 p0, Vec3 p1) {
    return {
        make_rect(V4R_PRIM_XY_RECT, p0.x, p1.x, p0.y, p1.y, p0.z, false),
        make_rect(V4R_PRIM_XY_RECT, p0.x, p1.x, p0.y, p1.y, p1.z, true),
        make_rect(V4R_PRIM_YZ_RECT, p0.y, p1.y, p0.z, p1.z, p0.x, false),
        make_rect(V4R_PRIM_YZ_RECT, p0.y, p1.y, p0.z, p1.z, p1.x, true),
        make_rect(V4R_PRIM_ZX_RECT, p0.z, p1.z, p0.x, p1.x, p0.y, false),
        make_rect(V4R_PRIM_ZX_RECT, p0.z, p1.z, p0.x, p1.x, p1.y, true),
    };
}

Hittable make_medium(std::vector<Hittable> boundary, float density) {
    if (boundary.empty()) throw std::runtime_error("constant medium without a boundary shape");
    for (auto &b : boundary)
        if (b.is_medium) throw std::runtime_error("nested constant media are not supported");
    Hittable h;
    h.is_medium = true;
    h.type = V4R_PRIM_MEDIUM;
    h.density = density;
    h.boundary = std::move(boundary);
    h.samplable = false;
    return h;
}

Hittable with_transform(Hittable h, const Affine &a) {
    if (h.has_xform) throw std::runtime_error("a hittable can carry one transform");
    h.has_xform = true;
    h.xform = a;
    return h;
}

static AABB object_box(const Hittable &h) {
    switch (h.type) {
        case V4R_PRIM_SPHERE: {  // sphere.rs:48-51
            Vec3 c(h.f[0], h.f[1], h.f[2]);
            Vec3 r(h.f[3], h.f[3], h.f[3]);
            return AABB::with_bounds(c - r, c + r);
        }
        case V4R_PRIM_XY_RECT:  // aa_rect.rs:34-39
            return AABB::with_bounds(Vec3(h.f[0], h.f[2], h.f[4] - EPSILON), Vec3(h.f[1], h.f[3], h.f[4] + EPSILON));
        case V4R_PRIM_YZ_RECT:  // aa_rect.rs:40-45
            return AABB::with_bounds(Vec3(h.f[4] - EPSILON, h.f[0], h.f[2]), Vec3(h.f[4] + EPSILON, h.f[1], h.f[3]));
        case V4R_PRIM_ZX_RECT:  // aa_rect.rs:46-51
            return AABB::with_bounds(Vec3(h.f[2], h.f[4] - EPSILON, h.f[0]), Vec3(h.f[3], h.f[4] + EPSILON, h.f[1]));
        case V4R_PRIM_TRIANGLE: {  // triangle.rs:37-48
            Vec3 lo = vmin(vmin(h.verts[0], h.verts[1]), h.verts[2]);
            Vec3 hi = vmax(vmax(h.verts[0], h.verts[1]), h.verts[2]);
            Vec3 diff = lo - hi;
            for (int i = 0; i < 3; ++i)
                if (diff[i] == 0.0f) {
                    lo[i] -= EPSILON;
                    hi[i] += EPSILON;
                }
            return AABB::with_bounds(lo, hi);
        }
        case V4R_PRIM_MEDIUM: {  // constant_medium.rs:29-31, hittables/group.rs:43-49
            AABB b = h.boundary[0].bounding_box();
            for (size_t i = 1; i < h.boundary.size(); ++i) b = b.join(h.boundary[i].bounding_box());
            return b;
        }
    }
    throw std::runtime_error("unknown primitive type");
}

AABB Hittable::bounding_box() const {
    AABB b = object_box(*this);
    if (!has_xform) return b;
    // transform.rs:37-56: box of the eight transformed corners, in this corner order.
    Vec3 lo = b.min, hi = b.max;
    Vec3 corners[8] = {{lo.x, lo.y, lo.z}, {lo.x, lo.y, hi.z}, {lo.x, hi.y, lo.z}, {lo.x, hi.y, hi.z},
                       {hi.x, lo.y, lo.z}, {hi.x, lo.y, hi.z}, {hi.x, hi.y, lo.z}, {hi.x, hi.y, hi.z}};
    Vec3 mn = xform.apply_point(corners[0]), mx = mn;
    for (int i = 1; i < 8; ++i) {
        Vec3 p = xform.apply_point(corners[i]);
        mn = vmin(mn, p);
        mx = vmax(mx, p);
    }
    return AABB::with_bounds(mn, mx);
}

// ---- BVH::build, src/bvh/bvh.rs:34-180 ---------------------------------------------------------
namespace {
struct Builder {
    const std::vector<AABB> &shapes;
    std::vector<Vec3> centroids;
    std::vector<BVHNode> &nodes;

    uint32_t build(const std::vector<uint32_t> &indices, uint32_t parent, uint32_t depth) {
        if (indices.size() == 1) {  // bvh.rs:43-53
            BVHNode n;
            n.parent_index = parent;
            n.depth = depth;
            n.leaf = true;
            n.shape_index = indices[0];
            nodes.push_back(n);
            return (uint32_t)nodes.size() - 1;
        }
        AABB cb = AABB::empty();  // bvh.rs:54-59
        for (uint32_t i : indices) cb = cb.grow(centroids[i]);
        uint32_t node_index = (uint32_t)nodes.size();
        nodes.push_back(BVHNode());  // dummy, bvh.rs:60-61
        Vec3 size = cb.size();
        int axis = 0;  // argmax: first strictly-greatest component (bvh.rs:62)
        float axis_size = size[0];
        for (int a = 1; a < 3; ++a)
            if (size[a] > axis_size) {
                axis = a;
                axis_size = size[a];
            }
        uint32_t child_index[2];
        AABB child_aabb[2];
        if (axis_size < EPSILON) {  // bvh.rs:63-83: halve the index list
            size_t half = indices.size() / 2;
            std::vector<uint32_t> part[2] = {std::vector<uint32_t>(indices.begin(), indices.begin() + half),
                                             std::vector<uint32_t>(indices.begin() + half, indices.end())};
            for (int c = 0; c < 2; ++c) {
                AABB b = AABB::empty();
                for (uint32_t i : part[c]) b = b.join(shapes[i]);
                child_aabb[c] = b;
                child_index[c] = build(part[c], node_index, depth + 1);
            }
        } else {  // bvh.rs:84-141: six buckets along the widest centroid axis
            constexpr int NB = 6;
            size_t count[NB] = {0, 0, 0, 0, 0, 0};
            AABB box[NB];
            std::vector<uint32_t> assign[NB];
            for (uint32_t i : indices) {
                float relative = (centroids[i][axis] - cb.min[axis]) / axis_size;
                float scaled = relative * ((float)NB - 0.01f);
                // Rust `as usize`: truncate toward zero, saturate, NaN -> 0
                size_t b = 0;
                if (scaled == scaled && scaled > 0.0f) b = (size_t)scaled;
                if (b >= (size_t)NB) b = NB - 1;  // cannot happen for relative in [0,1]; guards the array
                count[b] += 1;
                box[b] = box[b].join(shapes[i]);
                assign[b].push_back(i);
            }
            int min_bucket = 0;
            float min_cost = std::numeric_limits<float>::infinity();
            AABB best_l = AABB::empty(), best_r = AABB::empty();
            for (int i = 0; i < NB - 1; ++i) {
                size_t nl = 0, nr = 0;
                AABB bl = AABB::empty(), br = AABB::empty();
                for (int k = 0; k <= i; ++k) {
                    nl += count[k];
                    bl = bl.join(box[k]);
                }
                for (int k = i + 1; k < NB; ++k) {
                    nr += count[k];
                    br = br.join(box[k]);
                }
                float cost = (float)nl * bl.surface_area() + (float)nr * br.surface_area();
                if (cost < min_cost) {
                    min_bucket = i;
                    min_cost = cost;
                    best_l = bl;
                    best_r = br;
                }
            }
            std::vector<uint32_t> part[2];
            for (int k = 0; k < NB; ++k) {
                auto &dst = part[k <= min_bucket ? 0 : 1];
                dst.insert(dst.end(), assign[k].begin(), assign[k].end());
            }
            if (part[0].empty() || part[1].empty())
                throw std::runtime_error("BVH build: degenerate bucket split (non-finite bounding boxes?)");
            child_index[0] = build(part[0], node_index, depth + 1);
            child_index[1] = build(part[1], node_index, depth + 1);
            child_aabb[0] = best_l;
            child_aabb[1] = best_r;
        }
        if (child_aabb[0].is_empty() || child_aabb[1].is_empty())  // bvh.rs:142-143
            throw std::runtime_error("BVH build: empty child box");
        BVHNode &n = nodes[node_index];
        n.parent_index = parent;
        n.depth = depth;
        n.leaf = false;
        n.children_index[0] = child_index[0];
        n.children_index[1] = child_index[1];
        n.children_aabb[0] = child_aabb[0];
        n.children_aabb[1] = child_aabb[1];
        return node_index;
    }
};
}  // namespace

BVH BVH::build(const std::vector<AABB> &shapes) {
    BVH bvh;
    if (shapes.empty()) return bvh;
    bvh.nodes.reserve(shapes.size() * 2);
    Builder b{shapes, {}, bvh.nodes};
    b.centroids.reserve(shapes.size());
    for (auto &s : shapes) b.centroids.push_back(s.center());
    std::vector<uint32_t> indices(shapes.size());
    for (size_t i = 0; i < shapes.size(); ++i) indices[i] = (uint32_t)i;
    b.build(indices, 0, 0);
    return bvh;
}

void BVH::flatten(std::vector<v4r_bvh_node> &out, uint32_t &depth) const {
    out.clear();
    depth = 0;
    std::vector<uint32_t> remap(nodes.size(), 0);
    uint32_t n_interior = 0;
    for (size_t i = 0; i < nodes.size(); ++i)
        if (!nodes[i].leaf) remap[i] = n_interior++;
    out.resize(n_interior);
    for (size_t i = 0; i < nodes.size(); ++i) {
        const BVHNode &n = nodes[i];
        if (n.leaf) {
            depth = std::max(depth, n.depth);
            continue;
        }
        v4r_bvh_node &d = out[remap[i]];
        std::memset(&d, 0, sizeof d);
        n.children_aabb[0].min.store(d.lo0);
        n.children_aabb[0].max.store(d.hi0);
        n.children_aabb[1].min.store(d.lo1);
        n.children_aabb[1].max.store(d.hi1);
        uint32_t c[2];
        for (int k = 0; k < 2; ++k) {
            const BVHNode &ch = nodes[n.children_index[k]];
            c[k] = ch.leaf ? (V4R_LEAF_BIT | ch.shape_index) : remap[n.children_index[k]];
        }
        d.child0 = c[0];
        d.child1 = c[1];
    }
}

// ---- Camera::new, src/camera.rs:78-112 ---------------------------------------------------------
v4r_camera make_camera(const CameraParam &p, float default_aspect_ratio) {
    const float pi = 3.14159265358979323846f;
    float theta = p.vfov * pi / 180.0f;
    float h = std::tan(theta / 2.0f);
    float viewport_height = 2.0f * h;
    float aspect_ratio = p.has_aspect ? p.aspect_ratio : default_aspect_ratio;
    float viewport_width = aspect_ratio * viewport_height;
    Vec3 w = normalize(p.look_from - p.look_at);
    Vec3 up = p.has_up ? p.up : Vec3(0, 1, 0);
    Vec3 u = normalize(cross(up, w));
    Vec3 v = normalize(cross(w, u));
    float focus_dist = p.has_focus ? p.focus_dist : norm(p.look_from - p.look_at);
    Vec3 horizontal = u * (focus_dist * viewport_width);
    Vec3 vertical = v * (focus_dist * viewport_height);
    Vec3 llc = p.look_from - horizontal / 2.0f - vertical / 2.0f - w * focus_dist;
    float aperture = p.has_aperture ? p.aperture : 0.0f;
    v4r_camera c;
    p.look_from.store(c.origin);
    llc.store(c.lower_left_corner);
    horizontal.store(c.horizontal);
    vertical.store(c.vertical);
    u.store(c.u);
    v.store(c.v);
    c.lens_radius = aperture / 2.0f;
    c.time0 = p.has_time0 ? p.time0 : 0.0f;
    c.time1 = p.has_time1 ? p.time1 : 0.0f;
    return c;
}

// ---- Scene -------------------------------------------------------------------------------------
uint32_t Scene::add_solid_color(Vec3 color) {
    v4r_texture t{};
    t.kind = V4R_TEX_SOLID;
    color.store(t.color);
    textures_.push_back(t);
    return (uint32_t)textures_.size() - 1;
}
uint32_t Scene::add_checker(uint32_t odd, uint32_t even, float density) {
    if (odd >= textures_.size() || even >= textures_.size()) throw std::runtime_error("checker: unknown child texture");
    v4r_texture t{};
    t.kind = V4R_TEX_CHECKER;
    t.a = odd;
    t.b = even;
    t.s = density;
    textures_.push_back(t);
    return (uint32_t)textures_.size() - 1;
}
uint32_t Scene::add_image(Image img) {
    if (img.width == 0 || img.height == 0 || img.rgb.size() != (size_t)img.width * img.height * 3)
        throw std::runtime_error("image texture: bad dimensions");
    v4r_texture t{};
    t.kind = V4R_TEX_IMAGE;
    t.a = (uint32_t)images_.size();
    images_.push_back(std::move(img));
    textures_.push_back(t);
    return (uint32_t)textures_.size() - 1;
}
uint32_t Scene::add_noise(float scale, uint32_t depth, uint64_t seed) {
    v4r_texture t{};
    t.kind = V4R_TEX_NOISE;
    t.a = (uint32_t)perlin_.size();
    t.b = depth;
    t.s = scale;
    perlin_.push_back(PerlinTables::generate(seed));
    textures_.push_back(t);
    return (uint32_t)textures_.size() - 1;
}
static v4r_material mat(uint32_t kind) {
    v4r_material m{};
    m.kind = kind;
    return m;
}
uint32_t Scene::add_lambertian(uint32_t tex) {
    if (tex >= textures_.size()) throw std::runtime_error("lambertian: unknown texture");
    v4r_material m = mat(V4R_MAT_LAMBERTIAN);
    m.tex = tex;
    materials_.push_back(m);
    return (uint32_t)materials_.size() - 1;
}
uint32_t Scene::add_metal(Vec3 albedo, float fuzz) {
    v4r_material m = mat(V4R_MAT_METAL);
    albedo.store(m.albedo);
    m.fuzz = fuzz;
    materials_.push_back(m);
    return (uint32_t)materials_.size() - 1;
}
uint32_t Scene::add_dielectric(float ir) {
    v4r_material m = mat(V4R_MAT_DIELECTRIC);
    m.ir = ir;
    materials_.push_back(m);
    return (uint32_t)materials_.size() - 1;
}
uint32_t Scene::add_diffuse_light(uint32_t tex) {
    if (tex >= textures_.size()) throw std::runtime_error("diffuse light: unknown texture");
    v4r_material m = mat(V4R_MAT_DIFFUSE_LIGHT);
    m.tex = tex;
    materials_.push_back(m);
    return (uint32_t)materials_.size() - 1;
}
uint32_t Scene::add_isotropic(uint32_t tex) {
    if (tex >= textures_.size()) throw std::runtime_error("isotropic: unknown texture");
    v4r_material m = mat(V4R_MAT_ISOTROPIC);
    m.tex = tex;
    materials_.push_back(m);
    return (uint32_t)materials_.size() - 1;
}

void Scene::add(Hittable h, uint32_t material) {
    if (material >= materials_.size()) throw std::runtime_error("Scene::add: unknown material");
    hittables_.push_back(std::move(h));
    materials_of_.push_back(material);
}

void Scene::add_important(Hittable h, uint32_t material) {
    if (material >= materials_.size()) throw std::runtime_error("Scene::add_important: unknown material");
    if (!h.samplable) throw std::runtime_error("Scene::add_important: shape is not samplable");
    lights_.push_back((uint32_t)hittables_.size());
    hittables_.push_back(std::move(h));
    materials_of_.push_back(material);
}

BVH Scene::build_bvh() const {
    std::vector<AABB> boxes;
    boxes.reserve(hittables_.size());
    for (auto &h : hittables_) boxes.push_back(h.bounding_box());
    return BVH::build(boxes);
}

uint32_t Scene::push_prim(const Hittable &h, std::vector<v4r_prim> &dst, bool allow_bake) {
    v4r_prim p{};
    p.type = h.type;
    p.xform = V4R_NO_XFORM;
    p.aux = 0;
    bool bake = false;
    if (h.type == V4R_PRIM_TRIANGLE) {
        Vec3 v[3] = {h.verts[0], h.verts[1], h.verts[2]};
        v4r_tri_shading ts{};
        std::memcpy(ts.n, h.n, sizeof ts.n);
        std::memcpy(ts.uv, h.uv, sizeof ts.uv);
        bake = allow_bake && h.has_xform && h.xform.det() > 0.0f;
        if (bake) {
            // World-space triangle: vertices mapped by the affine, normals by its linear part
            // WITHOUT renormalising, so normalize(sum w_i * M n_i) equals the reference's
            // normalize(M * sum w_i n_i) (transform.rs:86, triangle.rs:83-87).
            for (int i = 0; i < 3; ++i) {
                v[i] = h.xform.apply_point(v[i]);
                h.xform.apply_vector(Vec3(h.n[i])).store(ts.n[i]);
            }
        }
        Vec3 e12 = v[1] - v[0], e13 = v[2] - v[0];  // triangle.rs:30-31
        v[0].store(p.f);
        e12.store(p.f + 3);
        e13.store(p.f + 6);
        p.aux = (uint32_t)f_tri_.size();
        f_tri_.push_back(ts);
    } else if (h.type == V4R_PRIM_MEDIUM) {
        v4r_medium m{};
        m.first = (uint32_t)f_medium_prims_.size();
        m.count = (uint32_t)h.boundary.size();
        m.neg_inv_density = -1.0f / h.density;  // constant_medium.rs:20
        // reserve the contiguous range first, then fill (push_prim may append xforms only)
        for (auto &b : h.boundary) push_prim(b, f_medium_prims_, false);
        p.aux = (uint32_t)f_media_.size();
        f_media_.push_back(m);
    } else {
        std::memcpy(p.f, h.f, sizeof p.f);
        if (h.positive) p.type |= V4R_PRIM_POSITIVE;
    }
    if (h.has_xform && !bake) {
        v4r_xform x{};
        Affine inv = h.xform.inverse();
        std::memcpy(x.fwd, h.xform.m, sizeof x.fwd);
        std::memcpy(x.fwd_t, h.xform.t, sizeof x.fwd_t);
        std::memcpy(x.inv, inv.m, sizeof x.inv);
        std::memcpy(x.inv_t, inv.t, sizeof x.inv_t);
        // share identical transforms (a mesh wraps every triangle in the same one, json.rs:639-648)
        uint32_t idx = V4R_NO_XFORM;
        if (!f_xforms_.empty() && std::memcmp(&f_xforms_.back(), &x, sizeof x) == 0) idx = (uint32_t)f_xforms_.size() - 1;
        if (idx == V4R_NO_XFORM) {
            for (size_t i = 0; i < f_xforms_.size() && i < 64; ++i)
                if (std::memcmp(&f_xforms_[i], &x, sizeof x) == 0) {
                    idx = (uint32_t)i;
                    break;
                }
        }
        if (idx == V4R_NO_XFORM) {
            idx = (uint32_t)f_xforms_.size();
            f_xforms_.push_back(x);
        }
        p.xform = idx;
    }
    dst.push_back(p);
    return (uint32_t)dst.size() - 1;
}

const v4r_scene_desc &Scene::flatten(const RendererParam &rp, const CameraParam &cp) {
    if (rp.width == 0 || rp.height == 0) throw std::runtime_error("renderer: width and height must be positive");
    f_nodes_.clear();
    f_prims_.clear();
    f_medium_prims_.clear();
    f_xforms_.clear();
    f_tri_.clear();
    f_media_.clear();
    f_images_.clear();
    f_texels_.clear();
    f_perlin_.clear();

    BVH bvh = build_bvh();
    uint32_t depth = 0;
    bvh.flatten(f_nodes_, depth);

    std::vector<bool> is_light(hittables_.size(), false);
    for (uint32_t l : lights_) is_light[l] = true;
    for (size_t i = 0; i < hittables_.size(); ++i) push_prim(hittables_[i], f_prims_, pretransform_ && !is_light[i]);

    for (auto &img : images_) {
        v4r_image d{};
        d.offset = f_texels_.size();
        d.width = img.width;
        d.height = img.height;
        f_texels_.insert(f_texels_.end(), img.rgb.begin(), img.rgb.end());
        f_images_.push_back(d);
    }
    for (auto &pt : perlin_) {
        v4r_perlin d;
        std::memcpy(d.vec, pt.vec, sizeof d.vec);
        std::memcpy(d.perm_x, pt.perm_x, 256);
        std::memcpy(d.perm_y, pt.perm_y, 256);
        std::memcpy(d.perm_z, pt.perm_z, 256);
        f_perlin_.push_back(d);
    }

    v4r_scene_desc &d = desc_;
    std::memset(&d, 0, sizeof d);
    d.abi_version = V4R_ABI_VERSION;
    d.flags = flags_;
    d.n_nodes = (uint32_t)f_nodes_.size();
    d.bvh_depth = depth;
    d.nodes = f_nodes_.data();
    d.n_prims = (uint32_t)f_prims_.size();
    d.prims = f_prims_.data();
    d.prim_material = materials_of_.data();
    d.n_xforms = (uint32_t)f_xforms_.size();
    d.xforms = f_xforms_.data();
    d.n_tri_shading = (uint32_t)f_tri_.size();
    d.tri_shading = f_tri_.data();
    d.n_media = (uint32_t)f_media_.size();
    d.media = f_media_.data();
    d.n_medium_prims = (uint32_t)f_medium_prims_.size();
    d.medium_prims = f_medium_prims_.data();
    d.n_materials = (uint32_t)materials_.size();
    d.materials = materials_.data();
    d.n_textures = (uint32_t)textures_.size();
    d.textures = textures_.data();
    d.n_images = (uint32_t)f_images_.size();
    d.images = f_images_.data();
    d.texels = f_texels_.data();
    d.texel_bytes = f_texels_.size();
    d.n_perlin = (uint32_t)f_perlin_.size();
    d.perlin = f_perlin_.data();
    d.n_lights = (uint32_t)lights_.size();
    d.lights = lights_.data();
    background_.store(d.background);
    environment_.store(d.environment);
    // Renderer::new, renderer.rs:84-93
    float default_aspect = (float)rp.width / (float)rp.height;
    d.camera = make_camera(cp, default_aspect);
    d.width = rp.width;
    d.height = rp.height;
    d.max_depth = rp.has_max_depth ? rp.max_depth : 20u;          // renderer.rs:331
    d.antialias = (rp.has_antialias && !rp.antialias) ? 0u : 1u;  // renderer.rs:103
    return d;
}

}  // namespace v4ray
