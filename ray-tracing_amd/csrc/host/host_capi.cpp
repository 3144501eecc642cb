// extern "C" surface of the host-side scene pipeline (see include/v4ray_host.h).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>

#include "scene.hpp"
#include "v4ray_host.h"

using namespace v4ray;

struct v4rh_scene {
    std::unique_ptr<Scene> scene;
    RendererParam renderer;
    CameraParam camera;
};

struct v4rh_shape {
    std::vector<Hittable> items;
};

namespace {
thread_local std::string g_error;

int fail(const std::exception &e) {
    g_error = e.what();
    return -1;
}
int fail(const char *msg) {
    g_error = msg;
    return -1;
}

template <class F>
int guarded(F &&f) {
    try {
        return f();
    } catch (const std::exception &e) {
        return fail(e);
    } catch (...) {
        return fail("unknown error");
    }
}

template <class F>
v4rh_shape *guarded_shape(F &&f) {
    try {
        auto *s = new v4rh_shape();
        s->items = f();
        return s;
    } catch (const std::exception &e) {
        fail(e);
    } catch (...) {
        fail("unknown error");
    }
    return nullptr;
}
}  // namespace

extern "C" {

const char *v4rh_last_error(void) { return g_error.c_str(); }

int v4rh_scene_new(const float background[3], const float environment[3], v4rh_scene **out) {
    return guarded([&] {
        if (!out || !background) throw std::runtime_error("v4rh_scene_new: NULL argument");
        auto *s = new v4rh_scene();
        Vec3 env = environment ? Vec3(environment) : Vec3(0, 0, 0);
        s->scene = std::make_unique<Scene>(Vec3(background), env);
        *out = s;
        return 0;
    });
}

void v4rh_scene_free(v4rh_scene *s) { delete s; }

static int adopt(LoadedScene &&ls, v4rh_scene **out) {
    auto *s = new v4rh_scene();
    s->scene = std::move(ls.scene);
    s->renderer = ls.renderer;
    s->camera = ls.camera;
    *out = s;
    return 0;
}

int v4rh_scene_load_json_file(const char *path, uint64_t perlin_seed, v4rh_scene **out) {
    return guarded([&] {
        if (!path || !out) throw std::runtime_error("v4rh_scene_load_json_file: NULL argument");
        return adopt(load_scene_file(path, perlin_seed), out);
    });
}

int v4rh_scene_load_json_text(const char *text, const char *base_dir, uint64_t perlin_seed, v4rh_scene **out) {
    return guarded([&] {
        if (!text || !out) throw std::runtime_error("v4rh_scene_load_json_text: NULL argument");
        return adopt(load_scene_json(text, base_dir ? base_dir : "", perlin_seed), out);
    });
}

int v4rh_texture_solid_color(v4rh_scene *s, const float color[3]) {
    return guarded([&] { return (int)s->scene->add_solid_color(Vec3(color)); });
}
int v4rh_texture_checker(v4rh_scene *s, int odd, int even, float density) {
    return guarded([&] {
        if (odd < 0 || even < 0) throw std::runtime_error("checker: negative texture id");
        return (int)s->scene->add_checker((uint32_t)odd, (uint32_t)even, density);
    });
}
int v4rh_texture_image(v4rh_scene *s, const uint8_t *rgb, uint32_t width, uint32_t height) {
    return guarded([&] {
        if (!rgb) throw std::runtime_error("image: NULL pixels");
        Image img;
        img.width = width;
        img.height = height;
        img.rgb.assign(rgb, rgb + (size_t)width * height * 3);
        return (int)s->scene->add_image(std::move(img));
    });
}
int v4rh_texture_image_file(v4rh_scene *s, const char *path) {
    return guarded([&] { return (int)s->scene->add_image(load_image(path)); });
}
int v4rh_texture_noise(v4rh_scene *s, float scale, uint32_t depth, uint64_t seed) {
    return guarded([&] { return (int)s->scene->add_noise(scale, depth, seed); });
}
int v4rh_material_lambertian(v4rh_scene *s, int texture) {
    return guarded([&] {
        if (texture < 0) throw std::runtime_error("lambertian: negative texture id");
        return (int)s->scene->add_lambertian((uint32_t)texture);
    });
}
int v4rh_material_metal(v4rh_scene *s, const float albedo[3], float fuzz) {
    return guarded([&] { return (int)s->scene->add_metal(Vec3(albedo), fuzz); });
}
int v4rh_material_dielectric(v4rh_scene *s, float ir) {
    return guarded([&] { return (int)s->scene->add_dielectric(ir); });
}
int v4rh_material_diffuse_light(v4rh_scene *s, int texture) {
    return guarded([&] {
        if (texture < 0) throw std::runtime_error("diffuse light: negative texture id");
        return (int)s->scene->add_diffuse_light((uint32_t)texture);
    });
}
int v4rh_material_isotropic(v4rh_scene *s, int texture) {
    return guarded([&] {
        if (texture < 0) throw std::runtime_error("isotropic: negative texture id");
        return (int)s->scene->add_isotropic((uint32_t)texture);
    });
}

v4rh_shape *v4rh_shape_sphere(const float center[3], float radius) {
    return guarded_shape([&] { return std::vector<Hittable>{make_sphere(Vec3(center), radius)}; });
}
v4rh_shape *v4rh_shape_rect(uint32_t rect_type, float a0, float a1, float b0, float b1, float k, int positive) {
    return guarded_shape([&] { return std::vector<Hittable>{make_rect(rect_type, a0, a1, b0, b1, k, positive != 0)}; });
}
v4rh_shape *v4rh_shape_triangle(const float vertices[9], const float *normals9, const float *uvs6) {
    return guarded_shape([&] {
        Vec3 v[3], n[3];
        float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        for (int k = 0; k < 3; ++k) v[k] = Vec3(vertices + 3 * k);
        if (normals9) {
            for (int k = 0; k < 3; ++k) n[k] = Vec3(normals9 + 3 * k);
        } else {  // json.rs:580-586
            Vec3 nrm = normalize(cross(v[1] - v[0], v[2] - v[1]));
            n[0] = n[1] = n[2] = nrm;
        }
        if (uvs6)
            for (int k = 0; k < 3; ++k) {
                uv[k][0] = uvs6[2 * k];
                uv[k][1] = uvs6[2 * k + 1];
            }
        return std::vector<Hittable>{make_triangle(v, n, uv)};
    });
}
v4rh_shape *v4rh_shape_cuboid(const float p0[3], const float p1[3]) {
    return guarded_shape([&] { return make_cuboid(Vec3(p0), Vec3(p1)); });
}
v4rh_shape *v4rh_shape_mesh_file(const char *path, int model, int subdivide_levels) {
    return guarded_shape([&] {
        std::vector<ObjMesh> models = load_obj(path);
        if (models.empty()) throw std::runtime_error("obj file holds no model");
        size_t idx = model < 0 ? 0 : (size_t)model;
        if (idx >= models.size()) throw std::runtime_error("model index out of range");
        if (subdivide_levels > 0) return mesh_to_triangles(subdivide(models[idx], subdivide_levels));
        return mesh_to_triangles(models[idx]);
    });
}
v4rh_shape *v4rh_shape_constant_medium(const v4rh_shape *boundary, float density) {
    return guarded_shape([&] {
        if (!boundary) throw std::runtime_error("constant medium: NULL boundary");
        return std::vector<Hittable>{make_medium(boundary->items, density)};
    });
}
int v4rh_shape_transform(v4rh_shape *shape, const float matrix9[9], const float translate[3]) {
    return guarded([&] {
        if (!shape) throw std::runtime_error("v4rh_shape_transform: NULL shape");
        Affine a;
        if (matrix9) std::memcpy(a.m, matrix9, sizeof a.m);
        if (translate) std::memcpy(a.t, translate, sizeof a.t);
        for (auto &h : shape->items) h = with_transform(std::move(h), a);
        return 0;
    });
}
int v4rh_shape_count(const v4rh_shape *shape) { return shape ? (int)shape->items.size() : 0; }
void v4rh_shape_free(v4rh_shape *shape) { delete shape; }

int v4rh_scene_add(v4rh_scene *s, const v4rh_shape *shape, int material, int important) {
    return guarded([&] {
        if (!s || !shape) throw std::runtime_error("v4rh_scene_add: NULL argument");
        if (material < 0) throw std::runtime_error("v4rh_scene_add: negative material id");
        for (auto &h : shape->items) {
            if (important) {
                if (h.samplable) s->scene->add_important(h, (uint32_t)material);
                else fprintf(stderr, "importance sampling on unsupported shape!\n");  // json.rs:692
            } else {
                s->scene->add(h, (uint32_t)material);
            }
        }
        return 0;
    });
}
int v4rh_scene_len(const v4rh_scene *s) { return s ? (int)s->scene->len() : 0; }

int v4rh_scene_set_renderer(v4rh_scene *s, const v4rh_renderer_param *p) {
    return guarded([&] {
        if (!s || !p) throw std::runtime_error("v4rh_scene_set_renderer: NULL argument");
        RendererParam r;
        r.width = p->width;
        r.height = p->height;
        r.has_max_depth = p->max_depth >= 0;
        if (r.has_max_depth) r.max_depth = (uint32_t)p->max_depth;
        r.has_antialias = p->antialias >= 0;
        r.antialias = p->antialias != 0;
        s->renderer = r;
        return 0;
    });
}
int v4rh_scene_get_renderer(const v4rh_scene *s, v4rh_renderer_param *p) {
    if (!s || !p) return fail("v4rh_scene_get_renderer: NULL argument");
    p->width = s->renderer.width;
    p->height = s->renderer.height;
    p->max_depth = s->renderer.has_max_depth ? (int32_t)s->renderer.max_depth : -1;
    p->antialias = s->renderer.has_antialias ? (s->renderer.antialias ? 1 : 0) : -1;
    return 0;
}
int v4rh_scene_set_camera(v4rh_scene *s, const v4rh_camera_param *p) {
    return guarded([&] {
        if (!s || !p) throw std::runtime_error("v4rh_scene_set_camera: NULL argument");
        CameraParam c;
        c.look_from = Vec3(p->look_from);
        c.look_at = Vec3(p->look_at);
        c.vfov = p->vfov;
        c.has_up = p->has & V4RH_CAM_UP;
        c.up = Vec3(p->up);
        c.has_aspect = p->has & V4RH_CAM_ASPECT;
        c.aspect_ratio = p->aspect_ratio;
        c.has_aperture = p->has & V4RH_CAM_APERTURE;
        c.aperture = p->aperture;
        c.has_focus = p->has & V4RH_CAM_FOCUS;
        c.focus_dist = p->focus_dist;
        c.has_time0 = p->has & V4RH_CAM_TIME0;
        c.time0 = p->time0;
        c.has_time1 = p->has & V4RH_CAM_TIME1;
        c.time1 = p->time1;
        s->camera = c;
        return 0;
    });
}
int v4rh_scene_get_camera(const v4rh_scene *s, v4rh_camera_param *p) {
    if (!s || !p) return fail("v4rh_scene_get_camera: NULL argument");
    const CameraParam &c = s->camera;
    std::memset(p, 0, sizeof *p);
    c.look_from.store(p->look_from);
    c.look_at.store(p->look_at);
    p->vfov = c.vfov;
    c.up.store(p->up);
    p->aspect_ratio = c.aspect_ratio;
    p->aperture = c.aperture;
    p->focus_dist = c.focus_dist;
    p->time0 = c.time0;
    p->time1 = c.time1;
    p->has = (c.has_up ? V4RH_CAM_UP : 0) | (c.has_aspect ? V4RH_CAM_ASPECT : 0) | (c.has_aperture ? V4RH_CAM_APERTURE : 0) |
             (c.has_focus ? V4RH_CAM_FOCUS : 0) | (c.has_time0 ? V4RH_CAM_TIME0 : 0) | (c.has_time1 ? V4RH_CAM_TIME1 : 0);
    return 0;
}
int v4rh_scene_set_options(v4rh_scene *s, int pretransform_triangles, uint32_t flags) {
    if (!s) return fail("v4rh_scene_set_options: NULL scene");
    s->scene->set_pretransform_triangles(pretransform_triangles != 0);
    s->scene->set_flags(flags);
    return 0;
}

int v4rh_scene_flatten(v4rh_scene *s, const v4r_scene_desc **out) {
    return guarded([&] {
        if (!s || !out) throw std::runtime_error("v4rh_scene_flatten: NULL argument");
        *out = &s->scene->flatten(s->renderer, s->camera);
        return 0;
    });
}

int v4rh_bvh_build(const float *boxes, uint32_t n, v4rh_bvh_node *nodes, uint32_t *n_nodes) {
    return guarded([&] {
        if ((!boxes && n) || !n_nodes) throw std::runtime_error("v4rh_bvh_build: NULL argument");
        std::vector<AABB> shapes(n);
        for (uint32_t i = 0; i < n; ++i) shapes[i] = AABB::with_bounds(Vec3(boxes + 6 * i), Vec3(boxes + 6 * i + 3));
        BVH bvh = BVH::build(shapes);
        *n_nodes = (uint32_t)bvh.nodes.size();
        if (nodes)
            for (size_t i = 0; i < bvh.nodes.size(); ++i) {
                const BVHNode &b = bvh.nodes[i];
                v4rh_bvh_node &o = nodes[i];
                std::memset(&o, 0, sizeof o);
                o.parent_index = b.parent_index;
                o.depth = b.depth;
                o.leaf = b.leaf ? 1u : 0u;
                o.shape_index = b.shape_index;
                if (!b.leaf)
                    for (int c = 0; c < 2; ++c) {
                        o.children_index[c] = b.children_index[c];
                        b.children_aabb[c].min.store(o.children_aabb[c]);
                        b.children_aabb[c].max.store(o.children_aabb[c] + 3);
                    }
            }
        return 0;
    });
}

int v4rh_scene_bounding_box(const v4rh_scene *s, uint32_t index, float out6[6]) {
    return guarded([&] {
        if (!s || !out6) throw std::runtime_error("v4rh_scene_bounding_box: NULL argument");
        if (index >= s->scene->len()) throw std::runtime_error("v4rh_scene_bounding_box: index out of range");
        AABB b = s->scene->hittables()[index].bounding_box();
        b.min.store(out6);
        b.max.store(out6 + 3);
        return 0;
    });
}

int v4rh_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb, uint64_t capacity) {
    return guarded([&] {
        if (!path || !width || !height) throw std::runtime_error("v4rh_image_load: NULL argument");
        Image img = load_image(path);
        *width = img.width;
        *height = img.height;
        if (rgb) {
            if (capacity < img.rgb.size()) throw std::runtime_error("v4rh_image_load: buffer too small");
            std::memcpy(rgb, img.rgb.data(), img.rgb.size());
        }
        return 0;
    });
}

// RenderResult::get_raw, src/renderer.rs:378-402: scale by 1/n, sqrt, scale by 256, clamp to
// [0.5, 255.5], truncate to u8; NaN -> 0.
int v4rh_tonemap_u8(const float *sums, uint64_t n_floats, uint32_t n_passes, uint8_t *out) {
    if (!sums || !out) return fail("v4rh_tonemap_u8: NULL argument");
    if (n_passes == 0) return fail("v4rh_tonemap_u8: no passes accumulated");
    float scale1 = 1.0f / (float)n_passes;
    for (uint64_t i = 0; i < n_floats; ++i) {
        float v = std::sqrt(sums[i] * scale1) * 256.0f;
        if (v != v) {
            out[i] = 0;
            continue;
        }
        v = v < 0.5f ? 0.5f : (v > 255.5f ? 255.5f : v);
        out[i] = (uint8_t)v;
    }
    return 0;
}

int v4rh_write_image(const char *path, const uint8_t *rgb, uint32_t width, uint32_t height) {
    return guarded([&] {
        if (!path || !rgb) throw std::runtime_error("v4rh_write_image: NULL argument");
        std::string p(path);
        std::ofstream f(p, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open output image: " + p);
        // the format follows the file name's extension, as `image::save` does for the reference (src/main.rs:70-72):
        // .png, .ppm, .tga, anything else (.bmp) a 24-bit BMP; a format this writer lacks (.jpg, ...) is an error
        auto ends_with = [&](const char *ext) {
            size_t n = std::strlen(ext);
            if (p.size() < n) return false;
            for (size_t k = 0; k < n; ++k)
                if (std::tolower((unsigned char)p[p.size() - n + k]) != ext[k]) return false;
            return true;
        };
        for (const char *ext : {".jpg", ".jpeg", ".gif", ".tif", ".tiff", ".webp", ".exr", ".hdr"})
            if (ends_with(ext)) throw std::runtime_error("output format not supported (use .png, .bmp, .ppm or .tga): " + p);
        bool ppm = ends_with(".ppm");
        if (ends_with(".png")) {
            // PNG, 8-bit RGB, filter 0 on every row, zlib stream of stored (uncompressed) deflate blocks
            auto put_be32 = [](std::vector<uint8_t> &v, uint32_t x) {
                for (int sh = 24; sh >= 0; sh -= 8) v.push_back((uint8_t)(x >> sh));
            };
            static uint32_t crc_table[256];
            static bool have_table = false;
            if (!have_table) {
                for (uint32_t n = 0; n < 256; ++n) {
                    uint32_t c = n;
                    for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                    crc_table[n] = c;
                }
                have_table = true;
            }
            auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
                std::vector<uint8_t> out;
                put_be32(out, (uint32_t)data.size());
                out.insert(out.end(), type, type + 4);
                out.insert(out.end(), data.begin(), data.end());
                uint32_t c = 0xFFFFFFFFu;
                for (size_t k = 4; k < out.size(); ++k) c = crc_table[(c ^ out[k]) & 255u] ^ (c >> 8);
                put_be32(out, c ^ 0xFFFFFFFFu);
                f.write((const char *)out.data(), (std::streamsize)out.size());
            };
            const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
            f.write((const char *)magic, 8);
            std::vector<uint8_t> ihdr;
            put_be32(ihdr, width);
            put_be32(ihdr, height);
            for (uint8_t b : {8, 2, 0, 0, 0}) ihdr.push_back(b);  // bit depth, colour type RGB, deflate, adaptive filtering, no interlace
            chunk("IHDR", ihdr);
            std::vector<uint8_t> raw;
            raw.reserve((size_t)height * ((size_t)width * 3 + 1));
            for (uint32_t y = 0; y < height; ++y) {
                raw.push_back(0);
                raw.insert(raw.end(), rgb + (size_t)y * width * 3, rgb + (size_t)(y + 1) * width * 3);
            }
            std::vector<uint8_t> z = {0x78, 0x01};
            uint32_t a = 1, b = 0;  // Adler-32 of the raw data
            for (size_t off = 0; off < raw.size() || off == 0;) {
                size_t n = std::min<size_t>(65535, raw.size() - off);
                z.push_back(off + n >= raw.size() ? 1 : 0);
                z.push_back((uint8_t)n);
                z.push_back((uint8_t)(n >> 8));
                z.push_back((uint8_t)~n);
                z.push_back((uint8_t)(~n >> 8));
                z.insert(z.end(), raw.begin() + (std::ptrdiff_t)off, raw.begin() + (std::ptrdiff_t)(off + n));
                for (size_t k = off; k < off + n; ++k) {
                    a = (a + raw[k]) % 65521u;
                    b = (b + a) % 65521u;
                }
                off += n;
                if (n == 0) break;
            }
            put_be32(z, (b << 16) | a);
            chunk("IDAT", z);
            chunk("IEND", {});
        } else if (ends_with(".tga")) {  // uncompressed true-colour TGA, top-left origin, BGR
            uint8_t hdr[18] = {0};
            hdr[2] = 2;
            hdr[12] = (uint8_t)width;
            hdr[13] = (uint8_t)(width >> 8);
            hdr[14] = (uint8_t)height;
            hdr[15] = (uint8_t)(height >> 8);
            hdr[16] = 24;
            hdr[17] = 0x20;
            if (width > 65535 || height > 65535) throw std::runtime_error("image too large for TGA: " + p);
            f.write((const char *)hdr, 18);
            std::vector<uint8_t> line((size_t)width * 3);
            for (uint32_t y = 0; y < height; ++y) {
                const uint8_t *src = rgb + (size_t)y * width * 3;
                for (uint32_t x = 0; x < width; ++x) {
                    line[3 * x] = src[3 * x + 2];
                    line[3 * x + 1] = src[3 * x + 1];
                    line[3 * x + 2] = src[3 * x];
                }
                f.write((const char *)line.data(), (std::streamsize)line.size());
            }
        } else if (ppm) {
            f << "P6\n" << width << " " << height << "\n255\n";
            f.write((const char *)rgb, (std::streamsize)((size_t)width * height * 3));
        } else {  // 24-bit BMP, bottom-up rows, BGR, rows padded to 4 bytes
            uint32_t row = (width * 3 + 3) & ~3u;
            uint32_t data_size = row * height, file_size = 54 + data_size;
            uint8_t hdr[54] = {0};
            hdr[0] = 'B';
            hdr[1] = 'M';
            auto put32 = [&](int off, uint32_t v) {
                hdr[off] = (uint8_t)v;
                hdr[off + 1] = (uint8_t)(v >> 8);
                hdr[off + 2] = (uint8_t)(v >> 16);
                hdr[off + 3] = (uint8_t)(v >> 24);
            };
            put32(2, file_size);
            put32(10, 54);
            put32(14, 40);
            put32(18, width);
            put32(22, height);
            hdr[26] = 1;
            hdr[28] = 24;
            put32(34, data_size);
            put32(38, 2835);
            put32(42, 2835);
            f.write((const char *)hdr, 54);
            std::vector<uint8_t> line(row, 0);
            for (uint32_t y = 0; y < height; ++y) {
                const uint8_t *src = rgb + (size_t)(height - 1 - y) * width * 3;
                for (uint32_t x = 0; x < width; ++x) {
                    line[3 * x] = src[3 * x + 2];
                    line[3 * x + 1] = src[3 * x + 1];
                    line[3 * x + 2] = src[3 * x];
                }
                f.write((const char *)line.data(), row);
            }
        }
        if (!f) throw std::runtime_error("failed writing output image: " + p);
        return 0;
    });
}

namespace {
struct CheckpointHeader {
    char magic[8];
    uint32_t width, height, n_passes, reserved;
    uint64_t seed, checksum, payload_bytes;
};
static_assert(sizeof(CheckpointHeader) == 48, "checkpoint header layout");
uint64_t fnv1a64(const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}
}  // namespace

int v4rh_checkpoint_save(const char *path, const float *sums, uint32_t width, uint32_t height, uint32_t n_passes, uint64_t seed) {
    return guarded([&] {
        if (!path || !sums) throw std::runtime_error("v4rh_checkpoint_save: NULL argument");
        CheckpointHeader h;
        std::memcpy(h.magic, "V4RCKPT1", 8);
        h.width = width;
        h.height = height;
        h.n_passes = n_passes;
        h.reserved = 0;
        h.seed = seed;
        h.payload_bytes = (uint64_t)width * height * 3 * sizeof(float);
        h.checksum = fnv1a64(sums, (size_t)h.payload_bytes);
        std::string p(path), tmp = p + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
            if (!f) throw std::runtime_error("cannot open checkpoint for writing: " + tmp);
            f.write((const char *)&h, sizeof h);
            f.write((const char *)sums, (std::streamsize)h.payload_bytes);
            f.flush();
            if (!f) throw std::runtime_error("failed writing checkpoint: " + tmp);
        }
        if (std::rename(tmp.c_str(), p.c_str()) != 0) throw std::runtime_error("cannot move checkpoint into place: " + p);
        return 0;
    });
}

int v4rh_checkpoint_load(const char *path, uint32_t *width, uint32_t *height, uint32_t *n_passes, uint64_t *seed, float *sums,
                         uint64_t capacity_floats) {
    return guarded([&] {
        if (!path) throw std::runtime_error("v4rh_checkpoint_load: NULL path");
        std::string p(path);
        std::ifstream f(p, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open checkpoint: " + p);
        CheckpointHeader h;
        f.read((char *)&h, sizeof h);
        if (!f || std::memcmp(h.magic, "V4RCKPT1", 8) != 0) throw std::runtime_error("not a checkpoint file: " + p);
        if (h.payload_bytes != (uint64_t)h.width * h.height * 3 * sizeof(float)) throw std::runtime_error("checkpoint header is inconsistent: " + p);
        if (width) *width = h.width;
        if (height) *height = h.height;
        if (n_passes) *n_passes = h.n_passes;
        if (seed) *seed = h.seed;
        if (!sums) return 0;
        if (capacity_floats < h.payload_bytes / sizeof(float)) throw std::runtime_error("checkpoint does not fit the buffer: " + p);
        f.read((char *)sums, (std::streamsize)h.payload_bytes);
        if (!f || (uint64_t)f.gcount() != h.payload_bytes) throw std::runtime_error("checkpoint is truncated: " + p);
        if (fnv1a64(sums, (size_t)h.payload_bytes) != h.checksum) throw std::runtime_error("checkpoint checksum mismatch: " + p);
        return 0;
    });
}

}  // extern "C"
