// Host-side sanitizer run of the host library's C API: the reference's scene files through the JSON loader, the BVH
// builder and the flattener; every image writer; the JPEG / PPM readers; checkpoint save / load with damaged files; JSON
// texts cut short.  Built from the host sources with -fsanitize=address,undefined (tests/test_host_sanitized.py gives
// the command); every call must end in a status code, the sanitizers abort on anything else.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "v4ray_host.h"

static int fail(const char *what) {
    std::fprintf(stderr, "host_sanitize: %s: %s\n", what, v4rh_last_error());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string assets = argv[1], tmp = argv[2];
    long checked = 0;
    // 1. scene files: load, flatten (BVH build), and the same text cut at many places (must be rejected or loaded, never crash)
    for (const char *name : {"scene.json", "zy_scene.json"}) {
        const std::string path = assets + "/" + name;
        v4rh_scene *s = nullptr;
        if (v4rh_scene_load_json_file(path.c_str(), 7, &s) != 0) return fail(name);
        const v4r_scene_desc *d = nullptr;
        if (v4rh_scene_flatten(s, &d) != 0 || !d || d->n_prims == 0) return fail("flatten");
        float box[6];
        if (v4rh_scene_bounding_box(s, 0, box) != 0) return fail("bounding box");
        (void)v4rh_scene_bounding_box(s, 0x7fffffffu, box);  // out of range: an error, not a fault
        v4rh_scene_free(s);
        std::ifstream f(path);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string text = ss.str();
        for (size_t cut = 1; cut < text.size(); cut += text.size() / 97 + 1) {
            v4rh_scene *t = nullptr;
            if (v4rh_scene_load_json_text(text.substr(0, cut).c_str(), assets.c_str(), 7, &t) == 0) v4rh_scene_free(t);
            ++checked;
        }
    }
    // 2. a scene made through the builder calls, with a mesh, a medium, a transform and an image texture
    {
        const float bg[3] = {0.1f, 0.2f, 0.3f}, env[3] = {0, 0, 0};
        v4rh_scene *s = nullptr;
        if (v4rh_scene_new(bg, env, &s) != 0) return fail("scene_new");
        const float white[3] = {0.8f, 0.8f, 0.8f};
        int tex = v4rh_texture_solid_color(s, white);
        int img = v4rh_texture_image_file(s, (assets + "/earthmap.jpg").c_str());
        int noise = v4rh_texture_noise(s, 4.0f, 7, 3);
        int chk = v4rh_texture_checker(s, tex, noise, 10.0f);
        if (tex < 0 || img < 0 || noise < 0 || chk < 0) return fail("textures");
        int lam = v4rh_material_lambertian(s, chk), earth = v4rh_material_lambertian(s, img), glass = v4rh_material_dielectric(s, 1.5f);
        int light = v4rh_material_diffuse_light(s, tex), iso = v4rh_material_isotropic(s, tex);
        const float c[3] = {0, 1, 0}, p0[3] = {-1, 0, -1}, p1[3] = {1, 2, 1};
        v4rh_shape *sp = v4rh_shape_sphere(c, 1.0f), *cube = v4rh_shape_cuboid(p0, p1), *rect = v4rh_shape_rect(1, -3, 3, -3, 3, 5, 0);
        v4rh_shape *mesh = v4rh_shape_mesh_file((assets + "/bunny.obj").c_str(), 0, 0), *fog = v4rh_shape_constant_medium(cube, 0.2f);
        if (!sp || !cube || !rect || !mesh || !fog) return fail("shapes");
        const float m[9] = {2, 0, 0, 0, 2, 0.5f, 0, 0, 2}, t[3] = {0, -1, 0};
        if (v4rh_shape_transform(mesh, m, t) != 0) return fail("transform");
        if (v4rh_scene_add(s, sp, earth, 0) != 0 || v4rh_scene_add(s, mesh, glass, 0) != 0 || v4rh_scene_add(s, rect, light, 1) != 0 ||
            v4rh_scene_add(s, fog, iso, 0) != 0 || v4rh_scene_add(s, cube, lam, 0) != 0)
            return fail("add");
        (void)v4rh_scene_add(s, sp, 12345, 0);  // unknown material: an error
        v4rh_renderer_param rp = {64, 48, 10, 1};
        if (v4rh_scene_set_renderer(s, &rp) != 0) return fail("renderer param");
        for (int pre = 0; pre < 2; ++pre) {
            if (v4rh_scene_set_options(s, pre, 0) != 0) return fail("options");
            const v4r_scene_desc *d = nullptr;
            if (v4rh_scene_flatten(s, &d) != 0 || !d) return fail("flatten built scene");
            // every child word of the tree must name a node or a primitive of the scene
            for (uint32_t i = 0; i < d->n_nodes; ++i)
                for (uint32_t w : {d->nodes[i].child0, d->nodes[i].child1})
                    if ((w & V4R_LEAF_BIT) ? (w & ~V4R_LEAF_BIT) >= d->n_prims : w >= d->n_nodes) return fail("child word out of range");
            checked += d->n_nodes;
        }
        for (v4rh_shape *x : {sp, cube, rect, mesh, fog}) v4rh_shape_free(x);
        v4rh_scene_free(s);
    }
    // 3. image writers and readers
    {
        const uint32_t w = 37, h = 23;
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (uint8_t)(i * 7);
        for (const char *ext : {".png", ".ppm", ".tga", ".bmp"})
            if (v4rh_write_image((tmp + "/img" + ext).c_str(), rgb.data(), w, h) != 0) return fail(ext);
        if (v4rh_write_image((tmp + "/img.jpg").c_str(), rgb.data(), w, h) == 0) return fail("a .jpg was written");
        uint32_t rw = 0, rh = 0;
        if (v4rh_image_load((tmp + "/img.ppm").c_str(), &rw, &rh, nullptr, 0) != 0 || rw != w || rh != h) return fail("ppm size");
        std::vector<uint8_t> back((size_t)w * h * 3);
        if (v4rh_image_load((tmp + "/img.ppm").c_str(), &rw, &rh, back.data(), back.size()) != 0 || back != rgb) return fail("ppm round trip");
        if (v4rh_image_load((tmp + "/img.ppm").c_str(), &rw, &rh, back.data(), 10) == 0) return fail("a short buffer was accepted");
        if (v4rh_image_load((assets + "/earthmap.jpg").c_str(), &rw, &rh, nullptr, 0) != 0) return fail("jpeg size");
        std::vector<uint8_t> earth((size_t)rw * rh * 3);
        if (v4rh_image_load((assets + "/earthmap.jpg").c_str(), &rw, &rh, earth.data(), earth.size()) != 0) return fail("jpeg");
        std::vector<float> sums(earth.size());
        for (size_t i = 0; i < sums.size(); ++i) sums[i] = (float)earth[i] / 64.0f;
        std::vector<uint8_t> tm(sums.size());
        if (v4rh_tonemap_u8(sums.data(), sums.size(), 4, tm.data()) != 0) return fail("tonemap");
        checked += 8;
    }
    // 4. checkpoints: round trip, then damaged and truncated copies must be refused
    {
        const uint32_t w = 31, h = 17;
        std::vector<float> sums((size_t)w * h * 3);
        for (size_t i = 0; i < sums.size(); ++i) sums[i] = (float)i * 0.25f;
        const std::string path = tmp + "/job.ckpt";
        if (v4rh_checkpoint_save(path.c_str(), sums.data(), w, h, 12, 0xABCDEFull) != 0) return fail("checkpoint save");
        uint32_t rw, rh, rn;
        uint64_t seed;
        std::vector<float> back(sums.size());
        if (v4rh_checkpoint_load(path.c_str(), &rw, &rh, &rn, &seed, back.data(), back.size()) != 0 || back != sums || rn != 12 || seed != 0xABCDEFull)
            return fail("checkpoint round trip");
        if (v4rh_checkpoint_load(path.c_str(), &rw, &rh, &rn, &seed, back.data(), 5) == 0) return fail("a short checkpoint buffer was accepted");
        std::ifstream f(path, std::ios::binary);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string good = ss.str();
        for (size_t k = 0; k < good.size(); k += good.size() / 61 + 1) {
            std::string bad = good;
            bad[k] = (char)(bad[k] ^ 0x5A);
            std::ofstream(tmp + "/bad.ckpt", std::ios::binary).write(bad.data(), (std::streamsize)bad.size());
            if (k >= 8 && k < 24) continue;  // a different but plausible size or pass count is caught by the length / checksum below or is a different job
            if (v4rh_checkpoint_load((tmp + "/bad.ckpt").c_str(), &rw, &rh, &rn, &seed, back.data(), back.size()) == 0 && k >= 48)
                return fail("a damaged checkpoint payload was accepted");
            std::ofstream(tmp + "/cut.ckpt", std::ios::binary).write(good.data(), (std::streamsize)k);
            if (v4rh_checkpoint_load((tmp + "/cut.ckpt").c_str(), &rw, &rh, &rn, &seed, back.data(), back.size()) == 0) return fail("a truncated checkpoint was accepted");
            ++checked;
        }
    }
    std::printf("host_sanitize ok: %ld checks\n", checked);
    return 0;
}
