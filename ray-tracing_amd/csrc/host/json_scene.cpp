// Scene files in the reference's CLI schema (SceneParam, src/json.rs:14-250) and the recursive,
// name-resolving builder (visit_texture / visit_material / visit_shape / visit_object /
// build_scene, src/json.rs:312-720).
#include <cstdio>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

#include "json.hpp"
#include "scene.hpp"

namespace v4ray {

namespace {
using json::Value;

float num(const Value &v, const char *what) { return (float)v.as_number(what); }

Vec3 vec3(const Value &v, const char *what) {
    if (!v.is_array() || v.arr.size() != 3) throw std::runtime_error(std::string("expected [x,y,z] for ") + what);
    return Vec3(num(v.arr[0], what), num(v.arr[1], what), num(v.arr[2], what));
}

uint32_t u32(const Value &v, const char *what) {
    double d = v.as_number(what);
    if (d < 0 || d > 4294967295.0 || d != (double)(uint64_t)d)
        throw std::runtime_error(std::string("expected an unsigned integer for ") + what);
    return (uint32_t)d;
}

struct Visit {
    Scene &scene;
    std::string base_dir;
    uint64_t perlin_seed;
    uint64_t noise_ordinal = 0;
    // results cached by name (json.rs:254-256)
    std::map<std::string, std::vector<Hittable>> shapes;
    std::map<std::string, uint32_t> materials, textures;
    std::set<std::string> visiting_shapes, visiting_materials, visiting_textures;
    // top-level named tables (json.rs:260-262)
    std::map<std::string, const Value *> name_shapes, name_materials, name_textures;

    std::string resolve_path(const std::string &file) const {
        if (file.empty() || file[0] == '/' || base_dir.empty()) return file;
        return base_dir + "/" + file;
    }

    const Value &named(const std::map<std::string, const Value *> &table, const std::string &name, const char *kind) {
        auto it = table.find(name);
        if (it == table.end()) throw std::runtime_error(std::string("unknown ") + kind + " name `" + name + "`");
        return *it->second;
    }
    const Value &deref_texture(const Value &v) { return v.is_string() ? named(name_textures, v.str, "texture") : v; }
    const Value &deref_material(const Value &v) { return v.is_string() ? named(name_materials, v.str, "material") : v; }
    const Value &deref_shape(const Value &v) { return v.is_string() ? named(name_shapes, v.str, "shape") : v; }

    // visit_texture, json.rs:312-366
    uint32_t texture(const Value &t) {
        if (!t.is_object()) throw std::runtime_error("texture must be an object or a name");
        const Value *name = t.find("name");
        if (name && name->is_string()) {
            auto it = textures.find(name->str);
            if (it != textures.end()) return it->second;
            if (visiting_textures.count(name->str)) throw std::runtime_error("texture cycle through `" + name->str + "`");
            visiting_textures.insert(name->str);
        }
        const std::string &type = t.at("type").as_string("texture type");
        uint32_t id;
        if (type == "solid-color") {
            id = scene.add_solid_color(vec3(t.at("color"), "color"));
        } else if (type == "checker") {
            uint32_t odd = texture(deref_texture(t.at("odd")));
            uint32_t even = texture(deref_texture(t.at("even")));
            id = scene.add_checker(odd, even, num(t.at("density"), "density"));
        } else if (type == "image") {
            id = scene.add_image(load_image(resolve_path(t.at("file").as_string("file"))));
        } else if (type == "noise") {
            id = scene.add_noise(num(t.at("scale"), "scale"), u32(t.at("depth"), "depth"), perlin_seed + noise_ordinal);
            ++noise_ordinal;
        } else {
            throw std::runtime_error("unknown texture type `" + type + "`");
        }
        if (name && name->is_string()) {
            visiting_textures.erase(name->str);
            textures[name->str] = id;
        }
        return id;
    }

    // visit_material, json.rs:368-424
    uint32_t material(const Value &m) {
        if (!m.is_object()) throw std::runtime_error("material must be an object or a name");
        const Value *name = m.find("name");
        if (name && name->is_string()) {
            auto it = materials.find(name->str);
            if (it != materials.end()) return it->second;
            if (visiting_materials.count(name->str)) throw std::runtime_error("material cycle through `" + name->str + "`");
            visiting_materials.insert(name->str);
        }
        const std::string &type = m.at("type").as_string("material type");
        uint32_t id;
        if (type == "lambertian") id = scene.add_lambertian(texture(deref_texture(m.at("texture"))));
        else if (type == "isotropic") id = scene.add_isotropic(texture(deref_texture(m.at("albedo"))));
        else if (type == "dielectric") id = scene.add_dielectric(num(m.at("ir"), "ir"));
        else if (type == "diffuse-light") id = scene.add_diffuse_light(texture(deref_texture(m.at("emit"))));
        else if (type == "metal") id = scene.add_metal(vec3(m.at("albedo"), "albedo"), num(m.at("fuzz"), "fuzz"));
        else throw std::runtime_error("unknown material type `" + type + "`");
        if (name && name->is_string()) {
            visiting_materials.erase(name->str);
            materials[name->str] = id;
        }
        return id;
    }

    static bool positive_of(const Value &s) {
        const Value *p = s.find("positive");
        return (p && !p->is_null()) ? p->as_bool("positive") : true;  // json.rs:518
    }

    // visit_shape, json.rs:426-664
    std::vector<Hittable> shape(const Value &s) {
        if (!s.is_object()) throw std::runtime_error("shape must be an object or a name");
        const Value *name = s.find("name");
        if (name && name->is_string()) {
            auto it = shapes.find(name->str);
            if (it != shapes.end()) return it->second;
            if (visiting_shapes.count(name->str)) throw std::runtime_error("shape cycle through `" + name->str + "`");
            visiting_shapes.insert(name->str);
        }
        // json.rs:441-467
        const Value *tr = s.find("transform"), *tl = s.find("translate");
        bool has_tr = tr && !tr->is_null(), has_tl = tl && !tl->is_null();
        bool projected = has_tr || has_tl;
        Affine proj;
        if (has_tr) {
            if (!tr->is_array() || tr->arr.size() != 3) throw std::runtime_error("transform must be a 3x3 array");
            for (int r = 0; r < 3; ++r) {
                Vec3 row = vec3(tr->arr[r], "transform row");
                proj.m[3 * r] = row.x;
                proj.m[3 * r + 1] = row.y;
                proj.m[3 * r + 2] = row.z;
            }
        }
        if (has_tl) vec3(*tl, "translate").store(proj.t);
        auto wrap = [&](Hittable h) { return projected ? with_transform(std::move(h), proj) : h; };

        const std::string &type = s.at("type").as_string("shape type");
        std::vector<Hittable> result;
        if (type == "sphere") {
            result.push_back(wrap(make_sphere(vec3(s.at("center"), "center"), num(s.at("radius"), "radius"))));
        } else if (type == "xy-rect") {
            result.push_back(wrap(make_rect(V4R_PRIM_XY_RECT, num(s.at("x0"), "x0"), num(s.at("x1"), "x1"), num(s.at("y0"), "y0"),
                                            num(s.at("y1"), "y1"), num(s.at("z"), "z"), positive_of(s))));
        } else if (type == "yz-rect") {
            result.push_back(wrap(make_rect(V4R_PRIM_YZ_RECT, num(s.at("y0"), "y0"), num(s.at("y1"), "y1"), num(s.at("z0"), "z0"),
                                            num(s.at("z1"), "z1"), num(s.at("x"), "x"), positive_of(s))));
        } else if (type == "zx-rect") {
            result.push_back(wrap(make_rect(V4R_PRIM_ZX_RECT, num(s.at("z0"), "z0"), num(s.at("z1"), "z1"), num(s.at("x0"), "x0"),
                                            num(s.at("x1"), "x1"), num(s.at("y"), "y"), positive_of(s))));
        } else if (type == "constant-medium") {  // json.rs:549-577
            std::vector<Hittable> inner = shape(deref_shape(s.at("shape")));
            result.push_back(wrap(make_medium(std::move(inner), num(s.at("density"), "density"))));
        } else if (type == "triangle") {  // json.rs:578-597
            const Value &vs = s.at("vertices");
            if (!vs.is_array() || vs.arr.size() != 3) throw std::runtime_error("triangle needs 3 vertices");
            Vec3 v[3], n[3];
            float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
            for (int k = 0; k < 3; ++k) v[k] = vec3(vs.arr[k], "vertex");
            const Value *ns = s.find("normals");
            if (ns && !ns->is_null()) {
                if (!ns->is_array() || ns->arr.size() != 3) throw std::runtime_error("triangle needs 3 normals");
                for (int k = 0; k < 3; ++k) n[k] = vec3(ns->arr[k], "normal");
            } else {
                Vec3 nrm = normalize(cross(v[1] - v[0], v[2] - v[1]));
                n[0] = n[1] = n[2] = nrm;
            }
            const Value *us = s.find("uvs");
            if (us && !us->is_null()) {
                if (!us->is_array() || us->arr.size() != 3) throw std::runtime_error("triangle needs 3 uvs");
                for (int k = 0; k < 3; ++k) {
                    if (!us->arr[k].is_array() || us->arr[k].arr.size() != 2) throw std::runtime_error("uv must be [u,v]");
                    uv[k][0] = num(us->arr[k].arr[0], "uv");
                    uv[k][1] = num(us->arr[k].arr[1], "uv");
                }
            }
            result.push_back(wrap(make_triangle(v, n, uv)));
        } else if (type == "cuboid") {  // json.rs:598-617
            for (auto &h : make_cuboid(vec3(s.at("p0"), "p0"), vec3(s.at("p1"), "p1"))) result.push_back(wrap(std::move(h)));
        } else if (type == "mesh") {  // json.rs:618-657
            std::vector<ObjMesh> models = load_obj(resolve_path(s.at("file").as_string("file")));
            if (models.empty()) throw std::runtime_error("obj file holds no model");
            const ObjMesh *mesh = &models[0];
            const Value *model = s.find("model");
            if (model && model->is_number()) {
                uint32_t i = u32(*model, "model");
                if (i >= models.size()) throw std::runtime_error("model index out of range");
                mesh = &models[i];
            } else if (model && model->is_string()) {
                mesh = nullptr;
                for (auto &m : models)
                    if (m.name == model->str) {
                        mesh = &m;
                        break;
                    }
                if (!mesh) throw std::runtime_error("cannot find the model");  // json.rs:634
            }
            // extension (not in the reference schema): "subdivide": n applies n rounds of 1:4
            // midpoint subdivision before the triangles are generated
            int levels = 0;
            if (const Value *sd = s.find("subdivide"); sd && !sd->is_null()) levels = (int)u32(*sd, "subdivide");
            if (levels > 8) throw std::runtime_error("subdivide: at most 8 levels");
            if (levels > 0) {
                ObjMesh fine = subdivide(*mesh, levels);
                for (auto &h : mesh_to_triangles(fine)) result.push_back(wrap(std::move(h)));
            } else {
                for (auto &h : mesh_to_triangles(*mesh)) result.push_back(wrap(std::move(h)));
            }
        } else {
            throw std::runtime_error("unknown shape type `" + type + "`");
        }
        if (name && name->is_string()) {
            visiting_shapes.erase(name->str);
            shapes[name->str] = result;
        }
        return result;
    }

    // visit_object, json.rs:666-700
    void object(const Value &o) {
        uint32_t mat = material(deref_material(o.at("material")));
        std::vector<Hittable> hs = shape(deref_shape(o.at("shape")));
        const Value *vis = o.find("visible"), *imp = o.find("important");
        bool visible = (vis && !vis->is_null()) ? vis->as_bool("visible") : true;
        bool important = (imp && !imp->is_null()) ? imp->as_bool("important") : false;
        if (visible && important) {
            for (auto &h : hs) {
                if (h.samplable) scene.add_important(std::move(h), mat);
                else fprintf(stderr, "importance sampling on unsupported shape!\n");  // json.rs:692 (shape is dropped)
            }
        } else if (visible) {
            for (auto &h : hs) scene.add(std::move(h), mat);
        }
    }
};

void index_named(const Value *list, std::map<std::string, const Value *> &table) {
    if (!list || list->is_null()) return;
    if (!list->is_array()) throw std::runtime_error("shapes/materials/textures must be arrays");
    for (auto &item : list->arr) {
        const Value *name = item.find("name");
        if (name && name->is_string()) table[name->str] = &item;  // later duplicates win, as HashMap::collect
    }
}
}  // namespace

LoadedScene load_scene_json(const std::string &json_text, const std::string &base_dir, uint64_t perlin_seed) {
    Value root = json::parse(json_text);
    if (!root.is_object()) throw std::runtime_error("scene file must hold a JSON object");
    LoadedScene out;
    const Value &r = root.at("renderer");
    out.renderer.width = u32(r.at("width"), "width");
    out.renderer.height = u32(r.at("height"), "height");
    if (const Value *md = r.find("max_depth"); md && !md->is_null()) {
        out.renderer.has_max_depth = true;
        out.renderer.max_depth = u32(*md, "max_depth");
    }
    if (const Value *aa = r.find("antialias"); aa && !aa->is_null()) {
        out.renderer.has_antialias = true;
        out.renderer.antialias = aa->as_bool("antialias");
    }
    const Value &c = root.at("camera");
    out.camera.look_from = vec3(c.at("look_from"), "look_from");
    out.camera.look_at = vec3(c.at("look_at"), "look_at");
    out.camera.vfov = num(c.at("vfov"), "vfov");
    auto opt = [&](const char *key, bool &has, float &dst) {
        const Value *v = c.find(key);
        if (v && !v->is_null()) {
            has = true;
            dst = num(*v, key);
        }
    };
    if (const Value *up = c.find("up"); up && !up->is_null()) {
        out.camera.has_up = true;
        out.camera.up = vec3(*up, "up");
    }
    opt("aspect_ratio", out.camera.has_aspect, out.camera.aspect_ratio);
    opt("aperture", out.camera.has_aperture, out.camera.aperture);
    opt("focus_dist", out.camera.has_focus, out.camera.focus_dist);
    opt("time0", out.camera.has_time0, out.camera.time0);
    opt("time1", out.camera.has_time1, out.camera.time1);

    Vec3 bg(0, 0, 0), env(0, 0, 0);  // json.rs:708-713
    if (const Value *v = root.find("background"); v && !v->is_null()) bg = vec3(*v, "background");
    if (const Value *v = root.find("environment"); v && !v->is_null()) env = vec3(*v, "environment");
    out.scene = std::make_unique<Scene>(bg, env);

    Visit visit{*out.scene, base_dir, perlin_seed, 0, {}, {}, {}, {}, {}, {}, {}, {}, {}};
    index_named(root.find("shapes"), visit.name_shapes);
    index_named(root.find("materials"), visit.name_materials);
    index_named(root.find("textures"), visit.name_textures);
    if (const Value *objs = root.find("objects"); objs && !objs->is_null()) {
        if (!objs->is_array()) throw std::runtime_error("objects must be an array");
        for (auto &o : objs->arr) visit.object(o);
    }
    return out;
}

LoadedScene load_scene_file(const std::string &path, uint64_t perlin_seed) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open scene file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash);
    return load_scene_json(ss.str(), dir, perlin_seed);
}

}  // namespace v4ray
