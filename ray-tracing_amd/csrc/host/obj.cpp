// Wavefront OBJ loading and the mesh -> triangle expander.
// The reference loads meshes with tobj::load_obj(triangulate = true) (src/json.rs:619-626) and
// expands them in src/hittables/obj.rs:30-104.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "scene.hpp"

namespace v4ray {

namespace {
struct Corner {
    long v = 0, vt = 0, vn = 0;  // 1-based or negative (relative); 0 = absent
};

bool parse_corner(const char *tok, Corner &c) {
    char *end = nullptr;
    c.v = std::strtol(tok, &end, 10);
    if (end == tok) return false;
    if (*end != '/') return true;
    const char *p = end + 1;
    if (*p != '/') {
        c.vt = std::strtol(p, &end, 10);
        p = end;
    }
    if (*p == '/') {
        ++p;
        c.vn = std::strtol(p, &end, 10);
    }
    return true;
}

long resolve(long idx, size_t count) {
    if (idx > 0) return idx - 1;
    if (idx < 0) return (long)count + idx;
    return -1;
}

struct ModelBuilder {
    ObjMesh mesh;
    std::unordered_map<long, uint32_t> vmap, nmap, tmap;
    bool any_face = false;
    uint32_t map_index(std::unordered_map<long, uint32_t> &m, long global, const std::vector<float> &src, int width,
                       std::vector<float> &dst) {
        auto it = m.find(global);
        if (it != m.end()) return it->second;
        if (global < 0 || (size_t)(global + 1) * width > src.size()) throw std::runtime_error("OBJ: index out of range");
        uint32_t local = (uint32_t)(dst.size() / width);
        for (int k = 0; k < width; ++k) dst.push_back(src[(size_t)global * width + k]);
        m.emplace(global, local);
        return local;
    }
};
}  // namespace

std::vector<ObjMesh> parse_obj(const std::string &text) {
    std::vector<float> V, N, T;
    std::vector<ObjMesh> models;
    ModelBuilder cur;
    cur.mesh.name = "unnamed_object";
    auto flush = [&](const std::string &next_name) {
        if (cur.any_face) models.push_back(std::move(cur.mesh));
        cur = ModelBuilder();
        cur.mesh.name = next_name;
    };
    std::istringstream in(text);
    std::string line;
    std::vector<Corner> poly;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        size_t s = line.find_first_not_of(" \t");
        if (s == std::string::npos || line[s] == '#') continue;
        std::istringstream ls(line.substr(s));
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            float x, y, z;
            if (!(ls >> x >> y >> z)) throw std::runtime_error("OBJ: bad vertex line");
            V.push_back(x);
            V.push_back(y);
            V.push_back(z);
        } else if (tag == "vn") {
            float x, y, z;
            if (!(ls >> x >> y >> z)) throw std::runtime_error("OBJ: bad normal line");
            N.push_back(x);
            N.push_back(y);
            N.push_back(z);
        } else if (tag == "vt") {
            float u = 0, v = 0;
            if (!(ls >> u)) throw std::runtime_error("OBJ: bad texcoord line");
            ls >> v;
            T.push_back(u);
            T.push_back(v);
        } else if (tag == "f") {
            poly.clear();
            std::string tok;
            while (ls >> tok) {
                Corner c;
                if (!parse_corner(tok.c_str(), c)) throw std::runtime_error("OBJ: bad face corner");
                poly.push_back(c);
            }
            if (poly.size() < 3) throw std::runtime_error("OBJ: face with fewer than 3 corners");
            cur.any_face = true;
            for (size_t k = 1; k + 1 < poly.size(); ++k) {  // fan triangulation
                const Corner *tri[3] = {&poly[0], &poly[k], &poly[k + 1]};
                for (const Corner *c : tri) {
                    long gv = resolve(c->v, V.size() / 3);
                    cur.mesh.indices.push_back(cur.map_index(cur.vmap, gv, V, 3, cur.mesh.positions));
                    if (c->vn != 0) {
                        long gn = resolve(c->vn, N.size() / 3);
                        cur.mesh.normal_indices.push_back(cur.map_index(cur.nmap, gn, N, 3, cur.mesh.normals));
                    }
                    if (c->vt != 0) {
                        long gt = resolve(c->vt, T.size() / 2);
                        cur.mesh.texcoord_indices.push_back(cur.map_index(cur.tmap, gt, T, 2, cur.mesh.texcoords));
                    }
                }
            }
        } else if (tag == "o" || tag == "g") {
            std::string name;
            std::getline(ls, name);
            size_t b = name.find_first_not_of(" \t");
            name = b == std::string::npos ? std::string("unnamed_object") : name.substr(b);
            flush(name);
        }
        // mtllib / usemtl / s and anything else: ignored (materials come from the scene file)
    }
    if (cur.any_face) models.push_back(std::move(cur.mesh));
    for (auto &m : models) {
        if (!m.normal_indices.empty() && m.normal_indices.size() != m.indices.size())
            throw std::runtime_error("OBJ: some but not all face corners have normals");
        if (!m.texcoord_indices.empty() && m.texcoord_indices.size() != m.indices.size())
            throw std::runtime_error("OBJ: some but not all face corners have texcoords");
    }
    return models;
}

std::vector<ObjMesh> load_obj(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open obj file: " + path);  // json.rs:626
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_obj(ss.str());
}

std::vector<Hittable> mesh_to_triangles(const ObjMesh &mesh) {
    size_t n_faces = mesh.indices.size() / 3;
    std::vector<Hittable> out;
    out.reserve(n_faces);
    auto point = [&](uint32_t i) { return Vec3(&mesh.positions[3 * (size_t)i]); };
    bool own_normals = mesh.normal_indices.empty();
    std::vector<Vec3> vertex_norms;
    if (own_normals) {  // obj.rs:35-39, 64-69, 85-96
        vertex_norms.assign(mesh.positions.size() / 3, Vec3(0, 0, 0));
        for (size_t f = 0; f < n_faces; ++f) {
            uint32_t i0 = mesh.indices[3 * f], i1 = mesh.indices[3 * f + 1], i2 = mesh.indices[3 * f + 2];
            Vec3 p1 = point(i0), p2 = point(i1), p3 = point(i2);
            Vec3 nrm = normalize(cross(p2 - p1, p3 - p2));
            vertex_norms[i0] = vertex_norms[i0] + nrm;
            vertex_norms[i1] = vertex_norms[i1] + nrm;
            vertex_norms[i2] = vertex_norms[i2] + nrm;
        }
        for (auto &v : vertex_norms) v = normalize(v);
    }
    for (size_t f = 0; f < n_faces; ++f) {
        Vec3 v[3], n[3];
        float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};  // obj.rs:82
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = mesh.indices[3 * f + k];
            v[k] = point(vi);
            if (own_normals) n[k] = vertex_norms[vi];
            else n[k] = Vec3(&mesh.normals[3 * (size_t)mesh.normal_indices[3 * f + k]]);
            if (!mesh.texcoord_indices.empty()) {
                uint32_t ti = mesh.texcoord_indices[3 * f + k];
                uv[k][0] = mesh.texcoords[2 * (size_t)ti];
                uv[k][1] = mesh.texcoords[2 * (size_t)ti + 1];
            }
        }
        out.push_back(make_triangle(v, n, uv));
    }
    return out;
}

ObjMesh subdivide(const ObjMesh &mesh, int levels) {
    ObjMesh cur;
    cur.name = mesh.name;
    cur.positions = mesh.positions;
    cur.indices = mesh.indices;  // normals / texcoords are dropped: vertex normals get re-averaged
    for (int l = 0; l < levels; ++l) {
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> mid;
        std::vector<uint32_t> next;
        next.reserve(cur.indices.size() * 4);
        auto midpoint = [&](uint32_t a, uint32_t b) {
            auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
            auto it = mid.find(key);
            if (it != mid.end()) return it->second;
            uint32_t idx = (uint32_t)(cur.positions.size() / 3);
            for (int k = 0; k < 3; ++k)
                cur.positions.push_back(0.5f * (cur.positions[3 * (size_t)key.first + k] + cur.positions[3 * (size_t)key.second + k]));
            mid.emplace(key, idx);
            return idx;
        };
        for (size_t f = 0; f + 2 < cur.indices.size(); f += 3) {
            uint32_t a = cur.indices[f], b = cur.indices[f + 1], c = cur.indices[f + 2];
            uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
            uint32_t tris[12] = {a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca};
            next.insert(next.end(), tris, tris + 12);
        }
        cur.indices.swap(next);
    }
    return cur;
}

}  // namespace v4ray
