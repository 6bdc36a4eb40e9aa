// mesh.hpp — load_mesh of the reference (src/mesh.rs:9-73) without the tobj crate: a minimal
// Wavefront OBJ / MTL reader that yields the same Vec<Triangle>.
//
// What mesh.rs takes from tobj 0.1.2 (crate not vendored; semantics restated):
//   * `v x y z`, `vn x y z`; `f` with `v`, `v/vt`, `v//vn`, `v/vt/vn` corners, 1-based, negative =
//     relative to the elements read so far; polygons are fanned (c0, ci, ci+1);
//   * `mtllib file` (relative to the OBJ's directory), `usemtl name`: the faces that follow belong
//     to that material (tobj closes the current model at each usemtl, mesh.rs reads one
//     material_id per model); an unknown name means "no material";
//   * MTL: `newmtl`, `Ka`, `Kd`, `Ns`, `d` (defaults 0, 0, 0, 1.0); everything else is ignored.
// Material mapping, mesh.rs:39-66: dissolve < 1 -> Dielectric{1.0, 1.3, Ka}; else
// Diffuse{speculaty: Ns, Kd}; no material -> backup_material.
// mesh.rs indexes `normals[...]` unconditionally, so a face corner without a normal is an error
// here (the reference panics).
//
// PARITY UNPINNED where it matters: mesh.rs:10-29 iterates tobj's `models` and pushes each model's
// triangles in turn, so the ORDER of the Vec<Triangle> — which decides leaf order and therefore which
// primitive wins a traversal — is tobj's model order.  This loader emits triangles in file
// order, which equals tobj's order as long as tobj appends a model whenever an `o` / `g` / `usemtl`
// statement closes the current one (its documented behaviour, which the test suite restates independently).  The crate is
// not under /root/reference and the reference holds no OBJ fixture or expected output, so that equivalence
// — like every other tobj rule above — is restated from the crate's published behaviour, not checked
// against it.
#pragma once

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "pathtracer.hpp"

namespace pt {
namespace detail {

struct MtlEntry {
    float ambient[3] = {0.f, 0.f, 0.f}, diffuse[3] = {0.f, 0.f, 0.f};
    float shininess = 0.f, dissolve = 1.f;
};

inline std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> out;
    std::istringstream is(line);
    std::string tok;
    while (is >> tok) out.push_back(tok);
    return out;
}

inline float parse_f32(const std::string& s, const std::string& where) {
    char* end = nullptr;
    const float v = std::strtof(s.c_str(), &end);  // correctly rounded, like str::parse::<f32>
    if (end == s.c_str() || *end != '\0') throw Error(PT_ERR_INVALID, "bad number '" + s + "' in " + where);
    return v;
}

inline void load_mtl(const std::string& path, std::vector<MtlEntry>& mats, std::map<std::string, size_t>& by_name) {
    std::ifstream f(path);
    if (!f) throw Error(PT_ERR_INVALID, "cannot open material library " + path);
    std::string line;
    MtlEntry* cur = nullptr;
    while (std::getline(f, line)) {
        const std::vector<std::string> t = split_ws(line);
        if (t.empty() || t[0][0] == '#') continue;
        if (t[0] == "newmtl" && t.size() >= 2) {
            by_name[t[1]] = mats.size();
            mats.emplace_back();
            cur = &mats.back();
        } else if (cur && (t[0] == "Ka" || t[0] == "Kd") && t.size() >= 4) {
            float* dst = t[0] == "Ka" ? cur->ambient : cur->diffuse;
            for (int k = 0; k < 3; ++k) dst[k] = parse_f32(t[1 + k], path);
        } else if (cur && t[0] == "Ns" && t.size() >= 2) {
            cur->shininess = parse_f32(t[1], path);
        } else if (cur && t[0] == "d" && t.size() >= 2) {
            cur->dissolve = parse_f32(t[1], path);
        }
    }
}

// one face corner "v", "v/vt", "v//vn" or "v/vt/vn" -> 0-based (position, normal) indices; normal -1 if absent
inline void parse_corner(const std::string& tok, size_t n_pos, size_t n_nrm, long* vi, long* ni, const std::string& where) {
    long idx[3] = {0, 0, 0};
    bool have[3] = {false, false, false};
    size_t start = 0;
    for (int k = 0; k < 3 && start <= tok.size(); ++k) {
        size_t slash = tok.find('/', start);
        const std::string part = tok.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
        if (!part.empty()) {
            char* end = nullptr;
            idx[k] = std::strtol(part.c_str(), &end, 10);
            if (*end != '\0') throw Error(PT_ERR_INVALID, "bad face corner '" + tok + "' in " + where);
            have[k] = true;
        }
        if (slash == std::string::npos) break;
        start = slash + 1;
    }
    if (!have[0]) throw Error(PT_ERR_INVALID, "face corner without a position in " + where);
    *vi = idx[0] < 0 ? (long)n_pos + idx[0] : idx[0] - 1;
    *ni = !have[2] ? -1 : (idx[2] < 0 ? (long)n_nrm + idx[2] : idx[2] - 1);
    if (*vi < 0 || *vi >= (long)n_pos) throw Error(PT_ERR_INVALID, "position index out of range in " + where);
    if (have[2] && (*ni < 0 || *ni >= (long)n_nrm)) throw Error(PT_ERR_INVALID, "normal index out of range in " + where);
}

}  // namespace detail

// mesh::load_mesh, src/mesh.rs:9-73
inline std::vector<Triangle> load_mesh(const std::string& path, const Material& backup_material) {
    std::ifstream f(path);
    if (!f) throw Error(PT_ERR_INVALID, "cannot open mesh " + path);  // .expect("error with file")
    const size_t dir_end = path.find_last_of('/');
    const std::string dir = dir_end == std::string::npos ? std::string() : path.substr(0, dir_end + 1);
    std::vector<Vec3> positions, normals;
    std::vector<detail::MtlEntry> mtl;
    std::map<std::string, size_t> mtl_by_name;
    long material_id = -1;
    std::vector<Triangle> triangles;
    std::string line;
    while (std::getline(f, line)) {
        const std::vector<std::string> t = detail::split_ws(line);
        if (t.empty() || t[0][0] == '#') continue;
        if ((t[0] == "v" || t[0] == "vn") && t.size() >= 4) {
            Vec3 p;
            for (int k = 0; k < 3; ++k) p[k] = detail::parse_f32(t[1 + k], path);
            (t[0] == "v" ? positions : normals).push_back(p);
        } else if (t[0] == "mtllib" && t.size() >= 2) {
            detail::load_mtl(dir + t[1], mtl, mtl_by_name);
        } else if (t[0] == "usemtl" && t.size() >= 2) {
            auto it = mtl_by_name.find(t[1]);
            material_id = it == mtl_by_name.end() ? -1 : (long)it->second;
        } else if (t[0] == "f") {
            if (t.size() < 4) throw Error(PT_ERR_INVALID, "face with fewer than 3 corners in " + path);
            std::vector<long> vi(t.size() - 1), ni(t.size() - 1);
            for (size_t k = 1; k < t.size(); ++k) {
                detail::parse_corner(t[k], positions.size(), normals.size(), &vi[k - 1], &ni[k - 1], path);
                if (ni[k - 1] < 0) throw Error(PT_ERR_INVALID, "face corner without a normal in " + path + " (the reference indexes normals unconditionally)");
            }
            Material m = backup_material;
            if (material_id >= 0) {
                const detail::MtlEntry& e = mtl[(size_t)material_id];
                m = e.dissolve < 1.0f ? Material::dielectric(1.0f, 1.3f, {e.ambient[0], e.ambient[1], e.ambient[2]})
                                      : Material::diffuse(e.shininess, {e.diffuse[0], e.diffuse[1], e.diffuse[2]});
            }
            for (size_t k = 1; k + 1 < vi.size(); ++k)
                triangles.push_back({positions[(size_t)vi[0]], positions[(size_t)vi[k]], positions[(size_t)vi[k + 1]],
                                     normals[(size_t)ni[0]], normals[(size_t)ni[k]], normals[(size_t)ni[k + 1]], m});
        }
    }
    return triangles;
}

}  // namespace pt
