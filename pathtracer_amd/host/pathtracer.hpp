// pathtracer.hpp — C++ host-side mirror of the reference's Scene / Camera / Material surface for
// the hot path, on top of the C ABI (include/ptcore.h).  Same names, argument meaning and error
// behaviour as the Rust types, except that failures throw pt::Error instead of panicking and the
// per-pixel generate()/sample() pair is one per-frame call (Camera::render).
//
//   reference                                   here
//   enum Material            src/material.rs:5   pt::Material::{diffuse, dielectric, emissive}
//   struct Triangle / Sphere primitive/*.rs      pt::Triangle / pt::Sphere
//   Scene::<Sphere>::default_scene() scene.rs:98 pt::Scene<pt::Sphere>::default_scene()
//   Scene::<Triangle>::scene(path)   scene.rs:35 pt::Scene<pt::Triangle>::scene(mesh_triangles, backup)
//   Camera::new(w, h, scene) camera.rs:89        pt::Camera<T>(w, h, std::move(scene), device)
//   Camera::handle_input     camera.rs:114       pt::Camera<T>::handle_input(keys)
//   main.rs:143-170 frame loop                   pt::Camera<T>::render(spp) + rgb24(buffer, pitch)
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptcore.h"

namespace pt {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

using Vec3 = std::array<float, 3>;

// src/material.rs:20-27
inline Vec3 light_color() { return {8.5f * 0.2f, 8.5f * 0.2f, 7.0f * 0.2f}; }

struct Material {
    pt_material m;
    static Material diffuse(float speculaty, Vec3 color) { return {{PT_MAT_DIFFUSE, speculaty, 0.f, {color[0], color[1], color[2]}}}; }
    static Material dielectric(float n1, float n2, Vec3 color) { return {{PT_MAT_DIELECTRIC, n1, n2, {color[0], color[1], color[2]}}}; }
    static Material emissive(Vec3 color) { return {{PT_MAT_EMISSIVE, 0.f, 0.f, {color[0], color[1], color[2]}}}; }
    bool operator==(const Material& o) const {
        return m.kind == o.m.kind && m.a == o.m.a && m.b == o.m.b && m.rgb[0] == o.m.rgb[0] && m.rgb[1] == o.m.rgb[1] && m.rgb[2] == o.m.rgb[2];
    }
};

struct Triangle {  // src/primitive/triangle.rs:14-22
    Vec3 position0, position1, position2, normal0, normal1, normal2;
    Material material;
    static Triangle light(Vec3 p0, Vec3 p1, Vec3 p2, Vec3 n0, Vec3 n1, Vec3 n2) {
        return {p0, p1, p2, n0, n1, n2, Material::emissive(light_color())};
    }
};

struct Sphere {  // src/primitive/sphere.rs:14-18
    Vec3 position;
    float radius;
    Material material;
    static Sphere light(Vec3 position, float radius) { return {position, radius, Material::emissive(light_color())}; }
};

// Scene<T: Primitive>, src/scene.rs:19-22 — objects + skybox; the BVH lives in the device context.
template <class T>
struct Scene {
    std::vector<T> objects;
    std::vector<float> skybox;  // 2500 x 1250 x 3

    // Scene::read_skybox, src/scene.rs:171-183 (raw native-endian f32 file)
    static std::vector<float> read_skybox(const std::string& path = "./assets/sky_15.raw") {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw Error(PT_ERR_INVALID, "cannot open skybox " + path);
        std::vector<float> v((size_t)2500 * 1250 * 3);
        size_t got = std::fread(v.data(), sizeof(float), v.size(), f);
        std::fclose(f);
        if (got != v.size()) throw Error(PT_ERR_INVALID, "skybox " + path + " is not 2500x1250x3 f32");
        return v;
    }

    // Scene::default_scene, src/scene.rs:98-160
    static Scene<Sphere> default_scene(std::vector<float> sky) {
        Scene<Sphere> s;
        s.skybox = std::move(sky);
        s.objects.push_back(Sphere::light({2.7f, 1.7f, -0.5f}, 0.3f));
        s.objects.push_back({{-0.8f, 0.0f, -2.0f}, 0.3f, Material::diffuse(0.8f, {1.0f, 0.2f, 0.2f})});
        s.objects.push_back({{0.0f, 0.0f, -2.0f}, 0.3f, Material::dielectric(1.0f, 1.3f, {0.1f, 1.0f, 0.1f})});
        s.objects.push_back({{0.8f, 0.0f, -2.0f}, 0.3f, Material::diffuse(0.8f, {0.2f, 0.2f, 1.0f})});
        s.objects.push_back({{-0.8f, -0.8f, -2.0f}, 0.5f, Material::diffuse(0.0f, {1.0f, 1.0f, 1.0f})});
        s.objects.push_back({{-0.0f, -0.8f, -2.0f}, 0.5f, Material::diffuse(0.0f, {1.0f, 1.0f, 1.0f})});
        s.objects.push_back({{0.8f, -0.8f, -2.0f}, 0.5f, Material::diffuse(0.0f, {1.0f, 1.0f, 1.0f})});
        return s;
    }

    // Scene::scene, src/scene.rs:35-96: mesh triangles, then the light, then the two floor triangles
    static Scene<Triangle> scene(std::vector<Triangle> mesh, std::vector<float> sky, float floor_speculaty = 0.5f) {
        Scene<Triangle> s;
        s.skybox = std::move(sky);
        s.objects = std::move(mesh);
        s.objects.push_back(Triangle::light({2, 2, 2}, {1, 2, 2}, {2, 2, 1}, {0, -1, 0}, {0, -1, 0}, {0, -1, 0}));
        Material floor = Material::diffuse(floor_speculaty, {0.9f, 0.9f, 0.9f});
        s.objects.push_back({{200, -0.3f, 200}, {200, -0.3f, -200}, {-200, -0.3f, 200}, {0, 1, 0}, {0, 1, 0}, {0, 1, 0}, floor});
        s.objects.push_back({{-200, -0.3f, -200}, {-200, -0.3f, 200}, {200, -0.3f, -200}, {0, 1, 0}, {0, 1, 0}, {0, 1, 0}, floor});
        return s;
    }
};

// the backup material of Scene::scene, src/scene.rs:36-40
inline Material backup_glass() { return Material::dielectric(1.0f, 1.5f, {0.01f, 0.01f, 0.9f}); }

enum class Key { A, D, W, S, E, Q, R, F, Up, Down, Left, Right, H, P };

// The arithmetic of Camera::handle_input, src/camera.rs:114-204, on (origin, target, depth): which keys
// move what, in the reference's order and with its constants (A/D/R/F: origin and target +-0.1 along
// right / up; W/S: origin only, +-0.1 along direction; E/Q: both, +-10 along direction; arrows: target
// only, +-0.1 along up / right; H: depth 512 <-> 2).  `target` is first reset to origin + direction
// (camera.rs:115) whether or not a key is down.  Returns `changed`; the caller then runs update().
// Device-free, so it can be checked on the CPU.
inline bool apply_input(const std::set<Key>& keys, const float direction[3], const float right[3], const float up[3], Vec3& origin,
                        Vec3& target, uint32_t& depth) {
    auto add = [](Vec3 a, float k, const float* v) { return Vec3{a[0] + k * v[0], a[1] + k * v[1], a[2] + k * v[2]}; };
    target = add(origin, 1.0f, direction);
    bool changed = false;
    if (keys.count(Key::A)) { origin = add(origin, -0.1f, right); target = add(target, -0.1f, right); changed = true; }
    if (keys.count(Key::D)) { origin = add(origin, 0.1f, right); target = add(target, 0.1f, right); changed = true; }
    if (keys.count(Key::W)) { origin = add(origin, 0.1f, direction); changed = true; }
    if (keys.count(Key::E)) { origin = add(origin, 10.0f, direction); target = add(target, 10.0f, direction); changed = true; }
    if (keys.count(Key::Q)) { origin = add(origin, -10.0f, direction); target = add(target, -10.0f, direction); changed = true; }
    if (keys.count(Key::H)) { depth = depth == 512 ? 2 : 512; changed = true; }  // field otherwise unused, as in the reference
    if (keys.count(Key::P)) std::printf("origin: (%g, %g, %g), direction: (%g, %g, %g)\n", origin[0], origin[1], origin[2], direction[0], direction[1], direction[2]);
    if (keys.count(Key::S)) { origin = add(origin, -0.1f, direction); changed = true; }
    if (keys.count(Key::R)) { origin = add(origin, 0.1f, up); target = add(target, 0.1f, up); changed = true; }
    if (keys.count(Key::F)) { origin = add(origin, -0.1f, up); target = add(target, -0.1f, up); changed = true; }
    if (keys.count(Key::Up)) { target = add(target, -0.1f, up); changed = true; }
    if (keys.count(Key::Down)) { target = add(target, 0.1f, up); changed = true; }
    if (keys.count(Key::Left)) { target = add(target, -0.1f, right); changed = true; }
    if (keys.count(Key::Right)) { target = add(target, 0.1f, right); changed = true; }
    return changed;
}

namespace detail {
inline void check(pt_ctx* c, int rc, const char* what) {
    if (rc != PT_OK) throw Error(rc, std::string(what) + ": " + (c ? pt_last_error(c) : pt_strerror(rc)));
}
inline uint32_t intern(std::vector<Material>& table, const Material& m) {
    for (size_t i = 0; i < table.size(); ++i)
        if (table[i] == m) return (uint32_t)i;
    table.push_back(m);
    return (uint32_t)(table.size() - 1);
}
inline void upload(pt_ctx* c, const Scene<Triangle>& s) {
    std::vector<Material> table;
    std::vector<float> pos, nrm;
    std::vector<uint32_t> ids;
    for (const Triangle& t : s.objects) {
        for (const Vec3* p : {&t.position0, &t.position1, &t.position2}) pos.insert(pos.end(), p->begin(), p->end());
        for (const Vec3* n : {&t.normal0, &t.normal1, &t.normal2}) nrm.insert(nrm.end(), n->begin(), n->end());
        ids.push_back(intern(table, t.material));
    }
    std::vector<pt_material> mats;
    for (const Material& m : table) mats.push_back(m.m);
    check(c, pt_set_materials(c, mats.data(), (uint32_t)mats.size()), "pt_set_materials");
    check(c, pt_set_triangles(c, pos.data(), nrm.data(), ids.data(), s.objects.size()), "pt_set_triangles");
}
inline void upload(pt_ctx* c, const Scene<Sphere>& s) {
    std::vector<Material> table;
    std::vector<float> cr;
    std::vector<uint32_t> ids;
    for (const Sphere& sp : s.objects) {
        cr.insert(cr.end(), sp.position.begin(), sp.position.end());
        cr.push_back(sp.radius);
        ids.push_back(intern(table, sp.material));
    }
    std::vector<pt_material> mats;
    for (const Material& m : table) mats.push_back(m.m);
    check(c, pt_set_materials(c, mats.data(), (uint32_t)mats.size()), "pt_set_materials");
    check(c, pt_set_spheres(c, cr.data(), ids.data(), s.objects.size()), "pt_set_spheres");
}
}  // namespace detail

// Camera<T>, src/camera.rs:18-37.  Owns the scene (as the reference's Camera owns Scene<T>) and the
// device context; the accumulator and spp counter of main.rs:100-101 live in the context.
template <class T>
class Camera {
public:
    Camera(uint32_t width, uint32_t height, Scene<T> scene, int device = 0) : width_(width), height_(height), scene_(std::move(scene)) {
        detail::check(nullptr, pt_create(device, &ctx_), "pt_create");
        try {
            detail::upload(ctx_, scene_);
            detail::check(ctx_, pt_set_sky(ctx_, scene_.skybox.data(), 2500, 1250), "pt_set_sky");
            detail::check(ctx_, pt_build(ctx_), "pt_build");  // BVH::new, src/bvh.rs:25
            update();
        } catch (...) {
            pt_destroy(ctx_);
            throw;
        }
    }
    ~Camera() { pt_destroy(ctx_); }
    Camera(const Camera&) = delete;
    Camera& operator=(const Camera&) = delete;

    void look_at(Vec3 origin, Vec3 target) {
        origin_ = origin;
        target_ = target;
        update();
    }

    // Camera::handle_input, src/camera.rs:114-212.  Returns true when the camera moved (the
    // accumulation is then reset, as main.rs:137-142 does).
    bool handle_input(const std::set<Key>& keys) {
        const bool changed = apply_input(keys, state_.direction, state_.right, state_.up, origin_, target_, depth_);
        if (changed) update();
        return changed;
    }
    Vec3 origin() const { return origin_; }
    Vec3 target() const { return target_; }
    uint32_t depth() const { return depth_; }

    // the frame loop body, src/main.rs:143-170, for `spp` frames of one sample each
    void render(uint32_t spp = 1, uint32_t max_depth = 32, uint64_t seed = 0) {
        detail::check(ctx_, pt_render(ctx_, spp, max_depth, seed, nullptr, 0), "pt_render");
    }
    // the same frame queued without waiting (pt_render_async): consecutive frames overlap on the device,
    // the sums stay in frame order; rgb24() / accum() / wait() wait for everything queued
    void render_async(uint32_t spp = 1, uint32_t max_depth = 32, uint64_t seed = 0) {
        detail::check(ctx_, pt_render_async(ctx_, spp, max_depth, seed, nullptr, 0), "pt_render_async");
    }
    void wait() { detail::check(ctx_, pt_wait(ctx_), "pt_wait"); }
    // vec_to_rgb(scale * accum) into an RGB24 buffer with SDL-style pitch, src/main.rs:160-164
    void rgb24(uint8_t* buffer, size_t pitch) { detail::check(ctx_, pt_read_rgb8(ctx_, buffer, pitch), "pt_read_rgb8"); }
    // the frame as far as it has got, without waiting for queued frames; returns the samples it holds (pt_read_rgb8_latest)
    float rgb24_latest(uint8_t* buffer, size_t pitch) {
        float spp = 0.f;
        detail::check(ctx_, pt_read_rgb8_latest(ctx_, buffer, pitch, &spp), "pt_read_rgb8_latest");
        return spp;
    }
    float accum(std::vector<float>& rgb) {
        rgb.resize((size_t)width_ * height_ * 3);
        float spp = 0;
        detail::check(ctx_, pt_read_accum(ctx_, rgb.data(), &spp), "pt_read_accum");
        return spp;
    }
    // fp32 sum + sample count into caller memory (W*H*3 floats) / back (resume)
    float read_accum(float* rgb) {
        float spp = 0;
        detail::check(ctx_, pt_read_accum(ctx_, rgb, &spp), "pt_read_accum");
        return spp;
    }
    void write_accum(const float* rgb, float spp) { detail::check(ctx_, pt_write_accum(ctx_, rgb, spp), "pt_write_accum"); }
    const pt_camera_state& state() const { return state_; }
    const Scene<T>& scene() const { return scene_; }
    pt_ctx* context() { return ctx_; }
    pt_counters counters() {
        pt_counters c;
        detail::check(ctx_, pt_get_counters(ctx_, &c), "pt_get_counters");
        return c;
    }

private:
    void update() {  // Camera::update, src/camera.rs:271-292 (runs on the device, incl. autofocus)
        detail::check(ctx_, pt_set_camera(ctx_, origin_.data(), target_.data(), width_, height_, lens_size_), "pt_set_camera");
        detail::check(ctx_, pt_get_camera(ctx_, &state_), "pt_get_camera");
    }
    uint32_t width_, height_;
    uint32_t depth_ = 512;                   // src/camera.rs:93
    float lens_size_ = 0.04f;                // src/camera.rs:94
    Vec3 origin_{-1.6f, 0.0f, -1.3f};        // src/camera.rs:95
    Vec3 target_{0.7f, 0.0f, 0.6f};          // src/camera.rs:99
    Scene<T> scene_;
    pt_ctx* ctx_ = nullptr;
    pt_camera_state state_{};
};

}  // namespace pt
