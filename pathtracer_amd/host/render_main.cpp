// render_main.cpp — src/main.rs without SDL: builds a scene, runs the progressive frame loop on the
// GPU through the C ABI and writes the tonemapped frame as a PPM.
//   render [--spheres | --cube | --obj mesh.obj] [--size WxH] [--frames N] [--spp-per-frame K] [--sky file.raw]
//          [--out frame.ppm] [--save-accum file] [--resume file] [--dump-mesh file]
// --obj        Scene::scene(path) of src/scene.rs:35: load_mesh + light + floor (needs vertex normals)
// --save-accum / --resume   fp32 accumulation dump and restart (W, H, spp header + W*H*3 floats)
// --dump-mesh  write what load_mesh returned (24 floats per triangle) and exit; needs no GPU
// --keys "W,A;D;Up,Left"   Camera::handle_input once per ';'-separated key set before rendering; prints the pose
//              after each step as hex float bits (changed, origin, target, direction, right, up, p1, p2, p3, focal distance)
// --input-check  device-free check of the key arithmetic: reads lines "<15 hex words: origin target direction right up>
//              <depth> <keys>" from stdin, prints "<changed> <6 hex words: origin target> <depth>" per line
// --async      queue the frames with render_async, show the latest complete state every iteration (rgb24_latest) and wait once
//              at the end (frames overlap on the device; same final image)
// --gpus N [--shard samples|tiles]   the frame over N GPUs of this node (multi_gpu.hpp): one host thread and one context
//              per device, replicated scene, ONE RCCL collective at readout — samples: every rank renders its own sample
//              indices, ncclReduce(sum) of the accumulators (frames x spp samples PER RANK); tiles: 64x64 tiles dealt
//              round-robin, ncclGather of the ranks' slabs.  This is the GPU form of the reference's thread fan-out
//              (main.rs:113-115,149-170).  Refuses N above the visible devices (exit 3).  Prints rays and time per rank.
// --devices a,b,..  the device of each rank instead of 0..N-1;  --combine host  moves the partial frames through host
//              memory instead of RCCL (the only way to put two ranks on ONE device: a rehearsal of the N > 1 logic on a
//              one-GPU box, not a measurement)
#include <iostream>
#include <sstream>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>

#include "mesh.hpp"
#include "multi_gpu.hpp"
#include "pathtracer.hpp"

using namespace pt;

// stand-in for assets/sky_15.raw (absent from the reference checkout): vertical gradient + horizon band
static std::vector<float> synthetic_sky() {
    std::vector<float> sky((size_t)2500 * 1250 * 3);
    for (int v = 0; v < 1250; ++v) {
        float t = (v + 0.5f) / 1250.0f;
        float up = t < 0.5f ? t / 0.5f : 1.0f, dn = t < 0.5f ? 0.0f : (t - 0.5f) / 0.5f;
        float band = 0.6f * std::exp(-((t - 0.5f) / 0.03f) * ((t - 0.5f) / 0.03f));
        float r = (t < 0.5f ? 0.25f + 0.85f * up * up : 1.10f - 0.92f * std::sqrt(dn)) + band;
        float g = (t < 0.5f ? 0.45f + 0.55f * up * up : 1.00f - 0.84f * std::sqrt(dn)) + 0.9f * band;
        float b = (t < 0.5f ? 0.95f - 0.10f * up * up : 0.85f - 0.71f * std::sqrt(dn)) + 0.7f * band;
        for (int u = 0; u < 2500; ++u) {
            float* p = &sky[((size_t)v * 2500 + u) * 3];
            p[0] = r; p[1] = g; p[2] = b;
        }
    }
    return sky;
}

// BASELINE config C2's cube (side 0.6 on the floor, centred under the default camera target (0.7, ., 0.6)).  The corner
// coordinates are written out: computed as 0.7f - 0.3f they come out one ulp off the float32 values every other host
// of this scene holds (scenes.glass_cube_scene rounds 0.7 - 0.3 from double), and a corner one ulp off is another scene.
static std::vector<Triangle> cube_mesh(Material m) {
    const float x0 = 0.4f, x1 = 1.0f, z0 = 0.3f, z1 = 0.9f, y0 = -0.3f, y1 = 0.3f;
    Vec3 v[8] = {{x0, y0, z0}, {x1, y0, z0}, {x1, y1, z0}, {x0, y1, z0},
                 {x0, y0, z1}, {x1, y0, z1}, {x1, y1, z1}, {x0, y1, z1}};
    int quads[6][4] = {{0, 3, 2, 1}, {4, 5, 6, 7}, {0, 4, 7, 3}, {1, 2, 6, 5}, {3, 7, 6, 2}, {0, 1, 5, 4}};
    Vec3 normals[6] = {{0, 0, -1}, {0, 0, 1}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
    std::vector<Triangle> out;
    for (int q = 0; q < 6; ++q) {
        const int* k = quads[q];
        out.push_back({v[k[0]], v[k[1]], v[k[2]], normals[q], normals[q], normals[q], m});
        out.push_back({v[k[0]], v[k[2]], v[k[3]], normals[q], normals[q], normals[q], m});
    }
    return out;
}

struct AccumFile {
    std::string save, resume;
    bool async = false;
};

static bool parse_keys(const std::string& list, std::set<Key>& out) {
    static const std::pair<const char*, Key> names[] = {{"A", Key::A}, {"D", Key::D}, {"W", Key::W}, {"S", Key::S}, {"E", Key::E},
                                                        {"Q", Key::Q}, {"R", Key::R}, {"F", Key::F}, {"Up", Key::Up}, {"Down", Key::Down},
                                                        {"Left", Key::Left}, {"Right", Key::Right}, {"H", Key::H}, {"P", Key::P}};
    std::stringstream ss(list);
    std::string k;
    while (std::getline(ss, k, ',')) {
        if (k.empty() || k == "-") continue;
        bool found = false;
        for (const auto& n : names)
            if (k == n.first) { out.insert(n.second); found = true; }
        if (!found) return false;
    }
    return true;
}
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static int input_check() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::stringstream ss(line);
        float v[15];
        for (float& x : v) { uint32_t u; ss >> std::hex >> u; x = from_bits(u); }
        uint32_t depth;
        std::string keys;
        ss >> std::dec >> depth >> keys;
        std::set<Key> ks;
        if (!ss || !parse_keys(keys, ks)) { std::fprintf(stderr, "bad line: %s\n", line.c_str()); return 2; }
        Vec3 origin{v[0], v[1], v[2]}, target{v[3], v[4], v[5]};
        const bool changed = apply_input(ks, v + 6, v + 9, v + 12, origin, target, depth);
        std::printf("%d %08x %08x %08x %08x %08x %08x %u\n", changed ? 1 : 0, bits(origin[0]), bits(origin[1]), bits(origin[2]), bits(target[0]),
                    bits(target[1]), bits(target[2]), depth);
    }
    return 0;
}

template <class T>
static bool play_keys(Camera<T>& camera, const std::string& steps, bool print = true) {
    std::stringstream ss(steps);
    std::string step;
    while (std::getline(ss, step, ';')) {
        std::set<Key> ks;
        if (!parse_keys(step, ks)) return false;
        const bool changed = camera.handle_input(ks);
        if (!print) continue;
        const pt_camera_state& s = camera.state();
        const Vec3 o = camera.origin(), t = camera.target();
        std::printf("pose %d", changed ? 1 : 0);
        for (float x : {o[0], o[1], o[2], t[0], t[1], t[2]}) std::printf(" %08x", bits(x));
        for (const float* p : {s.direction, s.right, s.up, s.p1, s.p2, s.p3})
            for (int a = 0; a < 3; ++a) std::printf(" %08x", bits(p[a]));
        std::printf(" %08x %u\n", bits(s.focal_distance), camera.depth());
    }
    return true;
}

// the tonemapped frame as a PPM and, on request, the fp32 sums (W, H, spp header + W*H*3 floats)
template <class T>
static int write_outputs(Camera<T>& camera, uint32_t W, uint32_t H, const std::string& out, const AccumFile& af, std::vector<uint8_t>& fb,
                         std::vector<float>& accum) {
    if (!af.save.empty()) {
        float done = camera.read_accum(accum.data());
        FILE* f = std::fopen(af.save.c_str(), "wb");
        if (!f) return 1;
        uint32_t hdr[2] = {W, H};
        std::fwrite(hdr, 4, 2, f);
        std::fwrite(&done, 4, 1, f);
        std::fwrite(accum.data(), 4, accum.size(), f);
        std::fclose(f);
    }
    FILE* f = std::fopen(out.c_str(), "wb");
    if (!f) return 1;
    std::fprintf(f, "P6\n%u %u\n255\n", W, H);
    std::fwrite(fb.data(), 1, fb.size(), f);
    std::fclose(f);
    return 0;
}

template <class T>
static int run(Camera<T>& camera, uint32_t W, uint32_t H, int frames, int spp, const std::string& out, const AccumFile& af) {
    std::vector<uint8_t> fb((size_t)W * H * 3);
    std::vector<float> accum((size_t)W * H * 3);
    if (!af.resume.empty()) {
        FILE* f = std::fopen(af.resume.c_str(), "rb");
        uint32_t hdr[2];
        float done = 0.f;
        if (!f || std::fread(hdr, 4, 2, f) != 2 || std::fread(&done, 4, 1, f) != 1 || hdr[0] != W || hdr[1] != H ||
            std::fread(accum.data(), 4, accum.size(), f) != accum.size()) {
            if (f) std::fclose(f);
            std::fprintf(stderr, "cannot resume from %s (missing, truncated or another size)\n", af.resume.c_str());
            return 1;
        }
        std::fclose(f);
        camera.write_accum(accum.data(), done);
        std::printf("resumed at %g spp\n", done);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) {
        if (af.async) {
            camera.render_async(spp, 32, 2026);              // queued: frames overlap on the device ...
            camera.rgb24_latest(fb.data(), (size_t)W * 3);   // ... and the window shows the frame as far as it has got
        } else {
            camera.render(spp, 32, 2026);   // spp += 1 per frame in the reference, main.rs:143
            camera.rgb24(fb.data(), (size_t)W * 3);
        }
    }
    if (af.async) camera.rgb24(fb.data(), (size_t)W * 3);  // waits for every queued frame
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pt_counters c = camera.counters();
    std::printf("%d frames x %d spp: %.1f ms/frame, %.1f Mrays/s, %.1f Mpaths/s, focal distance %g\n", frames, spp, dt * 1e3 / frames,
                c.rays_closest / dt / 1e6, c.paths / dt / 1e6, camera.state().focal_distance);
    return write_outputs(camera, W, H, out, af, fb, accum);
}

// The frame over several GPUs (multi_gpu.hpp).  Every rank builds the same scene on its own device; rank 0 ends up
// with the combined frame and writes it exactly as the one-GPU run does.
template <class T>
static int run_multi(const MultiGpuPlan& plan, const std::function<std::unique_ptr<Camera<T>>(int device)>& make, const std::string& keys,
                     uint32_t W, uint32_t H, int frames, int spp, const std::string& out, const AccumFile& af) {
    if (!af.resume.empty() || af.async) {
        std::fprintf(stderr, "--resume / --async are single-GPU options\n");
        return 2;
    }
    bool keys_ok = true;
    auto make_camera = [&](int rank, int device) {
        std::unique_ptr<Camera<T>> camera = make(device);
        if (rank == 0) { if (!play_keys(*camera, keys)) keys_ok = false; }   // every rank makes the same moves, rank 0 prints them
        else if (!play_keys(*camera, keys, false)) keys_ok = false;
        return camera;
    };
    auto on_root = [&](Camera<T>& camera, const std::vector<RankReport>& reports, double wall_ms) {
        if (!keys_ok) return 2;
        uint64_t rays = 0, paths = 0;
        for (size_t r = 0; r < reports.size(); ++r) {
            const RankReport& rr = reports[r];
            std::printf("rank %zu device %d: %llu paths, %llu rays, render %.2f ms, combine %.2f ms\n", r, rr.device, (unsigned long long)rr.paths,
                        (unsigned long long)rr.rays, rr.render_ms, rr.combine_ms);
            rays += rr.rays;
            paths += rr.paths;
        }
        std::vector<uint8_t> fb((size_t)W * H * 3);
        std::vector<float> accum((size_t)W * H * 3);
        camera.rgb24(fb.data(), (size_t)W * 3);
        const float total_spp = camera.read_accum(accum.data());
        std::printf("%zu ranks, shard %s, combine %s: %g spp in the frame, %.1f ms wall (render + combine), %.1f Mrays/s, %.1f Mpaths/s, focal distance %g\n",
                    reports.size(), plan.mode == ShardMode::Samples ? "samples" : "tiles", plan.transport == Transport::Rccl ? "rccl" : "host (rehearsal)",
                    total_spp, wall_ms, rays / wall_ms / 1e3, paths / wall_ms / 1e3, camera.state().focal_distance);
        return write_outputs(camera, W, H, out, af, fb, accum);
    };
    return run_ranks<T>(plan, frames, (uint32_t)spp, 32, 2026, make_camera, on_root);
}

// the devices of the ranks: --devices as given, else 0..N-1.  Refused (exit 3): more ranks than visible devices, a device
// index that does not exist, and — over RCCL — the same device twice (a communicator holds a device once).
static int plan_devices(int gpus, const std::string& list, Transport transport, std::vector<int>& devices) {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) visible = 0;
    if (!list.empty()) {
        std::stringstream ss(list);
        std::string d;
        while (std::getline(ss, d, ',')) devices.push_back(std::atoi(d.c_str()));
        if ((int)devices.size() != gpus) {
            std::fprintf(stderr, "--devices names %zu devices for --gpus %d\n", devices.size(), gpus);
            return 2;
        }
    } else {
        if (gpus > visible) {
            std::fprintf(stderr, "--gpus %d, but only %d device%s visible\n", gpus, visible, visible == 1 ? " is" : "s are");
            return 3;
        }
        for (int r = 0; r < gpus; ++r) devices.push_back(r);
    }
    for (size_t i = 0; i < devices.size(); ++i) {
        if (devices[i] < 0 || devices[i] >= visible) {
            std::fprintf(stderr, "device %d does not exist (%d visible)\n", devices[i], visible);
            return 3;
        }
        for (size_t j = 0; j < i; ++j)
            if (devices[j] == devices[i] && transport == Transport::Rccl) {
                std::fprintf(stderr, "device %d is named twice: RCCL cannot put two ranks on one device (use --combine host to rehearse)\n", devices[i]);
                return 3;
            }
    }
    return 0;
}

int main(int argc, char** argv) {
    uint32_t W = 800, H = 600;  // main.rs:77-78
    int frames = 16, spp = 1, gpus = 0;
    bool spheres = false;
    std::string out = "frame.ppm", sky_path, obj_path, dump_path, keys, device_list;
    AccumFile af;
    MultiGpuPlan plan;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--spheres") spheres = true;
        else if (a == "--cube") spheres = false;
        else if (a == "--size" && i + 1 < argc) std::sscanf(argv[++i], "%ux%u", &W, &H);
        else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (a == "--spp-per-frame" && i + 1 < argc) spp = std::atoi(argv[++i]);
        else if (a == "--sky" && i + 1 < argc) sky_path = argv[++i];
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--obj" && i + 1 < argc) obj_path = argv[++i];
        else if (a == "--dump-mesh" && i + 1 < argc) dump_path = argv[++i];
        else if (a == "--save-accum" && i + 1 < argc) af.save = argv[++i];
        else if (a == "--resume" && i + 1 < argc) af.resume = argv[++i];
        else if (a == "--keys" && i + 1 < argc) keys = argv[++i];
        else if (a == "--async") af.async = true;
        else if (a == "--input-check") return input_check();
        else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--devices" && i + 1 < argc) device_list = argv[++i];
        else if (a == "--shard" && i + 1 < argc) {
            const std::string m = argv[++i];
            if (m == "samples") plan.mode = ShardMode::Samples;
            else if (m == "tiles") plan.mode = ShardMode::Tiles;
            else { std::fprintf(stderr, "--shard samples|tiles\n"); return 2; }
        } else if (a == "--combine" && i + 1 < argc) {
            const std::string m = argv[++i];
            if (m == "rccl") plan.transport = Transport::Rccl;
            else if (m == "host") plan.transport = Transport::HostStaged;
            else { std::fprintf(stderr, "--combine rccl|host\n"); return 2; }
        }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (frames < 1 || spp < 1 || gpus < 0) { std::fprintf(stderr, "--frames, --spp-per-frame and --gpus must be positive\n"); return 2; }
    try {
        std::vector<Triangle> mesh;
        if (!obj_path.empty()) mesh = load_mesh(obj_path, backup_glass());
        if (!dump_path.empty()) {
            FILE* f = std::fopen(dump_path.c_str(), "wb");
            if (!f) return 1;
            for (const Triangle& t : mesh) {
                const Vec3* v[6] = {&t.position0, &t.position1, &t.position2, &t.normal0, &t.normal1, &t.normal2};
                for (const Vec3* p : v) std::fwrite(p->data(), 4, 3, f);
                const float m[6] = {(float)t.material.m.kind, t.material.m.a, t.material.m.b, t.material.m.rgb[0], t.material.m.rgb[1], t.material.m.rgb[2]};
                std::fwrite(m, 4, 6, f);
            }
            std::fclose(f);
            std::printf("%zu triangles\n", mesh.size());
            return 0;
        }
        if (gpus > 0)  // before anything is loaded or built: a run that cannot start says so at once
            if (int rc = plan_devices(gpus, device_list, plan.transport, plan.devices)) return rc;
        const std::vector<float> sky = sky_path.empty() ? synthetic_sky() : Scene<Sphere>::read_skybox(sky_path);
        // one maker per scene kind; a multi-GPU run calls it once per rank (replicated scene), the others once on device 0
        const std::function<std::unique_ptr<Camera<Sphere>>(int)> make_spheres = [&](int device) {
            auto camera = std::make_unique<Camera<Sphere>>(W, H, Scene<Sphere>::default_scene(sky), device);
            camera->look_at({0.0f, 0.2f, 1.0f}, {0.0f, -0.2f, -2.0f});  // the default pose looks away from the spheres
            return camera;
        };
        const std::function<std::unique_ptr<Camera<Triangle>>(int)> make_triangles = [&](int device) {
            if (!obj_path.empty()) return std::make_unique<Camera<Triangle>>(W, H, Scene<Triangle>::scene(mesh, sky), device);  // floor speculaty 0.5, scene.rs:63
            return std::make_unique<Camera<Triangle>>(W, H, Scene<Triangle>::scene(cube_mesh(Material::dielectric(1.0f, 1.5f, {0.01f, 0.9f, 0.01f})), sky, 0.0f), device);
        };
        if (gpus > 0) {
            if (spheres) return run_multi<Sphere>(plan, make_spheres, keys, W, H, frames, spp, out, af);
            return run_multi<Triangle>(plan, make_triangles, keys, W, H, frames, spp, out, af);
        }
        if (spheres) {
            auto camera = make_spheres(0);
            if (!play_keys(*camera, keys)) return 2;
            return run(*camera, W, H, frames, spp, out, af);
        }
        auto camera = make_triangles(0);
        if (!play_keys(*camera, keys)) return 2;
        return run(*camera, W, H, frames, spp, out, af);
    } catch (const Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.status, e.what());
        return 1;
    }
}
