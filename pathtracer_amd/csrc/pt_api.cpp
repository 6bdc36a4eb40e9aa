// pt_api.cpp — the C ABI of libptcore.so (include/ptcore.h): context, scene upload, render loop.
// Host code only; every float operation that decides a result lives in pt_bvh.cpp (build) or in
// the kernels.  There is no CPU rendering path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_bvh.h"
#include "pt_kernels.h"

using namespace pt;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

constexpr uint32_t kMaxMaterials = 1u << 15;   // material id must fit the 15 bits saved across a look-ahead
constexpr uint32_t kMaxDepth = 0xffffu;        // iteration counter has 16 bits
constexpr uint64_t kTargetSlots = 8u << 20;    // paths kept in flight per batch (about 8.4 M)
constexpr uint64_t kMaxSlots = 1ull << 31;

}  // namespace

struct pt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    int num_cu = 0;

    // host copy of the scene
    int prim_kind = -1;
    std::vector<float> pos, nrm, cr;
    std::vector<uint32_t> mat_id;
    std::vector<pt_material> mats;
    std::vector<float> sky;
    uint32_t sky_w = 0, sky_h = 0;
    HostBVH bvh;
    bool built = false;
    bool nee = false;  // some diffuse material has speculaty > 1 and the scene has a light

    // device scene
    DevScene ds{};
    DevBuf d_pairs, d_prim, d_meta, d_nrm, d_big, d_mats, d_sky, d_lights;

    // camera / frame
    DevBuf d_cam;
    pt_camera_state cam{};
    bool have_cam = false;
    DevBuf d_accum;
    float spp = 0.f;
    uint32_t samples_done = 0;

    // pipeline
    LaunchCfg cfg{};
    DevBuf d_spill;
    PathState ps{};
    DevBuf d_ro, d_rd, d_hit, d_tp, d_out, d_acc, d_pxs, d_q0, d_q1, d_qn;
    uint64_t cap_slots = 0;
    uint32_t qn_len = 0;
    DevBuf d_counters;
    double extend_ms = 0, render_ms = 0;
    uint64_t extend_launches = 0;
    uint32_t* h_qn = nullptr;  // pinned read-back word

    int fail(int code, const std::string& msg) {
        err = msg;
        return code;
    }
};

#define PT_HIP(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return (ctx)->fail(PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));             \
    } while (0)

static int dev_alloc(pt_ctx* c, DevBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return PT_OK;
    b.release();
    if (bytes == 0) bytes = 16;
    PT_HIP(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return PT_OK;
}
static int dev_upload(pt_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = dev_alloc(c, b, bytes);
    if (rc) return rc;
    if (bytes) PT_HIP(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return PT_OK;
}

extern "C" {

const char* pt_strerror(int status) {
    switch (status) {
        case PT_OK: return "ok";
        case PT_ERR_INVALID: return "invalid argument";
        case PT_ERR_STATE: return "call order violated";
        case PT_ERR_NO_DEVICE: return "no usable HIP device";
        case PT_ERR_HIP: return "HIP runtime error";
        case PT_ERR_LIMIT: return "scene exceeds a documented limit";
        case PT_ERR_BVH: return "BVH build produced an empty child (reference would mis-traverse)";
        default: return "unknown status";
    }
}

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pt_create(int device, pt_ctx** out) {
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (device < 0 || hipGetDeviceCount(&n) != hipSuccess || device >= n) return PT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PT_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PT_ERR_NO_DEVICE;  // the code object is gfx950 only
    pt_ctx* c = new pt_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return PT_ERR_HIP;
    }
    if (hipHostMalloc((void**)&c->h_qn, 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return PT_ERR_HIP;
    }
    if (dev_alloc(c, c->d_counters, CNT_N * sizeof(unsigned long long)) != PT_OK ||
        hipMemset(c->d_counters.p, 0, CNT_N * sizeof(unsigned long long)) != hipSuccess) {
        pt_destroy(c);
        return PT_ERR_HIP;
    }
    // persistent traversal grid: the LDS stacks (kLdsStack * 8 B per lane) allow 6 workgroups of 256 per CU
    const uint32_t lds_per_block = kLdsStack * kTraceBlock * 8;
    uint32_t per_cu = std::min<uint32_t>(8, (160u * 1024u) / lds_per_block);
    c->cfg.grid = (uint32_t)c->num_cu * per_cu;
    c->cfg.spill = nullptr;
    c->cfg.spill_stride = c->cfg.grid * kTraceBlock;
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->d_pairs, &c->d_prim, &c->d_meta, &c->d_nrm, &c->d_big, &c->d_mats, &c->d_sky, &c->d_lights, &c->d_cam,
                      &c->d_accum, &c->d_spill, &c->d_ro, &c->d_rd, &c->d_hit, &c->d_tp, &c->d_out, &c->d_acc, &c->d_pxs, &c->d_q0,
                      &c->d_q1, &c->d_qn, &c->d_counters};
    for (DevBuf* b : bufs) b->release();
    if (c->h_qn) (void)hipHostFree(c->h_qn);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pt_set_materials(pt_ctx* c, const pt_material* m, uint32_t count) {
    if (!c) return PT_ERR_INVALID;
    if (!m || count == 0) return c->fail(PT_ERR_INVALID, "pt_set_materials: empty material table");
    if (count > kMaxMaterials) return c->fail(PT_ERR_LIMIT, "pt_set_materials: more than 32768 materials");
    for (uint32_t i = 0; i < count; ++i)
        if (m[i].kind > PT_MAT_EMISSIVE) return c->fail(PT_ERR_INVALID, "pt_set_materials: unknown material kind");
    c->mats.assign(m, m + count);
    c->built = false;
    return PT_OK;
}

int pt_set_triangles(pt_ctx* c, const float* pos, const float* nrm, const uint32_t* mat_id, uint64_t n) {
    if (!c) return PT_ERR_INVALID;
    if (!pos || !nrm || !mat_id || n == 0) return c->fail(PT_ERR_INVALID, "pt_set_triangles: null input or zero triangles");
    if (n > kMaxPrims) return c->fail(PT_ERR_LIMIT, "pt_set_triangles: more than 2^26 primitives");
    c->prim_kind = PRIM_TRIANGLE;
    c->pos.assign(pos, pos + 9 * n);
    c->nrm.assign(nrm, nrm + 9 * n);
    c->mat_id.assign(mat_id, mat_id + n);
    c->cr.clear();
    c->built = false;
    return PT_OK;
}

int pt_set_spheres(pt_ctx* c, const float* cr, const uint32_t* mat_id, uint64_t n) {
    if (!c) return PT_ERR_INVALID;
    if (!cr || !mat_id || n == 0) return c->fail(PT_ERR_INVALID, "pt_set_spheres: null input or zero spheres");
    if (n > kMaxPrims) return c->fail(PT_ERR_LIMIT, "pt_set_spheres: more than 2^26 primitives");
    c->prim_kind = PRIM_SPHERE;
    c->cr.assign(cr, cr + 4 * n);
    c->mat_id.assign(mat_id, mat_id + n);
    c->pos.clear();
    c->nrm.clear();
    c->built = false;
    return PT_OK;
}

int pt_set_sky(pt_ctx* c, const float* rgb, uint32_t w, uint32_t h) {
    if (!c) return PT_ERR_INVALID;
    if (!rgb || w == 0 || h == 0) return c->fail(PT_ERR_INVALID, "pt_set_sky: null or empty image");
    PT_HIP(c, hipSetDevice(c->device));
    // pad with zeros up to the largest index the unclamped lookup can produce (scene.rs:163-166,174)
    const size_t n = (size_t)w * h * 3;
    const size_t need = ((size_t)2500 + (size_t)1250 * 2500 + 1) * 3;
    c->sky.assign(std::max(n, need), 0.0f);
    std::memcpy(c->sky.data(), rgb, n * sizeof(float));
    c->sky_w = w;
    c->sky_h = h;
    int rc = dev_upload(c, c->d_sky, c->sky.data(), c->sky.size() * sizeof(float));
    if (rc) return rc;
    c->ds.sky = (const float*)c->d_sky.p;
    return PT_OK;
}

static int finalize_scene(pt_ctx* c);

int pt_build(pt_ctx* c) {
    if (!c) return PT_ERR_INVALID;
    if (c->prim_kind < 0) return c->fail(PT_ERR_STATE, "pt_build: no primitives set");
    if (c->mats.empty()) return c->fail(PT_ERR_STATE, "pt_build: no materials set");
    const uint64_t n = c->mat_id.size();
    std::vector<PrimInfo> info;
    if (c->prim_kind == PRIM_TRIANGLE) prim_info_triangles(c->pos.data(), n, info);
    else prim_info_spheres(c->cr.data(), n, info);
    build_bvh(info, c->bvh);
    if (c->bvh.hazard) return c->fail(PT_ERR_BVH, "pt_build: a partition produced an empty child");
    return finalize_scene(c);
}

int pt_set_bvh(pt_ctx* c, const pt_bvh_node* nodes, uint64_t n_nodes, const uint64_t* indices, uint64_t n_indices) {
    if (!c) return PT_ERR_INVALID;
    if (c->prim_kind < 0) return c->fail(PT_ERR_STATE, "pt_set_bvh: no primitives set");
    if (c->mats.empty()) return c->fail(PT_ERR_STATE, "pt_set_bvh: no materials set");
    const uint64_t n = c->mat_id.size();
    if (!nodes || !indices || n_nodes == 0 || n_indices != n) return c->fail(PT_ERR_INVALID, "pt_set_bvh: bad arguments");
    // validate: every node reachable from the root exactly once, children in range, leaves cover
    // [0, n) without overlap, and no node deeper than the node count allows
    std::vector<uint8_t> seen(n_nodes, 0);
    std::vector<uint8_t> covered(n, 0);
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};
    uint32_t max_depth = 1;
    while (!todo.empty()) {
        auto [ni, d] = todo.back();
        todo.pop_back();
        if (ni >= n_nodes || seen[ni]) return c->fail(PT_ERR_BVH, "pt_set_bvh: node index out of range or shared");
        seen[ni] = 1;
        max_depth = std::max(max_depth, d);
        const pt_bvh_node& nd = nodes[ni];
        if (nd.count != 0) {
            if ((uint64_t)nd.left_first + nd.count > n) return c->fail(PT_ERR_BVH, "pt_set_bvh: leaf range out of bounds");
            for (uint32_t i = nd.left_first; i < nd.left_first + nd.count; ++i) {
                if (covered[i]) return c->fail(PT_ERR_BVH, "pt_set_bvh: leaves overlap");
                covered[i] = 1;
            }
        } else {
            todo.push_back({nd.left_first + 1, d + 1});
            todo.push_back({nd.left_first, d + 1});
        }
    }
    for (uint64_t i = 0; i < n; ++i)
        if (indices[i] >= n) return c->fail(PT_ERR_INVALID, "pt_set_bvh: primitive index out of range");
    c->bvh.nodes.assign(nodes, nodes + n_nodes);
    c->bvh.indices.resize(n);
    for (uint64_t i = 0; i < n; ++i) c->bvh.indices[i] = (uint32_t)indices[i];
    c->bvh.max_depth = max_depth;
    c->bvh.hazard = false;
    return finalize_scene(c);
}

int pt_bvh_build(int prim_kind, const float* geom, uint64_t n, pt_bvh_node* nodes, uint64_t nodes_cap, uint64_t* indices,
                 uint64_t* n_nodes_out, uint32_t* max_depth_out) {
    if (!geom || n == 0 || n > kMaxPrims || (prim_kind != 0 && prim_kind != 1)) return PT_ERR_INVALID;
    std::vector<PrimInfo> info;
    if (prim_kind == PRIM_TRIANGLE) prim_info_triangles(geom, n, info);
    else prim_info_spheres(geom, n, info);
    HostBVH bvh;
    build_bvh(info, bvh);
    if (n_nodes_out) *n_nodes_out = bvh.nodes.size();
    if (max_depth_out) *max_depth_out = bvh.max_depth;
    if (bvh.hazard) return PT_ERR_BVH;
    if (nodes) {
        if (nodes_cap < bvh.nodes.size()) return PT_ERR_INVALID;
        std::memcpy(nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(pt_bvh_node));
    }
    if (indices)
        for (uint64_t i = 0; i < n; ++i) indices[i] = bvh.indices[i];
    return PT_OK;
}

static int finalize_scene(pt_ctx* c) {
    PT_HIP(c, hipSetDevice(c->device));
    const uint64_t n = c->mat_id.size();
    for (uint64_t i = 0; i < n; ++i)
        if (c->mat_id[i] >= c->mats.size()) return c->fail(PT_ERR_INVALID, "pt_build: material id out of range");

    // lights list, bvh.rs:29-35
    c->bvh.lights.clear();
    for (uint64_t i = 0; i < n; ++i)
        if (c->mats[c->mat_id[i]].kind == PT_MAT_EMISSIVE) c->bvh.lights.push_back((uint32_t)i);
    bool neg_nee = false;
    for (const pt_material& m : c->mats)
        if (m.kind == PT_MAT_DIFFUSE && std::fmin(1.0f - m.a, 0.0f) != 0.0f) neg_nee = true;
    c->nee = neg_nee && !c->bvh.lights.empty();

    FlatBVH flat;
    int rc = flatten_bvh(c->bvh, flat);
    if (rc) return c->fail(rc, "pt_build: flatten failed");

    // primitives in leaf order
    std::vector<uint32_t> slot_of(n);
    for (uint64_t s = 0; s < n; ++s) slot_of[c->bvh.indices[s]] = (uint32_t)s;
    if (c->prim_kind == PRIM_TRIANGLE) {
        std::vector<float> rec(12 * n), nr(9 * n);
        for (uint64_t s = 0; s < n; ++s) {
            const uint32_t obj = c->bvh.indices[s];
            const float* p = c->pos.data() + 9 * (size_t)obj;
            float* r = rec.data() + 12 * s;
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2];
            r[3] = p[3] - p[0]; r[4] = p[4] - p[1]; r[5] = p[5] - p[2];  // edge1 = position1 - position0
            r[6] = p[6] - p[0]; r[7] = p[7] - p[1]; r[8] = p[8] - p[2];  // edge2 = position2 - position0
            uint32_t id = obj, mat = c->mat_id[obj], zero = 0;
            std::memcpy(r + 9, &id, 4);
            std::memcpy(r + 10, &mat, 4);
            std::memcpy(r + 11, &zero, 4);
            std::memcpy(nr.data() + 9 * s, c->nrm.data() + 9 * (size_t)obj, 9 * sizeof(float));
        }
        if ((rc = dev_upload(c, c->d_prim, rec.data(), rec.size() * sizeof(float)))) return rc;
        if ((rc = dev_upload(c, c->d_nrm, nr.data(), nr.size() * sizeof(float)))) return rc;
        if ((rc = dev_alloc(c, c->d_meta, 16))) return rc;
    } else {
        std::vector<float> rec(4 * n);
        std::vector<uint32_t> meta(2 * n);
        for (uint64_t s = 0; s < n; ++s) {
            const uint32_t obj = c->bvh.indices[s];
            std::memcpy(rec.data() + 4 * s, c->cr.data() + 4 * (size_t)obj, 16);
            meta[2 * s] = obj;
            meta[2 * s + 1] = c->mat_id[obj];
        }
        if ((rc = dev_upload(c, c->d_prim, rec.data(), rec.size() * sizeof(float)))) return rc;
        if ((rc = dev_upload(c, c->d_meta, meta.data(), meta.size() * sizeof(uint32_t)))) return rc;
        if ((rc = dev_alloc(c, c->d_nrm, 16))) return rc;
    }
    std::vector<uint32_t> light_slots;
    for (uint32_t obj : c->bvh.lights) light_slots.push_back(slot_of[obj]);
    if ((rc = dev_upload(c, c->d_pairs, flat.pairs.data(), flat.pairs.size() * sizeof(PairRecord)))) return rc;
    if ((rc = dev_upload(c, c->d_big, flat.big_leaf.data(), flat.big_leaf.size() * sizeof(uint32_t)))) return rc;
    if ((rc = dev_upload(c, c->d_mats, c->mats.data(), c->mats.size() * sizeof(pt_material)))) return rc;
    if ((rc = dev_upload(c, c->d_lights, light_slots.data(), light_slots.size() * sizeof(uint32_t)))) return rc;

    DevScene& S = c->ds;
    S.pairs = (const float4*)c->d_pairs.p;
    S.prim = (const float4*)c->d_prim.p;
    S.prim_meta = (const uint32_t*)c->d_meta.p;
    S.nrm = (const float*)c->d_nrm.p;
    S.big_leaf = (const uint32_t*)c->d_big.p;
    S.mats = (const pt_material*)c->d_mats.p;
    S.lights = (const uint32_t*)c->d_lights.p;
    S.n_lights = (uint32_t)light_slots.size();
    S.n_prims = (uint32_t)n;
    S.root_ref = flat.root_ref;
    S.stack_cap = c->bvh.max_depth > 0 ? c->bvh.max_depth - 1 : 0;
    for (int a = 0; a < 3; ++a) {
        S.root_min[a] = flat.root_min[a];
        S.root_max[a] = flat.root_max[a];
    }
    // overflow area of the LDS stacks: exactly the levels the tree can need beyond kLdsStack
    const uint32_t extra = S.stack_cap > (uint32_t)kLdsStack ? S.stack_cap - kLdsStack : 0;
    if ((rc = dev_alloc(c, c->d_spill, (size_t)std::max(extra, 1u) * c->cfg.spill_stride * sizeof(unsigned long long)))) return rc;
    c->cfg.spill = (unsigned long long*)c->d_spill.p;
    c->built = true;
    c->have_cam = false;
    return PT_OK;
}

int pt_get_bvh_sizes(const pt_ctx* c, uint64_t* n_nodes, uint64_t* n_indices, uint32_t* max_depth) {
    if (!c || !c->built) return PT_ERR_STATE;
    if (n_nodes) *n_nodes = c->bvh.nodes.size();
    if (n_indices) *n_indices = c->bvh.indices.size();
    if (max_depth) *max_depth = c->bvh.max_depth;
    return PT_OK;
}

int pt_get_bvh(const pt_ctx* c, pt_bvh_node* nodes, uint64_t* indices) {
    if (!c || !c->built) return PT_ERR_STATE;
    if (nodes) std::memcpy(nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(pt_bvh_node));
    if (indices)
        for (size_t i = 0; i < c->bvh.indices.size(); ++i) indices[i] = c->bvh.indices[i];
    return PT_OK;
}

int pt_set_camera(pt_ctx* c, const float origin[3], const float target[3], uint32_t width, uint32_t height, float lens_size) {
    if (!c) return PT_ERR_INVALID;
    if (!c->built) return c->fail(PT_ERR_STATE, "pt_set_camera: call pt_build first");
    if (!origin || !target || width == 0 || height == 0) return c->fail(PT_ERR_INVALID, "pt_set_camera: bad arguments");
    if ((uint64_t)width * height >= kMaxSlots) return c->fail(PT_ERR_LIMIT, "pt_set_camera: image too large");
    PT_HIP(c, hipSetDevice(c->device));
    DevCamera cam{};
    for (int a = 0; a < 3; ++a) {
        cam.origin[a] = origin[a];
        cam.target[a] = target[a];
    }
    cam.lens_size = lens_size;
    cam.width = width;
    cam.height = height;
    int rc = dev_upload(c, c->d_cam, &cam, sizeof(cam));
    if (rc) return rc;
    launch_focus(c->prim_kind, c->ds, (DevCamera*)c->d_cam.p, c->cfg, c->stream);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, hipMemcpy(&cam, c->d_cam.p, sizeof(cam), hipMemcpyDeviceToHost));
    pt_camera_state& s = c->cam;
    for (int a = 0; a < 3; ++a) {
        s.origin[a] = cam.origin[a]; s.target[a] = cam.target[a]; s.direction[a] = cam.direction[a];
        s.right[a] = cam.right[a]; s.up[a] = cam.up[a];
        s.p1[a] = cam.p1[a]; s.p2[a] = cam.p2[a]; s.p3[a] = cam.p3[a];
    }
    s.focal_distance = cam.focal_distance;
    s.lens_size = cam.lens_size;
    s.width = width;
    s.height = height;
    if ((rc = dev_alloc(c, c->d_accum, (size_t)width * height * 3 * sizeof(float)))) return rc;
    c->have_cam = true;
    return pt_reset_accum(c);
}

int pt_get_camera(const pt_ctx* c, pt_camera_state* out) {
    if (!c || !out) return PT_ERR_INVALID;
    if (!c->have_cam) return PT_ERR_STATE;
    *out = c->cam;
    return PT_OK;
}

int pt_reset_accum(pt_ctx* c) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_reset_accum: no camera");
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemsetAsync(c->d_accum.p, 0, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    c->spp = 0.f;
    c->samples_done = 0;
    return PT_OK;
}

int pt_generate(pt_ctx* c, const uint32_t* xy, uint64_t n, uint32_t sample, uint64_t seed, pt_ray* rays) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_generate: no camera");
    if (n == 0) return PT_OK;
    if (!xy || !rays) return c->fail(PT_ERR_INVALID, "pt_generate: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    DevBuf dxy, dr;
    int rc;
    if ((rc = dev_upload(c, dxy, xy, n * 8))) return rc;
    if ((rc = dev_alloc(c, dr, n * sizeof(pt_ray)))) { dxy.release(); return rc; }
    launch_generate((const DevCamera*)c->d_cam.p, (const uint32_t*)dxy.p, n, sample, (uint32_t)seed, (uint32_t)(seed >> 32), (pt_ray*)dr.p,
                    c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(rays, dr.p, n * sizeof(pt_ray), hipMemcpyDeviceToHost);
    dxy.release();
    dr.release();
    if (e != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_generate: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_trace(pt_ctx* c, const pt_ray* rays, uint64_t n, int any_hit, pt_hit* hits) {
    if (!c) return PT_ERR_INVALID;
    if (!c->built) return c->fail(PT_ERR_STATE, "pt_trace: call pt_build first");
    if (n == 0) return PT_OK;
    if (!rays || !hits) return c->fail(PT_ERR_INVALID, "pt_trace: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    DevBuf dr, dh;
    int rc;
    if ((rc = dev_upload(c, dr, rays, n * sizeof(pt_ray)))) return rc;
    if ((rc = dev_alloc(c, dh, n * sizeof(pt_hit)))) { dr.release(); return rc; }
    launch_trace(c->prim_kind, any_hit != 0, true, c->ds, (const pt_ray*)dr.p, n, (pt_hit*)dh.p, c->cfg,
                 (unsigned long long*)c->d_counters.p, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(hits, dh.p, n * sizeof(pt_hit), hipMemcpyDeviceToHost);
    dr.release();
    dh.release();
    if (e != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_trace: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_sample_skybox(pt_ctx* c, const float* dirs, uint64_t n, float* rgb) {
    if (!c) return PT_ERR_INVALID;
    if (!c->ds.sky) return c->fail(PT_ERR_STATE, "pt_sample_skybox: no sky set");
    if (n == 0) return PT_OK;
    if (!dirs || !rgb) return c->fail(PT_ERR_INVALID, "pt_sample_skybox: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    DevBuf dd, dc;
    int rc;
    if ((rc = dev_upload(c, dd, dirs, n * 12))) return rc;
    if ((rc = dev_alloc(c, dc, n * 12))) { dd.release(); return rc; }
    launch_sky(c->ds, (const float*)dd.p, n, (float*)dc.p, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(rgb, dc.p, n * 12, hipMemcpyDeviceToHost);
    dd.release();
    dc.release();
    if (e != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_sample_skybox: ") + hipGetErrorString(e));
    return PT_OK;
}

static int ensure_pipeline(pt_ctx* c, uint64_t slots, uint32_t rounds) {
    int rc;
    if (slots > c->cap_slots) {
        if ((rc = dev_alloc(c, c->d_ro, slots * 16))) return rc;
        if ((rc = dev_alloc(c, c->d_rd, slots * 16))) return rc;
        if ((rc = dev_alloc(c, c->d_hit, slots * 16))) return rc;
        if ((rc = dev_alloc(c, c->d_tp, slots * 16))) return rc;
        if ((rc = dev_alloc(c, c->d_out, slots * 16))) return rc;
        if ((rc = dev_alloc(c, c->d_pxs, slots * 8))) return rc;
        if ((rc = dev_alloc(c, c->d_q0, slots * 4))) return rc;
        if ((rc = dev_alloc(c, c->d_q1, slots * 4))) return rc;
        c->cap_slots = slots;
    }
    if (c->nee && c->d_acc.bytes < slots * 16)
        if ((rc = dev_alloc(c, c->d_acc, slots * 16))) return rc;
    if (rounds + 2 > c->qn_len) {
        if ((rc = dev_alloc(c, c->d_qn, (size_t)(rounds + 2) * 4))) return rc;
        c->qn_len = rounds + 2;
    }
    c->ps.ro = (float4*)c->d_ro.p;
    c->ps.rd = (float4*)c->d_rd.p;
    c->ps.hit = (float4*)c->d_hit.p;
    c->ps.tp = (float4*)c->d_tp.p;
    c->ps.out = (float4*)c->d_out.p;
    c->ps.acc = (float4*)c->d_acc.p;
    c->ps.pxs = (uint2*)c->d_pxs.p;
    return PT_OK;
}

int pt_render(pt_ctx* c, uint32_t spp, uint32_t max_depth, uint64_t seed, const pt_shard* shard, uint32_t flags) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_render: call pt_set_camera first");
    if (!c->ds.sky) return c->fail(PT_ERR_STATE, "pt_render: call pt_set_sky first");
    if (max_depth > kMaxDepth) return c->fail(PT_ERR_LIMIT, "pt_render: max_depth above 65535");
    if (spp == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));

    PixelMap M{};
    M.width = c->cam.width;
    M.height = c->cam.height;
    M.tile = 8;
    M.rank = 0;
    M.nranks = 1;
    uint32_t sample_begin = c->samples_done;
    if (shard) {
        sample_begin = shard->sample_begin;
        if (shard->kind == PT_SHARD_TILES) {
            if (shard->nranks == 0 || shard->rank >= shard->nranks || shard->tile == 0 || shard->tile % 8 != 0 || shard->tile > 1024)
                return c->fail(PT_ERR_INVALID, "pt_render: bad tile shard");
            M.tile = shard->tile;
            M.rank = shard->rank;
            M.nranks = shard->nranks;
        } else if (shard->kind != PT_SHARD_ALL) {
            return c->fail(PT_ERR_INVALID, "pt_render: unknown shard kind");
        }
    }
    M.tiles_x = (M.width + M.tile - 1) / M.tile;
    M.tiles_y = (M.height + M.tile - 1) / M.tile;
    const uint32_t n_tiles = M.tiles_x * M.tiles_y;
    M.n_local_tiles = n_tiles > M.rank ? (n_tiles - M.rank + M.nranks - 1) / M.nranks : 0;
    M.npix_pad = M.n_local_tiles * M.tile * M.tile;
    if (M.npix_pad == 0) {  // this rank owns no tile
        c->spp += (float)spp;
        c->samples_done = sample_begin + spp;
        return PT_OK;
    }

    uint64_t per_batch = std::max<uint64_t>(1, kTargetSlots / M.npix_pad);
    per_batch = std::min<uint64_t>(per_batch, spp);
    if (per_batch * M.npix_pad >= kMaxSlots) return c->fail(PT_ERR_LIMIT, "pt_render: batch too large");
    const uint32_t rounds = max_depth == 0 ? 0 : 2 * max_depth;  // each iteration traces at most twice
    int rc = ensure_pipeline(c, per_batch * M.npix_pad, rounds);
    if (rc) return rc;

    const bool timed = (flags & PT_RENDER_TIMED) != 0;
    const bool count = (flags & PT_RENDER_COUNT) != 0;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t* q[2] = {(uint32_t*)c->d_q0.p, (uint32_t*)c->d_q1.p};
    uint32_t* qn = (uint32_t*)c->d_qn.p;
    unsigned long long* counters = (unsigned long long*)c->d_counters.p;
    float* accum = (float*)c->d_accum.p;

    hipEvent_t ev_begin, ev_end;
    PT_HIP(c, hipEventCreate(&ev_begin));
    PT_HIP(c, hipEventCreate(&ev_end));
    std::vector<hipEvent_t> ext_events;
    PT_HIP(c, hipEventRecord(ev_begin, c->stream));

    uint32_t done = 0;
    while (done < spp) {
        const uint32_t ns = (uint32_t)std::min<uint64_t>(per_batch, spp - done);
        const uint32_t sample_base = sample_begin + done;
        PT_HIP(c, hipMemsetAsync(qn, 0, (size_t)(rounds + 2) * 4, c->stream));
        launch_raygen((const DevCamera*)c->d_cam.p, M, c->ps, ns, sample_base, k0, k1, c->nee, q[0], qn, counters, c->stream);
        if (max_depth == 0) {
            // `for _ in 0..0` never runs: every sample is (0,0,0)
            PT_HIP(c, hipMemsetAsync(c->d_out.p, 0, (size_t)ns * M.npix_pad * 16, c->stream));
        }
        for (uint32_t r = 0; r < rounds; ++r) {
            if (timed) {
                hipEvent_t a, b;
                PT_HIP(c, hipEventCreate(&a));
                PT_HIP(c, hipEventCreate(&b));
                PT_HIP(c, hipEventRecord(a, c->stream));
                launch_extend(c->prim_kind, count, c->ds, c->ps, q[r & 1], qn + r, c->cfg, counters, c->stream);
                PT_HIP(c, hipEventRecord(b, c->stream));
                ext_events.push_back(a);
                ext_events.push_back(b);
            } else {
                launch_extend(c->prim_kind, count, c->ds, c->ps, q[r & 1], qn + r, c->cfg, counters, c->stream);
            }
            c->extend_launches += 1;
            launch_shade(c->prim_kind, c->nee, c->ds, c->ps, M, q[r & 1], qn + r, q[(r + 1) & 1], qn + r + 1, max_depth, sample_base,
                         k0, k1, c->cfg, counters, c->stream);
            // stop launching once the queue has drained (checked every other round after the first few)
            if (r >= 3 && (r & 1) == 1) {
                PT_HIP(c, hipMemcpyAsync(c->h_qn, qn + r + 1, 4, hipMemcpyDeviceToHost, c->stream));
                PT_HIP(c, hipStreamSynchronize(c->stream));
                if (*c->h_qn == 0) break;
            }
        }
        launch_accumulate(M, c->ps.out, ns, accum, c->stream);
        PT_HIP(c, hipGetLastError());
        done += ns;
    }
    PT_HIP(c, hipEventRecord(ev_end, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    PT_HIP(c, hipEventElapsedTime(&ms, ev_begin, ev_end));
    c->render_ms += ms;
    for (size_t i = 0; i + 1 < ext_events.size(); i += 2) {
        float e = 0.f;
        PT_HIP(c, hipEventElapsedTime(&e, ext_events[i], ext_events[i + 1]));
        c->extend_ms += e;
    }
    for (hipEvent_t e : ext_events) (void)hipEventDestroy(e);
    (void)hipEventDestroy(ev_begin);
    (void)hipEventDestroy(ev_end);
    c->spp += (float)spp;
    c->samples_done = sample_begin + spp;
    return PT_OK;
}

int pt_read_accum(pt_ctx* c, float* rgb, float* spp_out) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_read_accum: no camera");
    PT_HIP(c, hipSetDevice(c->device));
    if (rgb) PT_HIP(c, hipMemcpy(rgb, c->d_accum.p, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (spp_out) *spp_out = c->spp;
    return PT_OK;
}

int pt_read_rgb8(pt_ctx* c, uint8_t* rgb, uint64_t pitch) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_read_rgb8: no camera");
    if (!rgb || pitch < (uint64_t)c->cam.width * 3) return c->fail(PT_ERR_INVALID, "pt_read_rgb8: null pointer or pitch < 3 * width");
    PT_HIP(c, hipSetDevice(c->device));
    DevBuf d;
    int rc = dev_alloc(c, d, (size_t)pitch * c->cam.height);
    if (rc) return rc;
    const float scale = 1.0f / c->spp;  // main.rs:145
    launch_resolve((const float*)c->d_accum.p, c->cam.width, c->cam.height, scale, (uint8_t*)d.p, pitch, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    // only the 3 * width payload bytes of each row are written; the caller's padding is left alone
    if (e == hipSuccess)
        e = hipMemcpy2D(rgb, pitch, d.p, pitch, (size_t)c->cam.width * 3, c->cam.height, hipMemcpyDeviceToHost);
    d.release();
    if (e != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_read_rgb8: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_accum_device_ptr(pt_ctx* c, void** ptr, uint64_t* n_floats) {
    if (!c || !ptr) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_accum_device_ptr: no camera");
    *ptr = c->d_accum.p;
    if (n_floats) *n_floats = (uint64_t)c->cam.width * c->cam.height * 3;
    return PT_OK;
}

int pt_add_external_samples(pt_ctx* c, float spp) {
    if (!c) return PT_ERR_INVALID;
    c->spp += spp;
    return PT_OK;
}

int pt_get_counters(pt_ctx* c, pt_counters* out) {
    if (!c || !out) return PT_ERR_INVALID;
    PT_HIP(c, hipSetDevice(c->device));
    unsigned long long cur[CNT_N];
    PT_HIP(c, hipMemcpy(cur, c->d_counters.p, sizeof(cur), hipMemcpyDeviceToHost));
    out->paths = cur[CNT_PATHS];
    out->rays_closest = cur[CNT_RAYS_CLOSEST];
    out->rays_any = cur[CNT_RAYS_ANY];
    out->node_tests = cur[CNT_NODE_TESTS];
    out->prim_tests = cur[CNT_PRIM_TESTS];
    out->extend_launches = c->extend_launches;
    out->extend_ms = c->extend_ms;
    out->render_ms = c->render_ms;
    return PT_OK;
}

int pt_reset_counters(pt_ctx* c) {
    if (!c) return PT_ERR_INVALID;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemset(c->d_counters.p, 0, CNT_N * sizeof(unsigned long long)));
    c->extend_launches = 0;
    c->extend_ms = 0;
    c->render_ms = 0;
    return PT_OK;
}

}  // extern "C"
