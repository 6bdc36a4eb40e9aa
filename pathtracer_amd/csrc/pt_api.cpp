// pt_api.cpp — the C ABI of libptcore.so (include/ptcore.h): context, scene upload, render loop.
// Host code only; every float operation that decides a result lives in pt_bvh.cpp (build) or in
// the kernels.  There is no CPU rendering path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_bvh.h"
#include "pt_build.h"
#include "pt_device.h"
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

struct ScopedBuf : DevBuf {  // a device buffer that lives for one call
    ~ScopedBuf() { release(); }
};

constexpr uint32_t kMaxMaterials = 1u << 15;   // material id must fit the 15 bits saved across a look-ahead
constexpr uint32_t kMaxDepth = 0xffffu;        // iteration counter has 16 bits
constexpr uint64_t kTargetSlots = 8u << 20;    // wavefront mode: paths kept in flight per batch (about 8.4 M, 112 B of state each)
constexpr uint64_t kPathBatchSlots = 512u << 20;  // path-kernel mode: paths per launch (16 B each); measured: the larger the better
constexpr uint64_t kMaxSlots = 1ull << 31;
constexpr uint64_t kMergeSlots = 128ull << 20;  // queued calls are merged up to this many paths (64 spp at 1080p: ~25 ms a launch)
constexpr uint32_t kMaxLanes = 8;              // batches in flight (one HIP stream + path-state set each)
constexpr uint64_t kTailMinPaths = 65536;      // hand a batch to the fused tail kernel below this many live paths ...
constexpr uint64_t kTailFraction = 16;         // ... or below 1/16 of the batch
constexpr uint64_t kRedoCap = 65536;           // paths per batch that may ask for the NEE block in a scene whose kernels leave it out

// ---- pt_trace_noalloc: everything pt_trace allocates per call lives in one static device array instead.
// Layout: [rays: kNoAllocMaxRays x pt_ray][hits: kNoAllocMaxRays x pt_hit][stack spill: levels x lanes x 8 B].
constexpr uint32_t kNoAllocMaxRays = 16384;      // rays per call (= lanes of the launch: one ray per lane)
constexpr uint32_t kNoAllocSpillLevels = 64;     // stack levels beyond the LDS part: trees up to kLdsStack + 64 levels deep
constexpr size_t kScratchRays = 0;
constexpr size_t kScratchHits = kScratchRays + (size_t)kNoAllocMaxRays * sizeof(pt_ray);
constexpr size_t kScratchSpill = (kScratchHits + (size_t)kNoAllocMaxRays * sizeof(pt_hit) + 255) & ~(size_t)255;
constexpr size_t kScratchBytes = kScratchSpill + (size_t)kNoAllocSpillLevels * kNoAllocMaxRays * sizeof(unsigned long long);  // 8.6 MiB
constexpr int kMaxDevices = 64;

// One batch in flight.
struct RenderJob;
struct RenderLane {
    enum State { IDLE, RUNNING, TAILING, READY, DRAINING };
    State state = IDLE;
    hipStream_t stream = nullptr;
    hipEvent_t ev_check = nullptr, ev_done = nullptr;
    uint32_t* h_qn = nullptr;  // pinned read-back word
    DevBuf d_ro, d_rd, d_hit, d_tp, d_out, d_acc, d_pxs, d_q0, d_q1, d_qn, d_spill, d_redo;
    bool nee = false;          // the batch in flight runs the kernels with the NEE code (the scene needs them, or a restart after
                               // more paths asked for the block than the redo list holds)
    PathState ps{};
    LaunchCfg cfg{};
    LaunchCfg cfg7{};   // for k_path's 7-wave form: larger grid, same spill area
    uint64_t cap_slots = 0;
    uint32_t qn_len = 0;
    uint32_t batch = 0, ns = 0, sample_base = 0, round = 0, live = 0;
    RenderJob* job = nullptr;  // the call this batch belongs to
    uint64_t seq = 0;                 // place of the batch in the context-wide accumulation order
};

// One pt_render / pt_render_async call: its frame parameters and how far its batches have got.
struct RenderJob {
    PixelMap M;
    uint32_t max_depth, rounds, k0, k1;
    bool timed, count;
    uint32_t tail_threshold;
    uint32_t spp, sample_begin, n_batches, max_lanes;
    uint64_t per_batch, batch_slots;
    uint32_t next_batch = 0, finished = 0;
    uint64_t seq0 = 0;  // place of batch 0 in the context-wide accumulation order (sample order across calls)
    std::chrono::steady_clock::time_point t0;
    std::vector<hipEvent_t> ext_events, tail_events;  // (begin, end) pairs of PT_RENDER_TIMED launches
};

}  // namespace

// The scratch memory of pt_trace_noalloc.  Every GPU the library runs on holds its own copy at its own
// address for the life of the process.  In host code this name is only a handle for
// hipGetSymbolAddress: the device address is resolved once per device in pt_create (g_scratch_ptr)
// and only that pointer is ever handed to kernels or copies.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
static std::atomic<unsigned char*> g_scratch_ptr[kMaxDevices] = {};  // (contexts may be created from several host threads)

// with `device` current: the address of this device's copy of `scratch`
static hipError_t resolve_scratch(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    unsigned char* p = g_scratch_ptr[device].load(std::memory_order_acquire);
    if (!p) {
        hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device].store(p, std::memory_order_release);  // every thread resolves the same address
    }
    *out = p;
    return hipSuccess;
}

struct pt_sched {
    std::deque<RenderJob> jobs;   // submitted, not yet retired (front = oldest)
    uint64_t next_seq = 0;        // sequence number the next submitted batch gets
    uint64_t acc_seq = 0;         // sequence number of the next batch to be added to the accumulator
    hipEvent_t ev_epoch = nullptr;  // time zero of the timed intervals below
    // [begin, end) in ms since ev_epoch of the timed launches that may still overlap one to come; whenever the queue runs
    // empty their union is added to *_union_done, the lists are cleared and the epoch is dropped (the next timed call sets a
    // new one), so neither the lists nor the offsets grow with the life of the context
    std::vector<std::pair<double, double>> ext_iv, tail_iv;
    double ext_union_done = 0, tail_union_done = 0;
    double ext_sum = 0, tail_sum = 0;
};


struct pt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    int num_cu = 0;

    // host copy of the scene
    int prim_kind = -1;
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
    uint8_t* h_rgb = nullptr; // pinned host staging of the same readouts (kept between calls)
    size_t h_rgb_bytes = 0;
    DevBuf d_rgb;             // RGB24 staging of pt_read_rgb8 / pt_read_rgb8_latest (kept between calls)
    DevBuf d_stage_in, d_stage_out;  // device copies of the seams' inputs / outputs (pt_trace, pt_generate, pt_sample_skybox); grow only
    float spp = 0.f;          // samples submitted (what the frame holds once everything queued has been added)
    float spp_added = 0.f;    // samples whose k_accumulate has been enqueued on the accumulate stream so far
    uint32_t samples_done = 0;      // global index the next shard-less call starts at (samples submitted)
    uint32_t samples_added_end = 0; // ... and the one behind the last batch whose accumulate has been enqueued

    // pipeline
    LaunchCfg cfg{};          // grid geometry; every lane has its own spill area
    LaunchCfg cfg7{};         // for k_path's 7-wave form (pt_kernels.h, kLdsStack7)
    uint64_t opt_waves = 0;   // PT_OPT_WAVES_PER_SIMD: 0 = by scene size, 6, 7
    bool auto_seven = false;  // the scene's records do not fit the Infinity Cache (decided at pt_build)
    size_t spill_bytes = 16;
    DevBuf d_spill;           // spill area of the seams (pt_trace, focus)
    unsigned char* scratch_ptr = nullptr;  // this device's copy of the static scratch array (pt_trace_noalloc)
    RenderLane lanes[kMaxLanes];
    hipStream_t tail_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t bulk_stream[kMaxLanes] = {};  // the round kernels of lane i run on bulk_stream[i % opt_bulk_streams]  // high priority: the few latency-bound tail waves go first
    hipStream_t acc_stream = nullptr;  // every k_accumulate, in batch order
    pt_sched* sched = nullptr;         // calls in flight (pt_render_async) and their timed intervals
    DevBuf d_counters;
    double render_ms = 0;
    uint64_t extend_launches = 0, tail_launches = 0;
    uint64_t opt_batch_slots = 0 /* 0 = the mode's default */, opt_tail_min = kTailMinPaths, opt_tail_fraction = kTailFraction, opt_lanes = 0 /* 0 = the mode's default */, opt_tail_streams = 4, opt_tail_paths_per_lane = 4, opt_first_check = 3, opt_simple_extend = 0, opt_bulk_streams = kMaxLanes, opt_tail_mode = 0, opt_render_mode = 1, opt_merge_calls = kMergeSlots;
    DevBuf d_raw_a, d_raw_b, d_raw_mat;  // the objects as handed over, on the device: triangles positions + normals (9 floats each),
                                    // spheres centre + radius; material ids.  The host keeps only the material ids (lights, checks):
                                    // copying 10 M triangles into host vectors first cost more than building their tree
    DeviceTree dtree;               // the tree of a device build, kept on the GPU until pt_get_bvh asks for it
    bool host_bvh_current = true;   // c->bvh.nodes / indices hold the tree (false: only dtree does)
    uint64_t opt_device_build = 1;  // pt_build builds the tree on the GPU (pt_build.hip) when the input allows it
    bool build_on_device = false;   // what the last pt_build did
    double build_ms = 0;            // ... and how long the tree build itself took (without flattening / upload)
    uint64_t opt_redo_cap = kRedoCap;
    uint64_t redo_paths = 0, redo_restarts = 0;  // paths finished by k_redo / batches run again with the NEE kernels, since the last reset
    uint64_t merged_calls = 0;  // pt_render_async calls that were added to a queued call instead of becoming a launch of their own
    std::vector<uint32_t> round_sizes;  // queue length of every round of the last finished batch

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

static int drain(pt_ctx* c);  // wait for every call submitted with pt_render_async (defined with the scheduler below)

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
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // numerically lower = higher priority
        for (int i = 0; i < 4; ++i)
            if (hipStreamCreateWithPriority(&c->tail_stream[i], hipStreamNonBlocking, i < 2 ? hi : lo) != hipSuccess) {
                pt_destroy(c);
                return PT_ERR_HIP;
            }
    }
    if (resolve_scratch(device, &c->scratch_ptr) != hipSuccess) {  // hipSetDevice(device) happened above
        pt_destroy(c);
        return PT_ERR_HIP;
    }
    c->sched = new pt_sched();
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // highest priority: accumulates and frame readouts are short kernels that must not queue behind the path
        // kernels' workgroups (and streams of another priority do not share a hardware queue with the launch streams)
        if (hipStreamCreateWithPriority(&c->acc_stream, hipStreamNonBlocking, hi) != hipSuccess) {
            pt_destroy(c);
            return PT_ERR_HIP;
        }
    }
    // (a memset of device memory on the null stream may still be running when the call returns, and the library's non-blocking
    // streams do not wait for the null stream: clear on the context's stream and wait for it)
    if (dev_alloc(c, c->d_counters, CNT_N * sizeof(unsigned long long)) != PT_OK ||
        hipMemsetAsync(c->d_counters.p, 0, CNT_N * sizeof(unsigned long long), c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        pt_destroy(c);
        return PT_ERR_HIP;
    }
    // persistent traversal grid: the LDS stacks (kLdsStack * 8 B per lane) bound the waves a CU can hold
    const uint32_t lds_per_wave = kLdsStack * 64 * 8;
    const uint32_t waves_per_cu = std::min<uint32_t>(kWavesPerCU, (160u * 1024u) / lds_per_wave);
    c->cfg.grid = std::max<uint32_t>(1, (uint32_t)c->num_cu * waves_per_cu / (kTraceBlock / 64));
    c->cfg.spill = nullptr;
    c->cfg.spill_stride = c->cfg.grid * kTraceBlock;
    const uint32_t waves7 = std::min<uint32_t>(kWavesPerCU7, (160u * 1024u) / (kLdsStack7 * 64 * 8));
    c->cfg7.grid = std::max<uint32_t>(1, (uint32_t)c->num_cu * waves7 / (kTraceBlock / 64));
    c->cfg7.spill = nullptr;
    c->cfg7.spill_stride = c->cfg7.grid * kTraceBlock;
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->sched) (void)drain(c);
    (void)hipDeviceSynchronize();
    if (c->sched) {
        if (c->sched->ev_epoch) (void)hipEventDestroy(c->sched->ev_epoch);
        delete c->sched;
        c->sched = nullptr;
    }
    if (c->acc_stream) (void)hipStreamDestroy(c->acc_stream);
    DevBuf* bufs[] = {&c->d_pairs, &c->d_prim, &c->d_meta, &c->d_nrm, &c->d_big, &c->d_mats, &c->d_sky, &c->d_lights, &c->d_cam,
                      &c->d_accum, &c->d_rgb, &c->d_raw_a, &c->d_raw_b, &c->d_raw_mat, &c->d_stage_in, &c->d_stage_out, &c->d_spill, &c->d_counters};
    for (DevBuf* b : bufs) b->release();
    if (c->h_rgb) (void)hipHostFree(c->h_rgb);
    c->dtree.release();
    for (RenderLane& L : c->lanes) {
        DevBuf* lb[] = {&L.d_ro, &L.d_rd, &L.d_hit, &L.d_tp, &L.d_out, &L.d_acc, &L.d_pxs, &L.d_q0, &L.d_q1, &L.d_qn, &L.d_spill, &L.d_redo};
        for (DevBuf* b : lb) b->release();
        if (L.h_qn) (void)hipHostFree(L.h_qn);
        if (L.ev_check) (void)hipEventDestroy(L.ev_check);
        if (L.ev_done) (void)hipEventDestroy(L.ev_done);
    }
    for (hipStream_t t : c->tail_stream)
        if (t) { (void)hipStreamSynchronize(t); (void)hipStreamDestroy(t); }
    for (hipStream_t t : c->bulk_stream)
        if (t) { (void)hipStreamSynchronize(t); (void)hipStreamDestroy(t); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pt_set_materials(pt_ctx* c, const pt_material* m, uint32_t count) {
    if (!c) return PT_ERR_INVALID;
    if (!m || count == 0) return c->fail(PT_ERR_INVALID, "pt_set_materials: empty material table");
    if (int rc = drain(c)) return rc;  // calls in flight still read the old scene
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
    if (int rc = drain(c)) return rc;  // calls in flight still read the old scene
    if (n > kMaxPrims) return c->fail(PT_ERR_LIMIT, "pt_set_triangles: more than 2^26 primitives");
    PT_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = dev_upload(c, c->d_raw_a, pos, 9 * n * sizeof(float)))) return rc;
    if ((rc = dev_upload(c, c->d_raw_b, nrm, 9 * n * sizeof(float)))) return rc;
    if ((rc = dev_upload(c, c->d_raw_mat, mat_id, n * sizeof(uint32_t)))) return rc;
    c->prim_kind = PRIM_TRIANGLE;
    c->mat_id.assign(mat_id, mat_id + n);
    c->built = false;
    return PT_OK;
}

int pt_set_spheres(pt_ctx* c, const float* cr, const uint32_t* mat_id, uint64_t n) {
    if (!c) return PT_ERR_INVALID;
    if (!cr || !mat_id || n == 0) return c->fail(PT_ERR_INVALID, "pt_set_spheres: null input or zero spheres");
    if (int rc = drain(c)) return rc;  // calls in flight still read the old scene
    if (n > kMaxPrims) return c->fail(PT_ERR_LIMIT, "pt_set_spheres: more than 2^26 primitives");
    PT_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = dev_upload(c, c->d_raw_a, cr, 4 * n * sizeof(float)))) return rc;
    if ((rc = dev_upload(c, c->d_raw_mat, mat_id, n * sizeof(uint32_t)))) return rc;
    c->d_raw_b.release();
    c->prim_kind = PRIM_SPHERE;
    c->mat_id.assign(mat_id, mat_id + n);
    c->built = false;
    return PT_OK;
}

int pt_set_sky(pt_ctx* c, const float* rgb, uint32_t w, uint32_t h) {
    if (!c) return PT_ERR_INVALID;
    if (!rgb || w == 0 || h == 0) return c->fail(PT_ERR_INVALID, "pt_set_sky: null or empty image");
    if (int rc = drain(c)) return rc;  // calls in flight still read the old scene
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
    if (int rc = drain(c)) return rc;
    if (c->mats.empty()) return c->fail(PT_ERR_STATE, "pt_build: no materials set");
    const uint64_t n = c->mat_id.size();
    const bool dbg = std::getenv("PTCORE_DEBUG") != nullptr;
    auto tick = [&](const char* what, std::chrono::steady_clock::time_point since) {
        if (dbg) std::fprintf(stderr, "[ptcore] pt_build: %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - since).count());
    };
    PT_HIP(c, hipSetDevice(c->device));
    c->dtree.release();
    c->build_on_device = false;
    c->built = false;
    if (c->opt_device_build) {  // same tree either way
        const auto t0 = std::chrono::steady_clock::now();
        std::string why;
        const int rc = build_bvh_device(c->prim_kind == PRIM_TRIANGLE ? 0 : 1, (const float*)c->d_raw_a.p, (uint32_t)n, c->dtree, c->stream, &why);
        if (rc == PT_ERR_HIP) {
            // the device build could not run (its temporaries are ~600 B per primitive: out of memory, typically): the host
            // build below gives the same tree; pt_get_build_info says which one ran
            (void)hipGetLastError();
            c->dtree.release();
            if (dbg) std::fprintf(stderr, "[ptcore] pt_build: device build failed (%s), building on the host\n", why.c_str());
        } else if (rc) {
            return c->fail(rc, why);
        } else {
            c->build_on_device = true;
            c->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            tick("tree (device)", t0);
            c->host_bvh_current = false;
            c->bvh.nodes.clear();
            c->bvh.indices.clear();
            c->bvh.max_depth = c->dtree.max_depth;
            c->bvh.hazard = c->dtree.hazard;
        }
    }
    if (!c->build_on_device) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<PrimInfo> info;
        {   // the host build reads the objects back (they live on the device only)
            const size_t per = c->prim_kind == PRIM_TRIANGLE ? 9 : 4;
            std::vector<float> geom(per * n);
            PT_HIP(c, hipMemcpy(geom.data(), c->d_raw_a.p, geom.size() * sizeof(float), hipMemcpyDeviceToHost));
            if (c->prim_kind == PRIM_TRIANGLE) prim_info_triangles(geom.data(), n, info);
            else prim_info_spheres(geom.data(), n, info);
        }
        build_bvh(info, c->bvh);
        c->host_bvh_current = true;
        c->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        tick("tree (host)", t0);
    }
    if (c->bvh.hazard) return c->fail(PT_ERR_BVH, "pt_build: a partition produced an empty child");
    const auto t_fin = std::chrono::steady_clock::now();
    const int rc = finalize_scene(c);
    tick("flatten, reorder, upload", t_fin);
    return rc;
}

int pt_set_bvh(pt_ctx* c, const pt_bvh_node* nodes, uint64_t n_nodes, const uint64_t* indices, uint64_t n_indices) {
    if (!c) return PT_ERR_INVALID;
    if (c->prim_kind < 0) return c->fail(PT_ERR_STATE, "pt_set_bvh: no primitives set");
    if (int rc = drain(c)) return rc;
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
        if (!covered[i]) return c->fail(PT_ERR_BVH, "pt_set_bvh: leaves do not cover every index (a primitive would vanish)");
    // indices must be a permutation of [0, n): a duplicate would leave another primitive out and
    // point the light / material tables at the wrong slot
    std::vector<uint8_t> used(n, 0);
    for (uint64_t i = 0; i < n; ++i) {
        if (indices[i] >= n) return c->fail(PT_ERR_INVALID, "pt_set_bvh: primitive index out of range");
        if (used[indices[i]]) return c->fail(PT_ERR_BVH, "pt_set_bvh: indices is not a permutation (duplicate entry)");
        used[indices[i]] = 1;
    }
    c->bvh.nodes.assign(nodes, nodes + n_nodes);
    c->bvh.indices.resize(n);
    for (uint64_t i = 0; i < n; ++i) c->bvh.indices[i] = (uint32_t)indices[i];
    c->bvh.max_depth = max_depth;
    c->bvh.hazard = false;
    c->dtree.release();
    c->host_bvh_current = true;
    c->build_on_device = false;
    c->build_ms = 0;
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
    const bool dbg = std::getenv("PTCORE_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (dbg) std::fprintf(stderr, "[ptcore] finalize: %-26s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // material ids in range; lights list, bvh.rs:29-35 (one pass over the ids: 10 M of them are a few milliseconds)
    std::vector<uint8_t> emissive(c->mats.size());
    for (size_t m = 0; m < c->mats.size(); ++m) emissive[m] = c->mats[m].kind == PT_MAT_EMISSIVE;
    c->bvh.lights.clear();
    const uint32_t n_mats = (uint32_t)c->mats.size();
    const uint32_t* ids = c->mat_id.data();
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t id = ids[i];
        if (id >= n_mats) return c->fail(PT_ERR_INVALID, "pt_build: material id out of range");
        if (emissive[id]) c->bvh.lights.push_back((uint32_t)i);
    }
    bool neg_nee = false;
    for (const pt_material& m : c->mats)
        if (m.kind == PT_MAT_DIFFUSE && std::fmin(1.0f - m.a, 0.0f) != 0.0f) neg_nee = true;
    c->nee = neg_nee && !c->bvh.lights.empty();
#ifdef PT_FORCE_NEE
    c->nee = !c->bvh.lights.empty();  // A/B: what the NEE code costs when it is compiled in and never runs (C3: -7.5 %)
#endif
    // Up to which |transport| the NEE block may be left out where its weight is 0 (shade_core has the argument): the block's
    // product is transport * (cos / mis <= pi) * (light colour * brdf), so with lcb = the largest |light colour * colour / pi|
    // over every emissive x diffuse material pair it stays finite for |transport| <= FLT_MAX / (3.25 * lcb).  A scene with a
    // product that is not finite (here) or a light whose area is not a finite number >= 0 (k_light_check below) always runs it.
    bool literal = false;
    float lcb = 0.0f;
    for (const pt_material& e : c->mats) {
        if (e.kind != PT_MAT_EMISSIVE) continue;
        for (const pt_material& d : c->mats) {
            if (d.kind != PT_MAT_DIFFUSE) continue;
            for (int ch = 0; ch < 3; ++ch) {
                const float v = std::fabs(e.rgb[ch] * (kFrac1Pi * d.rgb[ch]));
                if (!(v <= 3.402823466e38f)) literal = true;
                else lcb = std::max(lcb, v);
            }
        }
    }
    {
        const double t = lcb > 0.0f ? 3.4028234663852886e38 / (3.25 * (double)lcb) : 3.4028234663852886e38;
        c->ds.nee_thr = t >= 3.4028234663852886e38 ? 3.402823466e38f : (float)t;
    }

    // triangle leaves of kCoopLeafFrom primitives or more are marked for wave-cooperative testing (step_prim_wave);
    // sphere leaves stay sequential (Sphere::intersect compares with ray.distance), so only the > 31 escape applies
    const uint32_t big_from = (c->prim_kind == PRIM_TRIANGLE && kCoopLeafFrom != 0) ? kCoopLeafFrom : 32u;
    const bool on_device = !c->host_bvh_current;  // the tree is in c->dtree only
    int rc;
    uint32_t root_ref = 0;
    float root_min[3], root_max[3];
    if (on_device) {
        DeviceFlat flat;
        std::string why;
        if ((rc = flatten_bvh_device(c->dtree, big_from, flat, c->stream, &why))) return c->fail(rc, why);
        c->d_pairs.release();
        c->d_big.release();
        c->d_pairs.p = flat.pairs;
        c->d_pairs.bytes = flat.pairs_bytes;
        c->d_big.p = flat.big_leaf;
        c->d_big.bytes = flat.big_bytes;
        root_ref = flat.root_ref;
        for (int a = 0; a < 3; ++a) { root_min[a] = flat.root_min[a]; root_max[a] = flat.root_max[a]; }
    } else {
        FlatBVH flat;
        rc = flatten_bvh(c->bvh, flat, big_from);
        if (rc) return c->fail(rc, "pt_build: flatten failed");
        if ((rc = dev_upload(c, c->d_pairs, flat.pairs.data(), flat.pairs.size() * sizeof(PairRecord)))) return rc;
        if ((rc = dev_upload(c, c->d_big, flat.big_leaf.data(), flat.big_leaf.size() * sizeof(uint32_t)))) return rc;
        root_ref = flat.root_ref;
        for (int a = 0; a < 3; ++a) { root_min[a] = flat.root_min[a]; root_max[a] = flat.root_max[a]; }
    }
    tick("lights, flatten");

    // primitives in leaf order: the objects go up as they are and a kernel gathers them into their slots
    // (a host loop over 10 M triangles took 0.6 s, most of it first-touch of the 840 MB it filled)
    std::vector<uint32_t> slot_of;
    {
        ScopedBuf raw_idx, d_slot;
        const uint32_t n32 = (uint32_t)n;
        const uint32_t* d_idx = c->dtree.indices;
        if (!on_device) {
            if ((rc = dev_upload(c, raw_idx, c->bvh.indices.data(), n * sizeof(uint32_t)))) return rc;
            d_idx = (const uint32_t*)raw_idx.p;
        }
        if ((rc = dev_alloc(c, d_slot, n * sizeof(uint32_t)))) return rc;
        if (c->prim_kind == PRIM_TRIANGLE) {
            if ((rc = dev_alloc(c, c->d_prim, 4 * (size_t)kTriStride * n * sizeof(float)))) return rc;
            if ((rc = dev_alloc(c, c->d_nrm, (size_t)kNrmStride * n * sizeof(float)))) return rc;
            if ((rc = dev_alloc(c, c->d_meta, 16))) return rc;
            launch_reorder_triangles((const float*)c->d_raw_a.p, (const float*)c->d_raw_b.p, (const uint32_t*)c->d_raw_mat.p, d_idx, n32, (float*)c->d_prim.p,
                                     (float*)c->d_nrm.p, (uint32_t*)d_slot.p, c->stream);
        } else {
            if ((rc = dev_alloc(c, c->d_prim, 4 * n * sizeof(float)))) return rc;
            if ((rc = dev_alloc(c, c->d_meta, 2 * n * sizeof(uint32_t)))) return rc;
            if ((rc = dev_alloc(c, c->d_nrm, 16))) return rc;
            launch_reorder_spheres((const float*)c->d_raw_a.p, (const uint32_t*)c->d_raw_mat.p, d_idx, n32, (float*)c->d_prim.p, (uint32_t*)c->d_meta.p,
                                   (uint32_t*)d_slot.p, c->stream);
        }
        PT_HIP(c, hipGetLastError());
        PT_HIP(c, hipStreamSynchronize(c->stream));
        if (!c->bvh.lights.empty()) {
            slot_of.resize(n);
            PT_HIP(c, hipMemcpy(slot_of.data(), d_slot.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        tick("primitives to leaf order");
    }
    std::vector<uint32_t> light_slots;
    for (uint32_t obj : c->bvh.lights) light_slots.push_back(slot_of[obj]);
    if ((rc = dev_upload(c, c->d_mats, c->mats.data(), c->mats.size() * sizeof(pt_material)))) return rc;
    if ((rc = dev_upload(c, c->d_lights, light_slots.data(), light_slots.size() * sizeof(uint32_t)))) return rc;
    tick("upload tree");

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
    S.n_pairs = (uint32_t)(c->d_pairs.bytes / sizeof(PairRecord));
    S.root_ref = root_ref;
    S.stack_cap = c->bvh.max_depth > 0 ? c->bvh.max_depth - 1 : 0;
    for (int a = 0; a < 3; ++a) {
        S.root_min[a] = root_min[a];
        S.root_max[a] = root_max[a];
    }
    if (S.n_lights != 0) {
        ScopedBuf d_flag;
        if ((rc = dev_alloc(c, d_flag, 16))) return rc;
        PT_HIP(c, hipMemsetAsync(d_flag.p, 0, 16, c->stream));
        launch_light_check(c->prim_kind, S, (uint32_t*)d_flag.p, c->stream);
        PT_HIP(c, hipGetLastError());
        uint32_t bad_area = 0;
        PT_HIP(c, hipMemcpyAsync(&bad_area, d_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
        if (bad_area) literal = true;
    }
    S.nee_literal = literal && S.n_lights != 0 ? 1u : 0u;
    if (S.nee_literal) c->nee = true;  // such a scene runs the kernels with the NEE code throughout
    // overflow area of the LDS stacks: exactly the levels the tree can need beyond kLdsStack
    // (sized for both forms of k_path: the 7-wave one keeps one level less in LDS and has more lanes in flight)
    const uint32_t extra = S.stack_cap > (uint32_t)kLdsStack ? S.stack_cap - kLdsStack : 0;
    const uint32_t extra7 = S.stack_cap > (uint32_t)kLdsStack7 ? S.stack_cap - kLdsStack7 : 0;
    c->spill_bytes = std::max((size_t)std::max(extra, 1u) * c->cfg.spill_stride, (size_t)std::max(extra7, 1u) * c->cfg7.spill_stride) * sizeof(unsigned long long);
    if ((rc = dev_alloc(c, c->d_spill, c->spill_bytes))) return rc;
    c->cfg.spill = (unsigned long long*)c->d_spill.p;
    // Node and primitive records beyond what the 256-MB Infinity Cache holds come from HBM, a step then waits longest, and a
    // seventh wave per SIMD hides more than its 8 registers cost (C5 +3.4 %; C3 / C4, whose records fit, -1.4 % / -2.0 %)
    const size_t record_bytes = (size_t)(S.n_pairs + 1u) * sizeof(PairRecord) +
                                (size_t)S.n_prims * (c->prim_kind == PRIM_TRIANGLE ? 4u * kTriStride * sizeof(float) : 4u * sizeof(float));
    c->auto_seven = record_bytes > (size_t)256 << 20;
    c->built = true;
    c->have_cam = false;
    return PT_OK;
}

int pt_get_build_info(const pt_ctx* c, uint32_t* on_device, double* build_ms) {
    if (!c || !c->built) return PT_ERR_STATE;
    if (on_device) *on_device = c->build_on_device ? 1u : 0u;
    if (build_ms) *build_ms = c->build_ms;
    return PT_OK;
}

int pt_get_bvh_sizes(const pt_ctx* c, uint64_t* n_nodes, uint64_t* n_indices, uint32_t* max_depth) {
    if (!c || !c->built) return PT_ERR_STATE;
    if (n_nodes) *n_nodes = c->host_bvh_current ? c->bvh.nodes.size() : c->dtree.n_nodes;
    if (n_indices) *n_indices = c->host_bvh_current ? c->bvh.indices.size() : c->dtree.n_prims;
    if (max_depth) *max_depth = c->bvh.max_depth;
    return PT_OK;
}

int pt_get_bvh(const pt_ctx* cc, pt_bvh_node* nodes, uint64_t* indices) {
    if (!cc || !cc->built) return PT_ERR_STATE;
    pt_ctx* c = const_cast<pt_ctx*>(cc);  // a device-built tree comes to the host the first time it is asked for
    if (!c->host_bvh_current) {
        PT_HIP(c, hipSetDevice(c->device));
        std::string why;
        if (int rc = download_tree(c->dtree, c->bvh, c->stream, &why)) return c->fail(rc, why);
        c->host_bvh_current = true;
        c->dtree.release();
    }
    if (nodes) std::memcpy(nodes, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(pt_bvh_node));
    if (indices)
        for (size_t i = 0; i < c->bvh.indices.size(); ++i) indices[i] = c->bvh.indices[i];
    return PT_OK;
}

int pt_set_camera(pt_ctx* c, const float origin[3], const float target[3], uint32_t width, uint32_t height, float lens_size) {
    if (!c) return PT_ERR_INVALID;
    if (!c->built) return c->fail(PT_ERR_STATE, "pt_set_camera: call pt_build first");
    if (!origin || !target || width == 0 || height == 0) return c->fail(PT_ERR_INVALID, "pt_set_camera: bad arguments");
    if ((((uint64_t)width + 7) & ~7ull) * (((uint64_t)height + 7) & ~7ull) >= kMaxSlots)  // padded to whole 8x8 pixel tiles
        return c->fail(PT_ERR_LIMIT, "pt_set_camera: image too large");
    if (int rc = drain(c)) return rc;
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
    if (int rc = drain(c)) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemsetAsync(c->d_accum.p, 0, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    c->spp = c->spp_added = 0.f;
    c->samples_done = c->samples_added_end = 0;
    return PT_OK;
}

int pt_generate(pt_ctx* c, const uint32_t* xy, uint64_t n, uint32_t sample, uint64_t seed, pt_ray* rays) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_generate: no camera");
    if (n == 0) return PT_OK;
    if (!xy || !rays) return c->fail(PT_ERR_INVALID, "pt_generate: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = dev_upload(c, c->d_stage_in, xy, n * 8))) return rc;
    if ((rc = dev_alloc(c, c->d_stage_out, n * sizeof(pt_ray)))) return rc;
    launch_generate((const DevCamera*)c->d_cam.p, (const uint32_t*)c->d_stage_in.p, n, sample, (uint32_t)seed, (uint32_t)(seed >> 32),
                    (pt_ray*)c->d_stage_out.p, c->stream);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, hipMemcpy(rays, c->d_stage_out.p, n * sizeof(pt_ray), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_trace(pt_ctx* c, const pt_ray* rays, uint64_t n, int any_hit, pt_hit* hits) {
    if (!c) return PT_ERR_INVALID;
    if (!c->built) return c->fail(PT_ERR_STATE, "pt_trace: call pt_build first");
    if (n == 0) return PT_OK;
    if (!rays || !hits) return c->fail(PT_ERR_INVALID, "pt_trace: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = dev_upload(c, c->d_stage_in, rays, n * sizeof(pt_ray)))) return rc;
    if ((rc = dev_alloc(c, c->d_stage_out, n * sizeof(pt_hit)))) return rc;
    launch_trace(c->prim_kind, any_hit != 0, true, c->ds, (const pt_ray*)c->d_stage_in.p, n, (pt_hit*)c->d_stage_out.p, c->cfg,
                 (unsigned long long*)c->d_counters.p, c->stream);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, hipMemcpy(hits, c->d_stage_out.p, n * sizeof(pt_hit), hipMemcpyDeviceToHost));
    return PT_OK;
}

// pt_trace for callers that may not allocate: rays, hits and the stack-spill area live in this
// device's copy of the static `scratch` array (through the pointer resolved in pt_create, never
// through the array's name).  No hipMalloc / hipFree here or below.  Nothing in the scratch has to
// start at zero: staged rays and hits are written before they are read, and a stack level is
// always pushed before it is popped.  All contexts of one GPU share the array, so calls that
// overlap on one GPU race.
int pt_trace_noalloc(pt_ctx* c, const pt_ray* rays, uint64_t n, int any_hit, pt_hit* hits) {
    if (!c) return PT_ERR_INVALID;
    if (!c->built) return c->fail(PT_ERR_STATE, "pt_trace_noalloc: call pt_build first");
    if (n == 0) return PT_OK;
    if (!rays || !hits) return c->fail(PT_ERR_INVALID, "pt_trace_noalloc: null pointer");
    if (n > kNoAllocMaxRays)
        return c->fail(PT_ERR_LIMIT, "pt_trace_noalloc: more than 16384 rays in one call (the static scratch holds no more): split the batch or use pt_trace");
    const uint32_t extra = c->ds.stack_cap > (uint32_t)kLdsStack ? c->ds.stack_cap - kLdsStack : 0;
    if (extra > kNoAllocSpillLevels)
        return c->fail(PT_ERR_LIMIT, "pt_trace_noalloc: the tree is deeper than 76 levels (the static scratch holds 64 spill levels): use pt_trace");
    if (!c->scratch_ptr) return c->fail(PT_ERR_STATE, "pt_trace_noalloc: scratch not resolved");
    PT_HIP(c, hipSetDevice(c->device));
    unsigned char* base = c->scratch_ptr;
    pt_ray* d_rays = (pt_ray*)(base + kScratchRays);
    pt_hit* d_hits = (pt_hit*)(base + kScratchHits);
    LaunchCfg cfg;
    cfg.grid = kNoAllocMaxRays / kTraceBlock;  // one ray per lane at most: a lane's spill column is its global thread id
    cfg.spill = (unsigned long long*)(base + kScratchSpill);
    cfg.spill_stride = kNoAllocMaxRays;
    PT_HIP(c, hipMemcpyAsync(d_rays, rays, n * sizeof(pt_ray), hipMemcpyHostToDevice, c->stream));
    launch_trace(c->prim_kind, any_hit != 0, true, c->ds, d_rays, n, d_hits, cfg, (unsigned long long*)c->d_counters.p, c->stream);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipMemcpyAsync(hits, d_hits, n * sizeof(pt_hit), hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_sample_skybox(pt_ctx* c, const float* dirs, uint64_t n, float* rgb) {
    if (!c) return PT_ERR_INVALID;
    if (!c->ds.sky) return c->fail(PT_ERR_STATE, "pt_sample_skybox: no sky set");
    if (n == 0) return PT_OK;
    if (!dirs || !rgb) return c->fail(PT_ERR_INVALID, "pt_sample_skybox: null pointer");
    PT_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = dev_upload(c, c->d_stage_in, dirs, n * 12))) return rc;
    if ((rc = dev_alloc(c, c->d_stage_out, n * 12))) return rc;
    launch_sky(c->ds, (const float*)c->d_stage_in.p, n, (float*)c->d_stage_out.p, c->stream);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipStreamSynchronize(c->stream));
    PT_HIP(c, hipMemcpy(rgb, c->d_stage_out.p, n * 12, hipMemcpyDeviceToHost));
    return PT_OK;
}

// ------------------------------------------------------------------------------------------------
// The frame loop.  A pt_render call is cut into batches of whole samples (about kTargetSlots
// paths each).  Up to kMaxLanes batches are in flight at once, each on its own HIP stream with
// its own path-state buffers, so the latency-bound tail of one batch overlaps the
// bandwidth-bound first rounds of the next.  Per-pixel sums stay in sample order: the
// accumulate kernel of batch b waits for the one of batch b-1.
static int lane_alloc(pt_ctx* c, RenderLane& L, uint64_t slots, uint32_t rounds) {
    int rc;
    if (!L.ev_check) {
        PT_HIP(c, hipEventCreateWithFlags(&L.ev_check, hipEventDisableTiming));
        PT_HIP(c, hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
        PT_HIP(c, hipHostMalloc((void**)&L.h_qn, 64, hipHostMallocDefault));
    }
    if (c->opt_render_mode == 1) {  // k_path keeps the path state in registers: only the radiance slab is needed
        if ((rc = dev_alloc(c, L.d_out, slots * 16))) return rc;
    } else if (slots > L.cap_slots) {
        if ((rc = dev_alloc(c, L.d_ro, slots * 16))) return rc;
        if ((rc = dev_alloc(c, L.d_rd, slots * 16))) return rc;
        if ((rc = dev_alloc(c, L.d_hit, slots * 16))) return rc;
        if ((rc = dev_alloc(c, L.d_tp, slots * 16))) return rc;
        if ((rc = dev_alloc(c, L.d_out, slots * 16))) return rc;
        if ((rc = dev_alloc(c, L.d_pxs, slots * 8))) return rc;
        if ((rc = dev_alloc(c, L.d_q0, slots * 4))) return rc;
        if ((rc = dev_alloc(c, L.d_q1, slots * 4))) return rc;
        L.cap_slots = slots;
    }
    if (c->opt_render_mode != 1 && L.nee && L.d_acc.bytes < slots * 16)
        if ((rc = dev_alloc(c, L.d_acc, slots * 16))) return rc;
    if (rounds + 2 > L.qn_len) {  // queue lengths [0, rounds+2) then one fetch cursor per round
        if ((rc = dev_alloc(c, L.d_qn, (size_t)(rounds + 2) * 8 + 64))) return rc;  // + 8 region cursors of k_path
        L.qn_len = rounds + 2;
    }
    if (!L.d_spill.p || L.d_spill.bytes < c->spill_bytes)
        if ((rc = dev_alloc(c, L.d_spill, c->spill_bytes))) return rc;
    if ((rc = dev_alloc(c, L.d_redo, std::max<uint64_t>(c->opt_redo_cap, 1) * 4))) return rc;
    L.ps.redo.list = (uint32_t*)L.d_redo.p;
    L.ps.redo.count = (uint32_t*)L.d_qn.p + 2 * (rounds + 2) + 8;  // behind k_path's 8 region cursors; zeroed with them in lane_start
    L.ps.redo.cap = (uint32_t)c->opt_redo_cap;
    L.ps.ro = (float4*)L.d_ro.p;
    L.ps.rd = (float4*)L.d_rd.p;
    L.ps.hit = (float4*)L.d_hit.p;
    L.ps.tp = (float4*)L.d_tp.p;
    L.ps.out = (float4*)L.d_out.p;
    L.ps.acc = (float4*)L.d_acc.p;
    L.ps.pxs = (uint2*)L.d_pxs.p;
    L.cfg = c->cfg;
    L.cfg.spill = (unsigned long long*)L.d_spill.p;
    L.cfg7 = c->cfg7;
    L.cfg7.spill = L.cfg.spill;
    return PT_OK;
}

static void destroy_events(std::vector<hipEvent_t>& v) {
    for (hipEvent_t e : v) (void)hipEventDestroy(e);
    v.clear();
}

static int timed_begin(pt_ctx* c, RenderJob& J, std::vector<hipEvent_t>& list, hipStream_t st) {
    if (!J.timed) return PT_OK;
    hipEvent_t a, b;
    PT_HIP(c, hipEventCreate(&a));
    PT_HIP(c, hipEventCreate(&b));
    list.push_back(a);
    list.push_back(b);
    PT_HIP(c, hipEventRecord(a, st));
    return PT_OK;
}
static int timed_end(pt_ctx* c, RenderJob& J, std::vector<hipEvent_t>& list, hipStream_t st) {
    if (!J.timed) return PT_OK;
    PT_HIP(c, hipEventRecord(list.back(), st));
    return PT_OK;
}

// enqueue rounds [L.round, upto) of the lane's batch, then an async read-back of the next queue length
static int lane_issue_rounds(pt_ctx* c, RenderLane& L, RenderJob& J, uint32_t upto) {
    uint32_t* q[2] = {(uint32_t*)L.d_q0.p, (uint32_t*)L.d_q1.p};
    uint32_t* qn = (uint32_t*)L.d_qn.p;
    unsigned long long* counters = (unsigned long long*)c->d_counters.p;
    int rc;
    for (uint32_t r = L.round; r < upto; ++r) {
        if ((rc = timed_begin(c, J, J.ext_events, L.stream))) return rc;
        launch_extend(c->prim_kind, J.count, c->opt_simple_extend != 0, c->ds, L.ps, q[r & 1], qn + r, qn + J.rounds + 2 + r, L.cfg, counters, L.stream);
        if ((rc = timed_end(c, J, J.ext_events, L.stream))) return rc;
        c->extend_launches += 1;
        launch_shade(c->prim_kind, L.nee, c->ds, L.ps, J.M, q[r & 1], qn + r, q[(r + 1) & 1], qn + r + 1, J.max_depth, L.sample_base,
                     J.k0, J.k1, L.cfg, counters, L.stream);
    }
    L.round = upto;
    PT_HIP(c, hipMemcpyAsync(L.h_qn, qn + upto, 4, hipMemcpyDeviceToHost, L.stream));
    PT_HIP(c, hipMemcpyAsync(L.h_qn + 1, L.ps.redo.count, 4, hipMemcpyDeviceToHost, L.stream));
    PT_HIP(c, hipEventRecord(L.ev_check, L.stream));
    PT_HIP(c, hipGetLastError());
    return PT_OK;
}

static int lane_start(pt_ctx* c, RenderLane& L, RenderJob& J, uint32_t batch, uint32_t ns, uint32_t sample_base) {
    uint32_t* qn = (uint32_t*)L.d_qn.p;
    int rc;
    L.batch = batch;
    L.ns = ns;
    L.sample_base = sample_base;
    L.round = 0;
    L.state = RenderLane::RUNNING;
    L.h_qn[1] = 0;
    PT_HIP(c, hipMemsetAsync(qn, 0, (size_t)(J.rounds + 2) * 8 + 64, L.stream));
    if (c->opt_render_mode == 1) {
        // the whole batch in one persistent kernel (generate + sample, state in registers)
        if ((rc = timed_begin(c, J, J.tail_events, L.stream))) return rc;
        const bool seven = c->opt_waves == 7 || (c->opt_waves == 0 && c->auto_seven);
        launch_path(c->prim_kind, L.nee, J.count, c->ds, (const DevCamera*)c->d_cam.p, J.M, ns, sample_base, J.max_depth, J.k0, J.k1,
                    L.ps.out, qn + 2 * (J.rounds + 2), L.ps.redo, seven ? L.cfg7 : L.cfg, seven, (unsigned long long*)c->d_counters.p, L.stream);
        if ((rc = timed_end(c, J, J.tail_events, L.stream))) return rc;
        c->tail_launches += 1;
        PT_HIP(c, hipMemcpyAsync(L.h_qn + 1, L.ps.redo.count, 4, hipMemcpyDeviceToHost, L.stream));
        PT_HIP(c, hipEventRecord(L.ev_check, L.stream));
        PT_HIP(c, hipGetLastError());
        L.round = J.rounds;
        L.live = 0;
        L.state = RenderLane::TAILING;  // ev_check now stands for the whole batch
        return PT_OK;
    }
    launch_raygen((const DevCamera*)c->d_cam.p, J.M, L.ps, ns, sample_base, J.k0, J.k1, L.nee, (uint32_t*)L.d_q0.p, qn,
                  (unsigned long long*)c->d_counters.p, L.stream);
    if (J.max_depth == 0)  // `for _ in 0..0` never runs: every sample is (0,0,0)
        PT_HIP(c, hipMemsetAsync(L.d_out.p, 0, (size_t)ns * J.M.npix_pad * 16, L.stream));
    // the first rounds carry almost all rays; look at the queue after three of them
    return lane_issue_rounds(c, L, J, std::min<uint32_t>((uint32_t)c->opt_first_check, J.rounds));
}

// No stream ever waits on an event of another stream that is still pending (DESIGN.md section 6 has
// the history): the host polls ev_check / ev_done and only then enqueues the dependent work, so no
// hardware queue ever holds a blocked barrier packet.

// The paths still alive after the rounds go to the batch's tail stream (ev_check has completed).
static int lane_tail(pt_ctx* c, RenderLane& L, RenderJob& J, uint32_t live) {
    hipStream_t ts = c->tail_stream[L.batch % c->opt_tail_streams];
    int rc;
    if (live != 0 && L.round < J.rounds && c->opt_tail_mode == 1) {
        // tail as further wavefront rounds, all enqueued blindly on the tail stream with a grid sized
        // for the few paths left (rounds that find an empty queue cost a few microseconds)
        uint32_t* q[2] = {(uint32_t*)L.d_q0.p, (uint32_t*)L.d_q1.p};
        uint32_t* qn = (uint32_t*)L.d_qn.p;
        unsigned long long* counters = (unsigned long long*)c->d_counters.p;
        LaunchCfg small = L.cfg;
        small.grid = std::min<uint32_t>(L.cfg.grid, std::max<uint32_t>(16, (live + kTraceBlock - 1) / kTraceBlock));
        for (uint32_t r = L.round; r < J.rounds; ++r) {
            if ((rc = timed_begin(c, J, J.ext_events, ts))) return rc;
            launch_extend(c->prim_kind, J.count, c->opt_simple_extend != 0, c->ds, L.ps, q[r & 1], qn + r, qn + J.rounds + 2 + r, small, counters, ts);
            if ((rc = timed_end(c, J, J.ext_events, ts))) return rc;
            c->extend_launches += 1;
            launch_shade(c->prim_kind, L.nee, c->ds, L.ps, J.M, q[r & 1], qn + r, q[(r + 1) & 1], qn + r + 1, J.max_depth, L.sample_base,
                         J.k0, J.k1, small, counters, ts);
        }
        L.round = J.rounds;
    } else if (live != 0 && L.round < J.rounds) {
        uint32_t* qn = (uint32_t*)L.d_qn.p;
        if ((rc = timed_begin(c, J, J.tail_events, ts))) return rc;
        launch_tail(c->prim_kind, L.nee, J.count, (int)c->opt_tail_mode, c->ds, L.ps, (L.round & 1) ? (uint32_t*)L.d_q1.p : (uint32_t*)L.d_q0.p, qn + L.round,
                    qn + 2 * J.rounds + 3, (uint32_t)((live + c->opt_tail_paths_per_lane - 1) / c->opt_tail_paths_per_lane), J.max_depth, J.k0, J.k1, L.cfg, (unsigned long long*)c->d_counters.p, ts);
        if ((rc = timed_end(c, J, J.tail_events, ts))) return rc;
        c->tail_launches += 1;
    }
    PT_HIP(c, hipMemcpyAsync(L.h_qn + 1, L.ps.redo.count, 4, hipMemcpyDeviceToHost, ts));
    PT_HIP(c, hipEventRecord(L.ev_check, ts));
    PT_HIP(c, hipGetLastError());
    L.state = RenderLane::TAILING;
    return PT_OK;
}

// All paths of the batch are finished: add its samples.  Every accumulate runs on the one
// accumulate stream and the host enqueues them in batch order, so every pixel's sum is built in
// sample order (main.rs:159 adds one frame at a time) across batches and across calls.
static int lane_start(pt_ctx* c, RenderLane& L, RenderJob& J, uint32_t batch, uint32_t ns, uint32_t sample_base);
static int lane_alloc(pt_ctx* c, RenderLane& L, uint64_t slots, uint32_t rounds);

// On return the lane is DRAINING (accumulate enqueued) — or running its batch a second time (see below).
static int lane_accumulate(pt_ctx* c, RenderLane& L, RenderJob& J) {
    // Paths the kernels without the NEE code gave up (shade_core, SHADE_REDO) are finished first, on the accumulate stream
    // right in front of the batch's accumulate.  Everything the batch launched has completed (the host saw its event),
    // so the count read back behind the last kernel is final.
    const uint32_t n_redo = L.nee ? 0u : L.h_qn[1];
    if (n_redo > L.ps.redo.cap) {
        // more than the list holds: the batch runs again from the start with the NEE kernels, which never give a path up
        // (the frame is the same; the ray counters then include the abandoned attempt)
        c->redo_restarts += 1;
        L.nee = true;
        int rc = lane_alloc(c, L, J.batch_slots, J.rounds);
        if (rc) return rc;
        return lane_start(c, L, J, L.batch, L.ns, L.sample_base);
    }
    if (n_redo) {
        launch_redo(c->prim_kind, J.count, c->ds, (const DevCamera*)c->d_cam.p, J.M, L.ns, c->opt_render_mode == 1, L.sample_base, J.max_depth, J.k0,
                    J.k1, L.ps.out, L.ps.redo.list, n_redo, L.cfg, (unsigned long long*)c->d_counters.p, c->acc_stream);
        c->redo_paths += n_redo;
    }
    launch_accumulate(J.M, L.ps.out, L.ns, c->opt_render_mode == 1, (float*)c->d_accum.p, c->acc_stream);
    PT_HIP(c, hipEventRecord(L.ev_done, c->acc_stream));
    PT_HIP(c, hipGetLastError());
    c->spp_added += (float)L.ns;
    c->samples_added_end = L.sample_base + L.ns;
    L.state = RenderLane::DRAINING;
    return PT_OK;
}

// total length of the union of [begin, end) intervals
static double union_ms(std::vector<std::pair<double, double>> iv) {
    std::sort(iv.begin(), iv.end());
    double total = 0;
    double lo = 0, hi = 0;
    bool open = false;
    for (const auto& p : iv) {
        if (open && p.first <= hi) {
            hi = std::max(hi, p.second);
        } else {
            if (open) total += hi - lo;
            lo = p.first;
            hi = p.second;
            open = true;
        }
    }
    if (open) total += hi - lo;
    return total;
}

static void fold_intervals(pt_sched& S) {
    S.ext_union_done += union_ms(S.ext_iv);
    S.tail_union_done += union_ms(S.tail_iv);
    S.ext_iv.clear();
    S.tail_iv.clear();
    if (S.ev_epoch) {
        (void)hipEventDestroy(S.ev_epoch);
        S.ev_epoch = nullptr;
    }
}

// a finished job leaves the queue: its timed intervals are kept relative to the context's epoch event
static int retire_job(pt_ctx* c, RenderJob& J) {
    pt_sched& S = *c->sched;
    c->render_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - J.t0).count();
    for (int which = 0; which < 2; ++which) {
        std::vector<hipEvent_t>& ev = which ? J.tail_events : J.ext_events;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float b = 0.f, e = 0.f, d = 0.f;
            PT_HIP(c, hipEventElapsedTime(&b, S.ev_epoch, ev[i]));
            PT_HIP(c, hipEventElapsedTime(&e, S.ev_epoch, ev[i + 1]));
            PT_HIP(c, hipEventElapsedTime(&d, ev[i], ev[i + 1]));
            (which ? S.tail_iv : S.ext_iv).push_back({(double)b, (double)std::max(e, b)});
            (which ? S.tail_sum : S.ext_sum) += d;
        }
        destroy_events(ev);
    }
    return PT_OK;
}

// Something failed while jobs were in flight: let the device finish whatever is queued, drop all
// jobs and put every lane back to IDLE, so that the next call starts from a clean state.
static int abort_jobs(pt_ctx* c, int rc) {
    pt_sched& S = *c->sched;
    const std::string keep = c->err;
    (void)hipDeviceSynchronize();
    for (RenderJob& J : S.jobs) {
        destroy_events(J.ext_events);
        destroy_events(J.tail_events);
    }
    S.jobs.clear();
    S.acc_seq = S.next_seq;
    c->spp = c->spp_added;  // the frame holds what was added before the failure ...
    c->samples_done = c->samples_added_end;  // ... and the next shard-less call goes on behind its last sample, not behind the dropped ones
    for (RenderLane& L : c->lanes) {
        L.state = RenderLane::IDLE;
        L.job = nullptr;
    }
    c->err = keep;
    return rc;
}

// One pass of the scheduler over all lanes; never blocks.  Batches of all queued calls share the
// lanes, so the end of one call's last kernel (a few long paths in nearly empty waves) overlaps the
// start of the next call's.
static int pump_once(pt_ctx* c, bool& progressed) {
    pt_sched& S = *c->sched;
    uint32_t busy = 0;
    for (const RenderLane& L : c->lanes) busy += L.state != RenderLane::IDLE;
    for (uint32_t i = 0; i < kMaxLanes; ++i) {
        RenderLane& L = c->lanes[i];
        if (L.state == RenderLane::IDLE) {
            RenderJob* J = nullptr;
            for (RenderJob& cand : S.jobs)
                if (cand.next_batch < cand.n_batches) { J = &cand; break; }
            if (!J || busy >= J->max_lanes) continue;
            L.nee = c->nee;
            int rc = lane_alloc(c, L, J->batch_slots, J->rounds);
            if (rc) return rc;
            hipStream_t& bs = c->bulk_stream[i % c->opt_bulk_streams];
            if (!bs) PT_HIP(c, hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
            L.stream = bs;
            L.nee = c->nee;
            L.job = J;
            L.seq = J->seq0 + J->next_batch;
            const uint64_t done = (uint64_t)J->next_batch * J->per_batch;
            const uint32_t ns = (uint32_t)std::min<uint64_t>(J->per_batch, J->spp - done);
            rc = lane_start(c, L, *J, J->next_batch, ns, J->sample_begin + (uint32_t)done);
            if (rc) return rc;
            J->next_batch += 1;
            busy += 1;
            progressed = true;
            continue;
        }
        RenderJob& J = *L.job;
        if (L.state == RenderLane::RUNNING) {
            hipError_t q = hipEventQuery(L.ev_check);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_render: ") + hipGetErrorString(q));
            progressed = true;
            const uint32_t live = *L.h_qn;
            int rc = PT_OK;
            if (live == 0 || L.round >= J.rounds) {
                L.state = RenderLane::READY;
            } else if (live <= J.tail_threshold) {
                rc = lane_tail(c, L, J, live);
            } else {
                rc = lane_issue_rounds(c, L, J, std::min<uint32_t>(L.round + 2, J.rounds));
            }
            if (rc) return rc;
        } else if (L.state == RenderLane::TAILING) {
            hipError_t q = hipEventQuery(L.ev_check);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_render: ") + hipGetErrorString(q));
            L.state = RenderLane::READY;
            progressed = true;
        }
        if (L.state == RenderLane::READY && L.seq == S.acc_seq) {
            int rc = lane_accumulate(c, L, J);
            if (rc) return rc;
            if (L.state == RenderLane::DRAINING) S.acc_seq += 1;  // (otherwise the batch runs again with the NEE kernels)
            progressed = true;
        } else if (L.state == RenderLane::DRAINING) {
            hipError_t q = hipEventQuery(L.ev_done);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return c->fail(PT_ERR_HIP, std::string("pt_render: ") + hipGetErrorString(q));
            if (J.count) {  // diagnostics: queue length per round of the last finished batch
                c->round_sizes.assign(J.rounds + 1, 0);
                PT_HIP(c, hipMemcpy(c->round_sizes.data(), L.d_qn.p, (size_t)(J.rounds + 1) * 4, hipMemcpyDeviceToHost));
            }
            L.state = RenderLane::IDLE;
            L.job = nullptr;
            J.finished += 1;
            busy -= 1;
            progressed = true;
        }
    }
    while (!S.jobs.empty() && S.jobs.front().finished == S.jobs.front().n_batches) {
        int rc = retire_job(c, S.jobs.front());
        if (rc) return rc;
        S.jobs.pop_front();
        progressed = true;
        if (S.jobs.empty()) fold_intervals(S);  // nothing in flight: no launch to come can overlap what has been timed
    }
    return PT_OK;
}

// Run the scheduler until every submitted call has finished.  The host thread only enqueues and
// polls; between polls it yields at first and then sleeps (5 us doubling up to 100 us), so a long
// kernel does not cost a spinning core.
static int drain(pt_ctx* c) {
    pt_sched& S = *c->sched;
    if (S.jobs.empty()) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    const bool dbg = std::getenv("PTCORE_DEBUG") != nullptr;
    uint32_t idle_polls = 0;
    auto last_report = std::chrono::steady_clock::now();
    while (!S.jobs.empty()) {
        bool progressed = false;
        int rc = pump_once(c, progressed);
        if (rc) return abort_jobs(c, rc);
        if (progressed) {
            idle_polls = 0;
            continue;
        }
        idle_polls += 1;
        if (idle_polls < 64) {
            std::this_thread::yield();
        } else {
            const uint32_t step = std::min<uint32_t>((idle_polls - 64) / 16, 5);  // 5, 10, 20, 40, 80, 100 us
            std::this_thread::sleep_for(std::chrono::microseconds(std::min<uint32_t>(5u << step, 100u)));
        }
        if (dbg && std::chrono::steady_clock::now() - last_report > std::chrono::seconds(2)) {
            last_report = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[ptcore] waiting: %zu calls queued, next batch to add %llu, lanes", S.jobs.size(), (unsigned long long)S.acc_seq);
            for (const RenderLane& L : c->lanes) std::fprintf(stderr, " %d(#%llu)", (int)L.state, (unsigned long long)L.seq);
            std::fprintf(stderr, "\n");
        }
    }
    return PT_OK;
}

// Cut a call's J.spp samples into batches of whole samples and size what depends on the batch.
static int plan_batches(pt_ctx* c, RenderJob& J) {
    const PixelMap& M = J.M;
    const uint64_t want_slots = c->opt_batch_slots ? c->opt_batch_slots : (c->opt_render_mode == 1 ? kPathBatchSlots : kTargetSlots);
    uint64_t per_batch = std::max<uint64_t>(1, want_slots / M.npix_pad);
    per_batch = std::min<uint64_t>(per_batch, J.spp);
    if (per_batch * M.npix_pad >= kMaxSlots) return c->fail(PT_ERR_LIMIT, "pt_render: batch too large");
    J.per_batch = per_batch;
    J.batch_slots = per_batch * M.npix_pad;
    J.n_batches = (uint32_t)((J.spp + per_batch - 1) / per_batch);
    // path-kernel mode: launches in flight cover each other's ends (a launch ends with a few long paths in nearly empty
    // waves).  Small batches (few samples per call) are mostly end, so up to eight of them overlap (C3: 1 spp per call
    // 0.76 -> 1.25 Grays/s, 8 spp 4.2 -> 6.6, 64 spp 10.3 -> 10.8 with 2 -> 8 in flight); every launch in flight holds a
    // radiance slab of 16 B per path, so large batches stay at two (more measured no faster there): the slabs of all
    // lanes together are kept under 24 GiB.  The wavefront rounds need all eight.
    uint64_t want_lanes = c->opt_lanes;
    if (!want_lanes) want_lanes = c->opt_render_mode == 1 ? std::min<uint64_t>(kMaxLanes, std::max<uint64_t>(2, (24ull << 30) / (J.batch_slots * 16))) : kMaxLanes;
    J.max_lanes = (uint32_t)want_lanes;
    J.tail_threshold = c->opt_tail_fraction == 0 ? 0 : (uint32_t)std::max<uint64_t>(c->opt_tail_min, J.batch_slots / c->opt_tail_fraction);
    return PT_OK;
}

// A call that continues the newest queued call (same frame parameters, the next sample indices) while that one has
// not started a batch yet — every lane is busy with earlier calls — is added to it: a host that submits 1 sample per
// call faster than the GPU renders them then gets launches of many samples (which run an order of magnitude more
// rays per second than launches of one) instead of a growing queue of small ones.  Results do not
// change: a sample is a function of (pixel, global sample index, seed) and sums are built in sample order either way.
static bool same_frame(const RenderJob& a, const RenderJob& b) {
    return a.M.width == b.M.width && a.M.height == b.M.height && a.M.tile == b.M.tile && a.M.rank == b.M.rank && a.M.nranks == b.M.nranks &&
           a.max_depth == b.max_depth && a.k0 == b.k0 && a.k1 == b.k1 && a.timed == b.timed && a.count == b.count;
}

int pt_render_async(pt_ctx* c, uint32_t spp, uint32_t max_depth, uint64_t seed, const pt_shard* shard, uint32_t flags) {
    if (!c) return PT_ERR_INVALID;
    if (!c->built) return c->fail(PT_ERR_STATE, "pt_render: scene changed since pt_build (or never built): call pt_build first");
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_render: call pt_set_camera first");
    if (!c->ds.sky) return c->fail(PT_ERR_STATE, "pt_render: call pt_set_sky first");
    if (max_depth > kMaxDepth) return c->fail(PT_ERR_LIMIT, "pt_render: max_depth above 65535");
    if (spp == 0) return PT_OK;
    PT_HIP(c, hipSetDevice(c->device));
    pt_sched& S = *c->sched;

    RenderJob J{};
    PixelMap& M = J.M;
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
    {   // the tile-padded pixel count must fit the 32-bit slot arithmetic of the kernels
        const uint64_t tx = ((uint64_t)M.width + M.tile - 1) / M.tile, ty = ((uint64_t)M.height + M.tile - 1) / M.tile;
        if (tx * ty * M.tile * M.tile >= kMaxSlots) return c->fail(PT_ERR_LIMIT, "pt_render: tile-padded image too large");
    }
    pixelmap_init(M, M.width, M.height, M.tile, M.rank, M.nranks);
    if (M.npix_pad == 0) {  // this rank owns no tile
        c->spp += (float)spp;
        c->spp_added += (float)spp;
        c->samples_done = c->samples_added_end = sample_begin + spp;
        return PT_OK;
    }

    J.spp = spp;
    J.sample_begin = sample_begin;
    J.max_depth = max_depth;
    J.rounds = max_depth == 0 ? 0 : 2 * max_depth;  // each iteration traces at most twice
    J.k0 = (uint32_t)seed;
    J.k1 = (uint32_t)(seed >> 32);
    J.timed = (flags & PT_RENDER_TIMED) != 0;
    J.count = (flags & PT_RENDER_COUNT) != 0;
    if (c->opt_merge_calls && !S.jobs.empty()) {
        RenderJob& B = S.jobs.back();
        if (B.next_batch == 0 && same_frame(B, J) && (uint64_t)B.sample_begin + B.spp == sample_begin && ((uint64_t)B.spp + spp) * B.M.npix_pad <= c->opt_merge_calls) {
            RenderJob merged = B;  // planned on a copy: a failure leaves the queued call as it was
            merged.spp = B.spp + spp;
            if (int rc = plan_batches(c, merged)) return rc;
            S.next_seq = B.seq0 + merged.n_batches;  // B is the newest call, so its batches are the last in the order
            B = std::move(merged);
            c->spp += (float)spp;
            c->samples_done = sample_begin + spp;
            c->merged_calls += 1;
            bool progressed = false;
            int rc = pump_once(c, progressed);
            if (rc) return abort_jobs(c, rc);
            return PT_OK;
        }
    }
    if (int rc = plan_batches(c, J)) return rc;
    J.seq0 = S.next_seq;
    J.t0 = std::chrono::steady_clock::now();
    if (J.timed && !S.ev_epoch) {
        PT_HIP(c, hipEventCreate(&S.ev_epoch));
        PT_HIP(c, hipEventRecord(S.ev_epoch, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
    }
    S.next_seq += J.n_batches;
    S.jobs.push_back(std::move(J));
    c->spp += (float)spp;
    c->samples_done = sample_begin + spp;
    bool progressed = false;
    int rc = pump_once(c, progressed);
    if (rc) return abort_jobs(c, rc);
    return PT_OK;
}

int pt_wait(pt_ctx* c) {
    if (!c) return PT_ERR_INVALID;
    return drain(c);
}

int pt_render(pt_ctx* c, uint32_t spp, uint32_t max_depth, uint64_t seed, const pt_shard* shard, uint32_t flags) {
    int rc = pt_render_async(c, spp, max_depth, seed, shard, flags);
    if (rc) return rc;
    return drain(c);
}

int pt_read_accum(pt_ctx* c, float* rgb, float* spp_out) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_read_accum: no camera");
    if (int rc = drain(c)) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    if (rgb) PT_HIP(c, hipMemcpy(rgb, c->d_accum.p, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (spp_out) *spp_out = c->spp;
    return PT_OK;
}

int pt_write_accum(pt_ctx* c, const float* rgb, float spp) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_write_accum: no camera");
    if (!rgb || !(spp >= 0.0f) || spp > 4294967040.0f) return c->fail(PT_ERR_INVALID, "pt_write_accum: null image or bad sample count");
    if (int rc = drain(c)) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemcpy(c->d_accum.p, rgb, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), hipMemcpyHostToDevice));
    c->spp = c->spp_added = spp;
    c->samples_done = c->samples_added_end = (uint32_t)spp;
    return PT_OK;
}

// vec_to_rgb(accum / spp) on the accumulate stream, behind every k_accumulate enqueued so far, into caller memory
static int resolve_to_host(pt_ctx* c, uint8_t* rgb, uint64_t pitch, float spp, const char* who) {
    PT_HIP(c, hipSetDevice(c->device));
    int rc = dev_alloc(c, c->d_rgb, (size_t)pitch * c->cam.height);
    if (rc) return rc;
    const float scale = 1.0f / spp;  // main.rs:145
    launch_resolve((const float*)c->d_accum.p, c->cam.width, c->cam.height, scale, (uint8_t*)c->d_rgb.p, pitch, c->acc_stream);
    hipError_t e = hipGetLastError();
    // only the 3 * width payload bytes of each row are written; the caller's padding is left alone
    // Through pinned staging memory: a copy straight into the caller's pageable rows (hipMemcpy2DAsync) measured 7.4 ms
    // for a 1080p frame, 0.8 GB/s.  Only the 3 * width payload bytes of each row reach the caller; its padding is left alone.
    const size_t bytes = (size_t)pitch * c->cam.height;
    if (e == hipSuccess && c->h_rgb_bytes < bytes) {
        if (c->h_rgb) (void)hipHostFree(c->h_rgb);
        c->h_rgb = nullptr;
        c->h_rgb_bytes = 0;
        e = hipHostMalloc((void**)&c->h_rgb, bytes, hipHostMallocDefault);
        if (e == hipSuccess) c->h_rgb_bytes = bytes;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_rgb, c->d_rgb.p, bytes, hipMemcpyDeviceToHost, c->acc_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->acc_stream);
    if (e != hipSuccess) return c->fail(PT_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    const size_t row = (size_t)c->cam.width * 3;
    if (pitch == row) {
        std::memcpy(rgb, c->h_rgb, bytes);
    } else {
        for (uint32_t y = 0; y < c->cam.height; ++y) std::memcpy(rgb + (size_t)y * pitch, c->h_rgb + (size_t)y * pitch, row);
    }
    return PT_OK;
}

int pt_read_rgb8(pt_ctx* c, uint8_t* rgb, uint64_t pitch) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_read_rgb8: no camera");
    if (!rgb || pitch < (uint64_t)c->cam.width * 3) return c->fail(PT_ERR_INVALID, "pt_read_rgb8: null pointer or pitch < 3 * width");
    if (int rc = drain(c)) return rc;
    return resolve_to_host(c, rgb, pitch, c->spp, "pt_read_rgb8");
}

int pt_read_rgb8_latest(pt_ctx* c, uint8_t* rgb, uint64_t pitch, float* spp_out) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_read_rgb8_latest: no camera");
    if (!rgb || pitch < (uint64_t)c->cam.width * 3) return c->fail(PT_ERR_INVALID, "pt_read_rgb8_latest: null pointer or pitch < 3 * width");
    if (!c->sched->jobs.empty()) {  // let finished launches be added, start queued ones; never waits for a path kernel
        PT_HIP(c, hipSetDevice(c->device));
        bool progressed = false;
        int rc = pump_once(c, progressed);
        if (rc) return abort_jobs(c, rc);
    }
    if (spp_out) *spp_out = c->spp_added;
    if (c->spp_added == 0.f) {  // nothing has been added yet (the first launch is still running): a black frame, not 0 / 0
        for (uint32_t y = 0; y < c->cam.height; ++y) std::memset(rgb + (size_t)y * pitch, 0, (size_t)c->cam.width * 3);
        return PT_OK;
    }
    return resolve_to_host(c, rgb, pitch, c->spp_added, "pt_read_rgb8_latest");
}

int pt_accum_device_ptr(pt_ctx* c, void** ptr, uint64_t* n_floats) {
    if (!c || !ptr) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, "pt_accum_device_ptr: no camera");
    if (int rc = drain(c)) return rc;  // the caller is about to read or reduce the sums
    *ptr = c->d_accum.p;
    if (n_floats) *n_floats = (uint64_t)c->cam.width * c->cam.height * 3;
    return PT_OK;
}

int pt_add_external_samples(pt_ctx* c, float spp) {
    if (!c) return PT_ERR_INVALID;
    if (int rc = drain(c)) return rc;
    c->spp += spp;
    c->spp_added += spp;
    return PT_OK;
}

int pt_get_counters(pt_ctx* c, pt_counters* out) {
    if (!c || !out) return PT_ERR_INVALID;
    if (int rc = drain(c)) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    unsigned long long cur[CNT_N];
    PT_HIP(c, hipMemcpy(cur, c->d_counters.p, sizeof(cur), hipMemcpyDeviceToHost));
    if (std::getenv("PTCORE_PHASE_STATS")) {  // filled by -DPT_PHASE_STATS builds only
        static const char* names[6] = {"shade", "generate", "internal", "leaf", "drained", "lifetime"};
        std::fprintf(stderr, "[ptcore] drained mode: most iterations of one wave %llu, longest drained time of one wave %llu clocks\n",
                     cur[CNT_MAX_NODE_TESTS], cur[CNT_NODE_TESTS]);
        for (int i = 0; i < 6; ++i)
            std::fprintf(stderr, "[ptcore] phase %-16s exec %14llu lanes %16llu clocks %16llu\n", names[i], cur[CNT_PHASE + 3 * i],
                         cur[CNT_PHASE + 3 * i + 1], cur[CNT_PHASE + 3 * i + 2]);
    }
    out->paths = cur[CNT_PATHS];
    out->rays_closest = cur[CNT_RAYS_CLOSEST];
    out->rays_any = cur[CNT_RAYS_ANY];
    out->node_tests = cur[CNT_NODE_TESTS] + cur[CNT_TAIL_NODE_TESTS];
    out->prim_tests = cur[CNT_PRIM_TESTS] + cur[CNT_TAIL_PRIM_TESTS];
    out->tail_rays = cur[CNT_TAIL_RAYS];
    out->tail_node_tests = cur[CNT_TAIL_NODE_TESTS];
    out->tail_prim_tests = cur[CNT_TAIL_PRIM_TESTS];
    out->tail_launches = c->tail_launches;
    // timed launches may overlap (several batches or calls in flight): *_ms is the time during which at least one
    // such launch was running, *_ms_sum the plain sum of their durations (what a profiler's average x count gives)
    out->tail_ms = c->sched->tail_union_done + union_ms(c->sched->tail_iv);
    out->extend_launches = c->extend_launches;
    out->extend_ms = c->sched->ext_union_done + union_ms(c->sched->ext_iv);
    out->render_ms = c->render_ms;
    out->tail_ms_sum = c->sched->tail_sum;
    out->extend_ms_sum = c->sched->ext_sum;
    out->redo_paths = c->redo_paths;
    out->redo_restarts = c->redo_restarts;
    return PT_OK;
}

int pt_get_round_stats(pt_ctx* c, uint32_t* queue_sizes, uint32_t cap, uint32_t* n_rounds, uint64_t* max_node_tests) {
    if (!c) return PT_ERR_INVALID;
    if (int rc = drain(c)) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    const uint32_t n = (uint32_t)c->round_sizes.size();
    if (n_rounds) *n_rounds = n;
    if (queue_sizes)
        for (uint32_t i = 0; i < n && i < cap; ++i) queue_sizes[i] = c->round_sizes[i];
    if (max_node_tests) {
        unsigned long long cur[CNT_N];
        PT_HIP(c, hipMemcpy(cur, c->d_counters.p, sizeof(cur), hipMemcpyDeviceToHost));
        *max_node_tests = cur[CNT_MAX_NODE_TESTS];
    }
    return PT_OK;
}

int pt_set_option(pt_ctx* c, int option, uint64_t value) {
    if (!c) return PT_ERR_INVALID;
    if (int rc = drain(c)) return rc;  // lanes in flight were set up for the old options
    switch (option) {
        case PT_OPT_BATCH_PATHS: c->opt_batch_slots = value ? std::max<uint64_t>(value, 64) : 0; return PT_OK;
        case PT_OPT_LANES: c->opt_lanes = std::min<uint64_t>(value, kMaxLanes); return PT_OK;
        case PT_OPT_TAIL_MIN_PATHS: c->opt_tail_min = value; return PT_OK;
        case PT_OPT_TAIL_FRACTION: c->opt_tail_fraction = value; return PT_OK;
        case PT_OPT_TAIL_STREAMS: c->opt_tail_streams = std::min<uint64_t>(std::max<uint64_t>(value, 1), 4); return PT_OK;
        case PT_OPT_BULK_STREAMS: c->opt_bulk_streams = std::min<uint64_t>(std::max<uint64_t>(value, 1), kMaxLanes); return PT_OK;
        case PT_OPT_FIRST_CHECK_ROUND: c->opt_first_check = value; return PT_OK;
        case PT_OPT_RENDER_MODE: c->opt_render_mode = value; return PT_OK;
        case PT_OPT_TAIL_MODE: c->opt_tail_mode = value; return PT_OK;
        case PT_OPT_SIMPLE_EXTEND: c->opt_simple_extend = value; return PT_OK;
        case PT_OPT_DEVICE_BUILD: c->opt_device_build = value != 0; return PT_OK;
        case PT_OPT_MERGE_QUEUED_CALLS: c->opt_merge_calls = std::min<uint64_t>(value, kMaxSlots - 1); return PT_OK;
        case PT_OPT_TAIL_PATHS_PER_LANE: c->opt_tail_paths_per_lane = std::max<uint64_t>(value, 1); return PT_OK;
        case PT_OPT_REDO_CAPACITY: c->opt_redo_cap = std::min<uint64_t>(value, 1u << 28); return PT_OK;
        case PT_OPT_SKIP_REPEATED_MISS: c->ds.skip_repeated_miss = value ? 1u : 0u; return PT_OK;
        case PT_OPT_WAVES_PER_SIMD:
            if (value != 0 && value != 6 && value != 7) return c->fail(PT_ERR_INVALID, "pt_set_option: PT_OPT_WAVES_PER_SIMD takes 0 (by scene size), 6 or 7");
            c->opt_waves = value;
            return PT_OK;
        default: return c->fail(PT_ERR_INVALID, "pt_set_option: unknown option");
    }
}

int pt_shard_map(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t nranks, uint32_t* xy, uint64_t cap,
                 uint64_t* n_out) {
    if (width == 0 || height == 0 || tile == 0 || tile % 8 != 0 || tile > 1024 || nranks == 0 || rank >= nranks) return PT_ERR_INVALID;
    PixelMap M{};
    pixelmap_init(M, width, height, tile, rank, nranks);
    uint64_t n = 0;
    for (uint32_t idx = 0; idx < M.npix_pad; ++idx) {
        uint32_t x, y;
        if (!map_pixel(M, idx, &x, &y)) continue;
        if (xy && n < cap) {
            xy[2 * n] = x;
            xy[2 * n + 1] = y;
        }
        n += 1;
    }
    if (n_out) *n_out = n;
    return PT_OK;
}

static bool tile_shard_ok(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t nranks) {
    return width != 0 && height != 0 && tile != 0 && tile % 8 == 0 && tile <= 1024 && nranks != 0 && rank < nranks;
}

int pt_shard_slab_floats(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t nranks, uint64_t* n_floats) {
    if (!n_floats || !tile_shard_ok(width, height, tile, rank, nranks)) return PT_ERR_INVALID;
    PixelMap M{};
    pixelmap_init(M, width, height, tile, rank, nranks);
    *n_floats = (uint64_t)M.npix_pad * 3;
    return PT_OK;
}

static int shard_copy(pt_ctx* c, const pt_shard* shard, float* slab, uint64_t n_floats, bool unpack, const char* who) {
    if (!c) return PT_ERR_INVALID;
    if (!c->have_cam) return c->fail(PT_ERR_STATE, std::string(who) + ": no camera");
    if (!shard || shard->kind != PT_SHARD_TILES || !tile_shard_ok(c->cam.width, c->cam.height, shard->tile, shard->rank, shard->nranks))
        return c->fail(PT_ERR_INVALID, std::string(who) + ": needs a valid PT_SHARD_TILES shard");
    if (int rc = drain(c)) return rc;
    PixelMap M{};
    pixelmap_init(M, c->cam.width, c->cam.height, shard->tile, shard->rank, shard->nranks);
    if (M.npix_pad == 0) return PT_OK;  // this rank owns no tile
    if (!slab || n_floats < (uint64_t)M.npix_pad * 3) return c->fail(PT_ERR_INVALID, std::string(who) + ": slab too small (see pt_shard_slab_floats)");
    PT_HIP(c, hipSetDevice(c->device));
    launch_shard_copy(M, (float*)c->d_accum.p, slab, unpack, c->stream);
    PT_HIP(c, hipGetLastError());
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_shard_pack(pt_ctx* c, const pt_shard* shard, float* slab_dev, uint64_t cap_floats) {
    return shard_copy(c, shard, slab_dev, cap_floats, false, "pt_shard_pack");
}

int pt_shard_unpack(pt_ctx* c, const pt_shard* shard, const float* slab_dev, uint64_t n_floats) {
    return shard_copy(c, shard, const_cast<float*>(slab_dev), n_floats, true, "pt_shard_unpack");
}

int pt_reset_counters(pt_ctx* c) {
    if (!c) return PT_ERR_INVALID;
    if (int rc = drain(c)) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipMemsetAsync(c->d_counters.p, 0, CNT_N * sizeof(unsigned long long), c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));  // the launches that count run on other streams
    c->extend_launches = 0;
    c->tail_launches = 0;
    c->render_ms = 0;
    c->redo_paths = c->redo_restarts = 0;
    pt_sched& S = *c->sched;
    S.ext_iv.clear();
    S.tail_iv.clear();
    S.ext_sum = S.tail_sum = 0;
    S.ext_union_done = S.tail_union_done = 0;
    if (S.ev_epoch) {
        (void)hipEventDestroy(S.ev_epoch);
        S.ev_epoch = nullptr;
    }
    return PT_OK;
}

}  // extern "C"
