// multi_gpu.hpp — one frame over the GPUs of one node, from the native host, combined with RCCL.
//
// What it replaces: the reference fans a frame out over num_cpus threads that share one accumulator
// (scoped_threadpool, src/main.rs:113-115,149-170: static row bands, disjoint chunks_mut slices).  Here the
// fan-out is over GPUs: ONE process, one host thread + one pt_ctx per device (the scene, BVH and sky are
// replicated), no exchange while rendering, ONE collective when the frame is read out:
//
//   ShardMode::Samples  rank r renders its own global sample indices of the whole frame (frame f of the
//                       loop: samples [(f*N + r)*spp, +spp)); ncclReduce(sum, f32, W*H*3) IN PLACE on the
//                       library's accumulator (pt_accum_device_ptr) to rank 0, then
//                       pt_add_external_samples.  Equal to the 1-GPU frame of N*frames*spp samples up to
//                       fp32 summation order.
//   ShardMode::Tiles    rank r renders the 64x64 tiles t with t % N == r, packs them into one compact slab
//                       (pt_shard_pack), ONE ncclGather of N equal slabs to rank 0, which un-swizzles the
//                       other ranks' slabs into its frame (pt_shard_unpack).  Pixel sets are disjoint:
//                       bit-identical to the 1-GPU frame.
//
// RCCL is linked into this HOST program, not into libptcore.so (the library only exposes the device
// pointers and the slab kernels).  Transport::HostStaged moves the same buffers through host memory instead
// of RCCL: RCCL cannot put two ranks of one communicator on one device, so this is how the N > 1 logic
// (sample indices, slab order, un-swizzle, sample count) is rehearsed on a one-GPU box.  It is a
// functional rehearsal, never a measurement.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <exception>
#include <memory>
#include <functional>
#include <mutex>
#include <thread>

#include "pathtracer.hpp"

namespace pt {

enum class ShardMode { Samples, Tiles };
enum class Transport { Rccl, HostStaged };

struct MultiGpuPlan {
    std::vector<int> devices;  // rank r runs on devices[r]
    ShardMode mode = ShardMode::Samples;
    Transport transport = Transport::Rccl;
    uint32_t tile = 64;        // fixed 64x64 tiles dealt round-robin
};

struct RankReport {
    int device = -1;
    uint64_t paths = 0, rays = 0;
    double render_ms = 0;   // this rank's own rendering, submission to completion
    double combine_ms = 0;  // pack + collective + un-swizzle as seen by this rank
};

namespace detail {

// a reusable barrier for the rank threads (C++17: no std::barrier)
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const uint64_t gen = gen_;
        if (++arrived_ == n_) {
            arrived_ = 0;
            gen_ += 1;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen_ != gen; });
        }
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, arrived_ = 0;
    uint64_t gen_ = 0;
};

inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(PT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
inline void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Error(PT_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(r));
}

struct DeviceBuffer {
    float* p = nullptr;
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
    void alloc_zero(size_t n_floats) {
        hip_check(hipMalloc((void**)&p, n_floats * sizeof(float)), "hipMalloc(slab)");
        hip_check(hipMemset(p, 0, n_floats * sizeof(float)), "hipMemset(slab)");
        // the fill runs on the null stream and need not have finished when hipMemset returns; the library's streams do not
        // wait for the null stream (hipStreamNonBlocking), so a pack kernel could otherwise be overwritten by it
        hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize(null)");
    }
};

}  // namespace detail

// The communicator clique of the ranks (RCCL transport) or nothing (host-staged rehearsal).
class Clique {
public:
    explicit Clique(const MultiGpuPlan& plan) : plan_(plan), barrier_((int)plan.devices.size()) {
        const int n = (int)plan.devices.size();
        if (plan.transport == Transport::Rccl) {
            comms_.resize(n);
            detail::nccl_check(ncclCommInitAll(comms_.data(), n, plan.devices.data()), "ncclCommInitAll");
        } else {
            staged_.resize(n);
        }
    }
    ~Clique() {
        for (ncclComm_t c : comms_)
            if (c) (void)ncclCommDestroy(c);
    }
    Clique(const Clique&) = delete;
    Clique& operator=(const Clique&) = delete;

    int size() const { return (int)plan_.devices.size(); }
    void barrier() { barrier_.wait(); }

    // sum the n floats at `data` (device memory of rank `rank`) over all ranks into rank 0's, in place
    void reduce_to_root(int rank, float* data, size_t n, hipStream_t stream) {
        if (plan_.transport == Transport::Rccl) {
            detail::nccl_check(ncclReduce(data, data, n, ncclFloat, ncclSum, 0, comms_[rank], stream), "ncclReduce");
            detail::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(reduce)");
            return;
        }
        // a rank whose copy fails still reaches both barriers (the others must not wait for it for ever)
        std::exception_ptr bad;
        try {
            staged_[rank].resize(n);
            detail::hip_check(hipMemcpy(staged_[rank].data(), data, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
        } catch (...) { bad = std::current_exception(); }
        barrier();
        if (rank == 0 && !bad) {
            try {
                std::vector<float>& sum = staged_[0];
                for (int r = 1; r < size(); ++r) {
                    if (staged_[r].size() != n) throw Error(PT_ERR_HIP, "a rank did not deliver its accumulator");
                    for (size_t i = 0; i < n; ++i) sum[i] = sum[i] + staged_[r][i];  // rank order
                }
                detail::hip_check(hipMemcpy(data, sum.data(), n * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(H2D)");
            } catch (...) { bad = std::current_exception(); }
        }
        barrier();
        if (bad) std::rethrow_exception(bad);
    }

    // every rank's n floats at `send` end up at recv + r * n on rank 0 (recv may be null elsewhere)
    void gather_to_root(int rank, const float* send, float* recv, size_t n, hipStream_t stream) {
        if (plan_.transport == Transport::Rccl) {
            detail::nccl_check(ncclGather(send, recv, n, ncclFloat, 0, comms_[rank], stream), "ncclGather");
            detail::hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize(gather)");
            return;
        }
        std::exception_ptr bad;
        try {
            staged_[rank].resize(n);
            detail::hip_check(hipMemcpy(staged_[rank].data(), send, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
        } catch (...) { bad = std::current_exception(); }
        barrier();
        if (rank == 0 && !bad) {
            try {
                for (int r = 0; r < size(); ++r) {
                    if (staged_[r].size() != n) throw Error(PT_ERR_HIP, "a rank did not deliver its slab");
                    detail::hip_check(hipMemcpy(recv + (size_t)r * n, staged_[r].data(), n * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(H2D)");
                }
            } catch (...) { bad = std::current_exception(); }
        }
        barrier();
        if (bad) std::rethrow_exception(bad);
    }

private:
    MultiGpuPlan plan_;
    detail::Barrier barrier_;
    std::vector<ncclComm_t> comms_;
    std::vector<std::vector<float>> staged_;
};

// What one rank does between "scene resident" and "frame combined on rank 0".
// render: queues `frames` calls of `spp` samples with the rank's shard and waits for them.
template <class T>
void render_shard(Camera<T>& camera, const MultiGpuPlan& plan, int rank, int frames, uint32_t spp, uint32_t max_depth, uint64_t seed) {
    const uint32_t n = (uint32_t)plan.devices.size();
    pt_ctx* ctx = camera.context();
    for (int f = 0; f < frames; ++f) {
        pt_shard sh{};
        if (plan.mode == ShardMode::Samples) {
            sh.kind = PT_SHARD_ALL;
            sh.rank = 0;
            sh.nranks = 1;
            sh.sample_begin = ((uint32_t)f * n + (uint32_t)rank) * spp;  // frame-major, then rank: the union is [0, frames*N*spp)
        } else {
            sh.kind = PT_SHARD_TILES;
            sh.rank = (uint32_t)rank;
            sh.nranks = n;
            sh.tile = plan.tile;
            sh.sample_begin = (uint32_t)f * spp;
        }
        detail::check(ctx, pt_render_async(ctx, spp, max_depth, seed, &sh, 0), "pt_render_async");
    }
    detail::check(ctx, pt_wait(ctx), "pt_wait");
}

// combine: after it, rank 0's context holds the whole frame and its sample count.
template <class T>
void combine_on_root(Camera<T>& camera, Clique& clique, const MultiGpuPlan& plan, int rank, float spp_per_rank, hipStream_t stream) {
    const uint32_t n = (uint32_t)plan.devices.size();
    pt_ctx* ctx = camera.context();
    const pt_camera_state& s = camera.state();
    if (plan.mode == ShardMode::Samples) {
        void* acc = nullptr;
        uint64_t n_floats = 0;
        detail::check(ctx, pt_accum_device_ptr(ctx, &acc, &n_floats), "pt_accum_device_ptr");  // waits for queued calls
        clique.reduce_to_root(rank, (float*)acc, (size_t)n_floats, stream);
        if (rank == 0 && n > 1) detail::check(ctx, pt_add_external_samples(ctx, spp_per_rank * (float)(n - 1)), "pt_add_external_samples");
        return;
    }
    uint64_t slab_floats = 0;  // rank 0 owns the most tiles: the common slab size of the gather
    detail::check(ctx, pt_shard_slab_floats(s.width, s.height, plan.tile, 0, n, &slab_floats), "pt_shard_slab_floats");
    detail::DeviceBuffer slab, parts;
    slab.alloc_zero((size_t)slab_floats);
    if (rank == 0) parts.alloc_zero((size_t)slab_floats * n);
    pt_shard mine{PT_SHARD_TILES, (uint32_t)rank, n, plan.tile, 0};
    detail::check(ctx, pt_shard_pack(ctx, &mine, slab.p, slab_floats), "pt_shard_pack");
    clique.gather_to_root(rank, slab.p, parts.p, (size_t)slab_floats, stream);
    if (rank == 0)
        for (uint32_t r = 1; r < n; ++r) {
            pt_shard theirs{PT_SHARD_TILES, r, n, plan.tile, 0};
            detail::check(ctx, pt_shard_unpack(ctx, &theirs, parts.p + (size_t)r * slab_floats, slab_floats), "pt_shard_unpack");
        }
}

// The whole multi-GPU frame: `make_camera(rank, device)` builds the replicated scene on the rank's device,
// every rank renders its shard, the frame is combined on rank 0, and `on_root(camera0, reports, wall_ms)`
// reads it out there.  Returns 0, or 1 after printing the first rank's error.
template <class T>
int run_ranks(const MultiGpuPlan& plan, int frames, uint32_t spp, uint32_t max_depth, uint64_t seed,
              const std::function<std::unique_ptr<Camera<T>>(int rank, int device)>& make_camera,
              const std::function<int(Camera<T>&, const std::vector<RankReport>&, double wall_ms)>& on_root) {
    const int n = (int)plan.devices.size();
    Clique clique(plan);
    std::vector<RankReport> reports(n);
    std::mutex err_lock;
    std::string err;
    int root_rc = 0;
    double wall_ms = 0;
    std::chrono::steady_clock::time_point t_begin;
    auto fail = [&](const std::string& what) {
        std::lock_guard<std::mutex> g(err_lock);
        if (err.empty()) err = what;
    };
    auto failed = [&] {
        std::lock_guard<std::mutex> g(err_lock);
        return !err.empty();
    };
    auto body = [&](int rank) {
        const int device = plan.devices[rank];
        std::unique_ptr<Camera<T>> camera;
        hipStream_t stream = nullptr;
        try {
            detail::hip_check(hipSetDevice(device), "hipSetDevice");
            detail::hip_check(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
            camera = make_camera(rank, device);
        } catch (const std::exception& e) {
            fail("rank " + std::to_string(rank) + ": " + e.what());
        }
        clique.barrier();  // every rank holds its scene (or one has failed: all of them see it here)
        const bool ok = !failed();
        if (rank == 0) t_begin = std::chrono::steady_clock::now();
        clique.barrier();
        if (ok) {
            try {
                const auto t0 = std::chrono::steady_clock::now();
                render_shard(*camera, plan, rank, frames, spp, max_depth, seed);
                reports[rank].render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            } catch (const std::exception& e) {
                fail("rank " + std::to_string(rank) + ": " + e.what());
            }
        }
        clique.barrier();  // a rank that failed to render must not leave the others inside a collective
        if (ok && !failed()) {
            try {
                const auto t0 = std::chrono::steady_clock::now();
                combine_on_root(*camera, clique, plan, rank, (float)frames * (float)spp, stream);
                reports[rank].combine_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            } catch (const std::exception& e) {
                fail("rank " + std::to_string(rank) + ": " + e.what());
            }
        }
        clique.barrier();
        if (rank == 0) wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        if (ok && !failed()) {
            try {
                const pt_counters c = camera->counters();
                reports[rank].device = device;
                reports[rank].paths = c.paths;
                reports[rank].rays = c.rays_closest + c.rays_any;
            } catch (const std::exception& e) {
                fail("rank " + std::to_string(rank) + ": " + e.what());
            }
        }
        clique.barrier();
        if (rank == 0 && ok && !failed()) {
            try {
                root_rc = on_root(*camera, reports, wall_ms);
            } catch (const std::exception& e) {
                fail(std::string("rank 0: ") + e.what());
            }
        }
        camera.reset();
        if (stream) (void)hipStreamDestroy(stream);
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < n; ++r) threads.emplace_back(body, r);
    body(0);
    for (std::thread& t : threads) t.join();
    if (!err.empty()) {
        std::fprintf(stderr, "error: %s\n", err.c_str());
        return 1;
    }
    return root_rc;
}

}  // namespace pt
