/* ptcore.h — C ABI of libptcore.so, the MI355X (gfx950) path-tracing core.
 *
 * The reference (ReinierMaas/PathTracer, Rust, /root/reference) has no FFI or plugin
 * boundary; the seam replaced here is the per-pixel pair of calls in its frame loop,
 *     camera.generate(x, y)  ->  camera.sample(&mut ray, 32)      src/main.rs:157-159
 * plus accumulate + tonemap (src/main.rs:159-164), hoisted from per-pixel to per-frame,
 * and (secondary, for bit-exact checks) BVH::intersect_closest / intersect_any
 * (src/bvh.rs:180,227).  Each entry point below names the reference code it stands in for.
 *
 * Conventions
 *  - plain pointers and sizes only; all structs are POD, little-endian, fp32.
 *  - every call returns PT_OK (0) or a negative pt_status; pt_last_error(ctx) gives text.
 *    Nothing throws or aborts across the boundary (the reference panics instead).
 *  - input arrays are caller-owned host memory and are copied during the call;
 *    output arrays are caller-allocated host memory.
 *  - a context is bound to one HIP device and is externally synchronised (one thread at a
 *    time); distinct contexts are independent.  Calls are synchronous at return.
 *  - there is NO CPU fallback: pt_create fails with PT_ERR_NO_DEVICE when no gfx950
 *    device is usable.
 *  - primitive ids are ORIGINAL object indices (the value stored in bvh.indices[i],
 *    src/bvh.rs:200), independent of the flattened order on the device.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument (null pointer, zero size, wrong primitive kind ...) */
    PT_ERR_STATE = -2,     /* call order violated (render before build / set_camera ...) */
    PT_ERR_NO_DEVICE = -3, /* no usable HIP device / device index out of range */
    PT_ERR_HIP = -4,       /* a HIP runtime call failed (text in pt_last_error) */
    PT_ERR_LIMIT = -5,     /* scene exceeds a documented limit */
    PT_ERR_BVH = -6        /* builder hit the empty-child hazard of src/bvh.rs:87-97 */
} pt_status;

/* Material — src/material.rs:5-18 (enum Material). */
enum { PT_MAT_DIFFUSE = 0, PT_MAT_DIELECTRIC = 1, PT_MAT_EMISSIVE = 2 };
typedef struct pt_material {
    uint32_t kind; /* PT_MAT_* */
    float a;       /* Diffuse: speculaty        Dielectric: refraction_index_n1 */
    float b;       /* Diffuse: unused           Dielectric: refraction_index_n2 */
    float rgb[3];  /* color */
} pt_material;

/* Ray — src/ray.rs:14-18 (origin, direction, distance). */
typedef struct pt_ray {
    float o[3];
    float d[3];
    float tmax;
} pt_ray;

/* Result of one traversal: the fields of Option<Intersection> (src/ray.rs:6-11) that identify
 * the hit, plus ray.distance after the call.  prim == PT_MISS for None. */
#define PT_MISS 0xFFFFFFFFu
typedef struct pt_hit {
    uint32_t prim;
    float t;         /* ray.distance after the call (unchanged tmax on a miss) */
    uint32_t inside; /* Intersection::inside */
} pt_hit;

/* One BVH node exactly as src/bvh.rs:10-14 (BVHNode: AABB + left_first + count), 32 bytes. */
typedef struct pt_bvh_node {
    float min[3];
    uint32_t left_first;
    float max[3];
    uint32_t count;
} pt_bvh_node;

/* Camera state after Camera::update (src/camera.rs:271-292). */
typedef struct pt_camera_state {
    float origin[3], target[3], direction[3], right[3], up[3];
    float p1[3], p2[3], p3[3];
    float focal_distance, lens_size;
    uint32_t width, height;
} pt_camera_state;

/* Work split for multi-GPU frames (no counterpart in the reference, which is one process):
 *  PT_SHARD_ALL     whole image, sample indices [sample_begin, sample_begin+spp)
 *  PT_SHARD_TILES   only the tile x tile pixel tiles t (row-major tile index) with t % nranks == rank
 * sample indices are global, so any split of [0, S) over ranks reproduces the 1-GPU samples. */
enum { PT_SHARD_ALL = 0, PT_SHARD_TILES = 1 };
typedef struct pt_shard {
    uint32_t kind;
    uint32_t rank, nranks;
    uint32_t tile;          /* tile edge in pixels (multiple of 8), PT_SHARD_TILES only */
    uint32_t sample_begin;  /* global index of the first sample this call renders */
} pt_shard;

typedef struct pt_counters {
    uint64_t paths;          /* generate+sample pairs started */
    uint64_t rays_closest;   /* intersect_closest traversals executed (incl. look-ahead, re-trace) */
    uint64_t rays_any;       /* intersect_any traversals executed */
    uint64_t node_tests;     /* AABB records tested: 1 (root) + 2 per internal node entered (PT_RENDER_COUNT) */
    uint64_t prim_tests;     /* Primitive::intersect calls (PT_RENDER_COUNT) */
    uint64_t extend_launches;
    double extend_ms;        /* PT_RENDER_TIMED: time during which at least one round traversal kernel (k_extend) was running,
                                from HIP events on the launching streams (union of the launch intervals) */
    double render_ms;        /* host wall-clock from submission to completion of every render call since reset, summed */
    /* the part of the above done by the fused tail kernel (paths finished outside the rounds) */
    uint64_t tail_rays, tail_node_tests, tail_prim_tests, tail_launches;
    double tail_ms;          /* as extend_ms, for k_path (render mode 1) / the fused tail kernel (mode 0) */
    /* plain sums of the same launch durations (= a profiler's average duration x launches); they exceed
     * extend_ms / tail_ms exactly when launches overlapped (several batches or calls in flight) */
    double extend_ms_sum, tail_ms_sum;
    /* The NEE block of src/camera.rs:314-338 where the fast kernels leave it out (its weight is exactly 0 there) and it could
     * still change a sum (a transport too large, infinite or NaN for 0 * product to be 0): redo_paths = paths finished by the
     * second, literal pass since the last reset; redo_restarts = batches run again as a whole with the NEE kernels because
     * more paths asked than PT_OPT_REDO_CAPACITY (the ray counters then include the abandoned attempt). */
    uint64_t redo_paths, redo_restarts;
} pt_counters;

/* flags for pt_render */
enum {
    PT_RENDER_DEFAULT = 0,
    PT_RENDER_TIMED = 1,   /* bracket every traversal launch with HIP events (extend_ms) */
    PT_RENDER_COUNT = 2    /* instrumented traversal: fill node_tests / prim_tests (slower) */
};

const char *pt_strerror(int status);
const char *pt_last_error(const pt_ctx *ctx);

/* lifetime — replaces process start-up in src/main.rs:76-111 (no SDL). */
int pt_create(int device, pt_ctx **out);
void pt_destroy(pt_ctx *ctx);

/* scene input — replaces Vec<Triangle>/Vec<Sphere> handed to Scene::new (src/scene.rs:26-33).
 * A scene holds ONE primitive kind, like Scene<T: Primitive>.
 *  pos/nrm: N x 9 floats (position0..2 / normal0..2, src/primitive/triangle.rs:14-22)
 *  cr:      N x 4 floats (position.xyz, radius, src/primitive/sphere.rs:14-18) */
int pt_set_triangles(pt_ctx *ctx, const float *pos, const float *nrm, const uint32_t *mat_id, uint64_t n);
int pt_set_spheres(pt_ctx *ctx, const float *cr, const uint32_t *mat_id, uint64_t n);
int pt_set_materials(pt_ctx *ctx, const pt_material *m, uint32_t count);
/* skybox — replaces Scene::read_skybox (src/scene.rs:171-183): w x h x 3 floats, row-major.
 * The device copy is zero-padded so the unclamped lookup of src/scene.rs:163-166 stays in
 * range and reads black, as the reference's oversize Vec does (src/scene.rs:174). */
int pt_set_sky(pt_ctx *ctx, const float *rgb, uint32_t w, uint32_t h);

/* BVH::new (src/bvh.rs:25-179): binned-SAH build, bit-identical tree and leaf order, then flattening + upload.
 * Also collects the lights list (src/bvh.rs:29-35).  The tree is built on the GPU, one level at a time (SURVEY 8f row
 * N3; PT_OPT_DEVICE_BUILD, default 1), or by the threaded host build when the option is 0.  Same tree either way,
 * negative zeros, infinities and NaNs in the coordinates included. */
int pt_build(pt_ctx *ctx);
/* what the last pt_build did: *on_device = 1 when the tree was built on the GPU; *build_ms = the tree build alone
 * (host or device, without flattening and upload).  No counterpart in the reference. */
int pt_get_build_info(const pt_ctx *ctx, uint32_t *on_device, double *build_ms);
/* Adopt a tree the host built itself — the north-star arrangement, where the Rust host keeps
 * BVH::new (src/bvh.rs:25-179) and hands over bvh_nodes + indices (src/bvh.rs:17-22).  The tree
 * is validated (ranges, coverage, no sharing), flattened and uploaded; use instead of pt_build. */
int pt_set_bvh(pt_ctx *ctx, const pt_bvh_node *nodes, uint64_t n_nodes, const uint64_t *indices, uint64_t n_indices);
/* The same CPU build as pt_build with no context and no device: geom is N x 9 triangle
 * positions (prim_kind 0) or N x 4 spheres (prim_kind 1).  nodes needs room for 2N-1 entries. */
int pt_bvh_build(int prim_kind, const float *geom, uint64_t n, pt_bvh_node *nodes, uint64_t nodes_cap,
                 uint64_t *indices, uint64_t *n_nodes_out, uint32_t *max_depth_out);
/* read the tree back in the reference's own layout (parity of the build). */
int pt_get_bvh_sizes(const pt_ctx *ctx, uint64_t *n_nodes, uint64_t *n_indices, uint32_t *max_depth);
int pt_get_bvh(const pt_ctx *ctx, pt_bvh_node *nodes, uint64_t *indices);

/* Camera::new + update + focus (src/camera.rs:89-112,214-292); the autofocus trace runs on
 * the device.  Resets the accumulation, as src/main.rs:137-142 does on a camera change. */
int pt_set_camera(pt_ctx *ctx, const float origin[3], const float target[3], uint32_t width,
                  uint32_t height, float lens_size);
int pt_get_camera(const pt_ctx *ctx, pt_camera_state *out);

/* Camera::generate (src/camera.rs:426-449) for n pixels (xy: n x 2 u32) of global sample index
 * `sample`; the four uniforms come from the counter-based stream of (seed, pixel, sample). */
int pt_generate(pt_ctx *ctx, const uint32_t *xy, uint64_t n, uint32_t sample, uint64_t seed, pt_ray *rays);

/* The frame loop body, src/main.rs:143-170: for each of `spp` samples, every pixel gets
 * generate + sample(max_depth) added to the accumulator.  max_depth is the literal 32 of
 * src/main.rs:159.  shard may be NULL (= PT_SHARD_ALL, sample_begin = samples rendered so far). */
int pt_render(pt_ctx *ctx, uint32_t spp, uint32_t max_depth, uint64_t seed, const pt_shard *shard,
              uint32_t flags);
/* The same call without waiting for the result (no counterpart in the reference, whose loop renders one
 * sample per frame, src/main.rs:143,159, and would otherwise pay a whole kernel's fill-and-drain per
 * sample): the call is queued and returns; consecutive calls overlap on the device (the few long paths
 * that end one call's kernel run beside the start of the next call's) while the accumulator is still
 * summed in call order and sample order, so the sums are bit-identical to the same sequence of pt_render
 * calls.  pt_wait blocks until everything queued has been added; every entry point that reads or
 * changes the frame, camera, scene, options or counters waits first by itself.  Progress is made
 * inside pt_render_async / pt_wait calls (there is no background thread).  pt_render = pt_render_async
 * + pt_wait.  An error of a queued call is reported by the call that waits for it. */
int pt_render_async(pt_ctx *ctx, uint32_t spp, uint32_t max_depth, uint64_t seed, const pt_shard *shard,
                    uint32_t flags);
int pt_wait(pt_ctx *ctx);
/* The pixels a shard owns, in the order the device walks them (device-free; the same mapping
 * function is compiled into the kernels).  xy: up to cap (x, y) pairs; n_out: pixels owned.
 * tile = 8 with nranks = 1 is the PT_SHARD_ALL order. */
int pt_shard_map(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t nranks, uint32_t *xy,
                 uint64_t cap, uint64_t *n_out);
int pt_reset_accum(pt_ctx *ctx);
/* accum: W x H x 3 floats, the running SUM (src/main.rs:100,159); spp_out: samples summed. */
int pt_read_accum(pt_ctx *ctx, float *rgb, float *spp_out);
/* Resume (the reference restarts from zero, src/main.rs:139-142): puts a sum read
 * earlier with pt_read_accum back, with its sample count.  The next pt_render then continues with
 * sample index (uint32)spp, so an interrupted and resumed run adds up to exactly the uninterrupted frame. */
int pt_write_accum(pt_ctx *ctx, const float *rgb, float spp);
/* vec_to_rgb(scale * accum) (src/main.rs:71-75,160-164) into an RGB24 image with row pitch. */
int pt_read_rgb8(pt_ctx *ctx, uint8_t *rgb, uint64_t pitch_bytes);
/* The frame as far as it has got, without waiting for queued render calls (no counterpart in the reference, whose
 * loop blocks on every frame): vec_to_rgb(accum / n) for the n samples that have been added so far — always whole
 * calls' batches in submission order, so with one batch per call the image equals what pt_read_rgb8 would have
 * returned after that many pt_render calls.  *spp_out = n (0 before the first batch has been added: the image is
 * then black).  Merged queued calls (PT_OPT_MERGE_QUEUED_CALLS) are added as one, so n advances by their sum.  Waits only for accumulate kernels already on
 * their way, never for a path kernel: a display loop can call pt_render_async + pt_read_rgb8_latest every
 * iteration and keep the device busy. */
int pt_read_rgb8_latest(pt_ctx *ctx, uint8_t *rgb, uint64_t pitch_bytes, float *spp_out);
/* device address of the W x H x 3 float accumulator, for collectives that combine ranks. */
int pt_accum_device_ptr(pt_ctx *ctx, void **ptr, uint64_t *n_floats);
/* add `spp` to the sample count after an external reduce put other ranks' sums in accum. */
int pt_add_external_samples(pt_ctx *ctx, float spp);

/* Tile-sharded frames (BASELINE config C4; no counterpart in the reference): the tiles a rank owns as one
 * compact slab, so that ranks exchange their own pixels only (one RCCL gather of nranks slabs of
 * W*H*12/nranks bytes instead of a reduce of nranks whole frames).  A slab holds the shard's pixels in the
 * shard's own order (pt_shard_map order: tile by tile, 8x8 blocks inside a tile), 3 floats each, with the
 * padding pixels of edge tiles kept as zeros: pt_shard_slab_floats() floats (device-free; ranks differ by at
 * most one tile, pad to rank 0's size for a gather of equal parts).
 *  pt_shard_pack    slab <- this context's accumulator (slab_dev: DEVICE memory of >= that many floats)
 *  pt_shard_unpack  this context's accumulator <- slab, for the pixels of `shard` only (the gathering rank
 *                   calls it once per received slab with that rank's shard).  Add the other ranks' sample
 *                   count with pt_add_external_samples only when they rendered samples this rank did not.
 * Both wait for queued render calls, run on the context's stream and return when the copy is done; the caller
 * orders them against its own streams (the slab must not be in use by a collective still in flight). */
int pt_shard_slab_floats(uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t nranks, uint64_t *n_floats);
int pt_shard_pack(pt_ctx *ctx, const pt_shard *shard, float *slab_dev, uint64_t cap_floats);
int pt_shard_unpack(pt_ctx *ctx, const pt_shard *shard, const float *slab_dev, uint64_t n_floats);

/* BVH::intersect_closest / intersect_any (src/bvh.rs:180-271) over a batch of rays. */
int pt_trace(pt_ctx *ctx, const pt_ray *rays, uint64_t n, int any_hit, pt_hit *hits);
/* pt_trace without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * call hipMalloc — tight serving loops, harnesses that forbid allocation.  (It is NOT for a stream under HIP graph
 * capture: like pt_trace it copies rays from and hits to the caller's host memory and waits for the stream.)
 * Same arguments, results and counters as pt_trace; the device copies of rays / hits and the overflow of the
 * per-lane traversal stacks live in a static device array (8.6 MiB per GPU, held until the process ends)
 * instead of per-call buffers.  Limits: n <= 16384 rays per call and a tree of at most 76 levels
 * (PT_ERR_LIMIT otherwise; pt_trace has neither limit).  All contexts on one GPU share that array:
 * calls that overlap on one GPU race — serialise them. */
int pt_trace_noalloc(pt_ctx *ctx, const pt_ray *rays, uint64_t n, int any_hit, pt_hit *hits);
/* Scene::sample_skybox (src/scene.rs:162-169) for n directions (n x 3 floats) -> n x 3 floats. */
int pt_sample_skybox(pt_ctx *ctx, const float *dirs, uint64_t n, float *rgb);

/* Scheduling knobs of pt_render (no counterpart in the reference; none of them changes a result):
 *  PT_OPT_BATCH_PATHS     paths per batch, whole samples (0 = default: 512 Mi in mode 1, 8 Mi in mode 0)
 *  PT_OPT_LANES           batches in flight at once, 1..8 (0 = default: mode 1: 2 for large batches up to 8 for
 *                         small ones (few samples per call); mode 0: 8)
 *  PT_OPT_TAIL_MIN_PATHS  a batch is handed to the fused tail kernel when at most this many paths
 *  PT_OPT_TAIL_FRACTION   ... or at most batch/fraction paths are still alive (0 = never)
 *  PT_OPT_TAIL_STREAMS    streams the tail kernels are spread over, 1..4 (default 4)
 *  PT_OPT_TAIL_PATHS_PER_LANE  paths each tail-kernel lane works through on average (default 4)
 *  PT_OPT_BULK_STREAMS    streams the round kernels of the lanes are spread over, 1..8 (default 8)
 *  PT_OPT_RENDER_MODE     1 = one persistent kernel per batch, path state in registers (k_path, default),
 *                         0 = wavefront rounds (raygen, extend, shade + compaction) + fused tail kernel
 *  PT_OPT_FIRST_CHECK_ROUND  rounds enqueued before the live-path count is first looked at (default 3)
 *  PT_OPT_TAIL_MODE       0 = fused phase-scheduled tail kernel, 1 = further rounds on the tail stream with a small
 *                         grid, 2 = fused tail kernel that traces and shades in lock step
 *  PT_OPT_SIMPLE_EXTEND   1 = one-ray-per-lane traversal kernel instead of the persistent one (A/B, tests)
 *  PT_OPT_DEVICE_BUILD    1 = pt_build builds the tree on the GPU (default), 0 = on the host (threads)
 *  PT_OPT_MERGE_QUEUED_CALLS  a pt_render_async call that continues the newest queued call (same frame, depth, seed,
 *                         shard and flags, the next sample indices) while that call has not started yet is added to it, up
 *                         to this many paths per merged call (default 128 Mi; 0 = never merge).  Same results, fewer and
 *                         larger launches when calls arrive faster than they are rendered
 *  PT_OPT_REDO_CAPACITY  paths per batch that may ask for the literal NEE block before the whole batch is run again with the
 *                         NEE kernels (default 65536; tests set it to 0 to exercise the restart)
 *  PT_OPT_WAVES_PER_SIMD  which form of the path kernel runs: 6 = six waves per SIMD (80 registers, 12 stack levels in LDS), 7 = seven
 *                         (72 registers, 11 levels), 0 (default) = seven when node and primitive records exceed the 256-MB Infinity Cache
 *                         (their fetches then wait longest), else six.  Same results either way
 * and ONE knob that changes what is executed, though not a single sum (default 0; bench.py's metric run never sets it):
 *  PT_OPT_SKIP_REPEATED_MISS  1 = when the look-ahead trace of a diffuse bounce (src/camera.rs:362) missed everything, the
 *                         reference's re-trace of that very ray in the next iteration (quirk Q3: the hit is dropped, ray.distance
 *                         stays inf) is not run again: it would miss again, and the path ends in the sky at once.  Same frame bit
 *                         for bit, fewer intersect_closest traversals: rays_closest then counts what was executed, which is no
 *                         longer the reference's number of traversals */
enum { PT_OPT_BATCH_PATHS = 0, PT_OPT_LANES = 1, PT_OPT_TAIL_MIN_PATHS = 2, PT_OPT_TAIL_FRACTION = 3,
       PT_OPT_TAIL_STREAMS = 4, PT_OPT_TAIL_PATHS_PER_LANE = 5, PT_OPT_SIMPLE_EXTEND = 6, PT_OPT_BULK_STREAMS = 7, PT_OPT_TAIL_MODE = 8, PT_OPT_FIRST_CHECK_ROUND = 9, PT_OPT_RENDER_MODE = 10, PT_OPT_MERGE_QUEUED_CALLS = 11, PT_OPT_DEVICE_BUILD = 12, PT_OPT_REDO_CAPACITY = 13, PT_OPT_SKIP_REPEATED_MISS = 14, PT_OPT_WAVES_PER_SIMD = 15 };
int pt_set_option(pt_ctx *ctx, int option, uint64_t value);
int pt_get_counters(pt_ctx *ctx, pt_counters *out);
/* Diagnostics of the wavefront loop (no counterpart in the reference): after a pt_render with
 * PT_RENDER_COUNT, the number of live paths entering each extend round of the last batch, and
 * the largest node_tests any single traversal needed since the last pt_reset_counters. */
int pt_get_round_stats(pt_ctx *ctx, uint32_t *queue_sizes, uint32_t cap, uint32_t *n_rounds, uint64_t *max_node_tests);
int pt_reset_counters(pt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
