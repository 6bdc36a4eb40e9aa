// pt_build.hip — the reference's binned-SAH build (src/bvh.rs:25-179) on the GPU, same tree bit for bit.
//
// The reference builds depth first with an explicit stack; nothing in a node's split depends on any node outside its own
// range of bvh.indices, so the tree can be built one LEVEL at a time, every node of a level at once, and numbered the
// reference's way afterwards.  Per level (positions = places in bvh.indices, a node = a contiguous range of them):
//   k_bin      every primitive of an active node: its three bin numbers (bvh.rs:125-133) -> bin bounds / counts
//   k_sah      one thread per node: the sweep of bvh.rs:135-168 -> axis + pivot, or "no split beats the parent"
//   k_flag     left / right flag per primitive (centre[axis] <= pivot, bvh.rs:70)
//   scan       exclusive prefix sum of the flags over all positions -> rank of every left primitive in its node
//   k_children two child nodes per split node, the next level's active list
//   k_lpos / k_scatter  the reference's swap walk (bvh.rs:66-79) decides the order inside the two halves; its result
//              is a function of the flags alone (pt_bvh.cpp, partition_large, has the argument): the k-th left object
//              ends at place k, a right object at place p moves to l[p], l[l[p]] ... until its place is >= nL.  Child
//              bounds (and the children's centre bounds for the next level) are reduced in the same pass.
// Bounds are mins / maxes of input values (exact in any order), counts are integers, and the few float operations
// (delta, bin number, areas, costs, pivot) are the same single IEEE operations in the same order as on the host
// (-ffp-contract=off).  Two things depend on the ORDER of a min / max chain in the reference (f32::min / max: the
// non-NaN operand wins, the first operand wins a tie): NaNs are simply never taken, so they are skipped here too; and
// min(+0, -0) is the zero that came first.  No decision of the build can see the sign of a zero (areas, costs, bin
// numbers and the <= against the pivot come out the same), so zeros are reduced as one value and the sign of a node
// bound that ends up zero is settled separately: the object with a zero there that the reference's walk meets first
// (smallest position at the time of the walk) decides it.
// Numbering: the reference appends a node's two children when it splits the node, and splits in depth-first order, left
// subtree first.  With s(v) = number of splits in v's subtree, the split of v's left child is event e(v) + 1 and the
// right child's e(v) + 1 + s(left); children of the node split at event e are nodes 1 + 2e and 2 + 2e.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pt_build.h"
#include "pt_kernels.h"

#pragma clang fp contract(off)

namespace pt {
namespace {

constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kSmallSlot = 0xfffffffeu;  // "slot" of a node that one wave will finish on its own (k_small_wave)
#ifndef PT_BUILD_SMALL
#define PT_BUILD_SMALL 128
#endif
constexpr uint32_t kSmall = PT_BUILD_SMALL;      // ... nodes of at most this many primitives
constexpr int kSmallPerLane = kSmall / 64;    // k_small_wave: primitives of a node per lane
static_assert(kSmall % 64 == 0, "k_small_wave handles whole waves of primitives");
constexpr uint32_t kKeyPosInf = 0xff800000u;  // key(+inf): identity of min
constexpr uint32_t kKeyNegInf = 0x007fffffu;  // key(-inf): identity of max
constexpr int kBlock = 256;

// order-preserving map float -> uint32 (finite values and infinities; both zeros share a key, zero_vote() keeps their sign)
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0;  // one key for both zeros
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// f32::min / max of the reference (pt_bvh.cpp mn2 / mx2): the non-NaN operand wins, the first wins a tie
__device__ __forceinline__ float rmin(float a, float b) { return a != a ? b : (b < a ? b : a); }
__device__ __forceinline__ float rmax(float a, float b) { return a != a ? b : (b > a ? b : a); }
__device__ __forceinline__ void key_min(uint32_t* dst, float v) { if (v == v) atomicMin(dst, f2key(v)); }  // a NaN is never taken
__device__ __forceinline__ void key_max(uint32_t* dst, float v) { if (v == v) atomicMax(dst, f2key(v)); }
// who decides the sign of a zero bound: smallest (position << 1 | sign bit) among the objects with a zero there
__device__ __forceinline__ void zero_vote(uint32_t* dst, float v, uint32_t pos) { if (v == 0.0f) atomicMin(dst, (pos << 1) | (__float_as_uint(v) >> 31)); }
__device__ __forceinline__ float key2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

struct Rec {  // pt_bvh.cpp's record: bvh.indices[i] with that object's centre and bounds
    float c[3], mn[3], mx[3];
    uint32_t idx;
};

struct BNode {  // 88 B, creation order (level by level)
    uint32_t bmin[3], bmax[3];  // Primitive::bounds of the members, as keys
    uint32_t cmin[3], cmax[3];  // bounds of their centres, as keys
    uint32_t first, count;
    uint32_t left;              // id of the left child (right = left + 1); 0 = leaf
    uint32_t slot;              // place in the current level's active list, kNone when the node is finished
    uint32_t zs[6];             // zero_vote() words of bmin[3], bmax[3]
};

struct Bins {  // per active node
    uint32_t mn[24][3], mx[24][3];
    uint32_t n[24];
};
constexpr int kBinWords = sizeof(Bins) / 4;  // 168

struct Split {
    int axis;     // -1: no split beats the parent cost
    float pivot;
    uint32_t n_left, left_slot, right_slot, pair;  // filled by k_children
};

struct DBox {
    float lo[3], hi[3];
    __device__ void clear() {
        for (int a = 0; a < 3; ++a) { lo[a] = __builtin_inff(); hi[a] = -__builtin_inff(); }
    }
    __device__ void grow(const DBox& o) {  // AABB::combine
        for (int a = 0; a < 3; ++a) { lo[a] = rmin(lo[a], o.lo[a]); hi[a] = rmax(hi[a], o.hi[a]); }
    }
    __device__ void grow(const float* mn, const float* mx) {
        for (int a = 0; a < 3; ++a) { lo[a] = rmin(lo[a], mn[a]); hi[a] = rmax(hi[a], mx[a]); }
    }
    __device__ float area() const {  // AABB::area, aabb.rs:42-45
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return (2.0f * ex * ey + 2.0f * ex * ez) + 2.0f * ey * ez;
    }
};

// ---- Primitive::centre / bounds of every object (triangle.rs:80-90, sphere.rs:81-91), the same single operations as
// prim_info_triangles / prim_info_spheres on the host
__global__ void k_recs_tri(const float* __restrict__ pos, uint32_t n, Rec* rec) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pos + 9 * (size_t)i;
    Rec r;
    for (int a = 0; a < 3; ++a) {
        r.c[a] = ((p[a] + p[3 + a]) + p[6 + a]) * (1.0f / 3.0f);
        r.mn[a] = rmin(rmin(p[a], p[3 + a]), p[6 + a]);
        r.mx[a] = rmax(rmax(p[a], p[3 + a]), p[6 + a]);
    }
    r.idx = i;
    rec[i] = r;
}

__global__ void k_recs_sph(const float* __restrict__ cr, uint32_t n, Rec* rec) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* s = cr + 4 * (size_t)i;
    Rec r;
    for (int a = 0; a < 3; ++a) {
        r.c[a] = s[a];
        r.mn[a] = s[a] - s[3];
        r.mx[a] = s[a] + s[3];
    }
    r.idx = i;
    rec[i] = r;
}

// ---- level 0: root bounds (the reference's walk is over the objects in input order)
__global__ void k_root(const Rec* __restrict__ rec, uint32_t n, BNode* nodes) {
    __shared__ uint32_t acc[12];
    if (threadIdx.x < 12) acc[threadIdx.x] = (threadIdx.x % 6) < 3 ? kKeyPosInf : kKeyNegInf;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Rec r = rec[i];
        for (int a = 0; a < 3; ++a) {
            key_min(&acc[a], r.mn[a]);
            key_max(&acc[3 + a], r.mx[a]);
            key_min(&acc[6 + a], r.c[a]);
            key_max(&acc[9 + a], r.c[a]);
            zero_vote(&nodes[0].zs[a], r.mn[a], i);
            zero_vote(&nodes[0].zs[3 + a], r.mx[a], i);
        }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        uint32_t* dst = nodes[0].bmin + threadIdx.x;  // bmin, bmax, cmin, cmax are consecutive
        if ((threadIdx.x % 6) < 3) atomicMin(dst, acc[threadIdx.x]);
        else atomicMax(dst, acc[threadIdx.x]);
    }
}

__global__ void k_init_root(BNode* nodes, uint32_t n, uint32_t* seg, uint32_t* active) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        BNode r;
        for (int a = 0; a < 3; ++a) { r.bmin[a] = r.cmin[a] = kKeyPosInf; r.bmax[a] = r.cmax[a] = kKeyNegInf; }
        r.first = 0;
        r.count = n;
        r.left = 0;
        r.slot = n > 2 ? 0 : kNone;
        for (int a = 0; a < 6; ++a) r.zs[a] = kNone;
        nodes[0] = r;
        active[0] = 0;
    }
    if (i < n) seg[i] = n > 2 ? 0u : kNone;
}

__global__ void k_init_bins(Bins* bins, uint32_t n_active) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_active * kBinWords) return;
    const uint32_t w = (uint32_t)(i % kBinWords);
    ((uint32_t*)bins)[i] = w < 72 ? kKeyPosInf : (w < 144 ? kKeyNegInf : 0u);
}

// ---- binning.  Positions of one node are contiguous, so a whole workgroup usually lies inside one node and a wave in at
// most two: a wave reduces into its own LDS bins (one set per node), which are flushed with one global atomic per touched
// word (per workgroup when all of it is one node).  Only a wave with three or more nodes (nodes shorter than a wave: not
// with the small subtrees handed off at 128) updates the global bins per primitive.
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ uint32_t bin_identity(int w) { return w < 72 ? kKeyPosInf : (w < 144 ? kKeyNegInf : 0u); }
__device__ __forceinline__ void bin_flush(uint32_t* g, int w, uint32_t v) {
    if (w < 72) { if (v != kKeyPosInf) atomicMin(g + w, v); }
    else if (w < 144) { if (v != kKeyNegInf) atomicMax(g + w, v); }
    else if (v) atomicAdd(g + w, v);
}

// The active lanes of a wave (positions of unfinished nodes) belong to at most two nodes whenever nodes are longer than a
// wave — finished positions may lie in between: such a wave keeps one set of LDS accumulators per node.
struct WaveSegs {
    uint32_t a, b;  // slot of the first / last active lane's node (kNone: no active lane)
    bool two;       // every active lane is in a or b
};
__device__ __forceinline__ WaveSegs wave_segs(uint32_t s) {
    WaveSegs w;
    const unsigned long long am = __ballot(s != kNone);
    w.a = w.b = kNone;
    if (am) {
        w.a = __shfl(s, __ffsll((long long)am) - 1, 64);
        w.b = __shfl(s, 63 - __clzll((long long)am), 64);
    }
    w.two = __all((s == kNone || s == w.a || s == w.b) ? 1 : 0) != 0;
    return w;
}

__global__ void k_bin(const Rec* __restrict__ rec, const uint32_t* __restrict__ seg, uint32_t n, const BNode* __restrict__ nodes,
                      const uint32_t* __restrict__ active, Bins* bins) {
    __shared__ uint32_t lb[kWaves][2][kBinWords];
    __shared__ uint32_t s0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t s = i < n ? seg[i] : kNone;
    if (threadIdx.x == 0) s0 = s;
    for (int w = threadIdx.x; w < kWaves * 2 * kBinWords; w += blockDim.x) (&lb[0][0][0])[w] = bin_identity(w % kBinWords);
    __syncthreads();
    const WaveSegs ws = wave_segs(s);
    const uint32_t block_s = s0;
    const bool block_uniform = __syncthreads_and((i >= n || s == block_s) ? 1 : 0) != 0 && block_s != kNone;
    if (s != kNone) {
        const BNode& nd = nodes[active[s]];
        const Rec r = rec[i];
        Bins* B = ws.two ? (Bins*)lb[wave][s == ws.a ? 0 : 1] : bins + s;
        for (int a = 0; a < 3; ++a) {
            const float lo = key2f(nd.cmin[a]), hi = key2f(nd.cmax[a]);
            const float delta = (hi - lo) / 8.0f;                 // bvh.rs:121-123
            const float q = (r.c[a] - lo) / delta;                // bvh.rs:127-132: NaN (0 / 0) -> bin 0
            const int b = a * 8 + (int)fminf(fmaxf(q, 0.0f), 7.0f);
            for (int k = 0; k < 3; ++k) {
                key_min(&B->mn[b][k], r.mn[k]);
                key_max(&B->mx[b][k], r.mx[k]);
            }
            atomicAdd(&B->n[b], 1u);
        }
    }
    __syncthreads();
    if (block_uniform) {  // every wave is inside the one node (or past the end): combine the slices, one flush
        uint32_t* g = (uint32_t*)(bins + block_s);
        for (int w = threadIdx.x; w < kBinWords; w += blockDim.x) {
            uint32_t v = lb[0][0][w];
            for (int k = 1; k < kWaves; ++k) v = w < 72 ? min(v, lb[k][0][w]) : (w < 144 ? max(v, lb[k][0][w]) : v + lb[k][0][w]);
            bin_flush(g, w, v);
        }
    } else if (ws.two) {
        if (ws.a != kNone)
            for (int w = lane; w < kBinWords; w += 64) bin_flush((uint32_t*)(bins + ws.a), w, lb[wave][0][w]);
        if (ws.b != ws.a)
            for (int w = lane; w < kBinWords; w += 64) bin_flush((uint32_t*)(bins + ws.b), w, lb[wave][1][w]);
    }
}

// ---- bvh.rs:135-168: the sweep over the eight bins of one axis
__device__ __forceinline__ void sah_axis(const DBox* bb, const uint32_t* bn, int a, float& best, int& best_axis, int& best_bin) {
    DBox right[8];
    uint32_t nright[8];
    right[7] = bb[7];
    nright[7] = bn[7];
    for (int k = 6; k >= 0; --k) {
        right[k] = right[k + 1];
        right[k].grow(bb[k]);
        nright[k] = nright[k + 1] + bn[k];
    }
    DBox left;
    left.clear();
    uint32_t nleft = 0;
    for (int k = 0; k < 8; ++k) {
        left.grow(bb[k]);
        nleft += bn[k];
        const float cost = (float)nleft * left.area() + (float)nright[k] * right[k].area();  // right includes bin k, bvh.rs:155
        if (cost < best) { best_axis = a; best_bin = k; best = cost; }
    }
}

// ---- bvh.rs:104-169 for one node per thread
__global__ void k_sah(const uint32_t* __restrict__ active, uint32_t n_active, const BNode* __restrict__ nodes, const Bins* __restrict__ bins,
                      Split* split) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_active) return;
    const BNode& nd = nodes[active[s]];
    DBox nb, cb;
    for (int a = 0; a < 3; ++a) {
        nb.lo[a] = key2f(nd.bmin[a]); nb.hi[a] = key2f(nd.bmax[a]);
        cb.lo[a] = key2f(nd.cmin[a]); cb.hi[a] = key2f(nd.cmax[a]);
    }
    const float parent_cost = nb.area() * (float)nd.count;
    const Bins& B = bins[s];
    int best_axis = -1, best_bin = -1;
    float best = parent_cost;
    for (int a = 0; a < 3; ++a) {
        DBox bb[8];
        uint32_t bn[8];
        for (int k = 0; k < 8; ++k) {
            for (int c = 0; c < 3; ++c) { bb[k].lo[c] = key2f(B.mn[a * 8 + k][c]); bb[k].hi[c] = key2f(B.mx[a * 8 + k][c]); }
            bn[k] = B.n[a * 8 + k];
        }
        sah_axis(bb, bn, a, best, best_axis, best_bin);
    }
    Split sp{};
    sp.axis = best_axis;
    if (best_axis >= 0) {
        const float delta = (cb.hi[best_axis] - cb.lo[best_axis]) / 8.0f;
        sp.pivot = cb.lo[best_axis] + (float)(1 + best_bin) * delta;
    }
    sp.left_slot = sp.right_slot = kNone;
    split[s] = sp;
}

// ---- one WAVE finishes the subtree of a node of at most kSmall primitives, depth first as the reference does
// (bvh.rs:52-103 with its explicit stack).  The subtree's records live in LDS; every phase of a node (bins, flags,
// places, the children's bounds and centre bounds) runs over its primitives two per lane, the sweep one candidate
// (axis, bin) per lane, and the order inside the two halves comes from the flags by the rule of k_lpos / k_scatter.
// Bounds are reduced as keys with zero votes (position in the node = the order in which the reference's walk meets the
// objects), exactly as for the large nodes.  The subtree's nodes go to a private block — the children of its k-th
// split at block places 2k and 2k + 1, which is the reference's numbering relative to the block — and k_emit_small
// moves the block to its place once the split event of the subtree's root is known.
// (One THREAD per subtree, private bins in scratch memory, was the first version: 35 of the 72 ms of a 10 M-triangle
// build; one wave with the sweep on three lanes and a pass of its own for the centre bounds 16 of 54; this one 11 of
// 41.  The larger the subtrees it takes the better up to 128 — 256 and 512 measured slower, before and after.)
constexpr int kSmallStack = kSmall / 3 + 3;  // pending nodes are disjoint and hold at least three primitives each
struct SmallShared {
    Rec rec[kSmall];
    Bins bins;
    uint32_t cbk[2][6];  // bounds of the two children
    uint32_t cck[2][6];  // bounds of their centres
    uint32_t zs[2][6];
    uint32_t lpos[kSmall];
    int stack_id[kSmallStack];
    uint32_t stack_depth[kSmallStack], stack_rel[kSmallStack], stack_count[kSmallStack];
    float stack_box[kSmallStack][12];  // node bounds (min, max), centre bounds (min, max)
};

__global__ __launch_bounds__(64) void k_small_wave(const uint32_t* __restrict__ small_list, const uint32_t* __restrict__ small_level, uint32_t n_small,
                                                   const BNode* __restrict__ nodes, const Rec* __restrict__ small_rec, pt_bvh_node* small_nodes,
                                                   uint32_t* small_splits, uint32_t* final_idx, uint32_t* counters) {
    __shared__ SmallShared sh;
    const uint32_t t = blockIdx.x;  // one subtree per block of one wave, so the barriers below are uniform
    const uint32_t lane = threadIdx.x;
    const uint32_t v = small_list[t];
    const BNode root = nodes[v];
    pt_bvh_node* out = small_nodes + 2 * (size_t)root.first;
    for (uint32_t i = lane; i < root.count; i += 64) sh.rec[i] = small_rec[root.first + i];
    if (lane == 0) {
        sh.stack_id[0] = -1;
        sh.stack_depth[0] = 1;
        sh.stack_rel[0] = 0;
        sh.stack_count[0] = root.count;
        for (int a = 0; a < 3; ++a) {
            sh.stack_box[0][a] = key2f(root.bmin[a]); sh.stack_box[0][3 + a] = key2f(root.bmax[a]);
            sh.stack_box[0][6 + a] = key2f(root.cmin[a]); sh.stack_box[0][9 + a] = key2f(root.cmax[a]);
        }
    }
    int sp = 1;
    uint32_t nsplit = 0, maxd = 1;
    bool hazard = false;
    const unsigned long long lt = (1ull << lane) - 1ull;
    __syncthreads();
    while (sp > 0) {  // (only nodes of three or more primitives are ever pushed: the others are never split, bvh.rs:56)
        --sp;
        const int id = sh.stack_id[sp];
        const uint32_t d = sh.stack_depth[sp], rel = sh.stack_rel[sp], count = sh.stack_count[sp];
        DBox nb, cb;
        for (int a = 0; a < 3; ++a) {
            nb.lo[a] = sh.stack_box[sp][a]; nb.hi[a] = sh.stack_box[sp][3 + a];
            cb.lo[a] = sh.stack_box[sp][6 + a]; cb.hi[a] = sh.stack_box[sp][9 + a];
        }
        // choose_split, bvh.rs:104-169
        for (uint32_t w = lane; w < (uint32_t)kBinWords; w += 64) ((uint32_t*)&sh.bins)[w] = bin_identity((int)w);
        __syncthreads();
        float delta[3];
        for (int a = 0; a < 3; ++a) delta[a] = (cb.hi[a] - cb.lo[a]) / 8.0f;
        for (uint32_t i = lane; i < count; i += 64) {
            const Rec r = sh.rec[rel + i];
            for (int a = 0; a < 3; ++a) {
                const float q = (r.c[a] - cb.lo[a]) / delta[a];
                const int b = a * 8 + (int)fminf(fmaxf(q, 0.0f), 7.0f);
                for (int k = 0; k < 3; ++k) {
                    key_min(&sh.bins.mn[b][k], r.mn[k]);
                    key_max(&sh.bins.mx[b][k], r.mx[k]);
                }
                atomicAdd(&sh.bins.n[b], 1u);
            }
        }
        __syncthreads();
        // the sweep of bvh.rs:135-168, one candidate (axis, bin) per lane: the reference takes a candidate when its cost is
        // below the best so far, axis after axis and bin after bin, i.e. the FIRST candidate with the smallest cost, if
        // that is below the parent's; a NaN cost is never taken
        const float parent_cost = nb.area() * (float)count;
        float cost = 0.0f;
        bool cand = false;
        if (lane < 24) {
            const uint32_t a8 = lane & ~7u, k = lane & 7u;
            DBox left, right;
            left.clear();
            right.clear();
            uint32_t nleft = 0, nright = 0;
            for (uint32_t j = 0; j < 8; ++j) {
                DBox bj;
                for (int c = 0; c < 3; ++c) { bj.lo[c] = key2f(sh.bins.mn[a8 + j][c]); bj.hi[c] = key2f(sh.bins.mx[a8 + j][c]); }
                const uint32_t nj = sh.bins.n[a8 + j];
                if (j <= k) { left.grow(bj); nleft += nj; }
                if (j >= k) { right.grow(bj); nright += nj; }  // right side includes bin k, bvh.rs:155
            }
            cost = (float)nleft * left.area() + (float)nright * right.area();
            cand = cost < parent_cost;
        }
        uint32_t who = lane;
        if (!cand) cost = __builtin_inff();
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {  // lanes 0 .. 31: smallest cost, the lower lane on a tie
            const float oc = __shfl_xor(cost, off, 64);
            const uint32_t ow = __shfl_xor(who, off, 64);
            const bool ocand = __shfl_xor((int)cand, off, 64) != 0;
            const bool take = ocand && (!cand || oc < cost || (oc == cost && ow < who));
            if (take) { cost = oc; who = ow; cand = true; }
        }
        const bool any = __shfl((int)cand, 0, 64) != 0;
        if (!any) continue;
        const uint32_t win = (uint32_t)__shfl((int)who, 0, 64);
        const int axis = (int)(win >> 3), bin = (int)(win & 7u);
        const float pivot = (axis == 0 ? cb.lo[0] : axis == 1 ? cb.lo[1] : cb.lo[2]) + (float)(1 + bin) * (axis == 0 ? delta[0] : axis == 1 ? delta[1] : delta[2]);
        // partition, bvh.rs:63-103
        Rec r[kSmallPerLane];
        bool val[kSmallPerLane], fl[kSmallPerLane];
        uint32_t rank[kSmallPerLane], nl = 0;
#pragma unroll
        for (int h = 0; h < kSmallPerLane; ++h) {
            val[h] = lane + 64u * h < count;
            r[h] = sh.rec[rel + (val[h] ? lane + 64u * h : 0u)];
            const float c = axis == 0 ? r[h].c[0] : axis == 1 ? r[h].c[1] : r[h].c[2];
            fl[h] = val[h] && c <= pivot;
            const unsigned long long m = __ballot(fl[h]);
            rank[h] = nl + (uint32_t)__popcll(m & lt);
            nl += (uint32_t)__popcll(m);
            if (fl[h]) sh.lpos[rank[h]] = lane + 64u * h;
        }
        if (lane < 12) {
            (&sh.cbk[0][0])[lane] = (lane % 6) < 3 ? kKeyPosInf : kKeyNegInf;
            (&sh.cck[0][0])[lane] = (lane % 6) < 3 ? kKeyPosInf : kKeyNegInf;
            (&sh.zs[0][0])[lane] = kNone;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < kSmallPerLane; ++h) {
            if (!val[h]) continue;
            const uint32_t p = lane + 64u * h;
            uint32_t q;
            if (fl[h]) {
                q = rank[h];
            } else {
                q = p;
                while (q < nl) q = sh.lpos[q];
            }
            sh.rec[rel + q] = r[h];
            const uint32_t right = q >= nl ? 1u : 0u;
            for (int a = 0; a < 3; ++a) {
                key_min(&sh.cbk[right][a], r[h].mn[a]);
                key_max(&sh.cbk[right][3 + a], r[h].mx[a]);
                key_min(&sh.cck[right][a], r[h].c[a]);
                key_max(&sh.cck[right][3 + a], r[h].c[a]);
                zero_vote(&sh.zs[right][a], r[h].mn[a], p);
                zero_vote(&sh.zs[right][3 + a], r[h].mx[a], p);
            }
        }
        __syncthreads();
        // An empty child (bvh.rs:87-97 when float rounding makes the binning and the `<= pivot` test disagree): the full child
        // would be split the same way again and again, past this subtree's block of nodes.  The wave stops before it writes
        // or pushes the children; pt_build then fails with PT_ERR_BVH (nl comes from ballots: the same in every lane).
        if (nl == 0 || nl == count) {
            hazard = true;
            break;
        }
        const uint32_t L = 2 * nsplit;
        float box[2][6];
        for (int ch = 0; ch < 2; ++ch)
            for (int a = 0; a < 6; ++a) {
                float f = key2f(sh.cbk[ch][a]);
                if (f == 0.0f && sh.zs[ch][a] != kNone && (sh.zs[ch][a] & 1u)) f = -0.0f;
                box[ch][a] = f;
            }
        if (lane == 0) {
            pt_bvh_node c;
            for (int a = 0; a < 3; ++a) { c.min[a] = box[0][a]; c.max[a] = box[0][3 + a]; }
            c.left_first = root.first + rel;
            c.count = nl;
            out[L] = c;
            for (int a = 0; a < 3; ++a) { c.min[a] = box[1][a]; c.max[a] = box[1][3 + a]; }
            c.left_first = root.first + rel + nl;
            c.count = count - nl;
            out[L + 1] = c;
            if (id >= 0) {  // the node becomes internal: link to its children's block places
                out[id].left_first = L;
                out[id].count = 0;
            }
        }
        nsplit += 1;
        maxd = max(maxd, d + 1);
        if (nsplit >= kSmall) break;  // (cannot happen: at most count - 1 splits of non-empty sides)
        const uint32_t cnt[2] = {nl, count - nl};
        if (lane == 0) {
            int e = sp;
            for (int ch = 1; ch >= 0; --ch) {  // the left child is split first: pushed last
                if (cnt[ch] <= 2) continue;   // never split
                sh.stack_id[e] = (int)L + ch;
                sh.stack_depth[e] = d + 1;
                sh.stack_rel[e] = ch ? rel + nl : rel;
                sh.stack_count[e] = cnt[ch];
                for (int a = 0; a < 6; ++a) {
                    sh.stack_box[e][a] = box[ch][a];
                    sh.stack_box[e][6 + a] = key2f(sh.cck[ch][a]);
                }
                e += 1;
            }
        }
        sp += (cnt[0] > 2 ? 1 : 0) + (cnt[1] > 2 ? 1 : 0);
        __syncthreads();
    }
    __syncthreads();
    for (uint32_t i = lane; i < root.count; i += 64) final_idx[root.first + i] = sh.rec[i].idx;
    if (lane == 0) {
        small_splits[v] = nsplit;
        atomicMax(&counters[5], small_level[t] + maxd - 1);
        if (hazard) atomicOr(&counters[2], 1u);
    }
}

__global__ void k_emit_small(const uint32_t* __restrict__ small_list, uint32_t n_small, const BNode* __restrict__ nodes, const pt_bvh_node* __restrict__ small_nodes,
                             const uint32_t* __restrict__ small_splits, const uint32_t* __restrict__ event, pt_bvh_node* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_small) return;
    const uint32_t v = small_list[t];
    const uint32_t s = small_splits[v], base = 1 + 2 * event[v];
    const pt_bvh_node* blk = small_nodes + 2 * (size_t)nodes[v].first;
    for (uint32_t j = 0; j < 2 * s; ++j) {
        pt_bvh_node nd = blk[j];
        if (nd.count == 0) nd.left_first += base;
        out[base + j] = nd;
    }
}

__global__ void k_flag(const Rec* __restrict__ rec, const uint32_t* __restrict__ seg, uint32_t n, const Split* __restrict__ split, uint32_t* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg[i];
    uint32_t f = 0;
    if (s != kNone) {
        const Split sp = split[s];
        if (sp.axis >= 0) f = rec[i].c[sp.axis] <= sp.pivot ? 1u : 0u;
    }
    flags[i] = f;
}

// ---- exclusive prefix sum of n flags into P[0..n] (P[n] = total): block sums, one block over the sums, apply
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;

__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* total) {  // 256 threads
    __shared__ uint32_t wsum[kBlock / 64];
    uint32_t inc = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int k = 0; k < kBlock / 64; ++k) {
        if (k < w) base += wsum[k];
        all += wsum[k];
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

__global__ void k_scan_sums(const uint32_t* __restrict__ flags, uint32_t n, uint32_t* sums) {
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v = 0;
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) v += flags[base + k];
    uint32_t total;
    (void)block_exclusive(v, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void k_scan_top(uint32_t* sums, uint32_t nb, uint32_t* P, uint32_t n) {  // one block
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += kBlock) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive(v, &total);
        const uint32_t c = carry;
        if (i < nb) sums[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) P[n] = carry;
}

__global__ void k_scan_apply(const uint32_t* __restrict__ flags, uint32_t n, const uint32_t* __restrict__ sums, uint32_t* P) {
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t f[kScanItems], v = 0;
    for (int k = 0; k < kScanItems; ++k) {
        f[k] = base + k < n ? flags[base + k] : 0;
        v += f[k];
    }
    uint32_t total;
    uint32_t run = sums[blockIdx.x] + block_exclusive(v, &total);
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) P[base + k] = run;
        run += f[k];
    }
}

// ---- children of the nodes that split; the next level's active list
__device__ __forceinline__ uint32_t child_slot(uint32_t count, uint32_t id, uint32_t level, uint32_t* counters, uint32_t* next_active, uint32_t* small_list,
                                               uint32_t* small_level, uint32_t small_limit) {
    if (count <= 2) return kNone;  // never split
    if (count <= small_limit) {    // one wave finishes it (k_small_wave)
        const uint32_t k = atomicAdd(&counters[4], 1u);
        small_list[k] = id;
        small_level[k] = level;
        return kSmallSlot;
    }
    const uint32_t slot = atomicAdd(&counters[1], 1u);
    next_active[slot] = id;
    return slot;
}

__global__ void k_children(const uint32_t* __restrict__ active, uint32_t n_active, BNode* nodes, Split* split, const uint32_t* __restrict__ P,
                           uint32_t* counters /* [0] nodes, [1] next active, [2] hazard, [4] small roots, [5] deepest small level */,
                           uint32_t* next_active, uint32_t* small_list, uint32_t* small_level, uint32_t child_level, uint32_t small_limit) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_active) return;
    const uint32_t v = active[s];
    Split sp = split[s];
    if (sp.axis < 0) {
        nodes[v].slot = kNone;
        return;
    }
    const uint32_t first = nodes[v].first, count = nodes[v].count;
    const uint32_t nl = P[first + count] - P[first];
    if (nl == 0 || nl == count) {
        // the empty-child hazard: no children (the full one would be split the same way for ever, one level per pass and
        // two nodes each, past the node array); the node is left as it is and the host ends the build after this level
        atomicOr(&counters[2], 1u);
        sp.axis = -1;
        split[s] = sp;
        nodes[v].slot = kNone;
        return;
    }
    const uint32_t pair = atomicAdd(&counters[0], 2u);
    BNode c;
    for (int a = 0; a < 3; ++a) { c.bmin[a] = c.cmin[a] = kKeyPosInf; c.bmax[a] = c.cmax[a] = kKeyNegInf; }
    for (int a = 0; a < 6; ++a) c.zs[a] = kNone;
    c.left = 0;
    c.first = first;
    c.count = nl;
    c.slot = child_slot(nl, pair, child_level, counters, next_active, small_list, small_level, small_limit);
    nodes[pair] = c;
    sp.left_slot = c.slot;
    c.first = first + nl;
    c.count = count - nl;
    c.slot = child_slot(c.count, pair + 1, child_level, counters, next_active, small_list, small_level, small_limit);
    nodes[pair + 1] = c;
    sp.right_slot = c.slot;
    sp.n_left = nl;
    sp.pair = pair;
    split[s] = sp;
    nodes[v].left = pair;
    nodes[v].slot = kNone;
}

__global__ void k_lpos(const uint32_t* __restrict__ seg, uint32_t n, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ P,
                       const uint32_t* __restrict__ active, const BNode* __restrict__ nodes, uint32_t* lpos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const uint32_t first = nodes[active[seg[i]]].first;  // a set flag implies an active, split node
    lpos[first + (P[i] - P[first])] = i - first;
}

// one pass: every active primitive to its place in the next level, child bounds reduced on the way (per wave or per
// workgroup in LDS when it lies inside one node, as in k_bin)
__global__ void k_scatter(const Rec* __restrict__ rec, const uint32_t* __restrict__ seg, uint32_t n, const uint32_t* __restrict__ flags,
                          const uint32_t* __restrict__ P, const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ active, BNode* nodes,
                          const Split* __restrict__ split, Rec* rec_next, uint32_t* seg_next, uint32_t* final_idx, Rec* small_rec) {
    __shared__ uint32_t acc[kWaves][2][24];  // per wave and node of the wave: [child][bmin3 bmax3 cmin3 cmax3]
    __shared__ uint32_t s0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t s = i < n ? seg[i] : kNone;
    if (threadIdx.x == 0) s0 = s;
    if (threadIdx.x < kWaves * 48) (&acc[0][0][0])[threadIdx.x] = (threadIdx.x % 6) < 3 ? kKeyPosInf : kKeyNegInf;
    __syncthreads();
    const WaveSegs ws = wave_segs(s);
    const uint32_t block_s = s0;
    const bool block_uniform = __syncthreads_and((i >= n || s == block_s) ? 1 : 0) != 0 && block_s != kNone;
    if (i < n) {
        if (s == kNone) {
            seg_next[i] = kNone;
        } else {
            const Split sp = split[s];
            const Rec r = rec[i];
            if (sp.axis < 0) {  // the node stays a leaf: its primitives are where they will be
                final_idx[i] = r.idx;
                seg_next[i] = kNone;
            } else {
                const uint32_t first = nodes[active[s]].first;
                uint32_t q;
                if (flags[i]) {
                    q = P[i] - P[first];
                } else {
                    q = i - first;
                    while (q < sp.n_left) q = lpos[first + q];
                }
                const uint32_t right = q >= sp.n_left ? 1u : 0u;
                const uint32_t cs = right ? sp.right_slot : sp.left_slot;
                const uint32_t dest = first + q;
                if (cs == kNone) final_idx[dest] = r.idx;  // a child of one or two primitives is never split
                else if (cs == kSmallSlot) small_rec[dest] = r;
                else rec_next[dest] = r;
                seg_next[dest] = cs == kSmallSlot ? kNone : cs;
                uint32_t* b = ws.two ? acc[wave][s == ws.a ? 0 : 1] + 12 * right : nodes[sp.pair + right].bmin;  // bmin, bmax, cmin, cmax are consecutive
                uint32_t* zs = nodes[sp.pair + right].zs;
                for (int a = 0; a < 3; ++a) {
                    key_min(b + a, r.mn[a]);
                    key_max(b + 3 + a, r.mx[a]);
                    key_min(b + 6 + a, r.c[a]);
                    key_max(b + 9 + a, r.c[a]);
                    zero_vote(zs + a, r.mn[a], i);      // i: where the reference's walk over the parent meets this object
                    zero_vote(zs + 3 + a, r.mx[a], i);
                }
            }
        }
    }
    __syncthreads();
    // flush: the untouched identities are skipped, so a node that did not split (nothing accumulated) writes nothing
    if (block_uniform) {
        if (threadIdx.x < 24) {
            const bool is_min = (threadIdx.x % 6) < 3;
            uint32_t v = acc[0][0][threadIdx.x];
            for (int k = 1; k < kWaves; ++k) v = is_min ? min(v, acc[k][0][threadIdx.x]) : max(v, acc[k][0][threadIdx.x]);
            if (v != (is_min ? kKeyPosInf : kKeyNegInf)) {
                uint32_t* dst = nodes[split[block_s].pair + threadIdx.x / 12].bmin + threadIdx.x % 12;
                if (is_min) atomicMin(dst, v);
                else atomicMax(dst, v);
            }
        }
    } else if (ws.two && lane < 48) {
        const int which = lane / 24, w = lane % 24;
        const uint32_t node_s = which ? ws.b : ws.a;
        const bool is_min = (w % 6) < 3;
        const uint32_t v = acc[wave][which][w];
        if (node_s != kNone && !(which && ws.b == ws.a) && v != (is_min ? kKeyPosInf : kKeyNegInf)) {
            uint32_t* dst = nodes[split[node_s].pair + w / 12].bmin + w % 12;
            if (is_min) atomicMin(dst, v);
            else atomicMax(dst, v);
        }
    }
}

// ---- numbering (see the header comment): splits per subtree bottom-up, split events and ids top-down, then the output
__global__ void k_count_splits(const BNode* __restrict__ nodes, uint32_t lo, uint32_t hi, const uint32_t* __restrict__ small_splits, uint32_t* splits) {
    const uint32_t v = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= hi) return;
    const uint32_t l = nodes[v].left;
    splits[v] = l ? 1u + splits[l] + splits[l + 1] : small_splits[v];
}

__global__ void k_number(const BNode* __restrict__ nodes, uint32_t lo, uint32_t hi, const uint32_t* __restrict__ splits, uint32_t* event, uint32_t* newid) {
    const uint32_t v = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= hi) return;
    const uint32_t l = nodes[v].left;
    if (!l) return;
    const uint32_t e = event[v];
    event[l] = e + 1;
    event[l + 1] = e + 1 + splits[l];
    newid[l] = 1 + 2 * e;
    newid[l + 1] = 2 + 2 * e;
}

__global__ void k_emit(const BNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ newid, const uint32_t* __restrict__ event,
                       const uint32_t* __restrict__ small_splits, pt_bvh_node* out) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    const BNode nd = nodes[v];
    pt_bvh_node o;
    for (int a = 0; a < 3; ++a) {
        o.min[a] = key2f(nd.bmin[a]);
        o.max[a] = key2f(nd.bmax[a]);
        if (o.min[a] == 0.0f && nd.zs[a] != kNone && (nd.zs[a] & 1u)) o.min[a] = -0.0f;
        if (o.max[a] == 0.0f && nd.zs[3 + a] != kNone && (nd.zs[3 + a] & 1u)) o.max[a] = -0.0f;
    }
    const bool small_split = !nd.left && small_splits[v] != 0;  // split by k_small_wave: its children head its block
    o.left_first = nd.left ? newid[nd.left] : (small_split ? 1 + 2 * event[v] : nd.first);
    o.count = (nd.left || small_split) ? 0u : nd.count;
    out[newid[v]] = o;
}

// ---- primitives into leaf order (what the traversal kernels read): slot s holds object indices[s]
__global__ void k_reorder_tri(const float* __restrict__ pos, const float* __restrict__ nrm, const uint32_t* __restrict__ mat, const uint32_t* __restrict__ idx,
                              uint32_t n, float* __restrict__ prim, float* __restrict__ nrm_out, uint32_t* __restrict__ slot_of) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t obj = idx[s];
    const float* p = pos + 9 * (size_t)obj;
    float* r = prim + 4 * (size_t)kTriStride * s;
    const float p0[3] = {p[0], p[1], p[2]};
    for (int k = 0; k < 3; ++k) {
        r[k] = p0[k];
        r[3 + k] = p[3 + k] - p0[k];  // edge1 = position1 - position0
        r[6 + k] = p[6 + k] - p0[k];  // edge2 = position2 - position0
    }
    r[9] = __uint_as_float(obj);
    r[10] = __uint_as_float(mat[obj]);
    r[11] = 0.0f;
    if (kNrmStride == 16) {
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 4; ++k) nrm_out[16 * (size_t)s + 4 * v + k] = k < 3 ? nrm[9 * (size_t)obj + 3 * v + k] : 0.0f;
        for (int k = 12; k < 16; ++k) nrm_out[16 * (size_t)s + k] = 0.0f;
    } else {
        for (int k = 0; k < 9; ++k) nrm_out[9 * (size_t)s + k] = nrm[9 * (size_t)obj + k];
    }
    slot_of[obj] = s;
}

__global__ void k_reorder_sph(const float* __restrict__ cr, const uint32_t* __restrict__ mat, const uint32_t* __restrict__ idx, uint32_t n,
                              float* __restrict__ prim, uint32_t* __restrict__ meta, uint32_t* __restrict__ slot_of) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t obj = idx[s];
    for (int k = 0; k < 4; ++k) prim[4 * (size_t)s + k] = cr[4 * (size_t)obj + k];
    meta[2 * (size_t)s] = obj;
    meta[2 * (size_t)s + 1] = mat[obj];
    slot_of[obj] = s;
}

// ---- flattening (pt_bvh.cpp flatten_bvh) on the device: one 64-byte record per internal node holding its two children
__global__ void k_node_flags(const pt_bvh_node* __restrict__ nodes, uint32_t n_nodes, uint32_t big_from, uint32_t* internal, uint32_t* big) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t cnt = nodes[i].count;
    internal[i] = cnt == 0 ? 1u : 0u;
    big[i] = (cnt != 0 && !(cnt < big_from && cnt <= 31)) ? 1u : 0u;
}

__device__ __forceinline__ uint32_t make_ref(const pt_bvh_node& nd, uint32_t id, const uint32_t* pair_of, const uint32_t* big_of, uint32_t big_from,
                                             uint32_t* big_leaf) {
    if (nd.count == 0) return kPairDfs ? nd.left_first >> 1 : pair_of[id];
    if (nd.count < big_from && nd.count <= 31) return kLeafBit | (nd.count << kLeafCountShift) | nd.left_first;
    const uint32_t slot = big_of[id];
    big_leaf[2 * slot] = nd.left_first;
    big_leaf[2 * slot + 1] = nd.count;
    return kLeafBit | slot;
}

__global__ void k_flatten(const pt_bvh_node* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ pair_of, const uint32_t* __restrict__ big_of,
                          uint32_t big_from, PairRecord* pairs, uint32_t* big_leaf, uint32_t* root_ref) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const pt_bvh_node nd = nodes[i];
    if (i == 0) *root_ref = make_ref(nd, 0, pair_of, big_of, big_from, big_leaf);
    if (nd.count != 0) return;
    const pt_bvh_node l = nodes[nd.left_first], r = nodes[nd.left_first + 1];
    PairRecord p;
    for (int a = 0; a < 3; ++a) {
        p.lmin[a] = l.min[a]; p.lmax[a] = l.max[a];
        p.rmin[a] = r.min[a]; p.rmax[a] = r.max[a];
    }
    p.lref = make_ref(l, nd.left_first, pair_of, big_of, big_from, big_leaf);
    p.rref = make_ref(r, nd.left_first + 1, pair_of, big_of, big_from, big_leaf);
    p.lpad = p.rpad = 0;
    pairs[kPairDfs ? nd.left_first >> 1 : pair_of[i]] = p;
}

struct Buf {
    void* p = nullptr;
    bool owned = true;  // false: a piece of an Arena
    ~Buf() { if (p && owned) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T* as() const { return (T*)p; }
    void* take() { void* q = p; p = nullptr; return q; }
};

// The temporaries of one build come out of ONE device allocation (two dozen hipMalloc / hipFree pairs cost 3 ms, a
// quarter of the time of a 1 M-triangle tree): 256-byte aligned pieces, freed together.
struct Arena {
    char* base = nullptr;
    std::vector<std::pair<Buf*, size_t>> want;
    ~Arena() { if (base) (void)hipFree(base); }
    void add(Buf& b, size_t bytes) { want.push_back({&b, (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255}); }
    hipError_t commit() {
        size_t total = 0;
        for (const auto& w : want) total += w.second;
        hipError_t e = hipMalloc((void**)&base, total);
        if (e != hipSuccess) return e;
        size_t at = 0;
        for (const auto& w : want) {
            w.first->p = base + at;
            w.first->owned = false;
            at += w.second;
        }
        return hipSuccess;
    }
};

// exclusive prefix sum of n flags into P[0..n]
void scan_flags(const uint32_t* flags, uint32_t n, uint32_t* sums, uint32_t* P, hipStream_t st) {
    const uint32_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(kBlock), 0, st, flags, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, st, sums, nb, P, n);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(kBlock), 0, st, flags, n, (const uint32_t*)sums, P);
}

inline uint32_t blocks(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

}  // namespace

#define BUILD_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            if (err) *err = std::string("device build: " #call ": ") + hipGetErrorString(e_);   \
            return PT_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

void launch_reorder_triangles(const float* pos, const float* nrm, const uint32_t* mat, const uint32_t* idx, uint32_t n, float* prim, float* nrm_out,
                              uint32_t* slot_of, hipStream_t st) {
    hipLaunchKernelGGL(k_reorder_tri, dim3(blocks(n)), dim3(kBlock), 0, st, pos, nrm, mat, idx, n, prim, nrm_out, slot_of);
}

void launch_reorder_spheres(const float* cr, const uint32_t* mat, const uint32_t* idx, uint32_t n, float* prim, uint32_t* meta, uint32_t* slot_of,
                            hipStream_t st) {
    hipLaunchKernelGGL(k_reorder_sph, dim3(blocks(n)), dim3(kBlock), 0, st, cr, mat, idx, n, prim, meta, slot_of);
}

void DeviceTree::release() {
    if (nodes) (void)hipFree(nodes);
    if (indices) (void)hipFree(indices);
    nodes = nullptr;
    indices = nullptr;
    n_nodes = n_prims = max_depth = 0;
    hazard = false;
}

int build_bvh_device(int prim_kind, const float* d_geom, uint32_t n, DeviceTree& out, hipStream_t st, std::string* err) {
    out.release();
    if (n == 0 || n > kMaxPrims || !d_geom) return PT_ERR_INVALID;
    static_assert(sizeof(Rec) == 40 && sizeof(BNode) == 88 && sizeof(Bins) == 672, "layout");

    const bool dbg = std::getenv("PTCORE_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (dbg) std::fprintf(stderr, "[ptcore] device build: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // a node on the active list holds more than kSmall primitives, the root of a small subtree at least three
    const uint32_t max_nodes = 2 * n + 1, max_active = n / (kSmall + 1) + 2, max_small = n / 3 + 2, nb = (n + kScanTile - 1) / kScanTile;
    Buf rec_a, rec_b, seg_a, seg_b, flags, P, lpos, sums, nodes, act_a, act_b, bins, split, counters, final_idx, splits, event, newid, outn;
    Buf small_rec, small_nodes, small_list, small_level, small_splits;
    Arena arena;  // declared after the Bufs: released first, while they still only point into it
    arena.add(rec_a, (size_t)n * sizeof(Rec));
    arena.add(rec_b, (size_t)n * sizeof(Rec));
    arena.add(seg_a, (size_t)n * 4);
    arena.add(seg_b, (size_t)n * 4);
    arena.add(flags, (size_t)n * 4);
    arena.add(P, ((size_t)n + 1) * 4);
    arena.add(lpos, (size_t)n * 4);
    arena.add(sums, (size_t)nb * 4);
    arena.add(nodes, (size_t)max_nodes * sizeof(BNode));
    arena.add(act_a, (size_t)max_active * 4);
    arena.add(act_b, (size_t)max_active * 4);
    arena.add(bins, (size_t)max_active * sizeof(Bins));
    arena.add(split, (size_t)max_active * sizeof(Split));
    arena.add(counters, 32);
    arena.add(small_rec, (size_t)n * sizeof(Rec));
    arena.add(small_nodes, (size_t)n * 2 * sizeof(pt_bvh_node));
    arena.add(small_list, (size_t)max_small * 4);
    arena.add(small_level, (size_t)max_small * 4);
    arena.add(small_splits, (size_t)max_nodes * 4);
    arena.add(splits, (size_t)max_nodes * 4);  // (the level nodes are at most max_nodes)
    arena.add(event, (size_t)max_nodes * 4);
    arena.add(newid, (size_t)max_nodes * 4);
    BUILD_HIP(arena.commit());
    BUILD_HIP(final_idx.alloc((size_t)n * 4));  // becomes the tree's index array
    BUILD_HIP(hipMemsetAsync(small_splits.p, 0, (size_t)max_nodes * 4, st));
    uint32_t h_cnt[8] = {1, 0, 0, 0, 0, 0, 0, 0};  // nodes, next active, hazard, -, small roots, deepest level below a small root
    BUILD_HIP(hipMemcpyAsync(counters.p, h_cnt, 32, hipMemcpyHostToDevice, st));
    if (prim_kind == 0) hipLaunchKernelGGL(k_recs_tri, dim3(blocks(n)), dim3(kBlock), 0, st, d_geom, n, rec_a.as<Rec>());
    else hipLaunchKernelGGL(k_recs_sph, dim3(blocks(n)), dim3(kBlock), 0, st, d_geom, n, rec_a.as<Rec>());
    hipLaunchKernelGGL(k_init_root, dim3(blocks(n)), dim3(kBlock), 0, st, nodes.as<BNode>(), n, seg_a.as<uint32_t>(), act_a.as<uint32_t>());
    hipLaunchKernelGGL(k_root, dim3(std::min<uint32_t>(blocks(n), 2048)), dim3(kBlock), 0, st, rec_a.as<Rec>(), n, nodes.as<BNode>());
    BUILD_HIP(hipMemcpyAsync(h_cnt, counters.p, 16, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipStreamSynchronize(st));
    BUILD_HIP(hipGetLastError());
    tick("alloc, records, root");
    std::vector<uint32_t> level_start{0, 1};  // node ids of level L are [level_start[L], level_start[L + 1])
    // Subtrees of at most kSmall primitives are finished by one wave each (k_small_wave).  Measured with limits of 16 / 32 /
    // 64 / 128 / 256 / 512: 10 M triangles 85 / 74 / 62 / 54 / 60 / 100 ms, 871 k triangles 12.8 / 11.9 / 10.7 / 9.8 / 11.0 / 15.3 ms
    // (with the first wave kernel; the present one: 128 -> 41 ms, 256 -> 48 ms).
    const uint32_t small_limit = kSmall;
    uint32_t n_active = n > small_limit ? 1 : 0;
    Buf *rc = &rec_a, *rn = &rec_b, *sc = &seg_a, *sn = &seg_b, *ac = &act_a, *an = &act_b;
    if (n > 2 && n <= small_limit) {  // the root itself is a small subtree: one wave builds all of it
        const uint32_t one[2] = {0u, 1u};
        BUILD_HIP(hipMemcpyAsync(small_list.p, &one[0], 4, hipMemcpyHostToDevice, st));
        BUILD_HIP(hipMemcpyAsync(small_level.p, &one[1], 4, hipMemcpyHostToDevice, st));
        BUILD_HIP(hipMemcpyAsync(counters.as<uint32_t>() + 4, &one[1], 4, hipMemcpyHostToDevice, st));
        BUILD_HIP(hipMemcpyAsync(small_rec.p, rec_a.p, (size_t)n * sizeof(Rec), hipMemcpyDeviceToDevice, st));
        BUILD_HIP(hipStreamSynchronize(st));
    }
    if (n <= 2) {  // the root is the only node: its primitives stay in input order
        std::vector<uint32_t> id(n);
        for (uint32_t i = 0; i < n; ++i) id[i] = i;
        BUILD_HIP(hipMemcpyAsync(final_idx.p, id.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
        BUILD_HIP(hipStreamSynchronize(st));
    }
    while (n_active) {
        hipLaunchKernelGGL(k_init_bins, dim3(blocks((uint64_t)n_active * kBinWords)), dim3(kBlock), 0, st, bins.as<Bins>(), n_active);
        hipLaunchKernelGGL(k_bin, dim3(blocks(n)), dim3(kBlock), 0, st, rc->as<Rec>(), sc->as<uint32_t>(), n, nodes.as<BNode>(), ac->as<uint32_t>(), bins.as<Bins>());
        hipLaunchKernelGGL(k_sah, dim3(blocks(n_active)), dim3(kBlock), 0, st, ac->as<uint32_t>(), n_active, nodes.as<BNode>(), bins.as<Bins>(), split.as<Split>());
        hipLaunchKernelGGL(k_flag, dim3(blocks(n)), dim3(kBlock), 0, st, rc->as<Rec>(), sc->as<uint32_t>(), n, split.as<Split>(), flags.as<uint32_t>());
        scan_flags(flags.as<uint32_t>(), n, sums.as<uint32_t>(), P.as<uint32_t>(), st);
        BUILD_HIP(hipMemsetAsync(counters.as<uint32_t>() + 1, 0, 4, st));
        hipLaunchKernelGGL(k_children, dim3(blocks(n_active)), dim3(kBlock), 0, st, ac->as<uint32_t>(), n_active, nodes.as<BNode>(), split.as<Split>(), P.as<uint32_t>(),
                           counters.as<uint32_t>(), an->as<uint32_t>(), small_list.as<uint32_t>(), small_level.as<uint32_t>(), (uint32_t)level_start.size(), small_limit);
        hipLaunchKernelGGL(k_lpos, dim3(blocks(n)), dim3(kBlock), 0, st, sc->as<uint32_t>(), n, flags.as<uint32_t>(), P.as<uint32_t>(), ac->as<uint32_t>(), nodes.as<BNode>(),
                           lpos.as<uint32_t>());
        hipLaunchKernelGGL(k_scatter, dim3(blocks(n)), dim3(kBlock), 0, st, rc->as<Rec>(), sc->as<uint32_t>(), n, flags.as<uint32_t>(), P.as<uint32_t>(), lpos.as<uint32_t>(),
                           ac->as<uint32_t>(), nodes.as<BNode>(), split.as<Split>(), rn->as<Rec>(), sn->as<uint32_t>(), final_idx.as<uint32_t>(), small_rec.as<Rec>());
        BUILD_HIP(hipMemcpyAsync(h_cnt, counters.p, 12, hipMemcpyDeviceToHost, st));
        BUILD_HIP(hipStreamSynchronize(st));
        BUILD_HIP(hipGetLastError());
        if (h_cnt[2]) {
            if (err) *err = "pt_build: a partition produced an empty child";
            return PT_ERR_BVH;
        }
        if (h_cnt[0] > level_start.back()) level_start.push_back(h_cnt[0]);
        n_active = h_cnt[1];
        std::swap(rc, rn);
        std::swap(sc, sn);
        std::swap(ac, an);
        if (h_cnt[0] > max_nodes || n_active > max_active) {
            if (err) *err = "device build: node count out of range";
            return PT_ERR_BVH;
        }
    }
    const uint32_t n_level_nodes = h_cnt[0], levels = (uint32_t)level_start.size() - 1;
    tick("levels");
    BUILD_HIP(hipMemcpyAsync(h_cnt, counters.p, 32, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipStreamSynchronize(st));
    const uint32_t n_small = h_cnt[4];
    if (n_small)
        hipLaunchKernelGGL(k_small_wave, dim3(n_small), dim3(64), 0, st, small_list.as<uint32_t>(), small_level.as<uint32_t>(), n_small, nodes.as<BNode>(),
                           small_rec.as<Rec>(), small_nodes.as<pt_bvh_node>(), small_splits.as<uint32_t>(), final_idx.as<uint32_t>(), counters.as<uint32_t>());

    BUILD_HIP(hipMemsetAsync(event.p, 0, 4, st));  // the root: split event 0, node 0
    BUILD_HIP(hipMemsetAsync(newid.p, 0, 4, st));
    for (uint32_t L = levels; L-- > 0;)
        hipLaunchKernelGGL(k_count_splits, dim3(blocks(level_start[L + 1] - level_start[L])), dim3(kBlock), 0, st, nodes.as<BNode>(), level_start[L], level_start[L + 1],
                           small_splits.as<uint32_t>(), splits.as<uint32_t>());
    uint32_t total_splits = 0;
    BUILD_HIP(hipMemcpyAsync(&total_splits, splits.p, 4, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipMemcpyAsync(h_cnt, counters.p, 32, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipStreamSynchronize(st));
    BUILD_HIP(hipGetLastError());
    const uint32_t n_nodes = 1 + 2 * total_splits;  // every split adds two nodes
    BUILD_HIP(outn.alloc((size_t)n_nodes * sizeof(pt_bvh_node)));
    for (uint32_t L = 0; L < levels; ++L)
        hipLaunchKernelGGL(k_number, dim3(blocks(level_start[L + 1] - level_start[L])), dim3(kBlock), 0, st, nodes.as<BNode>(), level_start[L], level_start[L + 1],
                           splits.as<uint32_t>(), event.as<uint32_t>(), newid.as<uint32_t>());
    hipLaunchKernelGGL(k_emit, dim3(blocks(n_level_nodes)), dim3(kBlock), 0, st, nodes.as<BNode>(), n_level_nodes, newid.as<uint32_t>(), event.as<uint32_t>(),
                       small_splits.as<uint32_t>(), outn.as<pt_bvh_node>());
    if (n_small)
        hipLaunchKernelGGL(k_emit_small, dim3(blocks(n_small)), dim3(kBlock), 0, st, small_list.as<uint32_t>(), n_small, nodes.as<BNode>(), small_nodes.as<pt_bvh_node>(),
                           small_splits.as<uint32_t>(), event.as<uint32_t>(), outn.as<pt_bvh_node>());
    BUILD_HIP(hipStreamSynchronize(st));
    BUILD_HIP(hipGetLastError());
    out.nodes = (pt_bvh_node*)outn.take();
    out.indices = (uint32_t*)final_idx.take();
    out.n_nodes = n_nodes;
    out.n_prims = n;
    out.max_depth = std::max(levels, h_cnt[5]);
    out.hazard = h_cnt[2] != 0;
    tick("numbering");
    return PT_OK;
}

int flatten_bvh_device(const DeviceTree& t, uint32_t big_from, DeviceFlat& out, hipStream_t st, std::string* err) {
    const uint32_t nn = t.n_nodes;
    Buf internal, big, pair_of, big_of, sums, root_ref, pairs, big_leaf;
    Arena arena;
    arena.add(internal, (size_t)nn * 4);
    arena.add(big, (size_t)nn * 4);
    arena.add(pair_of, ((size_t)nn + 1) * 4);
    arena.add(big_of, ((size_t)nn + 1) * 4);
    arena.add(sums, ((size_t)(nn + kScanTile - 1) / kScanTile) * 4);
    arena.add(root_ref, 4);
    BUILD_HIP(arena.commit());
    hipLaunchKernelGGL(k_node_flags, dim3(blocks(nn)), dim3(kBlock), 0, st, (const pt_bvh_node*)t.nodes, nn, big_from, internal.as<uint32_t>(), big.as<uint32_t>());
    scan_flags(internal.as<uint32_t>(), nn, sums.as<uint32_t>(), pair_of.as<uint32_t>(), st);
    scan_flags(big.as<uint32_t>(), nn, sums.as<uint32_t>(), big_of.as<uint32_t>(), st);
    uint32_t counts[2] = {0, 0};
    BUILD_HIP(hipMemcpyAsync(&counts[0], pair_of.as<uint32_t>() + nn, 4, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipMemcpyAsync(&counts[1], big_of.as<uint32_t>() + nn, 4, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipStreamSynchronize(st));
    if (kPairDfs) counts[0] = (nn + 1) / 2;  // records are indexed by left_first >> 1 < nn / 2
    counts[0] += 1;                          // + 1: the record behind the last one may be touched (prefetch)
    BUILD_HIP(pairs.alloc((size_t)counts[0] * sizeof(PairRecord)));
    BUILD_HIP(hipMemsetAsync(pairs.p, 0, (size_t)counts[0] * sizeof(PairRecord), st));
    BUILD_HIP(big_leaf.alloc((size_t)counts[1] * 8));
    hipLaunchKernelGGL(k_flatten, dim3(blocks(nn)), dim3(kBlock), 0, st, (const pt_bvh_node*)t.nodes, nn, pair_of.as<uint32_t>(), big_of.as<uint32_t>(), big_from,
                       pairs.as<PairRecord>(), big_leaf.as<uint32_t>(), root_ref.as<uint32_t>());
    pt_bvh_node root;
    BUILD_HIP(hipMemcpyAsync(&out.root_ref, root_ref.p, 4, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipMemcpyAsync(&root, t.nodes, sizeof(root), hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipStreamSynchronize(st));
    BUILD_HIP(hipGetLastError());
    for (int a = 0; a < 3; ++a) { out.root_min[a] = root.min[a]; out.root_max[a] = root.max[a]; }
    out.pairs_bytes = std::max<size_t>((size_t)counts[0] * sizeof(PairRecord), 16);
    out.big_bytes = std::max<size_t>((size_t)counts[1] * 8, 16);
    out.pairs = pairs.take();
    out.big_leaf = big_leaf.take();
    return PT_OK;
}

int download_tree(const DeviceTree& t, HostBVH& out, hipStream_t st, std::string* err) {
    out.nodes.resize(t.n_nodes);
    out.indices.resize(t.n_prims);
    BUILD_HIP(hipMemcpyAsync(out.nodes.data(), t.nodes, (size_t)t.n_nodes * sizeof(pt_bvh_node), hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipMemcpyAsync(out.indices.data(), t.indices, (size_t)t.n_prims * 4, hipMemcpyDeviceToHost, st));
    BUILD_HIP(hipStreamSynchronize(st));
    out.max_depth = t.max_depth;
    out.hazard = t.hazard;
    return PT_OK;
}

}  // namespace pt
