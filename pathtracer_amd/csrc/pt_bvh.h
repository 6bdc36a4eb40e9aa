// pt_bvh.h — host-side binned-SAH BVH build and flattening for the device.
//
// build() reproduces BVH::new / subdivide / partition / surface_area_heuristic of
// /root/reference/src/bvh.rs:25-179 bit for bit (same tree, same leaf order): the node array it
// returns uses the reference's own 32-byte record (AABB + left_first + count, bvh.rs:10-14).
// flatten() turns that tree into the device layout (see DESIGN.md "Data layout in HBM").
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/ptcore.h"

namespace pt {

// Primitive::bounds() and Primitive::centre() of one object, computed once (both are pure
// functions of the object; the reference recomputes them on every use).
struct PrimInfo {
    float c[3];
    float mn[3];
    float mx[3];
};

struct HostBVH {
    std::vector<pt_bvh_node> nodes;   // reference layout, reference numbering
    std::vector<uint32_t> indices;    // bvh.indices after the build (leaf order -> object index)
    std::vector<uint32_t> lights;     // bvh.lights (object indices), bvh.rs:29-35
    uint32_t max_depth = 0;           // levels, root = 1
    bool hazard = false;              // a partition produced an empty side (bvh.rs:87-97 would mis-traverse)
};

void prim_info_triangles(const float* pos, uint64_t n, std::vector<PrimInfo>& out);
void prim_info_spheres(const float* cr, uint64_t n, std::vector<PrimInfo>& out);
// threads <= 0: PTCORE_BUILD_THREADS or min(16, hardware threads).  The tree does not depend on the thread count.
void build_bvh(const std::vector<PrimInfo>& prims, HostBVH& out, int threads = 0);

// Device references.  A child reference is one u32:
//   bit 31 = 0 : internal node, bits 30..0 = index of its 64-byte child-pair record
//   bit 31 = 1 : leaf, bits 30..26 = primitive count (1..31; 0 = look up kBigLeaf table,
//                bits 25..0 then index that table), bits 25..0 = first primitive slot
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafCountShift = 26;
constexpr uint32_t kLeafFirstMask = (1u << 26) - 1;
constexpr uint64_t kMaxPrims = 1ull << 26;

// One record per internal node: its two children, each as a 32-byte AABB node whose
// (left_first, count) pair is replaced by (reference, 0).
// Which record of `pairs` an internal node gets.  0: its rank among the internal nodes in node order.  1: its split event,
// left_first >> 1 — the reference appends a node's children when it splits the node and splits depth first, left child
// first (bvh.rs:52-62,87-97), so children are nodes 1 + 2e, 2 + 2e of split event e and records in event order are the
// tree in DFS pre-order: the record after a node's is its left child's (or, below a leaf, the next subtree's).
#ifndef PT_PAIR_DFS
#define PT_PAIR_DFS 1
#endif
constexpr bool kPairDfs = PT_PAIR_DFS != 0;

struct alignas(16) PairRecord {
    float lmin[3];
    uint32_t lref;
    float lmax[3];
    uint32_t lpad;
    float rmin[3];
    uint32_t rref;
    float rmax[3];
    uint32_t rpad;
};
static_assert(sizeof(PairRecord) == 64, "pair record must be 64 bytes");

struct FlatBVH {
    std::vector<PairRecord> pairs;
    std::vector<uint32_t> big_leaf;  // (first, count) pairs of the leaves referenced through the escape encoding (count field 0)
    float root_min[3], root_max[3];
    uint32_t root_ref = 0;
};

// Leaves of `big_from` primitives or more (and always those above 31, which the 5-bit count field cannot hold) are
// referenced through the escape encoding: the persistent kernels test such a leaf with the whole wave at once.
int flatten_bvh(const HostBVH& bvh, FlatBVH& out, uint32_t big_from = 32);

}  // namespace pt
