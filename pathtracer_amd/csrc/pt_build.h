// pt_build.h — the BVH build on the GPU (pt_build.hip): same tree as build_bvh() of pt_bvh.h, bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "pt_bvh.h"

namespace pt {

// What a device build leaves on the GPU (owned by the holder; release() frees it).
struct DeviceTree {
    pt_bvh_node* nodes = nullptr;  // the reference's node array, the reference's numbering
    uint32_t* indices = nullptr;   // bvh.indices after the build
    uint32_t n_nodes = 0, n_prims = 0, max_depth = 0;
    bool hazard = false;
    void release();
};

// The flattened tree (pt_bvh.h FlatBVH) in device memory; pairs / big_leaf are hipMalloc'ed and handed to the caller.
struct DeviceFlat {
    void* pairs = nullptr;
    void* big_leaf = nullptr;
    size_t pairs_bytes = 0, big_bytes = 0;
    uint32_t root_ref = 0;
    float root_min[3], root_max[3];
};

// Builds the tree on the current device from geometry that is already there (prim_kind 0: n x 9 triangle positions,
// 1: n x 4 spheres); work is enqueued on `st` and has finished on return.  Negative zeros, infinities and NaNs in the
// input give the reference's result too (pt_build.hip, header comment).
int build_bvh_device(int prim_kind, const float* d_geom, uint32_t n, DeviceTree& out, hipStream_t st, std::string* err);
// flatten_bvh() of pt_bvh.h on the device (pair records in node order; big-leaf slots in node order)
int flatten_bvh_device(const DeviceTree& t, uint32_t big_from, DeviceFlat& out, hipStream_t st, std::string* err);
// nodes + indices to the host (pt_get_bvh)
int download_tree(const DeviceTree& t, HostBVH& out, hipStream_t st, std::string* err);

// Primitives into leaf order, on the device (all pointers device memory): slot s gets object idx[s] — triangles as
// {p0, p1 - p0, p2 - p0, object id, material, 0} plus their nine normal floats, spheres as {centre, radius} plus
// {object id, material} — and slot_of[object] = s.
void launch_reorder_triangles(const float* pos, const float* nrm, const uint32_t* mat, const uint32_t* idx, uint32_t n, float* prim, float* nrm_out,
                              uint32_t* slot_of, hipStream_t st);
void launch_reorder_spheres(const float* cr, const uint32_t* mat, const uint32_t* idx, uint32_t n, float* prim, uint32_t* meta, uint32_t* slot_of,
                            hipStream_t st);

}  // namespace pt
