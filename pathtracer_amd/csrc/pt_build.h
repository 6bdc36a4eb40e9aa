// pt_build.h — the BVH build on the GPU (pt_build.hip): same tree as build_bvh() of pt_bvh.h, bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "pt_bvh.h"

namespace pt {

// Builds out.nodes / out.indices / out.max_depth / out.hazard on the current device (work enqueued on `st`, returns
// after it has finished).  *unsupported = true (and nothing built) when an input value is a negative zero, infinite or
// NaN: min / max of such values depend on operand order, which only the host build reproduces.
int build_bvh_device(const std::vector<PrimInfo>& prims, HostBVH& out, hipStream_t st, bool* unsupported, std::string* err);

// Primitives into leaf order, on the device (all pointers device memory): slot s gets object idx[s] — triangles as
// {p0, p1 - p0, p2 - p0, object id, material, 0} plus their nine normal floats, spheres as {centre, radius} plus
// {object id, material} — and slot_of[object] = s.
void launch_reorder_triangles(const float* pos, const float* nrm, const uint32_t* mat, const uint32_t* idx, uint32_t n, float* prim, float* nrm_out,
                              uint32_t* slot_of, hipStream_t st);
void launch_reorder_spheres(const float* cr, const uint32_t* mat, const uint32_t* idx, uint32_t n, float* prim, uint32_t* meta, uint32_t* slot_of,
                            hipStream_t st);

}  // namespace pt
