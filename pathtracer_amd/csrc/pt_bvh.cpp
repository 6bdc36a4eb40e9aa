// pt_bvh.cpp — see pt_bvh.h.  Host code; compiled with -ffp-contract=off so every float
// operation is the single IEEE-754 operation the Rust source performs.
#include "pt_bvh.h"

#include <cmath>
#include <limits>

#pragma clang fp contract(off)

namespace pt {
namespace {

constexpr float kInfH = std::numeric_limits<float>::infinity();

// f32::min / f32::max (non-NaN operand wins)
inline float mn2(float a, float b) { return std::fmin(a, b); }
inline float mx2(float a, float b) { return std::fmax(a, b); }

struct Box {
    float lo[3], hi[3];
    void clear() {  // AABB::new, aabb.rs:15-20
        for (int a = 0; a < 3; ++a) { lo[a] = kInfH; hi[a] = -kInfH; }
    }
    void grow(const float* mn, const float* mx) {  // AABB::combine, aabb.rs:30-38
        for (int a = 0; a < 3; ++a) { lo[a] = mn2(lo[a], mn[a]); hi[a] = mx2(hi[a], mx[a]); }
    }
    void grow_point(const float* p) {  // AABB::extent, aabb.rs:21-29
        for (int a = 0; a < 3; ++a) { lo[a] = mn2(lo[a], p[a]); hi[a] = mx2(hi[a], p[a]); }
    }
    float area() const {  // AABB::area, aabb.rs:42-45
        float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return (2.0f * ex * ey + 2.0f * ex * ez) + 2.0f * ey * ez;
    }
};

struct Builder {
    const std::vector<PrimInfo>& prims;
    HostBVH& out;
    std::vector<uint32_t> depth;  // level of each node

    Builder(const std::vector<PrimInfo>& p, HostBVH& o) : prims(p), out(o) {}

    // bvh.rs:104-169.  Returns false when no split beats the parent cost.
    bool choose_split(const pt_bvh_node& node, int* axis_out, float* pivot_out) const {
        const uint32_t first = node.left_first, count = node.count;
        Box nb;
        for (int a = 0; a < 3; ++a) { nb.lo[a] = node.min[a]; nb.hi[a] = node.max[a]; }
        const float parent_cost = nb.area() * (float)count;

        Box cb;
        cb.clear();
        for (uint32_t i = first; i < first + count; ++i) cb.grow_point(prims[out.indices[i]].c);
        float delta[3];
        for (int a = 0; a < 3; ++a) delta[a] = (cb.hi[a] - cb.lo[a]) / 8.0f;

        Box bin_box[24];
        uint32_t bin_n[24];
        for (int b = 0; b < 24; ++b) { bin_box[b].clear(); bin_n[b] = 0; }
        for (uint32_t i = first; i < first + count; ++i) {
            const PrimInfo& p = prims[out.indices[i]];
            for (int a = 0; a < 3; ++a) {
                // ((c - cmin) / delta).max(0.).min(7.) as usize  — NaN (0/0) becomes 0
                float q = (p.c[a] - cb.lo[a]) / delta[a];
                int b = a * 8 + (int)mn2(mx2(q, 0.0f), 7.0f);
                bin_box[b].grow(p.mn, p.mx);
                bin_n[b] += 1;
            }
        }

        int best_axis = -1, best_bin = -1;
        float best = parent_cost;
        for (int a = 0; a < 3; ++a) {
            const Box* bb = bin_box + a * 8;
            const uint32_t* bn = bin_n + a * 8;
            Box right[8];
            uint32_t nright[8];
            right[7] = bb[7];
            nright[7] = bn[7];
            for (int k = 6; k >= 0; --k) {
                right[k] = right[k + 1];
                right[k].grow(bb[k].lo, bb[k].hi);
                nright[k] = nright[k + 1] + bn[k];
            }
            Box left;
            left.clear();
            uint32_t nleft = 0;
            for (int k = 0; k < 8; ++k) {
                left.grow(bb[k].lo, bb[k].hi);
                nleft += bn[k];
                // right side includes bin k itself, exactly as bvh.rs:155
                float cost = (float)nleft * left.area() + (float)nright[k] * right[k].area();
                if (cost < best) { best_axis = a; best_bin = k; best = cost; }
            }
        }
        if (best_axis < 0) return false;
        *axis_out = best_axis;
        *pivot_out = cb.lo[best_axis] + (float)(1 + best_bin) * delta[best_axis];
        return true;
    }

    // bvh.rs:63-103
    bool partition(uint32_t ni) {
        int axis;
        float pivot;
        const pt_bvh_node node = out.nodes[ni];
        if (!choose_split(node, &axis, &pivot)) return false;
        const uint32_t first = node.left_first, count = node.count;
        Box lb, rb;
        lb.clear();
        rb.clear();
        uint32_t split = first;
        for (uint32_t i = first; i < first + count; ++i) {
            const PrimInfo& p = prims[out.indices[i]];
            if (p.c[axis] <= pivot) {
                lb.grow(p.mn, p.mx);
                uint32_t tmp = out.indices[split];
                out.indices[split] = out.indices[i];
                out.indices[i] = tmp;
                split += 1;
            } else {
                rb.grow(p.mn, p.mx);
            }
        }
        const uint32_t left = (uint32_t)out.nodes.size();
        out.nodes[ni].left_first = left;
        out.nodes[ni].count = 0;
        const uint32_t nl = split - first;
        out.nodes.push_back(pt_bvh_node{{lb.lo[0], lb.lo[1], lb.lo[2]}, first, {lb.hi[0], lb.hi[1], lb.hi[2]}, nl});
        out.nodes.push_back(pt_bvh_node{{rb.lo[0], rb.lo[1], rb.lo[2]}, split, {rb.hi[0], rb.hi[1], rb.hi[2]}, count - nl});
        if (nl == 0 || nl == count) out.hazard = true;
        return true;
    }

    void run() {
        const uint32_t n = (uint32_t)prims.size();
        out.indices.resize(n);
        Box root;
        root.clear();
        for (uint32_t i = 0; i < n; ++i) {
            out.indices[i] = i;
            root.grow(prims[i].mn, prims[i].mx);
        }
        out.nodes.clear();
        out.nodes.reserve(2 * (size_t)n);
        out.nodes.push_back(pt_bvh_node{{root.lo[0], root.lo[1], root.lo[2]}, 0, {root.hi[0], root.hi[1], root.hi[2]}, n});
        depth.assign(1, 1);
        out.max_depth = 1;
        out.hazard = false;
        // bvh.rs:52-62: explicit stack, left subtree first
        std::vector<uint32_t> todo{0};
        while (!todo.empty()) {
            uint32_t ni = todo.back();
            todo.pop_back();
            if (out.nodes[ni].count > 2 && partition(ni)) {
                uint32_t left = out.nodes[ni].left_first;
                depth.resize(out.nodes.size());
                depth[left] = depth[left + 1] = depth[ni] + 1;
                if (depth[left] > out.max_depth) out.max_depth = depth[left];
                todo.push_back(left + 1);
                todo.push_back(left);
            }
        }
    }
};

}  // namespace

void prim_info_triangles(const float* pos, uint64_t n, std::vector<PrimInfo>& out) {
    out.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = pos + 9 * i;
        PrimInfo& pi = out[i];
        for (int a = 0; a < 3; ++a) {
            // Triangle::centre, triangle.rs:80-82; Triangle::bounds, triangle.rs:83-90
            pi.c[a] = ((p[a] + p[3 + a]) + p[6 + a]) * (1.0f / 3.0f);
            pi.mn[a] = mn2(mn2(p[a], p[3 + a]), p[6 + a]);
            pi.mx[a] = mx2(mx2(p[a], p[3 + a]), p[6 + a]);
        }
    }
}

void prim_info_spheres(const float* cr, uint64_t n, std::vector<PrimInfo>& out) {
    out.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        const float* s = cr + 4 * i;
        PrimInfo& pi = out[i];
        for (int a = 0; a < 3; ++a) {
            pi.c[a] = s[a];  // Sphere::centre, sphere.rs:81-83; bounds, sphere.rs:84-91
            pi.mn[a] = s[a] - s[3];
            pi.mx[a] = s[a] + s[3];
        }
    }
}

void build_bvh(const std::vector<PrimInfo>& prims, HostBVH& out) {
    Builder b(prims, out);
    b.run();
}

int flatten_bvh(const HostBVH& bvh, FlatBVH& out) {
    const size_t nn = bvh.nodes.size();
    if (bvh.indices.size() > kMaxPrims) return PT_ERR_LIMIT;
    // number the internal nodes in reference order
    std::vector<uint32_t> pair_of(nn, 0);
    uint32_t npairs = 0;
    for (size_t i = 0; i < nn; ++i)
        if (bvh.nodes[i].count == 0) pair_of[i] = npairs++;
    out.pairs.assign(npairs, PairRecord{});
    out.big_leaf.clear();
    auto make_ref = [&](uint32_t node) -> uint32_t {
        const pt_bvh_node& n = bvh.nodes[node];
        if (n.count == 0) return pair_of[node];
        if (n.count <= 31) return kLeafBit | (n.count << kLeafCountShift) | n.left_first;
        uint32_t slot = (uint32_t)(out.big_leaf.size() / 2);
        out.big_leaf.push_back(n.left_first);
        out.big_leaf.push_back(n.count);
        return kLeafBit | slot;
    };
    for (size_t i = 0; i < nn; ++i) {
        const pt_bvh_node& n = bvh.nodes[i];
        if (n.count != 0) continue;
        if ((size_t)n.left_first + 1 >= nn) return PT_ERR_BVH;
        const pt_bvh_node& l = bvh.nodes[n.left_first];
        const pt_bvh_node& r = bvh.nodes[n.left_first + 1];
        PairRecord& p = out.pairs[pair_of[i]];
        for (int a = 0; a < 3; ++a) {
            p.lmin[a] = l.min[a]; p.lmax[a] = l.max[a];
            p.rmin[a] = r.min[a]; p.rmax[a] = r.max[a];
        }
        p.lref = make_ref(n.left_first);
        p.rref = make_ref(n.left_first + 1);
        p.lpad = p.rpad = 0;
    }
    for (int a = 0; a < 3; ++a) {
        out.root_min[a] = bvh.nodes[0].min[a];
        out.root_max[a] = bvh.nodes[0].max[a];
    }
    out.root_ref = make_ref(0);
    return PT_OK;
}

}  // namespace pt
