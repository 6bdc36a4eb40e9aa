// pt_bvh.cpp — see pt_bvh.h.  Host code; compiled with -ffp-contract=off so every float
// operation is the single IEEE-754 operation the Rust source performs.
#include "pt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

#pragma clang fp contract(off)

namespace pt {
namespace {

constexpr float kInfH = std::numeric_limits<float>::infinity();

// f32::min / f32::max: the non-NaN operand wins; of two equal operands (+0 / -0) the first is returned, as rustc's
// lowering of llvm.minnum / maxnum does (written out: libm's fminf returns the second)
inline float mn2(float a, float b) { return a != a ? b : (b < a ? b : a); }
inline float mx2(float a, float b) { return a != a ? b : (b > a ? b : a); }

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

// Parallel loop over [first, last) in contiguous pieces, one per thread (used on large ranges only).  The pieces depend
// only on (first, last, threads), so consecutive passes over one range see the same pieces.
template <class F>
void parallel_pieces(uint32_t first, uint32_t last, int threads, F&& body) {
    const uint32_t n = last - first;
    if (threads <= 1 || n < (1u << 16)) {
        body(0, first, last);
        return;
    }
    std::vector<std::thread> pool;
    const uint32_t per = (n + (uint32_t)threads - 1) / (uint32_t)threads;
    for (int t = 1; t < threads; ++t) {
        const uint32_t a = first + std::min<uint64_t>((uint64_t)t * per, n), b = first + std::min<uint64_t>((uint64_t)(t + 1) * per, n);
        pool.emplace_back([&body, t, a, b] { body(t, a, b); });
    }
    body(0, first, first + std::min(per, n));
    for (std::thread& th : pool) th.join();
}

int default_threads() {
    const char* env = std::getenv("PTCORE_BUILD_THREADS");
    int threads = env ? std::atoi(env) : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    return threads < 1 ? 1 : threads;
}

// A primitive as the build moves it around: bvh.indices[i] together with that object's centre and bounds, so that
// every pass over a node reads its range of one array front to back (the reference looks each object up through
// bvh.indices; following the indices into the object array made every pass a random gather).
struct Rec {
    float c[3], mn[3], mx[3];
    uint32_t idx;
};
static_assert(sizeof(Rec) == 40, "record layout");

constexpr uint32_t kParallelNode = 1u << 16;  // nodes of at least this many primitives are split by all threads of their subtree

// One subtree, built in the reference's order.  `nodes` is local to the subtree: nodes[0] is a copy
// of the subtree's root and child links are local indices, so subtrees can be built by different
// threads and spliced together afterwards (the reference's explicit stack finishes a node's whole
// subtree before it touches anything else, so a subtree's nodes are contiguous in its numbering).
// `work`, `tmp` and `lpos` are arrays shared by all subtrees; a subtree only touches its own range of them.
struct Builder {
    Rec* work;
    Rec* tmp;
    uint32_t* lpos;
    std::vector<pt_bvh_node> nodes;
    uint32_t max_depth = 1;  // levels below and including this subtree's root
    bool hazard = false;
    int threads = 1;         // helper threads for the passes over one large node

    Builder(Rec* w, Rec* t, uint32_t* l) : work(w), tmp(t), lpos(l) {}

    // bvh.rs:104-169.  Returns false when no split beats the parent cost.
    bool choose_split(const pt_bvh_node& node, int* axis_out, float* pivot_out) const {
        const uint32_t first = node.left_first, count = node.count;
        Box nb;
        for (int a = 0; a < 3; ++a) { nb.lo[a] = node.min[a]; nb.hi[a] = node.max[a]; }
        const float parent_cost = nb.area() * (float)count;

        // min / max / counts are order independent, so the pieces of a large node can be reduced in any order
        const int nt = threads > 1 && count >= kParallelNode ? threads : 1;
        std::vector<Box> cbs((size_t)nt);
        parallel_pieces(first, first + count, nt, [&](int t, uint32_t a, uint32_t b) {
            Box cb;
            cb.clear();
            for (uint32_t i = a; i < b; ++i) cb.grow_point(work[i].c);
            cbs[(size_t)t] = cb;
        });
        Box cb = cbs[0];
        for (int t = 1; t < nt; ++t) cb.grow(cbs[(size_t)t].lo, cbs[(size_t)t].hi);
        float delta[3];
        for (int a = 0; a < 3; ++a) delta[a] = (cb.hi[a] - cb.lo[a]) / 8.0f;

        struct Bins {
            Box box[24];
            uint32_t n[24];
        };
        std::vector<Bins> parts((size_t)nt);
        parallel_pieces(first, first + count, nt, [&](int t, uint32_t a0, uint32_t b0) {
            Bins& B = parts[(size_t)t];
            for (int b = 0; b < 24; ++b) { B.box[b].clear(); B.n[b] = 0; }
            for (uint32_t i = a0; i < b0; ++i) {
                const Rec& p = work[i];
                for (int a = 0; a < 3; ++a) {
                    // ((c - cmin) / delta).max(0.).min(7.) as usize  — NaN (0/0) becomes 0
                    float q = (p.c[a] - cb.lo[a]) / delta[a];
                    int b = a * 8 + (int)mn2(mx2(q, 0.0f), 7.0f);
                    B.box[b].grow(p.mn, p.mx);
                    B.n[b] += 1;
                }
            }
        });
        Box* bin_box = parts[0].box;
        uint32_t* bin_n = parts[0].n;
        for (int t = 1; t < nt; ++t)
            for (int b = 0; b < 24; ++b) {
                bin_box[b].grow(parts[(size_t)t].box[b].lo, parts[(size_t)t].box[b].hi);
                bin_n[b] += parts[(size_t)t].n[b];
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

    // The reference's partition (bvh.rs:63-103) walks the node once and swaps every object that belongs to the left
    // with the object at the running split position.  The swap sequence decides the leaf order, so the result must be
    // exactly that permutation — but it need not be produced by swapping.  Positions relative to the node, nL objects
    // on the left, l[k] = original position of the k-th of them:
    //   * the k-th left object ends at position k (the walk keeps their order);
    //   * [split, i) only ever holds right objects, and event k (the walk reaching l[k]) moves the object standing at
    //     position k to position l[k].  So a right object at position p stays if p >= nL, otherwise it moves to l[p]
    //     (> p), and from there again while its position is below nL.
    // Every object's final place follows from the flags and l[] alone: three passes that all threads share.
    void partition_large(uint32_t first, uint32_t count, int axis, float pivot, Box& lb, Box& rb, uint32_t& n_left) {
        const int nt = threads;
        std::vector<uint32_t> base((size_t)nt + 1, 0);
        std::vector<Box> lbs((size_t)nt), rbs((size_t)nt);
        parallel_pieces(first, first + count, nt, [&](int t, uint32_t a, uint32_t b) {
            Box l, r;
            l.clear();
            r.clear();
            uint32_t nl = 0;
            for (uint32_t i = a; i < b; ++i) {
                const Rec& p = work[i];
                if (p.c[axis] <= pivot) { l.grow(p.mn, p.mx); nl += 1; }
                else r.grow(p.mn, p.mx);
            }
            lbs[(size_t)t] = l;
            rbs[(size_t)t] = r;
            base[(size_t)t + 1] = nl;
        });
        lb.clear();
        rb.clear();
        for (int t = 0; t < nt; ++t) {
            lb.grow(lbs[(size_t)t].lo, lbs[(size_t)t].hi);
            rb.grow(rbs[(size_t)t].lo, rbs[(size_t)t].hi);
            base[(size_t)t + 1] += base[(size_t)t];
        }
        const uint32_t nL = base[(size_t)nt];
        n_left = nL;
        parallel_pieces(first, first + count, nt, [&](int t, uint32_t a, uint32_t b) {
            uint32_t k = base[(size_t)t];
            for (uint32_t i = a; i < b; ++i)
                if (work[i].c[axis] <= pivot) lpos[first + k++] = i - first;
        });
        parallel_pieces(first, first + count, nt, [&](int t, uint32_t a, uint32_t b) {
            uint32_t k = base[(size_t)t];
            for (uint32_t i = a; i < b; ++i) {
                const Rec& p = work[i];
                if (p.c[axis] <= pivot) {
                    tmp[first + k++] = p;
                } else {
                    uint32_t q = i - first;
                    while (q < nL) q = lpos[first + q];
                    tmp[first + q] = p;
                }
            }
        });
        parallel_pieces(first, first + count, nt, [&](int, uint32_t a, uint32_t b) { std::memcpy(work + a, tmp + a, (size_t)(b - a) * sizeof(Rec)); });
    }

    bool partition(uint32_t ni) {
        int axis;
        float pivot;
        const pt_bvh_node node = nodes[ni];
        if (!choose_split(node, &axis, &pivot)) return false;
        const uint32_t first = node.left_first, count = node.count;
        Box lb, rb;
        uint32_t nl = 0;
        if (threads > 1 && count >= kParallelNode) {
            partition_large(first, count, axis, pivot, lb, rb, nl);
        } else {
            lb.clear();
            rb.clear();
            uint32_t split = first;
            for (uint32_t i = first; i < first + count; ++i) {
                const Rec p = work[i];
                if (p.c[axis] <= pivot) {
                    lb.grow(p.mn, p.mx);
                    work[i] = work[split];
                    work[split] = p;
                    split += 1;
                } else {
                    rb.grow(p.mn, p.mx);
                }
            }
            nl = split - first;
        }
        const uint32_t left = (uint32_t)nodes.size();
        nodes[ni].left_first = left;
        nodes[ni].count = 0;
        nodes.push_back(pt_bvh_node{{lb.lo[0], lb.lo[1], lb.lo[2]}, first, {lb.hi[0], lb.hi[1], lb.hi[2]}, nl});
        nodes.push_back(pt_bvh_node{{rb.lo[0], rb.lo[1], rb.lo[2]}, first + nl, {rb.hi[0], rb.hi[1], rb.hi[2]}, count - nl});
        if (nl == 0 || nl == count) hazard = true;
        return true;
    }

    // bvh.rs:52-62: explicit stack, left subtree first
    void run(const pt_bvh_node& root) {
        nodes.clear();
        nodes.push_back(root);
        std::vector<uint32_t> depth(1, 1);
        max_depth = 1;
        std::vector<uint32_t> todo{0};
        while (!todo.empty()) {
            uint32_t ni = todo.back();
            todo.pop_back();
            if (nodes[ni].count > 2 && partition(ni)) {
                // an empty child (bvh.rs:87-97 with float rounding against the binning): the reference would go on splitting
                // the full child the same way for ever; the build stops here and reports PT_ERR_BVH
                if (hazard) return;
                uint32_t left = nodes[ni].left_first;
                depth.resize(nodes.size());
                depth[left] = depth[left + 1] = depth[ni] + 1;
                if (depth[left] > max_depth) max_depth = depth[left];
                todo.push_back(left + 1);
                todo.push_back(left);
            }
        }
    }
};

constexpr uint32_t kParallelMin = 8192;  // smaller subtrees are not worth a thread

// Builds the subtree below `root` with up to `threads` threads.  Result: b.nodes[0] = root, then
// the subtree's nodes in the reference's order with local child links.
void build_subtree(const pt_bvh_node& root, int threads, Builder& b) {
    if (threads <= 1 || root.count < kParallelMin) {
        b.threads = 1;
        b.run(root);
        return;
    }
    // split this node here (its passes use all our threads), then build the two sides concurrently
    b.threads = threads;
    b.nodes.clear();
    b.nodes.push_back(root);
    b.max_depth = 1;
    if (!(root.count > 2 && b.partition(0)) || b.hazard) return;
    const pt_bvh_node L = b.nodes[1], R = b.nodes[2];
    Builder bl(b.work, b.tmp, b.lpos), br(b.work, b.tmp, b.lpos);
    const uint64_t total = (uint64_t)L.count + R.count;
    int tl = (int)((uint64_t)threads * L.count / (total ? total : 1));
    tl = std::max(1, std::min(threads - 1, tl));
    std::thread worker([&] { build_subtree(L, tl, bl); });
    build_subtree(R, threads - tl, br);
    worker.join();
    // splice: [root, L, R, nodes below L..., nodes below R...]; an internal node's link moves with its block
    const uint32_t nl = (uint32_t)bl.nodes.size() - 1, nr = (uint32_t)br.nodes.size() - 1;
    b.nodes.resize(3 + (size_t)nl + nr);
    b.nodes[1] = bl.nodes[0];
    b.nodes[2] = br.nodes[0];
    if (nl) b.nodes[1].left_first += 2;       // bl-local index 1 -> 3
    if (nr) b.nodes[2].left_first += 2 + nl;  // br-local index 1 -> 3 + nl
    for (uint32_t i = 0; i < nl; ++i) {
        pt_bvh_node n = bl.nodes[1 + i];
        if (n.count == 0) n.left_first += 2;
        b.nodes[3 + i] = n;
    }
    for (uint32_t i = 0; i < nr; ++i) {
        pt_bvh_node n = br.nodes[1 + i];
        if (n.count == 0) n.left_first += 2 + nl;
        b.nodes[3 + (size_t)nl + i] = n;
    }
    b.max_depth = 1 + std::max(bl.max_depth, br.max_depth);
    b.hazard = b.hazard || bl.hazard || br.hazard;
}

}  // namespace

void prim_info_triangles(const float* pos, uint64_t n, std::vector<PrimInfo>& out) {
    out.resize(n);
    parallel_pieces(0, (uint32_t)n, default_threads(), [&](int, uint32_t a0, uint32_t b0) {
        for (uint64_t i = a0; i < b0; ++i) {
            const float* p = pos + 9 * i;
            PrimInfo& pi = out[i];
            for (int a = 0; a < 3; ++a) {
                // Triangle::centre, triangle.rs:80-82; Triangle::bounds, triangle.rs:83-90
                pi.c[a] = ((p[a] + p[3 + a]) + p[6 + a]) * (1.0f / 3.0f);
                pi.mn[a] = mn2(mn2(p[a], p[3 + a]), p[6 + a]);
                pi.mx[a] = mx2(mx2(p[a], p[3 + a]), p[6 + a]);
            }
        }
    });
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

void build_bvh(const std::vector<PrimInfo>& prims, HostBVH& out, int threads) {
    const uint32_t n = (uint32_t)prims.size();
    if (threads <= 0) threads = default_threads();
    std::vector<Rec> work(n), tmp(threads > 1 && n >= kParallelNode ? n : 0);
    std::vector<uint32_t> lpos(tmp.size());
    std::vector<Box> roots((size_t)threads);
    for (Box& b : roots) b.clear();
    parallel_pieces(0, n, threads, [&](int t, uint32_t a, uint32_t b) {
        Box r;
        r.clear();
        for (uint32_t i = a; i < b; ++i) {
            const PrimInfo& p = prims[i];
            Rec& w = work[i];
            for (int k = 0; k < 3; ++k) { w.c[k] = p.c[k]; w.mn[k] = p.mn[k]; w.mx[k] = p.mx[k]; }
            w.idx = i;
            r.grow(p.mn, p.mx);
        }
        roots[(size_t)t] = r;
    });
    Box root = roots[0];
    for (int t = 1; t < threads; ++t) root.grow(roots[(size_t)t].lo, roots[(size_t)t].hi);
    Builder b(work.data(), tmp.data(), lpos.data());
    build_subtree(pt_bvh_node{{root.lo[0], root.lo[1], root.lo[2]}, 0, {root.hi[0], root.hi[1], root.hi[2]}, n}, threads, b);
    out.indices.resize(n);
    parallel_pieces(0, n, threads, [&](int, uint32_t a, uint32_t b0) {
        for (uint32_t i = a; i < b0; ++i) out.indices[i] = work[i].idx;
    });
    out.nodes = std::move(b.nodes);
    out.max_depth = b.max_depth;
    out.hazard = b.hazard;
}

int flatten_bvh(const HostBVH& bvh, FlatBVH& out, uint32_t big_from) {
    const size_t nn = bvh.nodes.size();
    if (bvh.indices.size() > kMaxPrims) return PT_ERR_LIMIT;
    // number the internal nodes in reference order
    std::vector<uint32_t> pair_of(nn, 0);
    uint32_t npairs = 0;
    for (size_t i = 0; i < nn; ++i)
        if (bvh.nodes[i].count == 0) {
            if (kPairDfs) {  // children pairs are disjoint, so left_first >> 1 is unique whatever numbering an adopted tree uses
                pair_of[i] = bvh.nodes[i].left_first >> 1;
                npairs = std::max(npairs, pair_of[i] + 1);
            } else {
                pair_of[i] = npairs++;
            }
        }
    out.pairs.assign((size_t)npairs + 1, PairRecord{});  // + 1: the record behind the last one may be touched (prefetch)
    out.big_leaf.clear();
    auto make_ref = [&](uint32_t node) -> uint32_t {
        const pt_bvh_node& n = bvh.nodes[node];
        if (n.count == 0) return pair_of[node];
        if (n.count < big_from && n.count <= 31) return kLeafBit | (n.count << kLeafCountShift) | n.left_first;
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
