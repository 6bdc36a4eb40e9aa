// Driver of tests/test_bvh_build_cpu.py::test_threaded_build_under_thread_sanitizer: the product's threaded BVH build
// (pt_bvh.cpp) compiled with -fsanitize=thread; the tree must also equal the single-threaded one.
#include "pt_bvh.h"
#include <cstdio>
#include <cstring>
#include <random>
int main() {
    using namespace pt;
    for (int rep = 0; rep < 3; ++rep) {
        const uint64_t n = rep == 0 ? 120000 : (rep == 1 ? 20000 : 5);
        std::mt19937 rng(7 + rep);
        std::uniform_real_distribution<float> U(-1.f, 1.f), S(-0.02f, 0.02f);
        std::vector<float> pos(n * 9);
        for (uint64_t i = 0; i < n; ++i) {
            float c[3] = {U(rng), U(rng), U(rng)};
            for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) pos[i * 9 + v * 3 + k] = c[k] + S(rng);
        }
        std::vector<PrimInfo> info;
        prim_info_triangles(pos.data(), n, info);
        HostBVH a, b;
        build_bvh(info, a, 8);
        build_bvh(info, b, 1);
        bool same = a.nodes.size() == b.nodes.size() && std::memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(a.nodes[0])) == 0 && a.indices == b.indices;
        std::printf("n=%llu nodes=%zu threaded==single: %d\n", (unsigned long long)n, a.nodes.size(), (int)same);
        if (!same) return 1;
    }
    return 0;
}
