// pt_pixelmap.h — which pixels a render call owns and in what order (shared by host and device).
#pragma once
#include <stdint.h>

namespace pt {

// floor(n / d) for 0 <= n < 2^31 and a divisor fixed per launch, as one multiply-high and a shift (Granlund & Montgomery:
// with l = ceil(log2 d) and m = ceil(2^(31 + l) / d), floor(n * m / 2^(31 + l)) = floor(n / d); m < 2^32).  The slot ->
// pixel mapping divides four times per generated path and per shaded hit; the general 32-bit division is ~30 instructions.
struct FastDiv {
    uint32_t m;   // 0: d == 1
    uint32_t sh;  // l - 1
};
inline FastDiv fastdiv_make(uint32_t d) {
    if (d <= 1) return FastDiv{0u, 0u};
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    return FastDiv{(uint32_t)(((1ull << (31 + l)) + d - 1) / d), l - 1};
}

// slot -> pixel mapping of one render call: 8x8 pixel tiles per wave for coherent primary rays,
// grouped in `tile` x `tile` blocks that are dealt round-robin to ranks.
struct PixelMap {
    uint32_t width, height;
    uint32_t tile;        // edge of a rank tile (multiple of 8)
    uint32_t tiles_x, tiles_y;
    uint32_t rank, nranks;
    uint32_t n_local_tiles;
    uint32_t npix_pad;    // n_local_tiles * tile * tile
    FastDiv fd_per_tile, fd_tiles_x, fd_subs;  // by tile * tile, tiles_x, tile / 8
};

inline void pixelmap_init(PixelMap& M, uint32_t width, uint32_t height, uint32_t tile, uint32_t rank, uint32_t nranks) {
    M.width = width;
    M.height = height;
    M.tile = tile;
    M.rank = rank;
    M.nranks = nranks;
    M.tiles_x = (width + tile - 1) / tile;
    M.tiles_y = (height + tile - 1) / tile;
    const uint32_t n_tiles = M.tiles_x * M.tiles_y;
    M.n_local_tiles = n_tiles > rank ? (n_tiles - rank + nranks - 1) / nranks : 0;
    M.npix_pad = M.n_local_tiles * tile * tile;
    M.fd_per_tile = fastdiv_make(tile * tile);
    M.fd_tiles_x = fastdiv_make(M.tiles_x);
    M.fd_subs = fastdiv_make(tile >> 3);
}

}  // namespace pt
