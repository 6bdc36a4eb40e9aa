// pt_pixelmap.h — which pixels a render call owns and in what order (shared by host and device).
#pragma once
#include <stdint.h>

namespace pt {

// slot -> pixel mapping of one render call: 8x8 pixel tiles per wave for coherent primary rays,
// grouped in `tile` x `tile` blocks that are dealt round-robin to ranks.
struct PixelMap {
    uint32_t width, height;
    uint32_t tile;        // edge of a rank tile (multiple of 8)
    uint32_t tiles_x, tiles_y;
    uint32_t rank, nranks;
    uint32_t n_local_tiles;
    uint32_t npix_pad;    // n_local_tiles * tile * tile
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
}

}  // namespace pt
