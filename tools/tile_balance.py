"""How evenly the 64x64 tiles dealt round-robin (t % N, row-major tile index t; SURVEY 8e names Morton order) spread the work of a
tile-sharded frame: every rank's shard of one frame is rendered on this GPU, one after the other, and its rays and kernel time are
printed.  The frame time of N GPUs is the slowest rank's, so  max / mean  is the efficiency the split itself allows.
usage: python3 tools/tile_balance.py [workload=C4] [spp=8] [tile=64]"""
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: F401,E402
import bench  # noqa: E402
from pathtracer_amd import ptcore, scenes  # noqa: E402

opts = dict(a.split('=') for a in sys.argv[1:])
wl, spp, tile = opts.get('workload', 'C4'), int(opts.get('spp', 8)), int(opts.get('tile', 64))
make, W, H, cam_o, cam_t, desc = bench.workloads(scenes)[wl]
g = ptcore.PathTracer(0).set_scene(make(), scenes.make_sky())
g.set_camera(cam_o, cam_t, W, H)
g.render(1, 32, 1)
print(f"{desc}: {spp} spp, tiles of {tile}, {((W + tile - 1) // tile)} x {((H + tile - 1) // tile)} tiles")
for world in (1, 2, 4, 8):
    rays, ms = [], []
    for r in range(world):
        g.reset_accum()
        g.reset_counters()
        t = time.perf_counter()
        g.render(spp, 32, 7, ptcore.Shard(ptcore.SHARD_TILES, r, world, tile, 0), ptcore.RENDER_TIMED)
        dt = time.perf_counter() - t
        c = g.counters()
        rays.append(c['rays_closest'])
        ms.append(c['tail_ms'])
    mean_r, mean_t = sum(rays) / world, sum(ms) / world
    print(f"N={world}: rays per rank {[f'{x / 1e6:.1f}M' for x in rays]}  max/mean {max(rays) / mean_r:.4f}   k_path ms per rank {[f'{x:.1f}' for x in ms]}  "
          f"max/mean {max(ms) / mean_t:.4f}  -> frame {max(ms):.1f} ms, {sum(rays) / max(ms) / 1e6:.1f} Grays/s aggregate if ranks ran side by side", flush=True)
g.close()
