"""Host-inclusive rates of the seam entry points (pt_trace, pt_generate, pt_read_accum, pt_read_rgb8): how much of a
call is the copy between the caller's pageable memory and the device.  python tools/seam_rates.py [workload]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: F401,E402
from pathtracer_amd import ptcore, scenes  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
make, W, H, cam_o, cam_t, desc = bench.workloads(scenes)[wl]
g = ptcore.PathTracer(0).set_scene(make(), scenes.make_sky())
g.set_camera(cam_o, cam_t, W, H)
g.render(1, 32, 1)


def best(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


for n in (100_000, 1_000_000, 4_000_000):
    rays = scenes.random_rays(n, seed=3, aim=(0.7, 0.2, 0.6))
    for any_hit in (False, True):
        t = best(lambda: g.trace(rays, any_hit))
        print(f"pt_trace {n:>8} rays any={int(any_hit)}: {t * 1e3:8.2f} ms  {n / t / 1e6:8.1f} Mrays/s  ({(28 + 16) * n / t / 1e9:.2f} GB/s of ray + hit bytes)")
ys, xs = np.mgrid[0:H, 0:W]
xy = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.uint32)
t = best(lambda: g.generate(xy, 0, 7))
print(f"pt_generate {len(xy)} rays: {t * 1e3:.2f} ms  {len(xy) / t / 1e6:.1f} Mrays/s")
t = best(lambda: g.read_accum())
print(f"pt_read_accum {W}x{H}: {t * 1e3:.2f} ms  ({W * H * 12 / t / 1e9:.2f} GB/s)")
t = best(lambda: g.read_rgb8())
print(f"pt_read_rgb8 {W}x{H}: {t * 1e3:.2f} ms")
g.close()
