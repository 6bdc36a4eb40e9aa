"""Time of one pt_render call against spp on C3 (fixed cost per launch = intercept): python3 tools/scan_spp.py"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa: F401
from pathtracer_amd import ptcore, scenes
g = ptcore.PathTracer(0).set_scene(scenes.statue_scene(), scenes.make_sky())
g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 1920, 1080)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spps = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256]
g.render(256, depth, 1)
for spp in spps:
    best = 1e9
    for _ in range(3):
        g.reset_counters(); t = time.perf_counter(); g.render(spp, depth, 7, None, ptcore.RENDER_TIMED); dt = time.perf_counter() - t
        best = min(best, dt)
    c = g.counters()
    print(f"depth={depth} spp={spp:4d} {best * 1e3:8.2f} ms  kernel {c['tail_ms']:8.2f} ms  {c['rays_closest'] / best / 1e6:8.1f} Mrays/s", flush=True)
