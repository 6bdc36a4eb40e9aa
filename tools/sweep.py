"""Scheduling sweep on the C3 statue (development helper): python3 tools/sweep.py"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa: F401
from pathtracer_amd import ptcore, scenes
g = ptcore.PathTracer(0).set_scene(scenes.statue_scene(), scenes.make_sky())
g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 1920, 1080)
g.render(8, 32, 1)
def run(spp, flags=0, **o):
    for k, v in o.items(): g.set_option(getattr(ptcore, 'OPT_' + k.upper()), v)
    best = None
    for _ in range(2):
        g.reset_counters(); t = time.perf_counter(); g.render(spp, 32, 7, None, flags); dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    c = g.counters()
    print(f"spp={spp} {o} -> {c['rays_closest'] / best / 1e6:8.1f} Mrays/s  {best * 1e3:7.1f} ms ext={c['extend_launches']} tail={c['tail_launches']} tail_ms={c['tail_ms']:.0f}", flush=True)
T = ptcore.RENDER_TIMED
for spp in (64, 256, 1024):
    for bp in (1 << 26, 1 << 27, 1 << 28, 1 << 29):
        run(spp, T, render_mode=1, batch_paths=bp)
