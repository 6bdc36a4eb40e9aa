"""Wavefront mode (PT_OPT_RENDER_MODE 0) on C3 with a few scheduling settings: python3 tools/mode0.py"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa: F401
from pathtracer_amd import ptcore, scenes
g = ptcore.PathTracer(0).set_scene(scenes.statue_scene(), scenes.make_sky())
g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 1920, 1080)
g.set_option(ptcore.OPT_RENDER_MODE, 0)
g.render(8, 32, 1)
def run(spp, **o):
    for k, v in o.items(): g.set_option(getattr(ptcore, 'OPT_' + k.upper()), v)
    best = 1e9
    for _ in range(2):
        g.reset_counters(); t = time.perf_counter(); g.render(spp, 32, 7, None, ptcore.RENDER_TIMED); best = min(best, time.perf_counter() - t)
    c = g.counters()
    print(f"spp={spp} {o} -> {c['rays_closest'] / best / 1e6:8.1f} Mrays/s {best * 1e3:7.1f} ms ext={c['extend_launches']} tail={c['tail_launches']}", flush=True)
run(64)
run(64, first_check_round=1, tail_fraction=1, tail_min_paths=1 << 30, batch_paths=1 << 25)
run(256, first_check_round=1, tail_fraction=1, tail_min_paths=1 << 30, batch_paths=1 << 26)
