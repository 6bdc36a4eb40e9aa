"""Render the C3 statue once or a few times with options from the command line (profiling helper).
usage: python3 tools/one.py [spp] [reps] [opt=value ...]"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa: F401  (load torch's HIP runtime first)
from pathtracer_amd import ptcore, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = ptcore.PathTracer(0).set_scene(scenes.statue_scene(), scenes.make_sky())
g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 1920, 1080)
for a in sys.argv[3:]:
    k, v = a.split('=')
    g.set_option(getattr(ptcore, 'OPT_' + k.upper()), int(v))
g.render(2, 32, 1)
for _ in range(reps):
    g.reset_counters()
    t = time.perf_counter()
    g.render(spp, 32, 7, None, ptcore.RENDER_TIMED)
    dt = time.perf_counter() - t
    c = g.counters()
    print(f"spp={spp} {sys.argv[3:]} -> {c['rays_closest'] / dt / 1e6:8.1f} Mrays/s {dt * 1e3:7.1f} ms tail_ms={c['tail_ms']:.1f}", flush=True)
