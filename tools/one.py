"""Render one workload once or a few times with options from the command line (profiling helper).
usage: python3 tools/one.py [spp] [reps] [workload=C3] [opt=value ...]"""
import sys, time
sys.path.insert(0, '.')
import torch  # noqa: F401  (load torch's HIP runtime first)
import bench
from pathtracer_amd import ptcore, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
opts = dict(a.split('=') for a in sys.argv[3:])
make, W, H, cam_o, cam_t, desc = bench.workloads(scenes)[opts.pop('workload', 'C3')]
g = ptcore.PathTracer(0).set_scene(make(), scenes.make_sky())
g.set_camera(cam_o, cam_t, W, H)
for k, v in opts.items():
    g.set_option(getattr(ptcore, 'OPT_' + k.upper()), int(v))
g.render(2, 32, 1)
for _ in range(reps):
    g.reset_counters()
    t = time.perf_counter()
    g.render(spp, 32, 7, None, ptcore.RENDER_TIMED)
    dt = time.perf_counter() - t
    c = g.counters()
    print(f"{desc[:2]} spp={spp} {opts} -> {c['rays_closest'] / dt / 1e6:8.1f} Mrays/s {dt * 1e3:7.1f} ms k_path busy {c['tail_ms']:.1f} ms", flush=True)
