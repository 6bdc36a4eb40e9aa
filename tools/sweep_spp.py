"""Mrays/s against samples per call, queued and blocking (the spp_sweep of bench.py on its own): python3 tools/sweep_spp.py [workload] [lanes...]"""
import sys, json
sys.path.insert(0, '.')
import torch  # noqa: F401
import bench
from pathtracer_amd import ptcore, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
make, W, H, cam_o, cam_t, desc = bench.workloads(scenes)[wl]
g = ptcore.PathTracer(0).set_scene(make(), scenes.make_sky())
g.set_camera(cam_o, cam_t, W, H)
for lanes in [int(a) for a in sys.argv[2:]] or [0]:
    g.set_option(ptcore.OPT_LANES, lanes)
    r = bench.spp_sweep(g, ptcore, W, H, 32, 2026)
    print(wl, "lanes", lanes, {k: (round(v["mrays_per_s"]), round(v["ms_per_call"], 2)) for k, v in r.items()}, flush=True)
