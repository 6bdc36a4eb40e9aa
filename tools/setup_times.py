import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from pathtracer_amd import ptcore, scenes
import bench
sky = scenes.make_sky()
for wl in ("C3", "C5"):
    scene = bench.workloads(scenes)[wl][0]()
    for rep in range(2):
        t = [time.perf_counter()]
        g = ptcore.PathTracer(0); t.append(time.perf_counter())
        g.set_materials(scene[3]); t.append(time.perf_counter())
        g.set_triangles(scene[0], scene[1], scene[2]); t.append(time.perf_counter())
        g.set_sky(sky); t.append(time.perf_counter())
        g.build(); t.append(time.perf_counter())
        g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 1920, 1080); t.append(time.perf_counter())
        names = ["create", "set_materials", "set_triangles", "set_sky", "build", "set_camera"]
        print(wl, rep, " ".join("%s %.1f" % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])), flush=True)
        g.close()
