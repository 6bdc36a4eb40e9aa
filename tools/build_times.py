"""pt_build: tree build on the GPU against the threaded host build (same tree): python tools/build_times.py [C3 C4 C5]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: F401,E402
from pathtracer_amd import ptcore, scenes  # noqa: E402
import bench  # noqa: E402

sky = scenes.make_sky()
for wl in (sys.argv[1:] or ["C3", "C4", "C5"]):
    make = bench.workloads(scenes)[wl][0]
    t0 = time.time()
    scene = make()
    print(f"{wl}: {len(scene[0])} triangles (scene generated in {time.time() - t0:.1f} s)", flush=True)
    g = ptcore.PathTracer(0)
    g.set_materials(scene[3])
    g.set_triangles(scene[0], scene[1], scene[2])
    g.set_sky(sky)
    trees = {}
    for on in (1, 1, 0):
        g.set_option(ptcore.OPT_DEVICE_BUILD, on)
        t0 = time.time()
        g.build()
        wall = time.time() - t0
        dev, ms = g.build_info()
        nodes, idx, depth = g.bvh()
        trees[on] = (nodes.tobytes(), idx.tobytes(), depth)
        print(f"  {'device' if dev else 'host  '} build: tree {ms:9.1f} ms, pt_build in all {wall * 1e3:9.1f} ms, {len(nodes)} nodes, depth {depth}", flush=True)
    print("  identical:", trees[1] == trees[0], flush=True)
    g.close()
