import sys
sys.path.insert(0, '.')
import torch
from pathtracer_amd import ptcore, scenes
import bench
scene = bench.workloads(scenes)["C5"][0]()
g = ptcore.PathTracer(0)
g.set_materials(scene[3]); g.set_triangles(scene[0], scene[1], scene[2]); g.set_sky(scenes.make_sky())
for _ in range(3): g.build()
print(g.build_info())
