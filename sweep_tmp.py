import sys, time
sys.path.insert(0,'.')
import torch
from pathtracer_amd import ptcore, scenes
g = ptcore.PathTracer(0).set_scene(scenes.statue_scene(), scenes.make_sky())
g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 1920, 1080)
g.render(8, 32, 1)
def run(spp, flags=0, **o):
    for k,v in o.items(): g.set_option(getattr(ptcore,'OPT_'+k.upper()), v)
    g.reset_counters(); t=time.perf_counter(); g.render(spp,32,7,None,flags); dt=time.perf_counter()-t
    c=g.counters(); print(f"spp={spp} {o} -> {c['rays_closest']/dt/1e6:8.1f} Mrays/s  {dt*1e3:7.1f} ms ext={c['extend_launches']} tail={c['tail_launches']} tail_ms={c['tail_ms']:.0f}", flush=True)
T=ptcore.RENDER_TIMED
run(64, T, tail_mode=2)
run(64, T, tail_mode=0)
for ppl in (1,2,4,8):
    run(64, T, tail_mode=0, tail_paths_per_lane=ppl)
run(64, T, tail_mode=0, tail_paths_per_lane=4, tail_fraction=4)
run(64, T, tail_mode=0, tail_paths_per_lane=4, tail_fraction=1, tail_min_paths=1<<30)
run(64, T, tail_mode=0, tail_paths_per_lane=8, tail_fraction=1, tail_min_paths=1<<30)
run(64, T, tail_mode=0, tail_paths_per_lane=4, tail_fraction=16, tail_min_paths=65536, tail_streams=2)
run(256, T, tail_mode=0, tail_paths_per_lane=4, tail_fraction=16, tail_min_paths=65536, tail_streams=4)
