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
    c=g.counters(); print(f"spp={spp} {o} -> {c['rays_closest']/dt/1e6:8.1f} Mrays/s  {dt*1e3:7.1f} ms ext={c['extend_launches']} tail={c['tail_launches']} tail_ms={c['tail_ms']:.0f} paths={c['paths']}", flush=True)
T=ptcore.RENDER_TIMED
big=dict(tail_fraction=1, tail_min_paths=1<<30)
run(256, T, first_check_round=1, batch_paths=1<<26, **big)
for bp in (1<<23, 1<<25, 1<<26, 1<<27):
    run(256, T, render_mode=1, batch_paths=bp)
run(256, T, render_mode=1, batch_paths=1<<26, lanes=2)
run(256, T, render_mode=1, batch_paths=1<<26, lanes=1)
run(256, T, render_mode=1, batch_paths=1<<26, lanes=4, bulk_streams=1)
run(64, T, render_mode=1, batch_paths=1<<25, lanes=4, bulk_streams=8)
run(64, T, render_mode=1, batch_paths=1<<28, lanes=4, bulk_streams=8)
