"""The documented maximum, 2^26 primitives, end to end: python tools/max_size.py [log2_n=26] [oracle=1] [triangles|spheres]

A triangle soup of exactly 2^n triangles (light and floor included): tree built on the GPU against the threaded host
build and against the oracle's scalar build (same nodes, same index order), 2 M rays through the GPU and 100 k of them
through the oracle (same hits), one small frame rendered.  A tool, not a suite test: the scene alone takes minutes to
generate and the oracle's build as long again."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: F401,E402
from pathtracer_amd import ptcore, scenes  # noqa: E402
from oracle import orc  # noqa: E402

log2_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
with_oracle = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
spheres = len(sys.argv) > 3 and sys.argv[3] == "spheres"
T0 = time.time()


def say(*a):
    print(f"[{time.time() - T0:7.1f} s]", *a, flush=True)


alive = True


def heartbeat():
    while alive:
        time.sleep(45)
        if alive:
            say("...")


threading.Thread(target=heartbeat, daemon=True).start()

if spheres:  # small spheres all over the soup's box, the sphere scene's materials (every seventh sphere a light)
    rng = np.random.default_rng(11)
    n = 1 << log2_n
    pos = np.empty((n, 4), np.float32)
    pos[:, :3] = rng.uniform([-2, -0.3, -2], [2, 1.7, 2], size=(n, 3))
    pos[:, 3] = rng.uniform(0.0005, 0.004, size=n)
    mats = scenes.default_scene()[2]
    mat = (np.arange(n) % len(mats)).astype(np.uint32)
    nrm = None
    say(f"{n} spheres generated")
else:
    n = (1 << log2_n) - 3
    pos, nrm, mat, mats = scenes.soup_scene(n=n, seed=11)
    say(f"{len(pos)} triangles generated ({pos.nbytes / 2**30:.1f} GiB of positions)")
assert len(pos) == 1 << log2_n

sky = scenes.make_sky()
g = ptcore.PathTracer(0)
g.set_materials(mats)
if spheres:
    g.set_spheres(pos, mat)
else:
    g.set_triangles(pos, nrm, mat)
g.set_sky(sky)
trees = {}
for on in (1, 0):
    g.set_option(ptcore.OPT_DEVICE_BUILD, on)
    t0 = time.time()
    g.build()
    wall = time.time() - t0
    dev, ms = g.build_info()
    nodes, idx, depth = g.bvh()
    trees[on] = (nodes.copy(), idx.copy(), depth)
    say(f"{'device' if dev else 'host  '} build: tree {ms:9.1f} ms, pt_build in all {wall * 1e3:9.1f} ms, {len(nodes)} nodes, depth {depth}")
same = trees[1][2] == trees[0][2] and np.array_equal(trees[1][1], trees[0][1]) and trees[1][0].tobytes() == trees[0][0].tobytes()
say("device tree == host tree:", same)
ok = same

g.set_option(ptcore.OPT_DEVICE_BUILD, 1)
g.build()
rays = scenes.random_rays(2_000_000, seed=3, aim=(0.0, 0.7, 0.0), aim_sigma=1.0)
t0 = time.time()
hits = g.trace(rays)
say(f"2 M rays traced in {time.time() - t0:.2f} s (host buffers), {int((hits['prim'] != ptcore.MISS).sum())} hit")
g.set_camera((0.0, 1.0, 3.5), (0.0, 0.6, 0.0), 256, 256)
g.render(4, 8, seed=1)
acc, spp = g.read_accum()
say(f"256 x 256 frame, {spp} spp: mean {float(np.nanmean(acc)):.4f}, finite {bool(np.isfinite(acc).all())}")

if with_oracle:
    o = orc.Oracle()
    o.set_materials(mats)
    if spheres:
        o.set_spheres(pos, mat)
    else:
        o.set_triangles(pos, nrm, mat)
    o.set_sky(sky)
    t0 = time.time()
    o.build()
    onodes, oidx, odepth = o.bvh()
    say(f"oracle build {time.time() - t0:.1f} s, {len(onodes)} nodes, depth {odepth}")
    same_o = odepth == trees[1][2] and np.array_equal(oidx, trees[1][1]) and np.asarray(onodes).tobytes() == trees[1][0].tobytes()
    say("device tree == oracle tree:", same_o)
    sub = rays[:100_000]
    oh = o.trace(sub)
    gh = g.trace(sub)
    same_h = np.asarray(oh).tobytes() == np.asarray(gh).tobytes()
    say("100 k rays, GPU hits == oracle hits:", same_h)
    # the frame above through the oracle: k_path's own node and primitive fetches at the size limit (the last records of both arrays included)
    o.set_camera((0.0, 1.0, 3.5), (0.0, 0.6, 0.0), 256, 256)
    t0 = time.time()
    o.render(4, 8, seed=1)
    oacc, ospp = o.read_accum()
    same_f = ospp == spp and np.asarray(oacc).tobytes() == np.asarray(acc).tobytes()
    say(f"256 x 256 frame through the oracle in {time.time() - t0:.1f} s, GPU sums == oracle sums: {same_f}")
    ok = ok and same_o and same_h and same_f
alive = False
say("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
