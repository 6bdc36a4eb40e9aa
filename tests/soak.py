"""Randomised parity campaign, GPU (through the C ABI) against the CPU oracle, bit for bit:
   python3 tests/soak.py [minutes] [first_seed] [image_scale]
Every round draws a scene (soup / statue / spheres of random size and materials, sometimes with a
specular floor or a speculaty > 1 material so that shadow rays run), a camera, an image size, a
depth and a sample count, and compares BVH, 20 000 traces (closest and any), camera state, and
the frame sums in both pt_render modes and as queued one-sample calls (few and many, merged under a random cap).  Stops at the first difference (exit code 1).
From round 3 the frame is compared with the loop AS WRITTEN (oracle nee_mode 0: the NEE block at every diffuse hit), and one scene in six gets
colours that run the transport up to inf / NaN, so that the device's second, literal pass (k_redo) and its restart with the NEE kernels (random
redo-list capacity) are part of the campaign."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: F401,E402
from oracle import orc  # noqa: E402
from pathtracer_amd import ptcore, scenes  # noqa: E402


SCALE = 1  # image size scale (third argument)


def draw_scene(rng):
    kind = rng.integers(0, 5)
    if kind == 0:
        n = int(rng.integers(50, 60000))
        pos, nrm, mat, mats = scenes.soup_scene(n, int(rng.integers(1, 1 << 30)), floor_speculaty=float(rng.choice([0.0, 0.5, 1.0])))
        if rng.random() < 0.3:
            mats = mats.copy()
            mats["a"][rng.integers(0, 64)] = 1.5  # speculaty > 1: the NEE weight becomes non-zero (quirk Q2)
        return "soup%d" % n, (pos, nrm, mat, mats)
    if kind == 1:
        nu, nv = int(rng.integers(8, 160)), int(rng.integers(6, 90))
        glass = scenes.GLASS_BLUE if rng.random() < 0.5 else scenes.GLASS_GREEN
        return "statue%dx%d" % (nu, nv), scenes.statue_scene(nu, nv, int(rng.integers(4, 64)), int(rng.integers(1, 1000)), glass, float(rng.choice([0.0, 0.5])))
    if kind == 4:  # large overlapping triangles: deep trees, long stacks (beyond the LDS levels), many coincident boxes
        n = int(rng.integers(200, 5000))
        ctr = rng.uniform(-1, 1, size=(n, 3)) * [1, 0.3, 1] + [0.7, 0.1, 0.6]
        a, b = rng.normal(size=(n, 3)) * rng.uniform(0.2, 1.5), rng.normal(size=(n, 3)) * rng.uniform(0.2, 1.5)
        pos = np.concatenate([ctr, ctr + a, ctr + b], axis=1).astype(np.float32)
        if rng.random() < 0.5:
            pos[: n // 10] = pos[0]  # a stack of identical triangles -> a leaf with many primitives
        rows = [scenes.GLASS_BLUE, scenes.diffuse(float(rng.choice([0.0, 0.3, 1.2])), (0.7, 0.6, 0.5))]
        return "overlap%d" % n, scenes.wrap_mesh(pos, scenes._flat_normals(pos), rng.integers(0, 2, n).astype(np.uint32), rows)
    if kind == 2:
        return "cube", scenes.glass_cube_scene(floor_speculaty=float(rng.choice([0.0, 0.5])), lift=float(rng.choice([0.0, 0.2])))
    if rng.random() < 0.5:  # random spheres, some of them lights, seen from speculaty > 1 materials: sphere-light NEE (sphere.rs:98-118)
        n = int(rng.integers(3, 400))
        cr = np.concatenate([rng.uniform(-1.5, 1.5, size=(n, 3)) * [1, 0.6, 1] + [0, -0.3, -2.0], rng.uniform(0.03, 0.4, size=(n, 1))], axis=1).astype(np.float32)
        rows = [scenes.diffuse(float(rng.choice([0.0, 0.4, 1.5])), (0.8, 0.7, 0.6)), scenes.emissive(), scenes.GLASS_BLUE, scenes.diffuse(1.3, (0.5, 0.6, 0.9))]
        return "spheres", (cr, rng.integers(0, 4, n).astype(np.uint32), scenes.materials(rows))
    return "spheres", scenes.default_scene()


def spoil_colours(rng, scene):
    """Colours that make transports overflow: a glass whose Beer factor is exp(+large), diffuse colours far above 1, a zero channel."""
    mats = scene[-1].copy()
    for m in mats:
        if m["kind"] == scenes.MAT_DIELECTRIC and rng.random() < 0.7:
            m["rgb"] = tuple(float(x) for x in rng.choice([0.5, 60.0, 300.0], size=3))
        elif m["kind"] == scenes.MAT_DIFFUSE and rng.random() < 0.5:
            m["rgb"] = tuple(float(x) for x in rng.choice([0.0, 0.9, 3e12, 2e19], size=3))
    return scene[:-1] + (mats,)


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global SCALE
    SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    sky = scenes.make_sky()
    t_end = time.time() + 60 * minutes
    rounds = device_builds = redo_paths = redo_restarts = 0
    while time.time() < t_end:
        seed = seed0 + rounds
        rng = np.random.default_rng(seed)
        name, scene = draw_scene(rng)
        if rng.random() < 1.0 / 6.0:
            name, scene = name + "+spoilt", spoil_colours(rng, scene)
        g = ptcore.PathTracer(0).set_scene(scene, sky)
        g.set_option(ptcore.OPT_REDO_CAPACITY, int(rng.choice([65536, 65536, 5, 0])))
        g.set_option(ptcore.OPT_WAVES_PER_SIMD, int(rng.choice([0, 6, 7])))  # both forms of the path kernel (these scenes would all get the 6-wave one)
        device_builds += int(g.build_info()[0])
        o = orc.from_scene(scene, sky)
        what = "bvh"
        ok = all(a.tobytes() == b.tobytes() for a, b in zip(g.bvh()[:2], o.bvh()[:2]))
        if ok:
            what = "trace"
            rays = scenes.random_rays(20000, seed=seed)
            for any_hit in (False, True):
                a, b = g.trace(rays, any_hit), o.trace(rays, any_hit)
                ok = ok and a.tobytes() == b.tobytes()
        if ok:
            what = "render"
            spheres = name == "spheres"
            co = scenes.SPHERE_CAMERA_ORIGIN if spheres else scenes.CAMERA_ORIGIN
            ct = scenes.SPHERE_CAMERA_TARGET if spheres else scenes.CAMERA_TARGET
            co = tuple(np.float32(co) + rng.normal(scale=0.3, size=3).astype(np.float32))
            W, H = int(rng.integers(1, 200 * SCALE)), int(rng.integers(1, 120 * SCALE))
            depth, spp = int(rng.choice([0, 1, 2, 5, 32])), int(rng.integers(1, 5))
            o.set_camera(co, ct, W, H)
            o.render(spp, depth, seed=seed, nthreads=8, nee_mode=0)  # camera.rs:314-338 as written
            want = o.read_accum()[0]
            for mode in (1, 0):
                g.set_option(ptcore.OPT_RENDER_MODE, mode)
                g.set_camera(co, ct, W, H)
                g.render(spp, depth, seed=seed)
                ok = ok and g.read_accum()[0].tobytes() == want.tobytes()
            ok = ok and g.read_rgb8().tobytes() == o.read_rgb8().tobytes()
            # the same frame as queued one-sample calls (pt_render_async): overlapping launches, same sums
            g.set_option(ptcore.OPT_RENDER_MODE, 1)
            g.set_camera(co, ct, W, H)
            for _ in range(spp):
                g.render_async(1, depth, seed=seed)
            ok = ok and g.read_accum()[0].tobytes() == want.tobytes()
            # many queued calls behind few lanes: the library merges those that have not started (random cap); against
            # one blocking call of as many samples
            calls = int(rng.integers(6, 30))
            g.set_camera(co, ct, W, H)
            g.render(calls, depth, seed=seed)
            whole = g.read_accum()[0]
            g.set_option(ptcore.OPT_LANES, int(rng.integers(1, 4)))
            g.set_option(ptcore.OPT_MERGE_QUEUED_CALLS, int(rng.choice([0, 2 * W * H, 5 * W * H, 128 << 20])))
            g.set_camera(co, ct, W, H)
            for _ in range(calls):
                g.render_async(1, depth, seed=seed)
            ok = ok and g.read_accum()[0].tobytes() == whole.tobytes()
            what += " %dx%d depth %d spp %d, %d queued calls" % (W, H, depth, spp, calls)
        if ok and "render" in what:
            c = g.counters()
            redo_paths += int(c["redo_paths"])
            redo_restarts += int(c["redo_restarts"])
        g.close()
        o.close()
        rounds += 1
        print(("ok   " if ok else "DIFF ") + "seed %d %s %s" % (seed, name, what), flush=True)
        if not ok:
            sys.exit(1)
    print("soak: %d rounds (%d trees built on the GPU; %d paths finished by the literal second pass, %d batches run again with the NEE kernels), no difference"
          % (rounds, device_builds, redo_paths, redo_restarts))


if __name__ == "__main__":
    main()
