"""Regenerates the derived golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself holds no fixtures beyond the 24
known-answer rays (tests/test_oracle_kat.py); these goldens freeze the ORACLE's outputs so a
change to either side is caught even without the other."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sky = scenes.make_sky()
    out = {}
    o_s = orc.from_scene(scenes.default_scene(), sky)
    o_c = orc.from_scene(scenes.glass_cube_scene(), sky)
    for name, o in (("spheres", o_s), ("cube", o_c)):
        nodes, idx, _ = o.bvh()
        out[name + "_nodes"] = nodes
        out[name + "_indices"] = idx
    np.savez_compressed(os.path.join(HERE, "bvh_small.npz"), **out)

    # traversal goldens: fixed ray batches -> (prim, t, inside)
    tr = {}
    rays_s = scenes.random_rays(2048, seed=21, box=((-2, -1.5, -0.5), (2, 2, 2)), aim=(0, -0.3, -2.0), aim_sigma=0.7)
    rays_c = scenes.random_rays(2048, seed=22)
    for name, o, rays in (("spheres", o_s, rays_s), ("cube", o_c, rays_c)):
        tr[name + "_rays"] = rays
        tr[name + "_closest"] = o.trace(rays, False)
        tr[name + "_any"] = o.trace(rays, True)
    np.savez_compressed(os.path.join(HERE, "trace_small.npz"), **tr)

    # image goldens: 64 x 64, 1 spp sums (bit-exact targets) and camera state
    im = {}
    o_s.set_camera(scenes.SPHERE_CAMERA_ORIGIN, scenes.SPHERE_CAMERA_TARGET, 64, 64)
    o_c.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 64, 64)
    for name, o in (("spheres", o_s), ("cube", o_c)):
        cam = o.camera()
        im[name + "_cam"] = np.concatenate([cam[k].ravel() for k in ("direction", "right", "up", "p1", "p2", "p3")] +
                                           [np.array([cam["focal_distance"]], np.float32)])
        o.render(4, 32, seed=7, nee_mode=0)
        acc, _ = o.read_accum()
        im[name + "_accum4"] = acc
        im[name + "_rgb8"] = o.read_rgb8()
    np.savez_compressed(os.path.join(HERE, "image_small.npz"), **im)

    # sky lookup goldens
    rng = np.random.default_rng(3)
    d = rng.normal(size=(512, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.concatenate([d, [[0, 1, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0]]]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "sky_lookup.npz"), dirs=d, rgb=o_s.sample_skybox(d))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
