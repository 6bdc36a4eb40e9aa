"""Measures the algorithmic work of the BASELINE configs with the CPU oracle's counters (SURVEY 8d:
"n_node, n_tri, rays/path are measured by the oracle's counters per config and committed") and
writes tests/golden/workloads.json.  1 spp of each full-size frame, seed 2026, nee_mode 1 (shadow
rays only where their weight is non-zero, as the device does).  Run from the repo root:
`python tests/golden/make_workloads.py [C2 C3 C4 C5]` (C5 builds a 10 M-triangle tree: minutes)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED, DEPTH = 2026, 32
CONFIGS = {
    "C2": (lambda: scenes.glass_cube_scene(), 1920, 1080),
    "C3": (lambda: scenes.statue_scene(), 1920, 1080),
    "C4": (lambda: scenes.statue_scene(1042, 520, 1009, 7, scenes.GLASS_BLUE, 0.5), 3840, 2160),
    "C5": (lambda: scenes.soup_scene(), 3840, 2160),
}


def main():
    path = os.path.join(HERE, "workloads.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    sky = scenes.make_sky()
    for name in sys.argv[1:] or ["C2", "C3", "C4"]:
        make, W, H = CONFIGS[name]
        t0 = time.time()
        o = orc.from_scene(make(), sky)
        o.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, W, H)
        o.reset_counters()
        o.render(1, DEPTH, seed=SEED, nthreads=8, nee_mode=1)
        c = o.counters()
        rays = c["rays_closest"] + c["rays_any"]
        out[name] = {
            "width": W, "height": H, "spp": 1, "max_depth": DEPTH, "seed": SEED,
            "paths": int(c["paths"]), "rays_closest": int(c["rays_closest"]), "rays_any": int(c["rays_any"]),
            "node_tests": int(c["node_tests"]), "prim_tests": int(c["prim_tests"]),
            "rays_per_path": rays / c["paths"], "node_tests_per_ray": c["node_tests"] / rays,
            "prim_tests_per_ray": c["prim_tests"] / rays,
            "B_ray": (32 * c["node_tests"] + 36 * c["prim_tests"]) / rays + 40,
        }
        print(name, out[name], f"{time.time() - t0:.1f} s", flush=True)
        o.close()
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
