"""The oracle's two NEE modes.  nee_mode 0 is camera.rs:314-338 as written: the block runs at every diffuse hit of a scene
with a light, although its weight min(1 - speculaty, 0) is 0 for speculaty <= 1 (quirk Q2).  nee_mode 1 restates the
device's rule (pt_kernels.hip shade_core): the block is left out where the weight is 0 — except where its product may not
be finite, because 0 * inf = NaN is what the reference then adds.  Both modes must give the same sums, bit for bit, also in
scenes built to make the difference show; a mode that only looked at the weight (round 2's rule) must NOT."""
import numpy as np
import pytest

from nee_scenes import NEE_SCENES
from pathtracer_amd import scenes

W, H, SPP = 72, 48, 3


def frames(orc, name, sky):
    make, cam_o, cam_t, _ = NEE_SCENES[name]
    o = orc.from_scene(make(), sky)
    out = {}
    for mode in (0, 1, 2):
        o.set_camera(cam_o, cam_t, W, H)
        o.reset_counters()
        o.render(SPP, 32, seed=23, nthreads=4, nee_mode=mode)
        out[mode] = (o.read_accum()[0].copy(), dict(o.counters()))
    o.close()
    return out


@pytest.mark.parametrize("name", list(NEE_SCENES))
def test_device_rule_equals_the_literal_loop(orc, small_sky, name):
    out = frames(orc, name, small_sky)
    lit, rule = out[0][0], out[1][0]
    assert lit.tobytes() == rule.tobytes()
    if name != "degenerate_light":   # (its NaN area is never used: Triangle::intersect rejects the degenerate light itself)
        assert (~np.isfinite(lit)).sum() > 50, "the scene must really produce non-finite sums"
    if name in ("absorbing_glass", "infinite_light", "absorbing_glass_spheres"):
        # negative control: looking at the weight alone (round 2's rule) gives other sums here
        assert out[2][0].tobytes() != lit.tobytes()
    assert out[1][1]["rays_any"] > 0                       # the rule ran the block somewhere ...
    assert out[1][1]["rays_any"] <= out[0][1]["rays_any"]  # ... the literal loop at least as often


def test_ordinary_scenes_never_run_the_block(orc, small_sky):
    """BASELINE-like scenes (speculaty <= 1, ordinary colours): no shadow ray under the rule, same sums as the literal loop."""
    o = orc.from_scene(scenes.small_mesh_scene(), small_sky)
    res = {}
    for mode in (0, 1):
        o.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, W, H)
        o.reset_counters()
        o.render(SPP, 32, seed=23, nthreads=4, nee_mode=mode)
        res[mode] = (o.read_accum()[0].copy(), dict(o.counters()))
    o.close()
    assert res[0][0].tobytes() == res[1][0].tobytes()
    assert res[1][1]["rays_any"] == 0 and res[0][1]["rays_any"] > 0
