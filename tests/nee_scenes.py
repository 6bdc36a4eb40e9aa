"""Scenes in which leaving the NEE block out where its weight is 0 (quirk Q2, camera.rs:333) would change a sum:
the block's product is not finite there, and 0 * inf = NaN is what the reference adds.  Used by the CPU test of the
oracle's restatement of the device rule (nee_mode 1 against the literal nee_mode 0) and by the GPU parity tests."""
import numpy as np

from pathtracer_amd import scenes


def absorbing_glass_mesh():
    """Glass whose Beer factor exp((color - 1) * distance) (camera.rs:376-380) overflows inside the statue: every path that
    crosses it leaves with an infinite transport and then meets the diffuse floor (one colour channel 0: inf * 0 = NaN)."""
    pos, nrm = scenes.statue_mesh(24, 12, 8, 3)
    glass = scenes.dielectric(1.0, 1.5, (300.0, 300.0, 0.5))
    p, n, ids, mats = scenes.wrap_mesh(pos, nrm, np.zeros(len(pos), np.uint32), [glass], floor_speculaty=0.25)
    mats[-1]["rgb"] = (0.9, 0.0, 0.9)
    return p, n, ids, mats


def overflowing_mirrors_mesh():
    """Colours far above 1 on a half-specular floor and a diffuse statue: the specular branch multiplies the transport by
    the colour with no renormalisation (camera.rs:341-345), so it runs up to FLT_MAX and beyond within a few bounces —
    finite but past the threshold first, infinite later."""
    pos, nrm = scenes.statue_mesh(24, 12, 8, 3)
    body = scenes.diffuse(0.7, (2e13, 1e13, 3e13))
    p, n, ids, mats = scenes.wrap_mesh(pos, nrm, np.zeros(len(pos), np.uint32), [body], floor_speculaty=0.5)
    mats[-1]["rgb"] = (4e12, 4e12, 5e12)
    return p, n, ids, mats


def infinite_light_mesh():
    """A light whose colour is infinite: every completed NEE block adds 0 * inf = NaN in the reference, so such a scene
    always runs the block (nee_literal)."""
    p, n, ids, mats = scenes.small_mesh_scene()
    mats = mats.copy()
    k = int(np.nonzero(mats["kind"] == scenes.MAT_EMISSIVE)[0][0])
    mats[k]["rgb"] = (np.inf, 1.7, 1.4)
    return p, n, ids, mats


def degenerate_light_mesh():
    """A second light with two equal corners: Triangle::area() (triangle.rs:105-113) normalises a zero edge, so its area
    is NaN — the bound on the block's product does not hold for it, and the scene always runs the block."""
    p, n, ids, mats = scenes.small_mesh_scene()
    light = int(np.nonzero(mats["kind"] == scenes.MAT_EMISSIVE)[0][0])
    extra_p = np.array([[0.2, 1.5, 0.2, 0.2, 1.5, 0.2, 0.9, 1.5, 0.7]], np.float32)
    extra_n = np.array([[0, -1, 0] * 3], np.float32)
    return (np.concatenate([p, extra_p]), np.concatenate([n, extra_n]), np.concatenate([ids, np.array([light], np.uint32)]), mats)


def absorbing_glass_spheres():
    """Scene<Sphere>: the reference's seven spheres with overflowing glass in the middle and a zero colour channel below it."""
    cr, ids, mats = scenes.default_scene()
    mats = mats.copy()
    mats[2]["rgb"] = (400.0, 0.5, 400.0)
    mats[4]["rgb"] = (1.0, 0.0, 1.0)
    return cr, ids, mats


# name -> (scene maker, camera origin, camera target, must the fast kernels hand paths to the second pass?)
NEE_SCENES = {
    "absorbing_glass": (absorbing_glass_mesh, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, True),
    "overflowing_mirrors": (overflowing_mirrors_mesh, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, True),
    "infinite_light": (infinite_light_mesh, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, False),
    "degenerate_light": (degenerate_light_mesh, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, False),
    "absorbing_glass_spheres": (absorbing_glass_spheres, scenes.SPHERE_CAMERA_ORIGIN, scenes.SPHERE_CAMERA_TARGET, True),
}
