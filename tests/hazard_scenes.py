"""Inputs on which BVH::partition (bvh.rs:63-103) leaves one child empty: the split is chosen from the bins
(`(centre - min) / delta`, bvh.rs:122-127) but applied with `centre <= pivot`, pivot = min + (1 + bin) * delta (bvh.rs:166);
next to 2^24, where a float32 ulp is 2 and delta is below it, the pivot rounds past every centre.  The reference would then
treat the empty child as an internal node and split the full one the same way for ever (SURVEY 3.5 "Hazard"); every
builder here must stop and report PT_ERR_BVH instead.  Found by random search against the oracle (5 spheres suffice)."""
import numpy as np

BASE = np.float32(2.0 ** 24)
OFFSETS = [0, 14, 4, 6, 10]
RADII = [2.0, 0.25, 0.25, 8.0, 4.0]


def hazard_cluster(copies=1, at=(0.0, 0.0)):
    """(n, 4) spheres: the five-sphere pattern, each sphere `copies` times (coincident copies scale every SAH cost by the
    same factor, so the same split is chosen and the root itself — of 5 * copies primitives — hits the hazard)."""
    cr = np.zeros((5 * copies, 4), np.float32)
    cr[:, 0] = np.repeat(BASE + np.array(OFFSETS, np.float32), copies)
    cr[:, 1], cr[:, 2] = at
    cr[:, 3] = np.repeat(np.array(RADII, np.float32), copies)
    return cr


def ordinary_spheres(n, seed=2):
    rng = np.random.default_rng(seed)
    cr = np.zeros((n, 4), np.float32)
    cr[:, :3] = rng.uniform(-3, 3, size=(n, 3))
    cr[:, 3] = rng.uniform(0.01, 0.1, size=n)
    return cr


# name -> spheres.  "wave": the whole tree is one small subtree (one wave finishes it); "level": the root has more primitives
# than a wave takes, so the hazard shows in the level-by-level passes; "deep": ordinary geometry with the cluster far away —
# the first splits are fine and the hazard sits in a small subtree several levels down.
HAZARD_SCENES = {
    "wave": lambda: hazard_cluster(1),
    "level": lambda: hazard_cluster(60),
    "deep": lambda: np.concatenate([ordinary_spheres(4000), hazard_cluster(1)]),
    # the host builder's other code paths: a root that is split with several threads (>= 8 192 primitives) and one that takes
    # the all-threads partition (>= 65 536)
    "threads": lambda: hazard_cluster(2000),
    "large": lambda: hazard_cluster(14000),
}
