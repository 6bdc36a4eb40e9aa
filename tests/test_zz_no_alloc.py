"""pt_trace_noalloc (include/ptcore.h): the entry point for callers that cannot allocate.  Its rays,
hits and stack-spill area live in a static device array; results must equal pt_trace's bit for bit,
also on a second, smaller call that reuses the array with the first call's contents still in it.
Collected last on purpose (the file name sorts after every other test file: pytest -x would otherwise hide later tests behind a failure here)."""
import numpy as np
import pytest

from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def _deep_scene():
    """Thousands of large overlapping triangles: traversals postpone more nodes than the LDS part of the
    per-lane stack holds, so the spill levels inside the static scratch array are really used."""
    n = 6000
    rng = np.random.default_rng(4)
    ctr = rng.uniform(-1, 1, size=(n, 3)) * [1, 0.3, 1] + [0, 0.3, 0]
    a = rng.normal(size=(n, 3)) * 1.5
    b = rng.normal(size=(n, 3)) * 1.5
    pos = np.concatenate([ctr, ctr + a, ctr + b], axis=1).astype(np.float32)
    return scenes.wrap_mesh(pos, scenes._flat_normals(pos), np.zeros(n, np.uint32), [scenes.GLASS_BLUE])


def _same(a, b):
    return (np.array_equal(a["prim"], b["prim"]) and np.array_equal(a["inside"], b["inside"])
            and a["t"].tobytes() == b["t"].tobytes())


def test_noalloc_trace_equals_trace_twice_in_a_row():
    from pathtracer_amd import ptcore
    sky = scenes.make_sky()
    for scene, box, aim in ((_deep_scene(), ((-3, -0.2, -3), (3, 1, 3)), (0, 0.1, 0)),
                            (scenes.default_scene(), ((-2.5, -0.3, -2.5), (2.5, 2.5, 2.5)), (0, -0.3, -2.0))):
        g = ptcore.PathTracer(0).set_scene(scene, sky)
        try:
            big = scenes.random_rays(16384, seed=8, box=box, aim=aim, aim_sigma=1.5)
            big[::3, 6] = np.random.default_rng(5).uniform(0.1, 4.0, size=len(big[::3]))  # finite ray.distance too
            small = scenes.random_rays(5000, seed=9, box=box, aim=aim, aim_sigma=1.0)
            for any_hit in (False, True):
                want_big, want_small = g.trace(big, any_hit), g.trace(small, any_hit)
                assert (want_big["prim"] != ptcore.MISS).sum() > 1000
                # two calls in a row, the second smaller: the array still holds the first call's rays, hits and stacks
                got_big = g.trace_noalloc(big, any_hit)
                got_small = g.trace_noalloc(small, any_hit)
                assert _same(got_big, want_big)
                assert _same(got_small, want_small)
            c0 = g.counters()["rays_closest"]
            g.trace_noalloc(small[:100])
            assert g.counters()["rays_closest"] == c0 + 100  # same counters as pt_trace
            assert len(g.trace_noalloc(np.zeros((0, 7), np.float32))) == 0
            with pytest.raises(ptcore.PtError) as e:
                g.trace_noalloc(scenes.random_rays(16385, seed=1))
            assert e.value.status == ptcore.PT_ERR_LIMIT and "16384" in str(e.value)
        finally:
            g.close()


def test_noalloc_before_build_is_a_state_error():
    from pathtracer_amd import ptcore
    g = ptcore.PathTracer(0)
    try:
        with pytest.raises(ptcore.PtError) as e:
            g.trace_noalloc(scenes.random_rays(4, seed=1))
        assert e.value.status == ptcore.PT_ERR_STATE
    finally:
        g.close()
