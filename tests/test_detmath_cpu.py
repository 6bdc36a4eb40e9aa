"""include/ptcore_detmath.h is shared by the product (device code) and the oracle, so a bug in it is
invisible to every bit-exact comparison between the two.  This test bounds each pt_*f against two
independent references over the ranges the path uses:
  * the platform libm in fp32 (what the reference's f32::sin/cos/exp/acos/atan2 call), and
  * numpy in fp64 (the true value).
Stated bounds, in units of the last place (ulp) of the reference value:
  pt_sincosf  <= 2 ulp for x in [0, 2 pi] (phi = 2 pi r1, camera.rs:77) and in [-8, 8]; where the true
              value is below 0.25 the bound is 2^-23 absolute (a quadrant's polynomial error does not shrink
              with the result)
  pt_expf     <= 2 ulp for x in [-100, 0] (Beer absorption (color - 1) * t <= 0, camera.rs:376-380) and [0, 80]
  pt_acosf    <= 2 ulp on [-1, 1]; NaN outside (scene.rs:164 then reads texel row 0)
  pt_atan2f   <= 4 ulp for finite non-zero inputs; exact C99 results on the axes (scene.rs:163)
  pt_cbrtf    <= 2 ulp on [0, 1) (u.cbrt() of a uniform sample, sphere.rs:108) and over all finite magnitudes,
              subnormals included; zeros, infinities and NaN as libm
(Round 1 shipped pt_sincosf with a wrong Cody-Waite constant: 0.03 rad of error at x = 2 pi, found by this test.)"""
import numpy as np
import pytest

from oracle import orc

f32 = np.float32
SIN, COS, EXP, ACOS, ATAN2, CBRT = 0, 1, 2, 3, 4, 5


def ulps(got, want64):
    want32 = want64.astype(f32)
    unit = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / unit


def both_refs(fn, x, y=None):
    """(libm fp32 result as fp64, numpy fp64 result)"""
    libm = orc.math_fn(fn, x, y, mode=1).astype(np.float64)
    x64 = x.astype(np.float64)
    true = {SIN: np.sin, COS: np.cos, EXP: np.exp, ACOS: np.arccos}[fn](x64) if fn != ATAN2 else np.arctan2(x64, y.astype(np.float64))
    return libm, true


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(2026)


@pytest.mark.parametrize("fn", [SIN, COS])
def test_sincos(rng, fn):
    # every 37th value the renderer can pass (phi = 2 pi * closed01(u32 >> 8)), plus random arguments
    grid = (f32(2.0) * f32(np.pi)) * (np.arange(0, 1 << 24, 37, dtype=np.float64) * (1.0 / 16777215.0)).astype(f32)
    x = np.concatenate([grid, rng.uniform(0, 2 * np.pi, 1_000_000).astype(f32), rng.uniform(-8, 8, 500_000).astype(f32),
                        np.array([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi], f32)]).astype(f32)
    got = orc.math_fn(fn, x, mode=0)
    for ref in both_refs(fn, x):
        big = np.abs(ref) >= 0.25
        assert ulps(got[big], ref[big]).max() <= 2.0
        assert np.abs(got[~big].astype(np.float64) - ref[~big]).max() <= 2.0 ** -23
    assert np.all(np.abs(got) <= 1.0)


def test_exp(rng):
    x = np.concatenate([rng.uniform(-100, 0, 1_000_000), -np.exp(rng.uniform(np.log(1e-8), np.log(100.0), 500_000)),
                        rng.uniform(0, 80, 200_000), [0.0, -0.0, -87.0, -103.0]]).astype(f32)
    got = orc.math_fn(EXP, x, mode=0)
    for ref in both_refs(EXP, x):
        normal = ref > 1.2e-38
        assert ulps(got[normal], ref[normal]).max() <= 2.0
        assert np.abs(got[~normal].astype(np.float64) - ref[~normal]).max() <= 2.0 ** -148  # denormal results: 2 denormal steps
    special = orc.math_fn(EXP, np.array([-np.inf, -200.0, 100.0, np.inf, np.nan], f32), mode=0)
    assert special[0] == 0 and special[1] == 0 and np.isinf(special[2]) and np.isinf(special[3]) and np.isnan(special[4])


def test_acos(rng):
    x = np.concatenate([rng.uniform(-1, 1, 1_000_000), 1 - np.exp(rng.uniform(np.log(1e-7), 0, 200_000)),
                        -1 + np.exp(rng.uniform(np.log(1e-7), 0, 200_000)), [1.0, -1.0, 0.0, 0.5, -0.5]]).astype(f32)
    x = np.clip(x, -1, 1)
    got = orc.math_fn(ACOS, x, mode=0)
    for ref in both_refs(ACOS, x):
        nz = ref != 0
        assert ulps(got[nz], ref[nz]).max() <= 2.0
        assert np.all(got[~nz] == 0)
    outside = orc.math_fn(ACOS, np.array([1.0000001, -1.0000001, 2.0, np.nan], f32), mode=0)
    assert np.isnan(outside).all()


def test_atan2(rng):
    a = np.concatenate([rng.normal(size=1_000_000), rng.normal(size=200_000) * 1e-3, rng.normal(size=200_000)]).astype(f32)
    b = np.concatenate([rng.normal(size=1_000_000), rng.normal(size=200_000), rng.normal(size=200_000) * 1e-3]).astype(f32)
    keep = (a != 0) & (b != 0)
    a, b = a[keep], b[keep]
    got = orc.math_fn(ATAN2, a, b, mode=0)
    for ref in both_refs(ATAN2, a, b):
        assert ulps(got, ref).max() <= 4.0
    # axes and signed zeros follow C99 (the sky lookup passes atan2(d.x, -d.z) with d.z = +-0 for axis-parallel rays)
    ya = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, -1.0, 0.0, -0.0], f32)
    xa = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, -0.0, -0.0, 0.0, -0.0], f32)
    got = orc.math_fn(ATAN2, ya, xa, mode=0)
    want = orc.math_fn(ATAN2, ya, xa, mode=1)
    assert got.tobytes() == want.tobytes()


def test_cbrt(rng):
    xs = [rng.random(400_000).astype(f32),                                              # the range the path uses
          np.exp(rng.uniform(-103.0, 88.0, size=300_000)).astype(f32),                  # every binade, subnormals included
          np.array([2.0 ** -149, 2.0 ** -127, 2.0 ** -126, 1.0, 8.0, 27.0, 2.0 ** 127, 3.4e38], f32)]
    for x in xs:
        x = np.concatenate([x, -x])
        got = orc.math_fn(CBRT, x, mode=0)
        libm = orc.math_fn(CBRT, x, mode=1).astype(np.float64)
        true = np.cbrt(x.astype(np.float64))
        assert ulps(got, libm).max() <= 2.0
        assert ulps(got, true).max() <= 2.0
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan], f32)
    got, want = orc.math_fn(CBRT, special, mode=0), orc.math_fn(CBRT, special, mode=1)
    assert got[:4].tobytes() == want[:4].tobytes() and np.isnan(got[4])
    assert orc.math_fn(CBRT, np.array([8.0, 27.0, 1.0, 0.125], f32), mode=0).tolist() == [2.0, 3.0, 1.0, 0.5]
