"""The documented image limit, W x H x samples-per-batch < 2^31, end to end: python tools/max_image.py

The glass-cube scene at 46336 x 46336 (2.147 G pixels, 7 % under the limit), as a 3-pixel-wide column and as a
3-pixel-high row of 2^27 + 5 pixels: one sample per pixel, 6000 random pixels plus the four corners against the oracle
bit for bit, the RGB readout of the same pixels against the oracle's tonemap, and the first size over the limit refused."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: F401,E402
from pathtracer_amd import ptcore, scenes  # noqa: E402
from oracle import orc  # noqa: E402

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
sky = scenes.make_sky()
scene = scenes.glass_cube_scene()
g = ptcore.PathTracer(0)
g.set_scene(scene, sky)
o = orc.from_scene(scene, sky)
depth, seed = 8, 9
ok = True
for W, H in ((46336, 46336), (3, (1 << 27) + 5), ((1 << 27) + 5, 3)):
    g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, W, H)
    o.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, W, H)
    t0 = time.time()
    g.render(1, depth, seed=seed)
    dt = time.time() - t0
    acc, spp = g.read_accum()
    rng = np.random.default_rng(W % 1000)
    xy = np.stack([rng.integers(0, W, 6000), rng.integers(0, H, 6000)], axis=1).astype(np.uint32)
    xy[:4] = [[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1]]
    want = o.sample_pixels(xy, 0, depth, seed, nee_mode=1).astype(np.float32)
    got = acc[xy[:, 1], xy[:, 0]]
    same = got.tobytes() == want.tobytes()
    rgb = g.read_rgb8()
    want_rgb = np.array([[orc.tonemap(c) for c in px] for px in want], np.uint8)
    same_rgb = np.array_equal(np.asarray(rgb)[xy[:, 1], xy[:, 0]], want_rgb)
    say(f"{W} x {H} ({W * H / 1e9:.3f} G pixels): 1 spp in {dt:.2f} s, finite sum {float(np.nansum(np.where(np.isfinite(acc[::7]), acc[::7], 0), dtype=np.float64)):.6g} over every 7th row, "
        f"6000 pixels == oracle: {same}, RGB24 of them == oracle: {same_rgb}")
    ok = ok and same and same_rgb and spp == 1.0
    del acc, rgb
try:
    g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, 46344, 46344)
    say("46344 x 46344 was NOT refused")
    ok = False
except ptcore.PtError as e:
    say("46344 x 46344 refused:", e)
    ok = ok and e.status == ptcore.PT_ERR_LIMIT
alive = False
say("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
