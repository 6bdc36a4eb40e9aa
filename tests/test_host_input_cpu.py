"""Camera::handle_input (src/camera.rs:114-212) in the C++ host mirror (pt::apply_input,
pathtracer_amd/host/pathtracer.hpp), checked without a GPU through `render --input-check` against a
restatement of the reference's key arithmetic in numpy float32.  The reference holds no test or
fixture for handle_input, so the expected values are the restatement below: parity unpinned beyond
following camera.rs line by line."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDER = os.path.join(ROOT, "pathtracer_amd", "host", "render")
KEYS = ["A", "D", "W", "E", "Q", "H", "P", "S", "R", "F", "Up", "Down", "Left", "Right"]
f32 = np.float32


def reference_handle_input(keys, origin, target, direction, right, up, depth):
    """camera.rs:114-204, one statement per line of the reference (update() at :205-207 not included)."""
    origin, target = origin.copy(), target.copy()
    target = origin + direction                                    # :115
    changed = False
    if "A" in keys:                                                # :117-123
        origin = origin + f32(-0.1) * right
        target = target + f32(-0.1) * right
        changed = True
    if "D" in keys:                                                # :124-130
        origin = origin + f32(0.1) * right
        target = target + f32(0.1) * right
        changed = True
    if "W" in keys:                                                # :131-136
        origin = origin + f32(0.1) * direction
        changed = True
    if "E" in keys:                                                # :137-143
        origin = origin + f32(10.0) * direction
        target = target + f32(10.0) * direction
        changed = True
    if "Q" in keys:                                                # :144-150
        origin = origin + f32(-10.0) * direction
        target = target + f32(-10.0) * direction
        changed = True
    if "H" in keys:                                                # :151-157
        depth = 2 if depth == 512 else 512
        changed = True
    if "S" in keys:                                                # :161-166
        origin = origin + f32(-0.1) * direction
        changed = True
    if "R" in keys:                                                # :167-173
        origin = origin + f32(0.1) * up
        target = target + f32(0.1) * up
        changed = True
    if "F" in keys:                                                # :174-180
        origin = origin + f32(-0.1) * up
        target = target + f32(-0.1) * up
        changed = True
    if "Up" in keys:                                               # :181-186
        target = target + f32(-0.1) * up
        changed = True
    if "Down" in keys:                                             # :187-192
        target = target + f32(0.1) * up
        changed = True
    if "Left" in keys:                                             # :193-198
        target = target + f32(-0.1) * right
        changed = True
    if "Right" in keys:                                            # :199-204
        target = target + f32(0.1) * right
        changed = True
    return changed, origin, target, depth


def _hex(v):
    return " ".join(f"{int(x):08x}" for x in np.asarray(v, f32).view(np.uint32))


def test_key_arithmetic_matches_the_reference():
    assert os.path.exists(RENDER), "pathtracer_amd/host/render is not built"
    rng = np.random.default_rng(11)
    cases = []
    # every key alone, no key, all keys, and random subsets on random poses
    subsets = [[k] for k in KEYS] + [[], KEYS] + [list(rng.choice(KEYS, size=rng.integers(1, 6), replace=False)) for _ in range(200)]
    for keys in subsets:
        origin = rng.uniform(-3, 3, 3).astype(f32)
        target = rng.uniform(-3, 3, 3).astype(f32)
        d = target - origin
        direction = (d * (f32(1.0) / np.sqrt(np.dot(d, d), dtype=f32))).astype(f32)
        right = np.cross(np.array([0, 1, 0], f32), direction).astype(f32)
        up = np.cross(direction, right).astype(f32)
        depth = int(rng.choice([512, 2]))
        cases.append((keys, origin, target, direction, right, up, depth))
    text = "".join(f"{_hex(o)} {_hex(t)} {_hex(d)} {_hex(r)} {_hex(u)} {depth} {','.join(k) if k else '-'}\n"
                   for k, o, t, d, r, u, depth in cases)
    res = subprocess.run([RENDER, "--input-check"], input=text, capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if not ln.startswith("origin:")]  # key P prints the pose
    assert len(lines) == len(cases)
    for line, (keys, origin, target, direction, right, up, depth) in zip(lines, cases):
        want_changed, wo, wt, wdepth = reference_handle_input(set(keys), origin, target, direction, right, up, depth)
        w = line.split()
        assert int(w[0]) == int(want_changed), keys
        got = np.array([int(x, 16) for x in w[1:7]], np.uint32).view(f32)
        assert got[:3].tobytes() == wo.astype(f32).tobytes(), (keys, got[:3], wo)
        assert got[3:].tobytes() == wt.astype(f32).tobytes(), (keys, got[3:], wt)
        assert int(w[7]) == wdepth


def test_only_p_or_no_key_does_not_count_as_a_change():
    res = subprocess.run([RENDER, "--input-check"], input=f"{_hex(np.zeros(15))} 512 P\n{_hex(np.zeros(15))} 512 -\n",
                         capture_output=True, text=True, timeout=60)
    assert res.returncode == 0
    lines = [ln for ln in res.stdout.splitlines() if not ln.startswith("origin:")]
    assert [ln.split()[0] for ln in lines] == ["0", "0"]


def test_render_refuses_more_ranks_than_devices():
    """`render --gpus N` (the native multi-GPU host, pathtracer_amd/host/multi_gpu.hpp) says so at once, with a non-zero exit,
    when N exceeds the visible devices — before it loads or builds anything.  Here: whatever this machine has, plus one."""
    import os
    import subprocess
    render = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtracer_amd", "host", "render")
    assert os.path.exists(render), "pathtracer_amd/host/render is not built"
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([render, "--gpus", str(n)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "visible" in r.stderr
    r = subprocess.run([render, "--gpus", "1", "--shard", "rows"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    libs = subprocess.run(["ldd", render], capture_output=True, text=True).stdout
    assert "librccl" in libs   # the collective is linked into the host program ...
    core = os.path.join(os.path.dirname(render), "..", "libptcore.so")
    assert "librccl" not in subprocess.run(["ldd", core], capture_output=True, text=True).stdout   # ... not into the library
