"""SURVEY 8f row N2 — load_mesh (src/mesh.rs:9-73): the C++ host loader (pathtracer_amd/host/mesh.hpp,
through the render CLI's --dump-mesh) and the Python loader (pathtracer_amd/mesh.py) against the
restatement in oracle/obj_ref.py on hand-written OBJ/MTL fixtures.  Parity unpinned: the tobj crate
the reference parses with is not in /root/reference and the reference has no loader test or OBJ file;
the expected values asserted by hand below follow mesh.rs's material mapping."""
import os
import subprocess

import numpy as np
import pytest

from pathtracer_amd import mesh, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "mesh")
RENDER = os.path.join(os.path.dirname(HERE), "pathtracer_amd", "host", "render")
BACKUP = scenes.GLASS_BLUE  # Scene::scene's backup material, src/scene.rs:36-40


def _expected(path):
    from oracle import obj_ref
    tris = obj_ref.load_mesh(path, BACKUP)
    out = np.zeros((len(tris), 24), np.float32)
    for i, (p, n, (kind, a, b, rgb)) in enumerate(tris):
        out[i, 0:9] = np.asarray(p, np.float32).ravel()
        out[i, 9:18] = np.asarray(n, np.float32).ravel()
        out[i, 18:24] = [kind, a, b, *rgb]
    return out


def _python_loader(path):
    pos, nrm, mat_id, rows = mesh.load_mesh(path, BACKUP)
    mats = np.array([[k, a, b, *rgb] for (k, a, b, rgb) in rows], np.float32)
    return np.concatenate([pos, nrm, mats[mat_id]], axis=1).astype(np.float32)


def _need_render():
    """The CLI is a build product (pathtracer_amd/csrc/Makefile, __graft_entry__.build()); build it if it is missing."""
    if not os.path.exists(RENDER):
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(HERE), "pathtracer_amd", "csrc"), "../host/render"],
                       capture_output=True, text=True)
    if not os.path.exists(RENDER):
        pytest.fail("pathtracer_amd/host/render is not built: run __graft_entry__.build()")


def _cpp_loader(path, tmp_path):
    _need_render()
    out = str(tmp_path / "mesh.bin")
    r = subprocess.run([RENDER, "--obj", path, "--dump-mesh", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return np.fromfile(out, np.float32).reshape(-1, 24)


def test_mixed_fixture(tmp_path):
    path = os.path.join(FIX, "mixed.obj")
    want = _expected(path)
    assert want.shape == (9, 24)
    # by hand, mesh.rs:39-66: no usemtl yet -> backup; d = 0.5 -> Dielectric(1.0, 1.3, Ka); d = 1 -> Diffuse(Ns, Kd)
    f = np.float32
    np.testing.assert_array_equal(want[0, 18:], np.array([1, 1.0, 1.5, 0.01, 0.01, 0.9], f))
    np.testing.assert_array_equal(want[1, 18:], np.array([1, 1.0, f(1.3), f(0.9), f(0.8), f(0.7)]))
    np.testing.assert_array_equal(want[2, 18:], want[1, 18:])
    np.testing.assert_array_equal(want[3, 18:], np.array([0, 0.0, 0.0, f(0.64), f(0.48), f(0.32)]))
    np.testing.assert_array_equal(want[6, 18:], np.array([0, f(0.8), 0.0, f(0.2), f(0.2), 1.0]))
    np.testing.assert_array_equal(want[7, 18:], want[0, 18:])  # unknown material name -> backup
    np.testing.assert_array_equal(want[8, 18:], np.array([0, 0.0, 0.0, 0.25, 0.5, 0.75], f))
    # quad 1 2 3 4 -> (1,2,3), (1,3,4); pentagon with negative indices fanned from its first corner
    np.testing.assert_array_equal(want[2, 0:9], np.array([0, 0, 0, 1, 1, 0, 0, 1, 0], f))
    np.testing.assert_array_equal(want[5, 0:9], np.array([0, 0, 0, 0, 1, 0, 0.5, 1.5, 0.25], f))
    np.testing.assert_array_equal(want[5, 15:18], np.array([0, f(0.6), f(0.8)]))
    assert _python_loader(path).tobytes() == want.tobytes()
    assert _cpp_loader(path, tmp_path).tobytes() == want.tobytes()


def test_generated_mesh_round_trip(tmp_path):
    """A 2 000-triangle statue written out as OBJ (v//vn corners, 9 significant digits) loads back
    bit-identically through both loaders, and feeds Scene::scene's assembly (wrap_mesh)."""
    pos, nrm, mat_id, mats = scenes.statue_scene(40, 25, 8, 3)
    n = len(pos) - 3  # without the light and the two floor triangles
    path = str(tmp_path / "statue.obj")
    with open(path, "w") as f:
        for t in range(n):
            for k in range(3):
                f.write("v %.9g %.9g %.9g\n" % tuple(pos[t, 3 * k:3 * k + 3]))
                f.write("vn %.9g %.9g %.9g\n" % tuple(nrm[t, 3 * k:3 * k + 3]))
            f.write("f %d//%d %d//%d %d//%d\n" % (3 * t + 1, 3 * t + 1, 3 * t + 2, 3 * t + 2, 3 * t + 3, 3 * t + 3))
    p2, n2, m2, rows = mesh.load_mesh(path, BACKUP)
    assert p2.tobytes() == pos[:n].tobytes() and n2.tobytes() == nrm[:n].tobytes()
    assert rows == [BACKUP] and not m2.any()
    got = _cpp_loader(path, tmp_path)
    assert got[:, :9].tobytes() == pos[:n].tobytes() and got[:, 9:18].tobytes() == nrm[:n].tobytes()
    wp, wn, wm, wmats = scenes.wrap_mesh(p2, n2, m2, rows)
    assert len(wp) == n + 3 and wmats["kind"].tolist() == [1, 2, 0]


def test_missing_normals_and_files_are_errors(tmp_path):
    _need_render()
    bad = os.path.join(FIX, "no_normals.obj")
    with pytest.raises(ValueError):
        mesh.load_mesh(bad, BACKUP)
    r = subprocess.run([RENDER, "--obj", bad, "--dump-mesh", str(tmp_path / "x.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "normal" in r.stderr
    r = subprocess.run([RENDER, "--obj", str(tmp_path / "absent.obj"), "--dump-mesh", str(tmp_path / "x.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    with pytest.raises(OSError):
        mesh.load_mesh(str(tmp_path / "absent.obj"), BACKUP)
