"""The native host's multi-GPU frame (`render --gpus N`, pathtracer_amd/host/multi_gpu.hpp): one process, one host
thread + one context per device, RCCL (rccl.h, linked into the host program) for the one collective at readout.
It stands in for the reference's thread fan-out over one shared accumulator (src/main.rs:113-115,149-170).

What a one-GPU box can check:
  * the RCCL leg itself with one rank: ncclCommInitAll, ncclReduce in place on the library's accumulator /
    ncclGather of the tile slab, then the readout — the frame must be the unsharded frame bit for bit;
  * the N > 1 logic (sample indices per rank and frame, tile ownership, slab order, un-swizzle, sample count) with two
    and three ranks on the one device and the partial frames moved through host memory instead of RCCL
    (`--combine host`; RCCL cannot hold a device twice): tiles bit-identical, sample shards equal up to fp32
    summation order (2e-5 relative, as in test_full_size_workloads);
  * the refusals: more ranks than devices, a device named twice over RCCL.
The N > 1 rates over xGMI are the driver's to measure (SCALE_rNN.json); nothing here is a measurement.
"""
import os
import re
import subprocess

import numpy as np
import pytest

from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDER = os.path.join(ROOT, "pathtracer_amd", "host", "render")
OBJ = os.path.join(ROOT, "tests", "golden", "mesh", "mixed.obj")
W, H = 200, 136   # not a multiple of the 64-pixel tile: edge tiles carry padding pixels


@pytest.fixture(scope="module")
def sky_file(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("sky") / "sky.raw")
    np.ascontiguousarray(scenes.make_sky(), np.float32).tofile(path)
    return path


def render(tmp_path, name, sky_file, *args, scene=("--obj", OBJ), size=(W, H), expect=0):
    acc = str(tmp_path / f"{name}.acc")
    cmd = [RENDER, *scene, "--sky", sky_file, "--size", f"{size[0]}x{size[1]}", "--out", str(tmp_path / f"{name}.ppm"), "--save-accum", acc, *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == expect, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    if expect != 0:
        return None, None, r
    raw = np.fromfile(acc, np.uint8)
    hdr = raw[:8].view(np.uint32)
    assert (int(hdr[0]), int(hdr[1])) == size
    spp = float(raw[8:12].view(np.float32)[0])
    with open(str(tmp_path / f"{name}.ppm"), "rb") as f:
        ppm = f.read()
    return raw[12:].view(np.float32).reshape(size[1], size[0], 3).copy(), (spp, ppm), r


def rank_lines(stdout):
    """[(device, paths, rays)] from the per-rank report."""
    out = []
    for m in re.finditer(r"^rank (\d+) device (\d+): (\d+) paths, (\d+) rays", stdout, re.M):
        assert int(m.group(1)) == len(out)
        out.append((int(m.group(2)), int(m.group(3)), int(m.group(4))))
    return out


def test_render_is_built_with_rccl():
    assert os.path.exists(RENDER), "pathtracer_amd/host/render is not built"
    libs = subprocess.run(["ldd", RENDER], capture_output=True, text=True).stdout
    assert "librccl" in libs and "libptcore" in libs


@pytest.mark.parametrize("shard", ["tiles", "samples"])
def test_one_rank_through_rccl_equals_the_unsharded_frame(tmp_path, sky_file, shard):
    """`render --gpus 1 --shard ...`: communicator, collective on the library's own device memory, readout."""
    want, (want_spp, want_ppm), _ = render(tmp_path, "plain", sky_file, "--frames", "3", "--spp-per-frame", "2")
    got, (spp, ppm), r = render(tmp_path, "one", sky_file, "--frames", "3", "--spp-per-frame", "2", "--gpus", "1", "--shard", shard)
    assert "combine rccl" in r.stdout and f"shard {shard}" in r.stdout
    assert spp == want_spp == 6.0
    assert got.tobytes() == want.tobytes()
    assert ppm == want_ppm
    ranks = rank_lines(r.stdout)
    assert len(ranks) == 1 and ranks[0][0] == 0 and ranks[0][1] == W * H * 6 and ranks[0][2] > ranks[0][1]
    assert np.abs(want).sum() > 0


@pytest.mark.parametrize("world", [2, 3])
def test_tile_shards_of_virtual_ranks_reassemble_bit_for_bit(tmp_path, sky_file, world):
    want, (want_spp, want_ppm), _ = render(tmp_path, "plain", sky_file, "--frames", "2", "--spp-per-frame", "2")
    got, (spp, ppm), r = render(tmp_path, "tiles", sky_file, "--frames", "2", "--spp-per-frame", "2", "--gpus", str(world), "--devices",
                                ",".join("0" * world), "--shard", "tiles", "--combine", "host")
    assert "host (rehearsal)" in r.stdout
    assert spp == want_spp == 4.0
    assert got.tobytes() == want.tobytes()
    assert ppm == want_ppm
    ranks = rank_lines(r.stdout)
    assert len(ranks) == world
    assert sum(p for _, p, _ in ranks) == W * H * 4           # every pixel sample rendered exactly once
    n_tiles = ((W + 63) // 64) * ((H + 63) // 64)
    assert all(p > 0 for _, p, _ in ranks) or n_tiles < world


@pytest.mark.parametrize("world", [2, 3])
def test_sample_shards_of_virtual_ranks_add_up(tmp_path, sky_file, world):
    """world ranks x 2 frames x 2 spp = the one-GPU frame of 2 * world frames x 2 spp: the same sample indices, summed
    per rank first and then over ranks, so equal up to fp32 summation order."""
    want, (want_spp, _), _ = render(tmp_path, "plain", sky_file, "--frames", str(2 * world), "--spp-per-frame", "2")
    got, (spp, _), r = render(tmp_path, "samples", sky_file, "--frames", "2", "--spp-per-frame", "2", "--gpus", str(world), "--devices",
                              ",".join("0" * world), "--shard", "samples", "--combine", "host")
    assert spp == want_spp == 4.0 * world
    finite = np.isfinite(want)
    assert np.array_equal(finite, np.isfinite(got))
    np.testing.assert_allclose(got[finite], want[finite], rtol=2e-5, atol=1e-6)
    ranks = rank_lines(r.stdout)
    assert [p for _, p, _ in ranks] == [W * H * 4] * world   # every rank renders the whole frame


def test_spheres_scene_sharded(tmp_path, sky_file):
    """Scene<Sphere> (Scene::default_scene) through the same path: the template is instantiated for both primitive kinds."""
    want, _, _ = render(tmp_path, "plain", sky_file, "--frames", "2", scene=("--spheres",))
    got, (spp, _), _ = render(tmp_path, "tiles", sky_file, "--frames", "2", "--gpus", "2", "--devices", "0,0", "--shard", "tiles", "--combine", "host",
                              scene=("--spheres",))
    assert spp == 2.0 and got.tobytes() == want.tobytes()
    got, (spp, _), _ = render(tmp_path, "one", sky_file, "--frames", "2", "--gpus", "1", "--shard", "samples", scene=("--spheres",))
    assert spp == 2.0 and got.tobytes() == want.tobytes()


def test_refusals(tmp_path, sky_file):
    import torch
    visible = torch.cuda.device_count()
    _, _, r = render(tmp_path, "x", sky_file, "--gpus", str(visible + 1), expect=3)
    assert "visible" in r.stderr
    _, _, r = render(tmp_path, "x", sky_file, "--gpus", "2", "--devices", "0,0", expect=3)   # RCCL holds a device once
    assert "twice" in r.stderr
    _, _, r = render(tmp_path, "x", sky_file, "--gpus", "1", "--devices", str(visible), expect=3)
    assert "does not exist" in r.stderr
    _, _, r = render(tmp_path, "x", sky_file, "--gpus", "2", "--devices", "0", "--combine", "host", expect=2)


def test_cube_scene_of_the_cpp_host_is_baseline_c2(tmp_path, sky_file):
    """`render --cube` holds BASELINE config C2's scene — the very float32 vertices of scenes.glass_cube_scene():
    same accumulator and same tonemapped bytes as the Python host on that scene."""
    from pathtracer_amd import ptcore
    w, h = 192, 108
    got, (spp, ppm), _ = render(tmp_path, "cube", sky_file, "--frames", "2", "--spp-per-frame", "3", scene=("--cube",), size=(w, h))
    g = ptcore.PathTracer(0).set_scene(scenes.glass_cube_scene(), scenes.make_sky())
    try:
        g.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, w, h)
        for _ in range(2):
            g.render(3, 32, seed=2026)
        want, want_spp = g.read_accum()
        want_rgb = g.read_rgb8()
    finally:
        g.close()
    assert spp == want_spp == 6.0
    assert got.tobytes() == want.tobytes()
    assert ppm.endswith(want_rgb.tobytes())
