"""N > 1 path on the CPU: world_size-2 gloo processes shard one frame the way bench.py / the C ABI
do (tile sharding via pt_shard_map, sample sharding via global sample indices), render their
shard with the ORACLE (the checker stands in for the GPU here), combine with one reduce and
compare with the unsharded frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, SEED, TILE = 48, 40, 4, 32, 17, 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from pathtracer_amd import ptcore, scenes
    sky = scenes.make_sky()
    scene = scenes.glass_cube_scene(floor_speculaty=0.5)
    o = orc.from_scene(scene, sky)
    o.set_camera(scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET, W, H)

    # tile sharding: this rank's pixels only, all samples in order
    xy = ptcore.shard_map(W, H, TILE, rank, world)
    tiles = np.zeros((H, W, 3), np.float32)
    for s in range(SPP):
        tiles[xy[:, 1], xy[:, 0]] += o.sample_pixels(xy, s, DEPTH, SEED, nee_mode=1)
    t = torch.from_numpy(tiles.copy())
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)

    # the same shards combined the C4 way (pathtracer_amd/dist.py): every rank packs its own tiles into a
    # compact slab (numpy stands in for pt_shard_pack), ONE gather of equal slabs, rank 0 un-swizzles
    from pathtracer_amd import dist as ptdist
    n0 = ptcore.shard_slab_floats(W, H, TILE, 0, world)
    pix = ptdist.slab_pixel_index(W, H, TILE, rank, world)
    slab = np.zeros((n0 // 3, 3), np.float32)
    slab[:len(pix)][pix >= 0] = tiles.reshape(-1, 3)[pix[pix >= 0]]
    parts = ptdist.gather_slabs(torch.from_numpy(slab), dst=0)
    gathered = np.zeros((H * W, 3), np.float32)
    if rank == 0:
        for r, part in enumerate(parts):
            pr = ptdist.slab_pixel_index(W, H, TILE, r, world)
            gathered[pr[pr >= 0]] = part.numpy()[:len(pr)][pr >= 0]

    # sample sharding: global sample indices [rank * SPP/world, (rank+1) * SPP/world)
    per = SPP // world
    o.reset_accum()
    o.render(per, DEPTH, SEED, sample_begin=rank * per, nthreads=2, nee_mode=1)
    s_acc = torch.from_numpy(o.read_accum()[0].copy())
    dist.reduce(s_acc, dst=0, op=dist.ReduceOp.SUM)

    if rank == 0:
        o.reset_accum()
        o.render(SPP, DEPTH, SEED, nthreads=2, nee_mode=1)
        np.savez(out_path, tiles=t.numpy(), gathered=gathered.reshape(H, W, 3), samples=s_acc.numpy(), full=o.read_accum()[0])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_map_partitions_the_frame():
    sys.path.insert(0, ROOT)
    from pathtracer_amd import ptcore
    for (w, h, tile, n) in [(48, 40, 16, 2), (100, 60, 16, 3), (33, 17, 8, 4), (64, 64, 64, 8), (1920, 1080, 64, 8)]:
        seen = np.zeros((h, w), np.int32)
        sizes = []
        for r in range(n):
            xy = ptcore.shard_map(w, h, tile, r, n)
            assert xy[:, 0].max(initial=0) < w and xy[:, 1].max(initial=0) < h
            np.add.at(seen, (xy[:, 1], xy[:, 0]), 1)
            sizes.append(len(xy))
        assert (seen == 1).all(), "every pixel belongs to exactly one rank"
        if w * h > 10 * tile * tile * n:
            assert max(sizes) - min(sizes) <= 2 * tile * tile, "round-robin tiles keep ranks balanced"
    # nranks = 1, tile = 8 is the single-GPU walking order: 8x8 blocks, row-major inside
    xy = ptcore.shard_map(16, 8)
    assert xy[:9].tolist() == [[0, 0], [1, 0], [2, 0], [3, 0], [4, 0], [5, 0], [6, 0], [7, 0], [0, 1]]
    assert xy[64].tolist() == [8, 0]
    with pytest.raises(ptcore.PtError):
        ptcore.shard_map(16, 16, 12, 0, 1)  # tile must be a multiple of 8


def test_slab_order_is_the_device_order():
    """dist.slab_pixel_index (numpy) restates the order pt_shard_pack writes a slab in; its valid entries must be
    exactly pt_shard_map's pixels in the same order, and pt_shard_slab_floats must give its length."""
    sys.path.insert(0, ROOT)
    from pathtracer_amd import dist as ptdist
    from pathtracer_amd import ptcore
    for (w, h, tile, n) in [(48, 40, 16, 2), (100, 60, 16, 3), (33, 17, 8, 4), (3840, 2160, 64, 8), (70, 50, 64, 4)]:
        for r in range(n):
            pix = ptdist.slab_pixel_index(w, h, tile, r, n)
            assert len(pix) * 3 == ptcore.shard_slab_floats(w, h, tile, r, n)
            xy = ptcore.shard_map(w, h, tile, r, n).astype(np.int64)
            assert np.array_equal(pix[pix >= 0], xy[:, 1] * w + xy[:, 0])
        assert ptcore.shard_slab_floats(w, h, tile, 0, n) >= max(ptcore.shard_slab_floats(w, h, tile, r, n) for r in range(n))


def test_two_rank_gloo_shards_reassemble(tmp_path):
    out = str(tmp_path / "res.npz")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    # disjoint pixel sets: the reduce adds zeros, so the tile-sharded frame is bit-identical
    assert r["tiles"].tobytes() == r["full"].tobytes()
    assert r["gathered"].tobytes() == r["full"].tobytes()  # gather of compact slabs + un-swizzle: the same frame
    # sample shards are summed in a different order than the single frame loop: fp32 tolerance
    np.testing.assert_allclose(r["samples"], r["full"], rtol=1e-6, atol=1e-7)


def test_bench_refuses_more_ranks_than_gpus():
    """bench.py --gpus N without a launcher starts N ranks itself and must fail (exit code 2, nothing launched) when
    fewer than N devices are visible — here none is."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr
    # under a launcher whose WORLD_SIZE disagrees with --gpus the bench refuses as well
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
