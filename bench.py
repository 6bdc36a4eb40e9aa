"""bench.py — Mrays/s of the path-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload C3] [--spp 256] [--shard samples|tiles]

A "step" is one pass of the hot path over one batch: `--spp` samples per pixel of the named workload
(default C3: 871 414-triangle glass statue on a diffuse floor, 1920x1080, depth 32) rendered by
pt_render (default: the persistent path kernel k_path — raygen, traversal, shading and the bounce loop
in one launch per batch — then accumulate; `--mode wavefront` selects raygen -> extend/shade rounds ->
fused tail).  Scene, BVH and sky are resident in HBM before the timed region.

N > 1: one process per GPU over RCCL (torch.distributed, backend nccl).  Started by the driver through
torch.distributed.run, or — when RANK is not in the environment — by this script itself, which then
launches N ranks before anything touches a GPU and exits with their status.
  --shard samples (default)  every rank renders its own sample indices of the whole frame; ONE
                             reduce to rank 0 at readout, inside the timed region; weak scaling.
  --shard tiles              BASELINE C4's split: rank r renders the 64x64 tiles t with t % N == r; ONE
                             gather of compact tile slabs to rank 0 + un-swizzle; strong scaling.
value = rays traced by all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline     the dominant kernel (k_path) against the roof that binds it.  SURVEY 8(d)'s algorithmic
               bytes (B_ray = 32 * node_tests + 36 * prim_tests + 28 + 12 per ray, + 100 B per bounce + 36 B per
               path; counts from an instrumented, untimed replay of the very same steps) are reported under
               `algorithmic` — they are served by L1 / L2 / Infinity Cache, so against the HBM peak they can
               exceed 1 and are NOT the headline.  The headline fraction is the highest roof fraction that can
               bind: `hbm` = fabric traffic per launch from rocprofv3 FETCH_SIZE / WRITE_SIZE passes
               (profiles/r3_pmc.json, corrected as profiles/README.md states) / live kernel time / 8 TB/s, or
               `valu` = VALU wave-instructions per launch (SQ_INSTS_VALU pass) / live kernel time / the issue
               peak (1024 SIMDs x 2.4 GHz / 2 cycles); `lane_util` says how full those instructions were.
               The headline object carries the three numbers side by side — `hbm_frac`, `valu_frac` and `lane_frac`
               (= valu_frac x lane_util: the share of the lane-issue peak doing work) — with
               `algorithmic_over_hbm_peak` (above 1: cache-served).  Kernel time is measured live with HIP events on the
               launching stream; the counters cannot be read inside an ordinary run (they need rocprofv3 passes) and come
               from the committed passes over this very command (`pmc_source`), per launch.
  cpu_baseline the CPU oracle (a C++ restatement of the reference, kind "port") in the reference's
               threading mode on this box's host cores, on a bounded sample of the same workload.
  cpu_configs  BASELINE configs[0] (C1: sphere scene 512x512, 1 spp per frame) on the CPU path, with the
               GPU's rate on the same frames beside it.
  spp_sweep    Mrays/s at 1 / 8 / 64 samples per call, calls queued with pt_render_async (and 1 spp blocking).
  opt_in       options that are never part of the metric run, each with its own rate (PT_OPT_SKIP_REPEATED_MISS: Mpaths/s).
"""
import argparse
import json
import os

import numpy as np
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_GWIPS = 1024 * 2.4 / 2  # 256 CUs x 4 SIMDs, one plain wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md: v_fma_f32 wave64
                                 # 2 cyc; tools/calib.hip measures 1051 G/s at the clock the chip holds) = 1228.8 G wave-inst/s
PMC_FILE = os.path.join(ROOT, "profiles", "r3_pmc.json")  # counters of the round-3 kernels (tools/profile_round.sh, tools/make_pmc_json.py)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "C5", "tiny"])
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step")
    ap.add_argument("--mode", default="path", choices=["path", "wavefront"], help="pt_render scheduling (same results)")
    ap.add_argument("--shard", default="samples", choices=["samples", "tiles"], help="how N > 1 ranks split a frame")
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-replay", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the spp sweep and the C1 entry")
    ap.add_argument("--blocking-steps", action="store_true", help="pt_render (wait after every step) instead of pt_render_async + one pt_wait")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path on ONE GPU (all ranks on device 0, collectives staged through the host)")
    ap.add_argument("--verify", action="store_true", help="rank 0 re-renders the combined frame alone (untimed) and compares")
    return ap.parse_args()


def launch_ranks(n):
    """--gpus N without a launcher: start N ranks ourselves (before any GPU call in this process)."""
    import socket
    import torch
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if have < n:
        print(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def workloads(scenes):
    # name: (scene factory, width, height, camera origin, camera target, description)
    return {
        "C3": (lambda: scenes.statue_scene(), 1920, 1080, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET,
               "C3 glass statue 871414+3 tris on diffuse floor, 1920x1080, depth 32"),
        "C2": (lambda: scenes.glass_cube_scene(), 1920, 1080, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET,
               "C2 glass cube 12+3 tris on diffuse floor, 1920x1080, depth 32"),
        "C4": (lambda: scenes.statue_scene(1042, 520, 1009, 7, scenes.GLASS_BLUE, 0.5), 3840, 2160, scenes.CAMERA_ORIGIN,
               scenes.CAMERA_TARGET, "C4 glass statue 1087716+3 tris on semi-specular floor, 3840x2160, depth 32"),
        "C5": (lambda: scenes.soup_scene(), 3840, 2160, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET,
               "C5 10M-triangle soup, 3840x2160, depth 32"),
        "tiny": (lambda: scenes.small_mesh_scene(), 256, 144, scenes.CAMERA_ORIGIN, scenes.CAMERA_TARGET,
                 "tiny statue, 256x144 (CI)"),
    }


def cpu_baseline(scene, sky, cam_o, cam_t, width, height, depth, seconds_hint=20.0):
    """Oracle in reference-threading mode (static row bands, src/main.rs:150-156) on a bounded
    sample of the same workload: 1 spp of the frame at full resolution, shrunk if the box is slow."""
    from oracle import orc
    threads = min(16, os.cpu_count() or 1)  # the GPU box grants 16 CPUs per GPU
    o = orc.from_scene(scene, sky)
    # calibrate on a 1/16-area frame, then pick the largest frame that fits the time hint
    o.set_camera(cam_o, cam_t, width // 4, height // 4)
    t0 = time.time()
    o.render(1, depth, seed=1, nthreads=threads, nee_mode=0)
    dt = max(time.time() - t0, 1e-3)
    scale = 1
    while scale < 4 and dt * (4 // scale) ** 2 > seconds_hint:
        scale *= 2
    w, h = width // scale, height // scale
    o.set_camera(cam_o, cam_t, w, h)
    o.reset_counters()
    t0 = time.time()
    spp = max(1, min(8, int(seconds_hint / max(dt * (4 // scale) ** 2, 1e-3) / 8)))
    o.render(spp, depth, seed=2, nthreads=threads, nee_mode=0)
    dt = time.time() - t0
    c = o.counters()
    rays = c["rays_closest"] + c["rays_any"]
    o.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "mpaths_per_s": c["paths"] / dt / 1e6,
            "sample": f"{spp} spp of the {w}x{h} frame of the same scene ({c['paths']} paths, {rays} rays incl. the "
                      f"reference's shadow rays), C++ restatement of the reference (g++ -O2), static row bands, {dt:.1f} s wall"}


def c1_entry(ptcore, scenes, sky, device):
    """BASELINE configs[0]: the reference sphere scene, 512x512, 1 spp per frame, on the CPU threaded path
    (oracle, static row bands, literal NEE) — and the same frames on the GPU (queued 1-spp calls)."""
    from oracle import orc
    threads = min(16, os.cpu_count() or 1)
    scene = scenes.default_scene()
    W = H = 512
    frames = 40
    o = orc.from_scene(scene, sky)
    o.set_camera(scenes.SPHERE_CAMERA_ORIGIN, scenes.SPHERE_CAMERA_TARGET, W, H)
    o.render(2, 32, seed=1, nthreads=threads, nee_mode=0)  # warm
    o.reset_counters()
    t0 = time.time()
    for f in range(frames):
        o.render(1, 32, seed=3, sample_begin=2 + f, nthreads=threads, nee_mode=0)  # one frame of main.rs:143-170
    dt = time.time() - t0
    c = o.counters()
    cpu_rays = c["rays_closest"] + c["rays_any"]
    o.close()
    g = ptcore.PathTracer(device).set_scene(scene, sky)
    g.set_camera(scenes.SPHERE_CAMERA_ORIGIN, scenes.SPHERE_CAMERA_TARGET, W, H)
    for _ in range(8):
        g.render_async(1, 32, seed=3)
    g.wait()
    g.reset_counters()
    gframes = 400
    t0 = time.perf_counter()
    for _ in range(gframes):
        g.render_async(1, 32, seed=3)
    g.wait()
    gdt = time.perf_counter() - t0
    gc = g.counters()
    g.close()
    return {"C1": {"workload": "reference sphere scene (src/scene.rs:98-160), 512x512, 1 spp per frame, depth 32",
                   "cpu": {"mrays_per_s": cpu_rays / dt / 1e6, "mpaths_per_s": c["paths"] / dt / 1e6, "fps": frames / dt, "cores": threads,
                           "kind": "port", "frames": frames,
                           "note": "oracle in the reference's threading mode (static row bands, shadow rays traced as the reference does)"},
                   "gpu": {"mrays_per_s": (gc["rays_closest"] + gc["rays_any"]) / gdt / 1e6, "mpaths_per_s": gc["paths"] / gdt / 1e6,
                           "fps": gframes / gdt, "frames": gframes, "launches": gc["tail_launches"],
                           "note": "1-spp calls queued with pt_render_async, one wait at the end; calls queued behind busy lanes are merged into larger launches"}}}


def spp_sweep(g, ptcore, W, H, depth, seed):
    """Mrays/s against samples per call: queued calls (pt_render_async + one pt_wait) and, for 1 spp, blocking calls
    (the reference's loop shows every frame: main.rs:143-175).  Calls queued while every lane is busy are merged into
    larger launches by the library (PT_OPT_MERGE_QUEUED_CALLS, same results); `launches` says how many there were."""
    out = {}
    for spp, calls in ((1, 64), (8, 16), (64, 8)):
        g.reset_accum()
        for _ in range(calls):   # untimed pass of the same shape: every lane's buffers exist before the timed one
            g.render_async(spp, depth, seed)
        g.wait()
        g.reset_counters()
        t0 = time.perf_counter()
        for _ in range(calls):
            g.render_async(spp, depth, seed)
        g.wait()
        dt = time.perf_counter() - t0
        c = g.counters()
        out[f"{spp}_spp_per_call_queued"] = {"mrays_per_s": (c["rays_closest"] + c["rays_any"]) / dt / 1e6, "ms_per_call": dt * 1e3 / calls, "calls": calls,
                                             "launches": c["tail_launches"]}  # fewer than calls: queued calls that had not started were merged
    # the reference's frame loop with the display decoupled: one sample submitted and the newest finished frame fetched
    # as RGB24 (over PCIe, into a reused texture buffer) per iteration, without waiting for the queued launches
    texture = np.zeros((H, W * 3), dtype=np.uint8)
    for timed in (False, True):
        g.reset_accum()
        g.reset_counters()
        frames = 200
        t0 = time.perf_counter()
        for _ in range(frames):
            g.render_async(1, depth, seed)
            _, shown = g.read_rgb8_latest(out=texture)
        t_loop = time.perf_counter() - t0
        g.wait()
        dt = time.perf_counter() - t0
    c = g.counters()
    out["1_spp_per_frame_with_rgb_readout"] = {"mrays_per_s": (c["rays_closest"] + c["rays_any"]) / dt / 1e6, "frames_shown_per_s": frames / t_loop,
                                               "frames": frames, "launches": c["tail_launches"], "samples_in_the_last_frame_shown": shown,
                                               "note": "pt_render_async(1 spp) + pt_read_rgb8_latest per iteration: PCIe readout of every shown frame included"}
    g.reset_accum()
    g.render(1, depth, seed)
    g.reset_counters()
    t0 = time.perf_counter()
    for _ in range(16):
        g.render(1, depth, seed)
    dt = time.perf_counter() - t0
    c = g.counters()
    out["1_spp_per_call_blocking"] = {"mrays_per_s": (c["rays_closest"] + c["rays_any"]) / dt / 1e6, "ms_per_call": dt * 1e3 / 16, "calls": 16}
    return out


def opt_in_entry(g, ptcore, spp, depth, seed):
    """Options that are never part of the metric run, each with its own line.  skip_repeated_miss: PT_OPT_SKIP_REPEATED_MISS — the same frame bit
    for bit, but a look-ahead trace that missed is not traced a second time (quirk Q3's re-trace), so fewer traversals are EXECUTED: compare
    Mpaths/s with the headline's, not Mrays/s (the ray count is no longer the reference's)."""
    out = {}
    g.set_option(ptcore.OPT_SKIP_REPEATED_MISS, 1)
    try:
        g.reset_accum()
        g.render(spp, depth, seed)
        g.reset_counters()
        t0 = time.perf_counter()
        for _ in range(2):
            g.render(spp, depth, seed)
        dt = time.perf_counter() - t0
        c = g.counters()
        out["skip_repeated_miss"] = {"mpaths_per_s": c["paths"] / dt / 1e6, "mrays_executed_per_s": (c["rays_closest"] + c["rays_any"]) / dt / 1e6,
                                     "rays_executed_per_path": (c["rays_closest"] + c["rays_any"]) / max(c["paths"], 1), "ms_per_step": dt * 1e3 / 2, "spp_per_step": spp,
                                     "note": "same sums as the metric run; NOT the metric: executed traversals differ from the reference's"}
    finally:
        g.set_option(ptcore.OPT_SKIP_REPEATED_MISS, 0)
    return out


def main():
    args = parse_args()
    # Hardware queues of this process (read by the HIP runtime when it starts; its default is 4).  Queued render calls
    # run on up to eight streams, and streams that share a hardware queue run their kernels one after the other: with 8
    # queues the 1-spp-per-call rate of the spp sweep rises by 30 % (1.25 -> 1.6 Grays/s); the headline (one launch of
    # 256 spp per step) does not depend on it.  An existing setting in the environment wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch  # imported before libptcore so both share one HIP runtime
    import torch.distributed as dist
    from pathtracer_amd import dist as ptdist
    from pathtracer_amd import ptcore, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: same code path)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    one_gpu = args.backend == "gloo"   # rehearsal: every rank on device 0
    if use_dist and one_gpu:
        dist.init_process_group("gloo")
    elif use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if use_dist and not one_gpu else 0
    torch.cuda.set_device(device)

    make, W, H, cam_o, cam_t, desc = workloads(scenes)[args.workload]
    scene = make()
    sky = scenes.make_sky()
    t_build = time.perf_counter()
    g = ptcore.PathTracer(device).set_scene(scene, sky)
    t_build = time.perf_counter() - t_build
    built_on_device, tree_ms = g.build_info()
    g.set_camera(cam_o, cam_t, W, H)
    g.set_option(ptcore.OPT_RENDER_MODE, 1 if args.mode == "path" else 0)
    seed = 2026
    spp = args.spp
    tiles = args.shard == "tiles" and use_dist   # also with one rank under the launcher: pack, gather, un-swizzle all run

    def shard_for(step):
        if tiles:   # this rank's tiles, every sample of the step
            return ptdist.tile_shard(rank, world, ptdist.TILE, step * spp)
        # global sample indices: step-major, then rank (weak scaling: every rank adds its own samples)
        return ptcore.Shard(ptcore.SHARD_ALL, 0, 1, 0, ptdist.sample_begin(step, rank, world, spp))

    def run_steps(first, count, flags):
        # The steps are submitted the way the library is meant to be driven (pt_render_async, one pt_wait at the end of the
        # region): step k + 1 starts on the device while the last long paths of step k finish, as for any queued calls
        # (DESIGN section 6).  Every step is still its own launch of `spp` samples (531 M paths of C3 are above the merge cap),
        # all K are complete when the timed region ends.  --blocking-steps waits after every step instead.
        for s in range(first, first + count):
            if args.blocking_steps:
                g.render(spp, args.depth, seed, shard_for(s), flags)
            else:
                g.render_async(spp, args.depth, seed, shard_for(s), flags)
        g.wait()

    def combine():
        if not use_dist:
            return
        if tiles:
            ptdist.gather_tiles(g)
        else:
            ptdist.reduce_samples(g, spp_of_other_ranks=float((world - 1) * spp * args.steps))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- instrumented replay of the exact timed steps (untimed): algorithmic node / primitive tests
    alg = None
    if not args.no_roofline_replay:
        g.reset_counters()
        run_steps(args.warmup, args.steps, ptcore.RENDER_COUNT)
        c = g.counters()
        rays_replay = c["rays_closest"] + c["rays_any"]
        if args.mode == "path":
            # k_path does everything: every ray's traversal bytes plus the shading bytes of SURVEY 8d.
            # bounces (shaded hits) >= rays - paths: every ray after a path's first needs a hit before it.
            k_rays, k_nodes, k_prims = rays_replay, c["node_tests"], c["prim_tests"]
            shade_bytes = 100 * max(rays_replay - c["paths"], 0) + 36 * c["paths"]
        else:
            # the round kernel (k_extend) alone: rays finished by the fused tail kernel are not its work
            k_rays = rays_replay - c["tail_rays"]
            k_nodes = c["node_tests"] - c["tail_node_tests"]
            k_prims = c["prim_tests"] - c["tail_prim_tests"]
            shade_bytes = 0
        alg = {"bytes": 32 * k_nodes + 36 * k_prims + (28 + 12) * k_rays + shade_bytes, "rays": k_rays,
               "node_tests_per_ray": k_nodes / max(k_rays, 1), "prim_tests_per_ray": k_prims / max(k_rays, 1),
               "tail_share": c["tail_rays"] / max(rays_replay, 1), "tail_launches": c["tail_launches"]}

    # ---- warmup + timed region
    g.reset_accum()
    run_steps(0, args.warmup, 0)
    if use_dist:
        # warm the collective too (communicator channels and buffers for this message size are set up on first use)
        if tiles:
            ptdist.gather_tiles(g)
        elif not one_gpu:
            dist.reduce(torch.zeros(H, W, 3, dtype=torch.float32, device=f"cuda:{device}"), dst=0, op=dist.ReduceOp.SUM)
    g.reset_accum()   # the combined frame holds exactly the timed steps' samples (the warm-up's are dropped on every rank)
    g.reset_counters()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, ptcore.RENDER_TIMED)
    combine()  # readout: one collective per frame, not per bounce
    barrier()
    dt = time.perf_counter() - t0
    c = g.counters()
    rays = c["rays_closest"] + c["rays_any"]
    paths = c["paths"]

    stat_dev = "cpu" if one_gpu else f"cuda:{device}"
    tt = torch.tensor([dt], dtype=torch.float64, device=stat_dev)
    work = torch.tensor([rays, paths], dtype=torch.float64, device=stat_dev)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    rays_all, paths_all = float(work[0].item()), float(work[1].item())

    verified = verify_info = None
    if args.verify and rank == 0:
        # the combined frame against the same samples rendered by this rank alone
        got, got_spp = g.read_accum()
        g.reset_accum()
        if tiles:   # every rank rendered all samples of its tiles: the frame holds steps x spp samples, bit-identical
            for s_ in range(args.warmup, args.warmup + args.steps):
                g.render(spp, args.depth, seed, ptcore.Shard(ptcore.SHARD_ALL, 0, 1, 0, s_ * spp))
            want = g.read_accum()[0]
            verified = bool(got.tobytes() == want.tobytes() and got_spp == args.steps * spp)
        else:       # rank r of step s rendered samples [(s * world + r) * spp, ...): summed in another order -> fp32 tolerance
            for s_ in range(args.warmup, args.warmup + args.steps):
                g.render(spp * world, args.depth, seed, ptcore.Shard(ptcore.SHARD_ALL, 0, 1, 0, s_ * world * spp))
            want = g.read_accum()[0]
            finite = np.isfinite(want)
            same_kind = bool(np.array_equal(finite, np.isfinite(got)) and np.array_equal(np.isnan(want), np.isnan(got)))
            err = np.abs(got[finite] - want[finite]) / np.maximum(np.abs(want[finite]), 1e-3) if same_kind else np.array([np.inf])
            verify_info = {"max_rel_err": float(err.max()) if err.size else 0.0, "non_finite_values": int((~finite).sum()), "spp": got_spp}
            verified = bool(same_kind and verify_info["max_rel_err"] <= 2e-5 and got_spp == args.steps * spp * world)
    if rank == 0:
        if world == 1 and not tiles:
            par = "1 GPU"
        elif tiles:
            par = f"tile-sharded x{world} (64x64 tiles round-robin), one gather of tile slabs at readout"
        else:
            par = f"sample-sharded x{world}, one reduce at readout"
        out = {
            "metric": "Mrays/sec", "value": rays_all / dt_max / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max * 1e3 / max(args.steps, 1),
            "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "spp_per_step": spp, "max_depth": args.depth, "parallelism": par,
                       "steps_submitted": "blocking (pt_render)" if args.blocking_steps else "queued (pt_render_async per step, one pt_wait)",
                       "paths_per_step_per_gpu": (W * H * spp) // (world if tiles else 1)},
            "mpaths_per_s": paths_all / dt_max / 1e6,
            "rays_per_path": rays_all / max(paths_all, 1),
            "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
            "scene_setup": {"tree_built_on": "gpu" if built_on_device else "host", "tree_ms": tree_ms, "create_upload_build_ms": t_build * 1e3,
                            "note": "outside the timed region: pt_create (first HIP call of the process), objects + sky handed over and uploaded, the reference's BVH built, flattened, primitives put in leaf order"},
        }
        if one_gpu:
            out["rehearsal"] = f"{world} ranks on ONE GPU over gloo (collectives staged through the host): a functional check, not a scaling number"
        if verified is not None:
            out["verified"] = verified
            if verify_info:
                out["verify"] = verify_info
        k_ms = c["tail_ms"] if args.mode == "path" else c["extend_ms"]          # time with >= 1 launch running
        k_ms_sum = c["tail_ms_sum"] if args.mode == "path" else c["extend_ms_sum"]  # sum of launch durations
        launches = max(c["tail_launches"] if args.mode == "path" else c["extend_launches"], 1)
        if alg is not None and k_ms > 0:
            kernel = "k_path" if args.mode == "path" else "k_extend"
            alg_gbs = alg["bytes"] / (k_ms * 1e-3) / 1e9
            roof = {
                "kernel": kernel, "launches": launches, "avg_launch_ms": k_ms_sum / launches, "kernel_busy_ms": k_ms,
                "kernel_rays": alg["rays"], "kernel_mrays_per_s": alg["rays"] / (k_ms * 1e-3) / 1e6,
                "algorithmic": {"bytes_per_launch": alg["bytes"] / launches, "bytes_per_ray": alg["bytes"] / max(alg["rays"], 1),
                                "node_tests_per_ray": alg["node_tests_per_ray"], "prim_tests_per_ray": alg["prim_tests_per_ray"],
                                "gb_per_s": alg_gbs, "over_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                                "note": "SURVEY 8(d) bytes the algorithm asks for; served mostly by L1 / L2 / Infinity Cache, so this is not an HBM fraction"},
            }
            pmc = cal = None
            if os.path.exists(PMC_FILE):
                with open(PMC_FILE) as f:
                    table = json.load(f)
                pmc, cal = table.get(f"{args.workload}:{args.mode}:{spp}:{args.depth}"), table.get("_calibration")
            ms_per_launch = k_ms / launches
            if pmc and world == 1:
                hbm_gbs = pmc["hbm_bytes_per_launch"] / (ms_per_launch * 1e-3) / 1e9
                valu_gwips = pmc["valu_wave_insts_per_launch"] / (ms_per_launch * 1e-3) / 1e9
                hbm_frac, valu_frac = hbm_gbs / HBM_PEAK_GBS, valu_gwips / VALU_PEAK_GWIPS
                roof["hbm"] = {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                               "fetch_bytes_per_launch": pmc["fetch_bytes_per_launch"], "write_bytes_per_launch": pmc["write_bytes_per_launch"],
                               "note": pmc.get("hbm_note", "")}
                roof["valu"] = {"achieved": valu_gwips, "peak": VALU_PEAK_GWIPS, "unit": "G wave-inst/s", "frac": valu_frac,
                                "lane_util": pmc["lane_util"], "lane_frac": valu_frac * pmc["lane_util"],
                                "valu_wave_insts_per_ray": pmc["valu_wave_insts_per_launch"] * launches / max(alg["rays"], 1),
                                "note": "wave64 VALU instructions issued / (1024 SIMDs x 2.4 GHz / 2 cycles: the rate of plain fp32 instructions; divides, v_mul_hi and packed ops issue slower, so the pipe is busier than this says); lane_util = active lanes per instruction / 64; "
                                        "lane_frac = fraction of the lane-issue peak doing work"}
                if cal:
                    req = pmc["ea_read_requests_per_launch"] / (ms_per_launch * 1e-3) / 1e9
                    peak_ic = cal["fabric_gather_peak_g_requests_per_s_infinity_cache"]
                    roof["fabric_gather"] = {"achieved": req, "peak": peak_ic, "unit": "G 64-B read requests/s", "frac": req / peak_ic,
                                             "peak_hbm_resident": cal["fabric_gather_peak_g_requests_per_s_hbm"], "frac_of_hbm_resident_peak": req / cal["fabric_gather_peak_g_requests_per_s_hbm"],
                                             "l2_hit_rate": pmc["l2_hit_rate"],
                                             "note": "L2-miss read requests (TCC_EA0_RDREQ) per second against the rate a pure random gather of 64-B records reaches on "
                                                     "this GPU when every wave follows a dependent chain with 4 loads in flight (tools/coop.hip, profiles/r3_calib.json: 64-MiB table resident in the "
                                                     "Infinity Cache (frac) / 1-GiB table in HBM (frac_of_hbm_resident_peak: the one that applies to C5, whose working set does not fit the cache)). "
                                                     "A MEASURED peak, not a spec figure, so it is reported beside the headline, not as it"}
                    vpk = cal.get("vmem_peak_g_wave_loads_per_s_l1_resident_full_waves")
                    if vpk and "vmem_rd_insts_per_launch" in pmc:
                        vm = pmc["vmem_rd_insts_per_launch"] / (ms_per_launch * 1e-3) / 1e9
                        roof["vector_memory_pipe"] = {"achieved": vm, "peak": vpk, "unit": "G wave-level load instructions/s", "frac": vm / vpk,
                                                      "peak_with_16_lanes_active": cal.get("vmem_peak_g_wave_loads_per_s_l1_resident_16_lanes"),
                                                      "peak_from_l2_full_waves": cal.get("vmem_peak_g_wave_loads_per_s_l2_resident_full_waves"),
                                                      "note": "SQ_INSTS_VMEM_RD per second against the rate at which the 256 CUs take buffer_load_dwordx4 instructions of full waves whose lines "
                                                              "are all L1 hits (tools/coop.hip: 20.5 cycles per instruction and CU; 16 cycles is the floor, whatever the width of the load and "
                                                              "however few lanes are active; every lane that misses L1 adds 0.36 cycles).  The busiest unit of the chip under k_path, and the "
                                                              "reason an extra load per step costs 6 - 19 %.  MEASURED peak, beside the headline"}
                roof["traffic"] = pmc["hbm_bytes_per_launch"]
                roof["pmc_source"] = pmc.get("source", "profiles/r3_pmc.json")
                top = "valu" if valu_frac >= hbm_frac else "hbm"
                roof.update({"bound": top, "achieved": roof[top]["achieved"], "peak": roof[top]["peak"], "unit": roof[top]["unit"],
                             "frac": roof[top]["frac"], "hbm_frac": hbm_frac, "valu_frac": valu_frac, "lane_frac": valu_frac * pmc["lane_util"],
                             "algorithmic_over_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                             "note": "frac = the larger of hbm_frac (fabric bytes / 8 TB/s) and valu_frac (VALU wave-instructions / issue peak); lane_frac = valu_frac x "
                                     "active lanes per instruction; algorithmic_over_hbm_peak > 1 because most of the bytes the algorithm asks for are served by caches"})
            else:
                roof.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                             "note": "no PMC passes committed for this configuration (profiles/r3_pmc.json): only the algorithmic figure is live"})
            out["roofline"] = roof
            if args.mode != "path":
                out["tail"] = {"share_of_rays": alg["tail_share"], "launches": c["tail_launches"], "ms_busy": c["tail_ms"]}
        if world == 1 and not args.no_extras:
            out["spp_sweep"] = spp_sweep(g, ptcore, W, H, args.depth, seed)
            out["opt_in"] = opt_in_entry(g, ptcore, spp, args.depth, seed)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, sky, cam_o, cam_t, W, H, args.depth)
            if not args.no_extras:
                out["cpu_configs"] = c1_entry(ptcore, scenes, sky, device)
        print(json.dumps(out), flush=True)
    g.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
