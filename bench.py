"""bench.py — Mrays/s of the path-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: `--spp` samples per pixel of the named
workload (default C3: 871 414-triangle glass statue on a diffuse floor, 1920x1080, depth 32)
rendered by pt_render (default: the persistent path kernel k_path — raygen, traversal, shading
and the bounce loop in one launch per batch — then accumulate; `--mode wavefront` selects the
raygen -> extend/shade rounds -> fused tail variant).  Scene, BVH and sky are
resident in HBM before the timed region.  For N > 1 (one process per GPU, torch.distributed over
RCCL) every rank renders its own sample indices of the whole frame (sample sharding, weak
scaling: per-GPU work is fixed) and the partial sums are combined by ONE reduce to rank 0 at
readout, inside the timed region.  value = rays traced by all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline     the dominant kernel (k_path; k_extend in wavefront mode) against the HBM roof:
               algorithmic bytes B_ray = 32 * node_tests + 36 * prim_tests + 28 + 12 per ray plus,
               for k_path, B_shade = 100 B per bounce + 36 B per path (the counts
               come from an instrumented, untimed replay of the very same steps) divided by the
               HIP-event time of that kernel's launches in the timed region.  `traffic` is the
               L2-miss traffic per launch from rocprofv3 PMC passes (profiles/r1_traffic.json).
  cpu_baseline the CPU oracle (a C++ restatement of the reference, kind "port") in the
               reference's threading mode on this box's host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before libptcore so both share one HIP runtime)
import torch.distributed as dist  # noqa: E402

from pathtracer_amd import ptcore, scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s

WORKLOADS = {
    # name: (scene factory, width, height, camera origin, camera target, description)
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
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "mpaths_per_s": c["paths"] / dt / 1e6,
            "sample": f"{spp} spp of the {w}x{h} frame of the same scene ({c['paths']} paths, {rays} rays incl. the "
                      f"reference's shadow rays), C++ restatement of the reference (g++ -O2), static row bands, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", choices=list(WORKLOADS))
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step")
    ap.add_argument("--mode", default="path", choices=["path", "wavefront"], help="pt_render scheduling (same results)")
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-replay", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: same code path)
    if use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if use_dist else 0
    torch.cuda.set_device(device)

    make, W, H, cam_o, cam_t, desc = WORKLOADS[args.workload]
    scene = make()
    sky = scenes.make_sky()
    g = ptcore.PathTracer(device).set_scene(scene, sky)
    g.set_camera(cam_o, cam_t, W, H)
    g.set_option(ptcore.OPT_RENDER_MODE, 1 if args.mode == "path" else 0)
    seed = 2026
    spp = args.spp

    def sample_begin(step):
        # global sample indices: step-major, then rank (weak scaling: every rank adds its own samples)
        return (step * world + rank) * spp

    def run_steps(first, count, flags):
        for s in range(first, first + count):
            g.render(spp, args.depth, seed, ptcore.Shard(ptcore.SHARD_ALL, 0, 1, 0, sample_begin(s)), flags)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- instrumented replay of the exact timed steps (untimed): algorithmic node / primitive tests
    bytes_total = None
    if not args.no_roofline_replay:
        g.reset_counters()
        run_steps(args.warmup, args.steps, ptcore.RENDER_COUNT)
        c = g.counters()
        rays_replay = c["rays_closest"] + c["rays_any"]
        if args.mode == "path":
            # k_path does everything: every ray's traversal bytes plus the shading bytes of SURVEY 8d.
            # bounces (shaded hits) >= rays - paths: every ray after a path's first needs a hit before it.
            k_rays, k_nodes, k_prims = rays_replay, c["node_tests"], c["prim_tests"]
            bounces = max(rays_replay - c["paths"], 0)
            shade_bytes = 100 * bounces + 36 * c["paths"]
        else:
            # the round kernel (k_extend) alone: rays finished by the fused tail kernel are not its work
            k_rays = rays_replay - c["tail_rays"]
            k_nodes = c["node_tests"] - c["tail_node_tests"]
            k_prims = c["prim_tests"] - c["tail_prim_tests"]
            shade_bytes = 0
        bytes_total = 32 * k_nodes + 36 * k_prims + (28 + 12) * k_rays + shade_bytes
        nodes_per_ray = k_nodes / max(k_rays, 1)
        prims_per_ray = k_prims / max(k_rays, 1)
        tail_share = c["tail_rays"] / max(rays_replay, 1)

    # ---- warmup + timed region
    g.reset_accum()
    run_steps(0, args.warmup, 0)
    if use_dist:
        # warm the collective too (communicator channels and buffers for this message size are set up on first use)
        dist.reduce(torch.zeros(H, W, 3, dtype=torch.float32, device=f"cuda:{device}"), dst=0, op=dist.ReduceOp.SUM)
    g.reset_counters()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, ptcore.RENDER_TIMED)
    if use_dist:
        acc = torch.as_tensor(ptcore.AccumView(g), device=f"cuda:{device}")
        dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)  # readout: one collective per frame, not per bounce
    barrier()
    dt = time.perf_counter() - t0
    c = g.counters()
    rays = c["rays_closest"] + c["rays_any"]
    paths = c["paths"]

    tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{device}")
    work = torch.tensor([rays, paths], dtype=torch.float64, device=f"cuda:{device}")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    rays_all, paths_all = float(work[0].item()), float(work[1].item())

    if rank == 0:
        out = {
            "metric": "Mrays/sec", "value": rays_all / dt_max / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max * 1e3 / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "spp_per_step": spp, "max_depth": args.depth,
                       "parallelism": "1 GPU" if world == 1 else f"sample-sharded x{world}, one reduce at readout",
                       "paths_per_step_per_gpu": W * H * spp},
            "mpaths_per_s": paths_all / dt_max / 1e6,
            "rays_per_path": rays_all / max(paths_all, 1),
        }
        k_ms = c["tail_ms"] if args.mode == "path" else c["extend_ms"]
        if bytes_total is not None and k_ms > 0:
            launches = max(c["tail_launches"] if args.mode == "path" else c["extend_launches"], 1)
            achieved = bytes_total / (k_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                key = f"{args.workload}:{args.mode}:{spp}:{args.depth}"
                if key in tj:
                    traffic = tj[key]["bytes_per_launch"]
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": "k_path" if args.mode == "path" else "k_extend", "launches": launches,
                "avg_launch_ms": k_ms / launches, "bytes_per_launch": bytes_total / launches,
                "bytes_per_ray": bytes_total / max(k_rays, 1), "node_tests_per_ray": nodes_per_ray,
                "prim_tests_per_ray": prims_per_ray, "kernel_rays": k_rays,
                "kernel_mrays_per_s": k_rays / (k_ms * 1e-3) / 1e6,
                "note": "HBM roof as the contract asks; the kernel's real limiters are VALU issue and memory latency "
                        "(divergent 64-B node fetches served by L2 / Infinity Cache), see DESIGN.md and profiles/",
            }
            if args.mode != "path":
                out["tail"] = {"share_of_rays": tail_share, "launches": c["tail_launches"], "ms_total": c["tail_ms"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, sky, cam_o, cam_t, W, H, args.depth)
        print(json.dumps(out), flush=True)
    g.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
