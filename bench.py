#!/usr/bin/env python3
"""bench.py — Mrays/s on BASELINE.json's headline config: Stanford Dragon 1080p 64 spp
(config 3: ~870k-triangle model in a box room, 3 spot lights, soft shadows reps 16, aperture
0.05, depth 16).  No STL ships with the image, so the Dragon is the seeded procedural stand-in
of raytracer-go_amd/scenes.py (871,200 triangles) unless RTGO_DRAGON_STL names a file.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one whole frame.  At N > 1 the frame's samples are split over the ranks
(RT_PART_SAMPLES), each rank renders into its own fp32 accumulation buffer in HBM, one RCCL
reduce(sum) lands the frame on rank 0, which divides and quantises: total work is fixed, so
"scaling" is "strong".  value = rays traced by all ranks / wall time of the timed region (scene
already resident in HBM; the timed region ends with the RGBA8 frame in device memory on rank 0).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=480, help="rows of the 1-spp slice the CPU baseline renders")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rt = importlib.import_module("raytracer-go_amd")
    abi = rt._abi
    backend = importlib.import_module("raytracer-go_amd.backend")
    scenes = importlib.import_module("raytracer-go_amd.scenes")
    rtdist = importlib.import_module("raytracer-go_amd.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsal only) ranks share devices and the
    # reduce goes through gloo, since RCCL refuses two ranks on one device
    n_dev = torch.cuda.device_count()
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend_name = os.environ.get("RTGO_DIST_BACKEND", "nccl" if n_dev >= world else "gloo")
        if backend_name == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend_name, rank=rank, world_size=world)

    spec, scene = scenes.config3(args.width, args.height, args.spp, 16, asset_dir=GOLD)
    dscene = backend.DeviceScene(scene, device=local_rank)
    info = dscene.info()
    W, H = spec.Width, spec.Height
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    totals = {"rays": 0, "primary": 0, "continuation": 0, "shadow": 0, "shadow_ref_eq": 0,
              "ms_extend": 0.0, "ms_shadow": 0.0, "ms_shade": 0.0}

    def step(record):
        accum.zero_()

        def share(part):
            st = dscene.render_accum_device(spec, accum.data_ptr(), stream=stream, part=part)
            if record:
                totals["rays"] += st.rays
                totals["primary"] += st.rays_primary
                totals["continuation"] += st.rays_continuation
                totals["shadow"] += st.rays_shadow
                totals["shadow_ref_eq"] += st.rays_shadow_reference_equivalent
                totals["ms_extend"] += st.ms_extend
                totals["ms_shadow"] += st.ms_shadow
                totals["ms_shade"] += st.ms_shade

        rtdist.render_frame(share, accum, spec, abi.RT_PART_SAMPLES, rank, world,
                            reduce_fn=(lambda a: dist.reduce(a, dst=0, op=dist.ReduceOp.SUM)) if world > 1 else None)
        if rank == 0:
            backend.resolve_device(accum.data_ptr(), W, H, spec.AntiAliasingFactor, rgba.data_ptr(), stream=stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0

    agg = torch.tensor([elapsed] + [float(totals[k]) for k in ("rays", "primary", "continuation", "shadow", "shadow_ref_eq",
                                                               "ms_extend", "ms_shadow", "ms_shade")],
                       dtype=torch.float64, device=dev)
    if world > 1:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        agg[0] = tmax[0]
    agg = agg.cpu().numpy()
    elapsed = float(agg[0])
    rays, n_prim, n_cont, n_shadow, n_shadow_ref, ms_extend, ms_shadow, ms_shade = [float(x) for x in agg[1:]]

    if rank == 0:
        with open(os.path.join(GOLD, "bray.json")) as f:
            bray_all = json.load(f)
            bray = bray_all["config3"]
        # dominant kernel: k_shadow (any-hit traversal of the soft-shadow rays)
        shadow_bytes = n_shadow * bray["shadow"]["bytes_per_ray"]
        shadow_s = ms_shadow / 1e3 / world  # mean over ranks of the kernel's summed launch durations
        achieved = shadow_bytes / world / shadow_s / 1e9 if shadow_s > 0 else 0.0
        extend_bytes = n_prim * bray["primary"]["bytes_per_ray"] + n_cont * bray["continuation"]["bytes_per_ray"]
        traffic = None  # HBM-side bytes per k_shadow launch, from the committed rocprofv3 PMC passes of this same workload
        issue = None    # the same passes' VALU figures: what actually bounds the kernel
        tpath = os.path.join(ROOT, "profiles", "r01", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if world == 1 and tj["workload"] == {"width": W, "height": H, "spp": spec.AntiAliasingFactor, "config": "config3"}:  # per-launch figures of the 1-GPU run
                traffic = tj["hbm_bytes_per_launch"]
                isu = tj.get("issue")
                if isu and shadow_s > 0 and world == 1:
                    issue = {"valu_insts_per_ray": isu["valu_insts_per_frame"] / (n_shadow / args.steps),
                             "valu_active_lane_frac": isu["valu_active_lane_frac"],
                             "valu_busy_frac_profiled": isu["valu_busy_seconds_per_frame"] / (shadow_s / args.steps),
                             "source": "profiles/r01/m_final_pmc.txt"}
        out = {
            "metric": "Mrays/s (whole node), Stanford Dragon 1080p 64spp",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "sec_per_frame": elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: Dragon stand-in (seeded torus-knot tube, 871,200 tris) in 10-tri room, "
                                   "3 spot lights + ambient, reps 16, aperture 0.05, depth 16",
                       "width": W, "height": H, "spp": spec.AntiAliasingFactor, "shapes": info["n_shapes"],
                       "tree_blocks": info["n_inner_nodes"], "child_records": info["n_child_records"],
                       "tree_depth": info["tree_depth"], "partition": "samples" if world > 1 else "all",
                       "real_stl": bool(os.environ.get("RTGO_DRAGON_STL"))},
            "rays_per_frame": {"primary": n_prim / args.steps, "continuation": n_cont / args.steps,
                               "shadow": n_shadow / args.steps, "shadow_reference_equivalent": n_shadow_ref / args.steps},
            "kernel_ms_per_frame": {"k_extend": ms_extend / args.steps / world, "k_shadow": ms_shadow / args.steps / world,
                                    "shade_phong_other": ms_shade / args.steps / world},
            "roofline": {"bound": "hbm", "kernel": "k_shadow", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_ray": bray["shadow"]["bytes_per_ray"], "issue_bound": issue,
                         # the same rate priced with what THIS tree reads per ray (fewer, fuller blocks than the reference's octree)
                         "achieved_device_tree": achieved * bray_all["config3_device_tree"]["bytes_per_ray"] / bray["shadow"]["bytes_per_ray"],
                         "k_extend_achieved": (extend_bytes / world / (ms_extend / 1e3 / world) / 1e9) if ms_extend > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, spec, args.cpu_rows)
        print(json.dumps(out), flush=True)
    dscene.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(scene, spec, rows):
    """The CPU oracle (float64, best-first traversal, std::thread workers — a restatement of the
    reference's goroutine renderer, not the Go binary: no Go toolchain exists here) on a bounded
    sample of the same workload: `rows` middle rows of the frame at 1 spp."""
    from oracle import bridge
    import copy

    threads = min(16, os.cpu_count() or 1)
    s1 = copy.copy(spec)
    r0 = (spec.Height - rows) // 2
    with bridge.OracleScene(scene) as osc:
        t0 = time.perf_counter()
        _, _, c = osc.render(s1, mode=bridge.F64_DIJKSTRA, threads=threads, rows=(r0, r0 + rows), samples=(0, 1), rgba=False)
        dt = time.perf_counter() - t0
    rays = c["rays_primary"] + c["rays_continuation"] + c["rays_shadow"]
    frac = rows / spec.Height / spec.AntiAliasingFactor
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{rows} middle rows x {spec.Width} px x 1 spp of the same frame ({rays} reference-equivalent rays, "
                      f"{dt:.1f} s); extrapolated full frame {dt / frac:.0f} s",
            "sec_per_frame_extrapolated": dt / frac}


if __name__ == "__main__":
    main()
