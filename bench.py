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
"scaling" is "strong".  value = rays whose outcome all ranks determined / wall time of the timed region
(scene already resident in HBM; the timed region ends with the RGBA8 frame in device memory on rank 0):
primary and scattered rays + shadow rays, whether a shadow ray was walked through the tree on its own or
proven unoccluded with the other repetitions of its (hit, light) pair by one conservative beam walk; rays
the reference traces but whose outcome cannot matter (zero shading term, outside the spot cone) are not counted.
sec_per_frame_one_shot is what the reference's own clock covers (raytracer.go:69-105): tree build +
upload + the first frame.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BOUNDS = os.path.join(ROOT, "profiles", "r02", "bounds.json")  # per-frame counter totals of k_shadow + calibrated unit rates


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

    import torch

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
        rtdist.init_process_group(rank, world, rtdist.pick_backend(world, n_dev), dev)

    t_cold = time.perf_counter()
    spec, scene = scenes.config3(args.width, args.height, args.spp, 16)
    t_scene = time.perf_counter()
    dscene = backend.DeviceScene(scene, device=local_rank)  # = NewBoundingVolumeHierarchy + upload (raytracer.go:69)
    t_created = time.perf_counter()
    info = dscene.info()
    W, H = spec.Width, spec.Height
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    keys = ("rays", "rays_primary", "rays_continuation", "rays_shadow", "rays_shadow_reference_equivalent",
            "ms_extend", "ms_shadow", "ms_shade", "ms_render", "rays_shadow_beam", "ms_shadow_busy")

    lib_build = {"s": 0.0}

    def share(part):
        st = dscene.render_accum_device(spec, accum.data_ptr(), stream=stream, part=part)
        lib_build["s"] = st.ms_build / 1e3
        return {k: float(getattr(st, k)) for k in keys}

    loop = rtdist.FrameLoop(spec, abi.RT_PART_SAMPLES, rank, world, accum, share,
                            resolve=lambda a: backend.resolve_device(a.data_ptr(), W, H, spec.AntiAliasingFactor, rgba.data_ptr(), stream=stream))
    # the first frame, cold: with the tree build and the upload before it, this is one GenerateImage call
    loop.step()
    loop.fence()
    one_shot = time.perf_counter() - t_scene
    one_shot_parts = {"scene_create_s": t_created - t_scene, "of_which_library_s": lib_build["s"],
                      "first_frame_s": one_shot - (t_created - t_scene)}
    for _ in range(max(0, args.warmup - 1)):
        loop.step()
    loop.fence()
    totals = dict.fromkeys(keys, 0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for k, v in loop.step().items():
            totals[k] += v
    loop.fence()
    elapsed, sums = loop.aggregate(time.perf_counter() - t0, [totals[k] for k in keys] + [one_shot])
    rays, n_prim, n_cont, n_shadow, n_shadow_ref, ms_extend, ms_shadow, ms_shade, _ms_render, n_beam, ms_shadow_busy, one_shot_sum = sums
    # after the timed region: this rank's share once more on ONE lane (RTGO_LANES=1), where no other kernel runs
    # beside k_shadow, so its launch durations are those of the kernel alone (roofline["alone"])
    os.environ["RTGO_LANES"] = "1"
    alone = share(loop.part)
    del os.environ["RTGO_LANES"]

    if rank == 0:
        with open(os.path.join(ROOT, "profiles", "bray.json")) as f:
            bray_all = json.load(f)
        bray = bray_all["config3"]
        # k_shadow seconds per frame: the time during which k_shadow was running on at least one of the call's streams
        # (mean over ranks).  The launches of the lanes overlap, so their summed durations (kernel_ms_per_frame) count
        # the same wall time several times.
        shadow_s = ms_shadow_busy / 1e3 / world / args.steps
        out = {
            "metric": "Mrays/s (whole node), Stanford Dragon 1080p 64spp",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            # `value` counts every ray whose outcome the device determined; the shadow rays that a pair's beam walk proved
            # unoccluded (never generated one by one, rays_per_frame.of_which_decided_by_pair_beams) are left out here
            "value_rays_traced_one_by_one": (rays - n_beam) / elapsed / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "sec_per_frame": elapsed / args.steps,
            "sec_per_frame_one_shot": one_shot_sum / world,
            "one_shot_parts_rank0": one_shot_parts,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: Dragon stand-in (seeded torus-knot tube, 871,200 tris) in 10-tri room, "
                                   "3 spot lights + ambient, reps 16, aperture 0.05, depth 16",
                       "width": W, "height": H, "spp": spec.AntiAliasingFactor, "shapes": info["n_shapes"],
                       "tree_blocks": info["n_inner_nodes"], "child_records": info["n_child_records"],
                       "tree_depth": info["tree_depth"], "partition": "samples" if world > 1 else "all",
                       "real_stl": bool(os.environ.get("RTGO_DRAGON_STL"))},
            "rays_per_frame": {"primary": n_prim / args.steps, "continuation": n_cont / args.steps,
                               "shadow": n_shadow / args.steps, "of_which_decided_by_pair_beams": n_beam / args.steps,
                               "shadow_reference_equivalent": n_shadow_ref / args.steps},
            "kernel_ms_per_frame": {"k_extend": ms_extend / args.steps / world, "k_shadow": ms_shadow / args.steps / world,
                                    "k_shadow_busy": ms_shadow_busy / args.steps / world,
                                    "shade_phong_other": ms_shade / args.steps / world},
            "roofline": roofline(shadow_s, n_shadow / args.steps / world, world, W, H, spec.AntiAliasingFactor, bray),
        }
        # the same kernel with the GPU to itself (rank 0's share, one lane): what the kernel sustains when no
        # HBM-bound kernel of the other lane takes time slices from it
        r1 = roofline(alone["ms_shadow"] / 1e3, alone["rays_shadow"], world, W, H, spec.AntiAliasingFactor, bray)
        out["roofline"]["alone"] = {"kernel_s_per_frame": r1["kernel_s_per_frame"], "achieved": r1["achieved"], "frac": r1["frac"],
                                    "frame_ms_one_lane": alone["ms_render"]}
        out["roofline"]["lanes"] = int(os.environ.get("RTGO_LANES", "4"))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, spec, args.cpu_rows)
        print(json.dumps(out), flush=True)
    dscene.close()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def roofline(shadow_s, shadow_rays, world, W, H, spp, bray):
    """Which unit of the chip k_shadow keeps busiest, as a fraction of what that unit sustains.

    The kernel's tree and shapes (~60 MB) stay in L2 / Infinity Cache, so HBM is not its roof (hbm_frac below).
    The candidates are VALU issue and the vector-memory path (TA + vector L1).  Their per-frame work is counted by
    rocprofv3 PMC passes over this same command (profiles/r02/bounds.json: instruction and access counts are
    properties of the workload, identical from run to run for a seed); what each unit sustains per clock comes
    from the micro-benchmarks of scripts/calib.hip and scripts/calib_mix.hip run on the same box (same file).
    peak (VALU) = the rate at which the SIMDs could issue this kernel's instruction mix.  frac = busy time of the
    busiest unit / the kernel's measured time (HIP events around every k_shadow launch of the timed region).
    A render call runs its batches on four streams (rtgo_hip.hip Lane), whose k_shadow launches overlap one another and
    the other lanes' HBM-bound kernels: the kernel's time in the timed region is the time during which at least one
    k_shadow launch was running (rt_stats.ms_shadow_busy, the union of the launches' event intervals), which still
    includes the time slices the other kernels got, so `frac` is lower than the kernel's own; roofline["alone"] is the
    same measurement over one more frame rendered on a single lane, where a launch has the GPU to itself."""
    out = {"bound": None, "kernel": "k_shadow", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
           "kernel_s_per_frame": shadow_s}
    # SURVEY 8d's figure, kept for the record: the REFERENCE's traversal bytes (its octree, 224 B per node) per second
    out["reference_equivalent_GBps"] = shadow_rays * bray["shadow"]["bytes_per_ray"] / shadow_s / 1e9 if shadow_s > 0 else None
    if not os.path.exists(BOUNDS) or shadow_s <= 0:
        return out
    with open(BOUNDS) as f:
        b = json.load(f)
    if world != 1 or b["workload"] != {"width": W, "height": H, "spp": spp, "config": "config3"}:
        return out  # the committed counts are per frame of the 1-GPU headline run
    pf, c = b["per_frame"], b["calib"]
    clk = c["clock_hz"]
    # VALU: instructions by class (SQ_INSTS_VALU_* of a PMC pass) x the SIMD cycles an instruction of that class takes
    # (scripts/calib_mix.hip on the same box).  A class holds cheap and dear instructions (a v_mov takes 2.6 cycles, a
    # v_cndmask 4.3; both count as "other"), so the busy time is a bracket: its lower end, every class at its cheapest
    # instruction, is what frac reports.
    lo = sum(pf["valu_mix"][k] * c["valu_class_cycles"][k][0] for k in pf["valu_mix"])
    hi = sum(pf["valu_mix"][k] * c["valu_class_cycles"][k][1] for k in pf["valu_mix"])
    valu_s, valu_s_hi = lo / (c["simds"] * clk), hi / (c["simds"] * clk)
    hits = pf["tcp_accesses"] - pf["tcp_l2_read_req"]
    ta_s = (hits / c["tcp_l1_hit_accesses_per_clk_per_cu"] + pf["tcp_l2_read_req"] / c["tcp_l2_served_accesses_per_clk_per_cu"]) / (c["cus"] * clk)
    hbm_bytes = 2 * pf["hbm_fetch_bytes"] + pf["hbm_write_bytes"]  # gfx950 FETCH_SIZE counts half the bytes (guide, HBM section)
    fr = {"valu_issue": valu_s / shadow_s, "ta_l1": ta_s / shadow_s, "hbm": hbm_bytes / shadow_s / (HBM_PEAK_GBS * 1e9)}
    bound = max(fr, key=fr.get)
    if bound == "valu_issue":
        out.update(achieved=pf["valu_insts"] / shadow_s / 1e9, peak=pf["valu_insts"] / valu_s / 1e9, unit="G wave64-inst/s")
    elif bound == "ta_l1":
        out.update(achieved=pf["tcp_accesses"] / shadow_s / 1e9, peak=pf["tcp_accesses"] / ta_s / 1e9, unit="G L1-access/s")
    else:
        out.update(achieved=hbm_bytes / shadow_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
    fr["valu_issue_if_every_class_cost_its_dearest_instruction"] = valu_s_hi / shadow_s
    out.update(bound=bound, frac=fr[bound], traffic=hbm_bytes / b["launches_per_frame"], fractions=fr,
               ta_busy_frac_profiled=pf.get("ta_busy_frac"), source="profiles/r02/bounds.json")
    return out


def cpu_threads():
    """The host threads this process may use: its CPU affinity, capped by OMP_NUM_THREADS where the box sets a share
    (a one-GPU box is given 16 of the machine's CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)


def cpu_baseline(scene, spec, rows):
    """The CPU oracle (float64, best-first traversal, std::thread workers — a restatement of the
    reference's goroutine renderer, not the Go binary: no Go toolchain exists here) on a bounded
    sample of the same workload: `rows` middle rows of the frame at 1 spp."""
    from oracle import bridge
    import copy

    threads = cpu_threads()
    s1 = copy.copy(spec)
    r0 = (spec.Height - rows) // 2
    with bridge.OracleScene(scene) as osc:
        t0 = time.perf_counter()
        _, _, c = osc.render(s1, mode=bridge.F64_DIJKSTRA, threads=threads, rows=(r0, r0 + rows), samples=(0, 1), rgba=False)
        dt = time.perf_counter() - t0
    rays = c["rays_primary"] + c["rays_continuation"] + c["rays_shadow"]
    frac = rows / spec.Height / spec.AntiAliasingFactor
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{rows} middle rows x {spec.Width} px x 1 spp of the same frame ({rays} reference-equivalent rays, "
                      f"{dt:.1f} s); extrapolated full frame {dt / frac:.0f} s",
            "sec_per_frame_extrapolated": dt / frac}


if __name__ == "__main__":
    main()
