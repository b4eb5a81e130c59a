#!/usr/bin/env python3
"""bench.py — Mrays/s on BASELINE.json's headline config: Stanford Dragon 1080p 64 spp
(config 3: ~870k-triangle model in a box room, 3 spot lights, soft shadows reps 16, aperture
0.05, depth 16).  No STL ships with the image, so the Dragon is the seeded procedural stand-in
of raytracer-go_amd/scenes.py (871,200 triangles) unless RTGO_DRAGON_STL names a file.

    python bench.py --gpus N --steps K --warmup W                     (no launcher needed, any N)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --config 5 --mode tiles                  (the other partition north_star names)

A step = one whole frame.  At N > 1 the frame is split over the GPUs — by sample ranges (RT_PART_SAMPLES, the
default, config 4's partition) or by interleaved 64x64 tiles (RT_PART_TILES, config 5's) — each GPU renders into
its own fp32 accumulation buffer in HBM, ONE RCCL reduce(sum) lands the frame on GPU 0, which divides and
quantises: total work is fixed, so "scaling" is "strong".  Two launchers, same partition, same kernels:

  rt_multi   (default when WORLD_SIZE is not set)  ONE process: the C-ABI's rt_multi_create / rt_multi_render
             drive devices 0..N-1 from one host thread each and reduce with ncclReduce inside the library —
             what a cgo caller binds, since the reference's GenerateImage is one blocking call that fans out
             inside (raytracer.go:73-87).  Its timed region ends with the RGBA8 frame in HOST memory.
  torchrun   (when a launcher has set WORLD_SIZE)  one process per GPU, torch.distributed backend nccl = RCCL
             (raytracer-go_amd/dist.py); the timed region ends with the RGBA8 frame in device memory on rank 0.

At N = 1 the default is the single-device path of the torchrun shape (`--launcher rt_multi` forces the other).
Fewer visible GPUs than N is an error: a measured line never comes from GPUs that share a device.

value = TRAVERSALS launched on the device ("one closest-hit or any-hit traversal actually launched":
primary and scattered rays, the shadow rays walked through the tree one by one, and the conservative any-hit walks
of whole (hit, light) pairs, k_pair_beam, one traversal each) / wall time of the timed region, scene already
resident in HBM.  value_rays_decided counts instead every shadow ray whose visibility the device decided, also the
ones proven unoccluded together with the other repetitions of their pair by that pair's one beam walk and never
generated; rays the reference traces but whose outcome cannot matter (zero shading term, outside the spot cone) are in
neither.  The better the beams, the fewer traversals a frame needs: sec_per_frame is the number to compare builds on.  sec_per_frame_one_shot is what the reference's own clock covers
(raytracer.go:69-105): tree build + upload + the first frame.
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
BOUNDS = os.path.join(ROOT, "profiles", "r03", "bounds.json")  # per-frame counter totals of k_shadow + calibrated unit rates
KEYS = ("rays", "rays_primary", "rays_continuation", "rays_shadow", "rays_shadow_reference_equivalent",
        "ms_extend", "ms_shadow", "ms_shade", "ms_render", "rays_shadow_beam", "ms_shadow_busy", "beam_walks")
# in the stats of one rt_multi_render call these are the slowest share's (max), the others are sums over the shares
MULTI_MAX_KEYS = ("ms_extend", "ms_shadow", "ms_shade", "ms_render")


def stats_dict(st):
    return {k: float(getattr(st, k)) for k in KEYS}


class TorchDriver:
    """One process per GPU (or the only one): scene on this rank's device, FrameLoop over torch.distributed."""

    name = "torchrun"

    def __init__(self, args, mode, mods):
        """the runtime: torch, this rank's device, the process group (not part of a GenerateImage call)"""
        import torch

        self.torch, self.mods, self.mode = torch, mods, mode
        abi, backend, rtdist = mods["abi"], mods["backend"], mods["dist"]
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={self.world} does not match --gpus {args.gpus}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        n_dev = torch.cuda.device_count()
        self.rehearsal = False
        if n_dev < self.world:
            if not args.allow_shared_devices:
                raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible: a measured line needs one GPU per rank "
                                 "(--allow-shared-devices rehearses the code path over gloo and marks the line)")
            self.rehearsal = True  # ranks share devices, the reduce goes through gloo: RCCL refuses two ranks on one device
        local_rank %= n_dev
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if self.world > 1:
            rtdist.init_process_group(self.rank, self.world, rtdist.pick_backend(self.world, n_dev), dev)
        if self.world == 1:
            self.name = "single"
        self.dev, self.local_rank = dev, local_rank
        torch.zeros(1, device=dev)  # the runtime's own first-use costs stay out of the frame's clock
        torch.cuda.synchronize()
        self.rccl_ranks = self.world if (self.world > 1 and not self.rehearsal) else 0

    def create(self, spec, scene):
        torch, backend, rtdist, dev = self.torch, self.mods["backend"], self.mods["dist"], self.dev
        t0 = time.perf_counter()
        self.dscene = backend.DeviceScene(scene, device=self.local_rank)  # = NewBoundingVolumeHierarchy + upload (raytracer.go:69)
        self.create_s = time.perf_counter() - t0
        self.info = self.dscene.info()
        W, H = spec.Width, spec.Height
        self.accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        self.rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.spec = spec
        self.lib_build_s = 0.0
        self.loop = rtdist.FrameLoop(spec, self.mode, self.rank, self.world, self.accum, self._share,
                                     resolve=lambda a: backend.resolve_device(a.data_ptr(), W, H, spec.AntiAliasingFactor,
                                                                              self.rgba.data_ptr(), stream=self.stream))

    def _share(self, part):
        st = self.dscene.render_accum_device(self.spec, self.accum.data_ptr(), stream=self.stream, part=part)
        self.lib_build_s = st.ms_build / 1e3
        return stats_dict(st)

    def step(self):
        return self.loop.step()

    def fence(self):
        self.loop.fence()

    def alone(self):
        """this rank's share once more (the caller has set RTGO_LANES=1)"""
        return self._share(self.loop.part)

    def aggregate(self, elapsed, sums):
        return self.loop.aggregate(elapsed, sums)

    def close(self):
        self.dscene.close()
        if self.world > 1:
            import torch.distributed as dist

            dist.destroy_process_group()


class MultiDriver:
    """ONE process, one blocking call per frame: rt_multi_* over devices 0..N-1 (ncclReduce inside the library)."""

    name = "rt_multi"

    def __init__(self, args, mode, mods):
        """the runtime: which devices (rt_init also starts each device's HIP warm-up, as a cgo caller's init would)"""
        backend = mods["backend"]
        self.mods, self.mode, self.args = mods, mode, args
        self.world, self.rank, self.rehearsal = args.gpus, 0, False
        # visible GPUs, counted through the library itself (rt_init also starts each device's runtime warm-up); torch
        # stays out of this process
        n_dev = 0
        try:
            while n_dev < args.gpus:
                backend.init(n_dev)
                n_dev += 1
        except backend.RtgoError as e:
            if n_dev == 0:
                raise SystemExit(f"bench.py needs an MI355X: the HIP path has no CPU fallback ({e})")
        if n_dev < args.gpus:
            if not args.allow_shared_devices:
                raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible: a measured line needs one GPU per share "
                                 "(--allow-shared-devices rehearses the code path on shared devices and marks the line)")
            self.rehearsal = True
        self.devices = [r % n_dev for r in range(args.gpus)]
        self.rccl_ranks = 0

    def create(self, spec, scene):
        import numpy as np

        backend, args = self.mods["backend"], self.args
        t0 = time.perf_counter()
        self.m = backend.MultiDeviceScene(scene, self.devices)
        self.create_s = time.perf_counter() - t0
        mi = self.m.info()
        self.rccl_ranks = mi["rccl_ranks"]
        if args.gpus > 1 and not self.rehearsal and self.rccl_ranks != args.gpus:
            raise SystemExit(f"rt_multi made {self.rccl_ranks} RCCL ranks for {args.gpus} GPUs")
        self.info = None
        self.spec = spec
        self.out = np.zeros((spec.Height, spec.Width, 4), dtype=np.uint8)
        self.lib_build_s = 0.0

    def step(self):
        _, st = self.m.render(self.spec, mode=self.mode, out=self.out)
        self.lib_build_s = st.ms_build / 1e3
        return stats_dict(st)

    def fence(self):
        pass  # rt_multi_render returns with the frame in host memory and every device idle

    def alone(self):
        return self.step()

    def aggregate(self, elapsed, sums):
        return elapsed, sums

    def close(self):
        self.m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=(3, 5), help="3 = the headline (Dragon 1080p 64 spp); 5 = Bunny + Dragon + glass + Metal floor at 4K")
    ap.add_argument("--mode", choices=("samples", "tiles"), default=None, help="partition over the GPUs (default: samples for config 3, tiles for config 5)")
    ap.add_argument("--launcher", choices=("auto", "single", "rt_multi", "torchrun"), default="auto")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=480, help="rows of the 1-spp slice the CPU baseline renders")
    ap.add_argument("--allow-shared-devices", action="store_true", help="rehearse N shares on fewer GPUs (the line says so; never a measurement)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    rt = importlib.import_module("raytracer-go_amd")
    mods = {"abi": rt._abi, "backend": importlib.import_module("raytracer-go_amd.backend"),
            "scenes": importlib.import_module("raytracer-go_amd.scenes"), "dist": importlib.import_module("raytracer-go_amd.dist")}
    abi, backend, scenes = mods["abi"], mods["backend"], mods["scenes"]

    launcher = args.launcher
    if launcher == "auto":
        launcher = "torchrun" if "WORLD_SIZE" in os.environ else ("rt_multi" if args.gpus > 1 else "single")
    if launcher == "single" and args.gpus != 1:
        raise SystemExit("--launcher single renders on one GPU")
    mode_name = args.mode or ("tiles" if args.config == 5 else "samples")
    mode = abi.RT_PART_TILES if mode_name == "tiles" else abi.RT_PART_SAMPLES

    drv = (MultiDriver if launcher == "rt_multi" else TorchDriver)(args, mode, mods)
    t_scene = time.perf_counter()
    if args.config == 5:
        W, H = args.width or 3840, args.height or 2160
        spec, scene = scenes.config5(W, H, args.spp, 16)
    else:
        W, H = args.width or 1920, args.height or 1080
        spec, scene = scenes.config3(W, H, args.spp, 16)
    t_described = time.perf_counter()
    drv.create(spec, scene)
    world, rank = drv.world, drv.rank

    # the first frame, cold: with the tree build and the upload before it, this is one GenerateImage call
    drv.step()
    drv.fence()
    one_shot = time.perf_counter() - t_described
    first_frame_s = one_shot - drv.create_s
    for _ in range(max(0, args.warmup - 1)):
        drv.step()
    drv.fence()
    totals = dict.fromkeys(KEYS, 0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for k, v in drv.step().items():
            totals[k] += v
    drv.fence()
    elapsed, sums = drv.aggregate(time.perf_counter() - t0, [totals[k] for k in KEYS] + [one_shot, first_frame_s])
    tot = dict(zip(KEYS, sums))
    one_shot_sum, first_frame_sum = sums[len(KEYS)], sums[len(KEYS) + 1]
    if launcher == "rt_multi":  # one process saw everything: its one-shot is the job's; max-type times are already per share
        per_rank = lambda k: tot[k] / args.steps / (1 if k in MULTI_MAX_KEYS else world)
        one_shot_mean, first_frame_mean = one_shot_sum, first_frame_sum
    else:
        per_rank = lambda k: tot[k] / args.steps / world
        one_shot_mean, first_frame_mean = one_shot_sum / world, first_frame_sum / world
    # after the timed region: the same frame once more on ONE lane (RTGO_LANES=1), where no other kernel runs
    # beside k_shadow, so its launch durations are those of the kernel alone (roofline["alone"])
    lanes_before = os.environ.get("RTGO_LANES")
    os.environ["RTGO_LANES"] = "1"
    alone = drv.alone()
    if lanes_before is None:
        del os.environ["RTGO_LANES"]
    else:
        os.environ["RTGO_LANES"] = lanes_before

    if rank == 0:
        with open(os.path.join(ROOT, "profiles", "bray.json")) as f:
            bray_all = json.load(f)
        bray = bray_all.get(f"config{args.config}")
        sec_per_frame = elapsed / args.steps
        rays, n_beam = tot["rays"], tot["rays_shadow_beam"]
        # k_shadow seconds per frame and GPU: the time during which k_shadow was running on at least one of a render
        # call's streams (mean over the GPUs).  The launches of the lanes overlap, so their summed durations
        # (kernel_ms_per_frame.k_shadow) count the same wall time several times.
        shadow_s = per_rank("ms_shadow_busy") / 1e3
        build = backend.build_info()
        workloads = {3: "config3: Dragon stand-in (seeded torus-knot tube, 871,200 tris) in 10-tri room, 3 spot lights + ambient, "
                        "reps 16, aperture 0.05, depth 16",
                     5: "config5: Dragon + Bunny stand-ins (953,120 tris) + 6 Dielectric spheres on a single-sided Metal floor in a "
                        "room, 4 spot lights + ambient, reps 16, aperture 0.03, depth 16"}
        out = {
            "metric": "Mrays/s (whole node), Stanford Dragon 1080p 64spp" if args.config == 3 else "Mrays/s (whole node), config 5 (Bunny + Dragon + glass, 4K 64spp)",
            "value": (rays - n_beam + tot["beam_walks"]) / elapsed / 1e6,
            "unit": "Mrays/s",
            # + the shadow rays a pair's beam walk proved unoccluded without generating them (rays_per_frame.of_which_decided_by_pair_beams)
            "value_rays_decided": rays / elapsed / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_per_frame * 1e3,
            "sec_per_frame": sec_per_frame,
            "sec_per_frame_one_shot": one_shot_mean,
            "first_frame_minus_steady_s": first_frame_mean - sec_per_frame,
            "one_shot_parts_rank0": {"scene_create_s": drv.create_s, "of_which_library_s": drv.lib_build_s, "first_frame_s": first_frame_s,
                                     "describe_scene_s_not_included": t_described - t_scene},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "launcher": drv.name, "rccl_ranks": drv.rccl_ranks,
            "config": {"workload": workloads[args.config],
                       "width": W, "height": H, "spp": spec.AntiAliasingFactor, "shapes": sum(len(s) if hasattr(s, "__len__") else 1 for s in scene.Shapes),
                       "partition": mode_name if world > 1 else "all",
                       "real_stl": bool(os.environ.get("RTGO_DRAGON_STL"))},
            "build": build,
            "rays_per_frame": {"primary": tot["rays_primary"] / args.steps, "continuation": tot["rays_continuation"] / args.steps,
                               "shadow": tot["rays_shadow"] / args.steps, "of_which_decided_by_pair_beams": n_beam / args.steps,
                               "pair_beam_walks": tot["beam_walks"] / args.steps,
                               "shadow_reference_equivalent": tot["rays_shadow_reference_equivalent"] / args.steps},
            "kernel_ms_per_frame_per_gpu": {"k_extend": per_rank("ms_extend"), "k_shadow": per_rank("ms_shadow"),
                                            "k_shadow_busy": per_rank("ms_shadow_busy"), "shade_phong_other": per_rank("ms_shade")},
            "roofline": roofline(shadow_s, (tot["rays_shadow"] - n_beam) / args.steps / world, world, W, H, spec.AntiAliasingFactor, bray, args.config, build),
        }
        if drv.info is not None:
            out["config"].update(tree_blocks=drv.info["n_inner_nodes"], child_records=drv.info["n_child_records"], tree_depth=drv.info["tree_depth"])
        if drv.rehearsal:
            out["rehearsal"] = "shares on shared devices: NOT a measurement"
        # the same kernel with the GPU to itself (one lane): what the kernel sustains when no HBM-bound kernel of
        # another lane takes time slices from it
        alone_rays = (alone["rays_shadow"] - alone["rays_shadow_beam"]) / (world if launcher == "rt_multi" else 1)
        r1 = roofline(alone["ms_shadow"] / 1e3, alone_rays, world, W, H, spec.AntiAliasingFactor, bray, args.config, build)
        out["roofline"]["alone"] = {"kernel_s_per_frame": r1["kernel_s_per_frame"], "achieved": r1["achieved"], "frac": r1["frac"],
                                    "frame_ms_one_lane": alone["ms_render"]}
        out["roofline"]["lanes"] = int(lanes_before or "4")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, spec, args.cpu_rows)
        print(json.dumps(out), flush=True)
    drv.close()


def roofline(shadow_s, shadow_rays, world, W, H, spp, bray, config, build):
    """Which unit of the chip k_shadow keeps busiest, as a fraction of what that unit sustains.

    The kernel's tree and shapes (~60 MB) stay in L2 / Infinity Cache, so HBM is not its roof (fractions.hbm below).
    The candidates are VALU issue and the vector-memory path (TA + vector L1).  Their per-frame work is counted by
    rocprofv3 PMC passes over this same command (profiles/r03/bounds.json: instruction and access counts are
    properties of the workload and the build, identical from run to run for a seed); what each unit sustains per clock
    comes from the micro-benchmarks of scripts/calib.hip and scripts/calib_mix.hip run on the same box (same file).
    The file names the build it was counted on (rt_build_info: source hash + kernel knobs): if the loaded library is
    another, the line says "stale": true and carries no frac.
    peak (VALU) = the rate at which the SIMDs could issue this kernel's instruction mix.  frac = busy time of the
    busiest unit / the kernel's measured time (HIP events around every k_shadow launch of the timed region).
    A render call runs its batches on four streams (rtgo_hip.hip Lane), whose k_shadow launches overlap one another and
    the other lanes' HBM-bound kernels: the kernel's time in the timed region is the time during which at least one
    k_shadow launch was running (rt_stats.ms_shadow_busy, the union of the launches' event intervals), which still
    includes the time slices the other kernels got, so `frac` is lower than the kernel's own; roofline["alone"] is the
    same measurement over one more frame rendered on a single lane, where a launch has the GPU to itself.
    At N > 1 GPUs each GPU does 1/N of the frame's k_shadow work (counts divided by N, `per_gpu_counts_scaled`)."""
    out = {"bound": None, "kernel": "k_shadow", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
           "kernel_s_per_frame": shadow_s}
    # SURVEY 8d's figure, kept for the record: the REFERENCE's traversal bytes (its octree, 224 B per node) per second
    if bray and shadow_s > 0:
        out["reference_equivalent_GBps"] = shadow_rays * bray["shadow"]["bytes_per_ray"] / shadow_s / 1e9
    if not os.path.exists(BOUNDS) or shadow_s <= 0:
        out["stale"] = True
        out["why"] = "no committed counts"
        return out
    with open(BOUNDS) as f:
        b = json.load(f)
    out["source"] = os.path.relpath(BOUNDS, ROOT)
    if b["workload"] != {"width": W, "height": H, "spp": spp, "config": f"config{config}"}:
        out["stale"] = True
        out["why"] = "the committed counts are per frame of the headline workload"
        return out
    if b.get("build") != build:
        out["stale"] = True
        out["why"] = f"counted on build {b.get('build')}, loaded library is {build}"
        return out
    pf, c = b["per_frame"], b["calib"]
    clk = c["clock_hz"]
    share = 1.0 / world
    # VALU: instructions by class (SQ_INSTS_VALU_* of a PMC pass) x the SIMD cycles an instruction of that class takes
    # (scripts/calib_mix.hip on the same box).  A class holds cheap and dear instructions (a v_mov takes 2.6 cycles, a
    # v_cndmask 4.3; both count as "other"), so the busy time is a bracket: its lower end, every class at its cheapest
    # instruction, is what frac reports.
    lo = share * sum(pf["valu_mix"][k] * c["valu_class_cycles"][k][0] for k in pf["valu_mix"])
    hi = share * sum(pf["valu_mix"][k] * c["valu_class_cycles"][k][1] for k in pf["valu_mix"])
    valu_s, valu_s_hi = lo / (c["simds"] * clk), hi / (c["simds"] * clk)
    tcp, tcp_l2 = share * pf["tcp_accesses"], share * pf["tcp_l2_read_req"]
    ta_s = ((tcp - tcp_l2) / c["tcp_l1_hit_accesses_per_clk_per_cu"] + tcp_l2 / c["tcp_l2_served_accesses_per_clk_per_cu"]) / (c["cus"] * clk)
    hbm_bytes = share * (2 * pf["hbm_fetch_bytes"] + pf["hbm_write_bytes"])  # gfx950 FETCH_SIZE counts half the bytes (guide, HBM section)
    fr = {"valu_issue": valu_s / shadow_s, "ta_l1": ta_s / shadow_s, "hbm": hbm_bytes / shadow_s / (HBM_PEAK_GBS * 1e9)}
    bound = max(fr, key=fr.get)
    insts = share * pf["valu_insts"]
    if bound == "valu_issue":
        out.update(achieved=insts / shadow_s / 1e9, peak=insts / valu_s / 1e9, unit="G wave64-inst/s")
    elif bound == "ta_l1":
        out.update(achieved=tcp / shadow_s / 1e9, peak=tcp / ta_s / 1e9, unit="G L1-access/s")
    else:
        out.update(achieved=hbm_bytes / shadow_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
    fr["valu_issue_if_every_class_cost_its_dearest_instruction"] = valu_s_hi / shadow_s
    # of the issued VALU lane slots the fraction that did work (SQ_THREAD_CYCLES_VALU): frac x this = useful lane throughput
    out.update(bound=bound, frac=fr[bound], traffic=hbm_bytes / b["launches_per_frame"], fractions=fr,
               valu_active_lanes_frac=pf.get("valu_active_lanes_frac"), ta_busy_frac_profiled=pf.get("ta_busy_frac"),
               stale=False, per_gpu_counts_scaled=world > 1)
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
    rows = min(rows, spec.Height)
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
