"""N > 1 path on CPU: a world_size-2 gloo run of the frame loop that bench.py --gpus N drives
(raytracer-go_amd/dist.py: backend choice, process-group set-up, FrameLoop.step = zero + share + reduce(sum) +
resolve on rank 0, FrameLoop.aggregate = max time / summed counters).  Only the share renderer is replaced: the
CPU oracle restricted to the rank's sample range stands in for the device, the checker in the GPU's place."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    rt = importlib.import_module("raytracer-go_amd")
    rtdist = importlib.import_module("raytracer-go_amd.dist")
    from oracle import bridge

    os.environ["MASTER_PORT"] = str(port)
    backend = rtdist.pick_backend(world, 0)  # no device here: gloo
    assert backend == "gloo"
    rtdist.init_process_group(rank, world, backend)
    spec, scene = rt.ExampleRegression(96, 57, os.path.join(ROOT, "tests", "golden"))
    spec.AntiAliasingFactor = 6
    accum = torch.zeros((spec.Height, spec.Width, 3), dtype=torch.float64)
    frames = []
    with bridge.OracleScene(scene) as osc:
        def share(part):
            assert (part.rank, part.world_size, part.mode) == (rank, world, rt._abi.RT_PART_SAMPLES)
            s0, s1 = rtdist.sample_range(spec.AntiAliasingFactor, part.rank, part.world_size)
            a, _, c = osc.render(spec, seed=part.seed, mode=bridge.F64_DIJKSTRA, threads=2, samples=(s0, s1), rgba=False)
            accum.add_(torch.from_numpy(a))
            return {"rays": c["rays_primary"] + c["rays_continuation"] + c["rays_shadow"], "paths": (s1 - s0) * spec.Width * spec.Height}

        loop = rtdist.FrameLoop(spec, rt._abi.RT_PART_SAMPLES, rank, world, accum, share, resolve=lambda a: frames.append(a.clone().numpy()))
        counters = [loop.step() for _ in range(2)]  # two frames: the buffer is zeroed per step
        loop.fence()
        elapsed, sums = loop.aggregate(1.0 + rank, [counters[-1]["rays"], counters[-1]["paths"]])
    assert elapsed == float(world)  # the slowest rank's time
    assert sums[1] == spec.Width * spec.Height * spec.AntiAliasingFactor
    assert (len(frames) == 2) == (rank == 0)
    if rank == 0:
        assert np.array_equal(frames[0], frames[1])
        np.save(out_path, frames[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_split_equals_single(tmp_path, rt, bridge):
    import torch.multiprocessing as mp
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, 29731, out), nprocs=2, join=True)
    got = np.load(out)
    spec, scene = rt.ExampleRegression(96, 57, os.path.join(ROOT, "tests", "golden"))
    spec.AntiAliasingFactor = 6
    with bridge.OracleScene(scene) as osc:
        ref, _, _ = osc.render(spec, mode=bridge.F64_DIJKSTRA, threads=2, rgba=False)
    # same samples (Philox counter = global sample index); only the summation order differs
    assert np.abs(got - ref).max() < 1e-12
