"""N > 1 path on CPU: world_size-2 gloo run of the frame driver (raytracer-go_amd/dist.py).  The
device renderer is replaced by the CPU oracle restricted to the rank's sample range — the checker
standing in for the GPU — so this covers the partition + reduce(sum) + resolve logic."""
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

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec, scene = rt.ExampleRegression(96, 57, os.path.join(ROOT, "tests", "golden"))
    spec.AntiAliasingFactor = 6
    accum = torch.zeros((spec.Height, spec.Width, 3), dtype=torch.float64)
    with bridge.OracleScene(scene) as osc:
        def share(part):
            s0, s1 = rtdist.sample_range(spec.AntiAliasingFactor, part.rank, part.world_size)
            a, _, _ = osc.render(spec, seed=part.seed, mode=bridge.F64_DIJKSTRA, threads=2, samples=(s0, s1), rgba=False)
            accum.add_(torch.from_numpy(a))
        rtdist.render_frame(share, accum, spec, rt._abi.RT_PART_SAMPLES, rank, world,
                            reduce_fn=lambda a: dist.reduce(a, dst=0, op=dist.ReduceOp.SUM))
    if rank == 0:
        np.save(out_path, accum.numpy())
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
