"""Multi-GPU frame rendering with one process per GPU: torch.distributed, backend "nccl" (= RCCL over xGMI) on
the GPU box, "gloo" where ranks share a device or there is none (CPU tests).  (One process driving several
GPUs is the C-ABI's rt_multi_*; this module is the torchrun shape that bench.py --gpus N uses.)

The path shards by construction — every (pixel, sample) is independent — so each rank renders its share into
a full-frame fp32 RGB accumulation buffer of un-normalised sums (zeros where it owns nothing) and ONE
reduce(sum) to rank 0 per frame is the only exchange.  Rank 0 then divides by spp and quantises
(raytracer.go:119, :93-96).  Shares:
  RT_PART_SAMPLES  rank r renders samples [r*spp/G, (r+1)*spp/G) of every pixel (config 4)
  RT_PART_TILES    rank r renders 64x64 tiles k with k % G == r (config 5)
The Philox counter carries the global pixel and sample index, so a sample's value does not depend on G; the
reduced image differs from the single-GPU one only by fp32 summation order (sample split) or not at all
(tile split).
"""
import os

from . import _abi


def sample_range(spp, rank, world):
    """Same split as rt_render_accum_device (RT_PART_SAMPLES) and rt_partition_share."""
    return (spp * rank) // world, (spp * (rank + 1)) // world


def tile_owner(tile_index, world):
    return tile_index % world


def pick_backend(world, n_devices):
    """RCCL needs one device per rank; ranks that share a device (a rehearsal) or have none reduce over gloo."""
    return os.environ.get("RTGO_DIST_BACKEND", "nccl" if n_devices >= world else "gloo")


def init_process_group(rank, world, backend, device=None):
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


class FrameLoop:
    """One frame per step(), split over the ranks of the default process group.

    render_share(part) ADDS this rank's share into `accum` (a (H, W, 3) tensor on the rank's device) and returns
    a dict of counters; resolve(accum), on rank 0 only, turns the reduced sums into the frame."""

    def __init__(self, spec, mode, rank, world, accum, render_share, resolve=None, seed=0x5EED, tile_size=64):
        self.spec, self.rank, self.world, self.accum = spec, rank, world, accum
        self.render_share, self.resolve = render_share, resolve
        self.part = _abi.partition(seed=seed, mode=mode if world > 1 else _abi.RT_PART_ALL, rank=rank, world_size=world,
                                   tile_size=tile_size)

    def step(self):
        self.accum.zero_()
        counters = self.render_share(self.part)
        if self.world > 1:
            import torch.distributed as dist

            dist.reduce(self.accum, dst=0, op=dist.ReduceOp.SUM)
        if self.rank == 0 and self.resolve is not None:
            self.resolve(self.accum)
        return counters

    def fence(self):
        import torch

        if self.world > 1:
            import torch.distributed as dist

            dist.barrier()
        if self.accum.is_cuda:
            torch.cuda.synchronize()

    def aggregate(self, elapsed, sums):
        """(max over ranks of elapsed, sums added over ranks) — the figures a whole-job rate is made of."""
        import torch

        t = torch.tensor([float(elapsed)], dtype=torch.float64, device=self.accum.device)
        v = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=self.accum.device)
        if self.world > 1:
            import torch.distributed as dist

            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return float(t.cpu()[0]), [float(x) for x in v.cpu()]
