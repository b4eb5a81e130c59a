"""Multi-GPU frame rendering: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

The path shards by construction — every (pixel, sample) is independent — so each rank renders
its share into a full-frame fp32 RGB accumulation buffer of un-normalised sums (zeros where it
owns nothing) and ONE reduce(sum) to rank 0 per frame is the only exchange.  Rank 0 then
divides by spp and quantises (raytracer.go:119, :93-96).  Shares:
  RT_PART_SAMPLES  rank r renders samples [r*spp/G, (r+1)*spp/G) of every pixel (config 4)
  RT_PART_TILES    rank r renders 64x64 tiles k with k % G == r (config 5)
The Philox counter carries the global pixel and sample index, so a sample's value does not
depend on G; the reduced image differs from the single-GPU one only by fp32 summation order
(sample split) or not at all (tile split).
"""
from . import _abi


def sample_range(spp, rank, world):
    """Same split as rt_render_accum_device (RT_PART_SAMPLES)."""
    return (spp * rank) // world, (spp * (rank + 1)) // world


def tile_owner(tile_index, world):
    return tile_index % world


def render_frame(render_share, accum, spec, mode, rank, world, seed=0x5EED, reduce_fn=None, tile_size=64):
    """render_share(part) must ADD this rank's share into `accum` (a tensor/array of shape
    (H, W, 3)); reduce_fn(accum) sums it onto rank 0 (torch.distributed.reduce).  Returns the
    partition that was rendered."""
    part = _abi.partition(seed=seed, mode=mode if world > 1 else _abi.RT_PART_ALL, rank=rank, world_size=world,
                          tile_size=tile_size)
    render_share(part)
    if world > 1 and reduce_fn is not None:
        reduce_fn(accum)
    return part
