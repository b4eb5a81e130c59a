"""The entry point that allocates nothing (rt_render_accum_device_static, include/rtgo.h) against the one it is a
variant of (rt_render_accum_device with the same batch shape): bit-identical frames and equal ray counts, on two
calls in a row — the second with a smaller input, over whatever the first left in the static workspace."""
import copy

import pytest

pytestmark = pytest.mark.gpu


def _both(dev, abi, spec, hb):
    import numpy as np

    part = abi.partition(max_paths=1 << 18)
    shape = (spec.Height, spec.Width, 3)
    nbytes = spec.Height * spec.Width * 3 * 4
    a, b = hb.zeros(nbytes), hb.zeros(nbytes)
    st_a = dev.render_accum_device(spec, a, part=part)
    st_b = dev.render_accum_device_static(spec, b, part=part)
    fa, fb = hb.to_host(a, shape, np.float32), hb.to_host(b, shape, np.float32)
    assert st_b.paths == st_a.paths == spec.Width * spec.Height * spec.AntiAliasingFactor
    for k in ("rays_primary", "rays_continuation", "rays_shadow", "rays_shadow_reference_equivalent", "phong_hits"):
        assert getattr(st_a, k) == getattr(st_b, k), k
    assert fa.max() > 0.1
    assert np.array_equal(fa, fb), f"{int((fa != fb).any(-1).sum())} pixels differ"


def test_static_entry_point_matches_render_accum_device(backend, rt, scenes, gold_dir, hip_buffers):
    abi = rt._abi
    hb = hip_buffers
    # config 3 with all 871,210 shapes: the tree's worst case needs the spill region (27 entries per lane beyond LDS)
    spec, scene = scenes.config3(192, 108, 8, 16, asset_dir=gold_dir)
    with backend.DeviceScene(scene) as dev:
        assert dev.info()["stack_bound"] + 1 > 20
        _both(dev, abi, spec, hb)
        small = copy.copy(spec)
        small.Width, small.Height, small.AntiAliasingFactor = 96, 54, 4  # second call, smaller input
        _both(dev, abi, small, hb)
        # beyond the static workspace: one 1024 x 1024 tile is 2^20 paths
        b = hb.zeros(spec.Height * spec.Width * 12)
        with pytest.raises(backend.RtgoError) as e:
            dev.render_accum_device_static(spec, b, part=abi.partition(tile_size=1024))
        assert e.value.code == abi.RT_ERR_LIMIT
    # the README scene: Metal / Dielectric continuation rays, image and checker textures, no spill
    spec1, scene1 = rt.ExampleRegression(160, 95, gold_dir)
    spec1.AntiAliasingFactor = 8
    with backend.DeviceScene(scene1) as dev:
        _both(dev, abi, spec1, hb)
        deep = copy.copy(spec1)
        deep.RayTracingMaxDepth = 65
        b = hb.zeros(95 * 160 * 12)
        with pytest.raises(backend.RtgoError) as e:
            dev.render_accum_device_static(deep, b)
        assert e.value.code == abi.RT_ERR_LIMIT
