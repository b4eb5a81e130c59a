"""Experiment: does one GPU render a frame faster as two concurrent shares (two host threads, two streams, rt_multi
with the same device twice) than as one?  k_shadow is VALU-bound; the other kernels are HBM-bound and could overlap."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-go_amd")
backend = importlib.import_module("raytracer-go_amd.backend")
scenes = importlib.import_module("raytracer-go_amd.scenes")
abi = rt._abi
spec, scene = scenes.config3(1920, 1080, 64, 16)
for devs in ([0], [0, 0], [0, 0, 0]):
    for mode, name in ((abi.RT_PART_SAMPLES, "samples"), (abi.RT_PART_TILES, "tiles")):
        with backend.MultiDeviceScene(scene, devs) as m:
            m.render(spec, mode=mode)
            t0 = time.perf_counter()
            for _ in range(2):
                _, st = m.render(spec, mode=mode)
            dt = (time.perf_counter() - t0) / 2
        print(f"shares {len(devs)} by {name}: {dt * 1e3:.1f} ms per frame, ms_render {st.ms_render:.1f}", flush=True)
