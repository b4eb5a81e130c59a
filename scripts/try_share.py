"""Frame time of ONE rank's share of config 3 under RT_PART_SAMPLES for a world of 1, 2, 4, 8 (what a rank of the
multi-GPU bench renders; the reduce is not included)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracer-go_amd")
backend = importlib.import_module("raytracer-go_amd.backend")
scenes = importlib.import_module("raytracer-go_amd.scenes")
abi = rt._abi
spec, scene = scenes.config3(1920, 1080, 64, 16)
with backend.DeviceScene(scene) as dev:
    for world in (1, 2, 4, 8):
        part = abi.partition(mode=abi.RT_PART_SAMPLES, rank=0, world_size=world)
        dev.render_accum(spec, part=part)
        t0 = time.perf_counter()
        for _ in range(3):
            _, st = dev.render_accum(spec, part=part)
        dt = (time.perf_counter() - t0) / 3
        print(f"world {world}: {dt * 1e3:.1f} ms per share (x{world} = {dt * world * 1e3:.1f}), ms_render {st.ms_render:.1f}", flush=True)
