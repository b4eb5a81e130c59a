"""Frame time of the other BASELINE.json configs on one MI355X (information only; bench.py is the contract).
    python scripts/bench_configs.py            # configs 1, 2, 5 at their named sizes, config 4 at 1/8 of its samples
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
backend = importlib.import_module("raytracer-go_amd.backend")
scenes = importlib.import_module("raytracer-go_amd.scenes")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(name, spec, scene, note=""):
    with backend.DeviceScene(scene) as dev:
        dev.render_accum(spec)  # warm-up (workspace allocation)
        t0 = time.perf_counter()
        _, st = dev.render_accum(spec)
        dt = time.perf_counter() - t0
        info = dev.info()
    rays = st.rays_primary + st.rays_continuation + st.rays_shadow
    print(json.dumps({"config": name, "note": note, "width": spec.Width, "height": spec.Height, "spp": spec.AntiAliasingFactor,
                      "shapes": info["n_shapes"], "sec_per_frame": round(dt, 4), "Mrays_per_s": round(rays / dt / 1e6, 1),
                      "rays": {"primary": st.rays_primary, "continuation": st.rays_continuation, "shadow": st.rays_shadow,
                               "shadow_reference_equivalent": st.rays_shadow_reference_equivalent},
                      "ms_build": round(st.ms_build, 1)}), flush=True)


if __name__ == "__main__":
    spec, scene = scenes.config1(640, 480, GOLD)
    run("1: README scene 640x480, 32 spp, reps 16", spec, scene)
    for reps in (1, 16):
        spec, scene = scenes.config2(1920, 1080, 16, reps)
        run(f"2: Bunny stand-in 1080p 16 spp, reps {reps}", spec, scene)
    spec, scene = scenes.config3(3840, 2160, 128, 16, asset_dir=GOLD)
    run("4: Dragon stand-in 4K, one GPU's share (128 of 1024 spp)", spec, scene, note="the share of one of 8 GPUs under RT_PART_SAMPLES")
    spec, scene = scenes.config5(3840, 2160, 64, 16, asset_dir=GOLD)
    run("5: Bunny + Dragon + glass + Metal floor 4K 64 spp (whole frame on one GPU)", spec, scene)
