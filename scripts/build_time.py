"""Host-side scene set-up time (tree build + flatten + upload) of config 3 for a few builder thread counts.
Run on the GPU box: python scripts/build_time.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scenes = importlib.import_module("raytracer-go_amd.scenes")
backend = importlib.import_module("raytracer-go_amd.backend")

spec, scene = scenes.config3()
backend.init(0)
for threads in ("1", "4", "8", "16", ""):
    if threads:
        os.environ["RTGO_BUILD_THREADS"] = threads
    else:
        os.environ.pop("RTGO_BUILD_THREADS", None)
    t0 = time.time()
    ds = backend.DeviceScene(scene)
    dt = time.time() - t0
    spec.Width, spec.Height, spec.AntiAliasingSamples = 64, 36, 1
    _, st = ds.render(spec)
    print(f"threads {threads or 'default':>7}: DeviceScene {dt * 1e3:7.1f} ms, of which tree build + upload {st.ms_build:7.1f} ms", flush=True)
    ds.close()
