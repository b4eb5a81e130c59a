"""Converged reference image for the stochastic-config RMSE gate (SURVEY 8c iv): config 1 at 160x95,
4096 spp, CPU oracle in float64 / best-first mode (the reference's arithmetic), seed 0xC0FFEE (a
different seed from the one the device test renders with).  Output: tests/golden/c1_converged_160x95.npy
(float32 mean radiance per pixel, 182 KB)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-go_amd")
from oracle import bridge
spec, scene = rt.ExampleRegression(160, 95, os.path.join(ROOT, "tests", "golden"))
spec.AntiAliasingFactor = 4096
t = time.time()
with bridge.OracleScene(scene) as osc:
    acc, _, ctr = osc.render(spec, seed=0xC0FFEE, mode=bridge.F64_DIJKSTRA, rgba=False)
np.save(os.path.join(ROOT, "tests", "golden", "c1_converged_160x95.npy"), (acc / spec.AntiAliasingFactor).astype(np.float32))
print("done in", time.time() - t, "s;", ctr["rays_primary"] + ctr["rays_continuation"] + ctr["rays_shadow"], "rays")
