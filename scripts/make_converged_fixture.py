"""Converged reference images for the stochastic-config RMSE gates (SURVEY 8c iv; north_star: "Stochastic configs are
checked by RMSE against a converged reference image"): the CPU oracle in float64 / best-first mode (the reference's
arithmetic, its insertion octree, every shadow ray traced), seed 0xC0FFEE — a seed the device tests never render with.

    python scripts/make_converged_fixture.py c1     config 1 at 160x95, 4096 spp -> tests/golden/c1_converged_160x95.npy
    python scripts/make_converged_fixture.py c3     config 3 (the headline scene, all 871,210 shapes) at 96x54, 1024 spp
                                                    -> tests/golden/c3_converged_96x54.npy
    python scripts/make_converged_fixture.py c5     config 5 (953,136 shapes, glass + Metal floor) at 96x54, 1024 spp
                                                    -> tests/golden/c5_converged_96x54.npy
Output: float32 mean radiance per pixel (182 KB / 62 KB)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-go_amd")
scenes = importlib.import_module("raytracer-go_amd.scenes")
from oracle import bridge
which = sys.argv[1] if len(sys.argv) > 1 else "c1"
gold = os.path.join(ROOT, "tests", "golden")
if which == "c1":
    spec, scene = rt.ExampleRegression(160, 95, gold)
    spec.AntiAliasingFactor = 4096
    name = "c1_converged_160x95.npy"
elif which == "c3":
    spec, scene = scenes.config3(96, 54, 1024, 16, asset_dir=gold)
    name = "c3_converged_96x54.npy"
elif which == "c5":
    spec, scene = scenes.config5(96, 54, 1024, 16, asset_dir=gold)
    name = "c5_converged_96x54.npy"
else:
    raise SystemExit("c1 | c3 | c5")
t = time.time()
with bridge.OracleScene(scene) as osc:
    acc, _, ctr = osc.render(spec, seed=0xC0FFEE, mode=bridge.F64_DIJKSTRA, rgba=False)
np.save(os.path.join(gold, name), (acc / spec.AntiAliasingFactor).astype(np.float32))
print(name, "done in", time.time() - t, "s;", ctr["rays_primary"] + ctr["rays_continuation"] + ctr["rays_shadow"], "rays")
