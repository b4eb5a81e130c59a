import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
backend = importlib.import_module("raytracer-go_amd.backend")
backend.LIB_PATH = os.path.join(ROOT, "raytracer-go_amd/csrc/librtgo_prof.so")
scenes = importlib.import_module("raytracer-go_amd.scenes")
import numpy as np
spec, scene = scenes.config3(1920, 270, 64, 16, asset_dir=os.path.join(ROOT, "tests/golden"))
dev = backend.DeviceScene(scene)
acc, st = dev.render_accum(spec)
out = (ctypes.c_ulonglong * 8)()
backend.lib().rt_debug_prof(out)
v = list(out)
waves = st.rays_shadow / 64
print("shadow rays", st.rays_shadow, "ms_shadow", st.ms_shadow)
print("node iters: wave-level %.1f per wave-ray, lane-level %.1f per ray, util %.2f" % (v[0] / waves, v[1] / st.rays_shadow, v[1] / (64 * v[0])))
print("leaf iters: wave-level %.1f per wave-ray, lane-level %.1f per ray, util %.2f" % (v[2] / waves, v[3] / st.rays_shadow, v[3] / (64 * v[2])))
print("prim tests: wave-level %.1f per wave-ray, lane-level %.1f per ray, util %.2f" % (v[4] / waves, v[5] / st.rays_shadow, v[5] / (64 * v[4])))
print("uniform node iters: %.1f per wave-ray (%.2f of node iters)" % (v[6] / waves, v[6] / v[0]))
