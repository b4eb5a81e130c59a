"""Quick GPU sanity run (developer script): C1 parity against the oracle + timing."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-go_amd")
backend = importlib.import_module("raytracer-go_amd.backend")
from oracle import bridge

spec, scene = rt.ExampleRegression(640, 380, os.path.join(ROOT, "tests/golden"))
dev = backend.DeviceScene(scene)
print("info", dev.info(), flush=True)
osc = bridge.OracleScene(scene)

# camera rays
rng = np.random.default_rng(1)
n = 20000
px = np.stack([rng.integers(0, 640, n), rng.integers(0, 380, n), rng.integers(0, 32, n)], 1).astype(np.int32)
g = dev.camera_rays(spec, px)
o = osc.camera_rays(spec, px, mode=bridge.F32_STACK).astype(np.float32)
print("camera rays bit-exact:", np.array_equal(g.view(np.uint32), o.view(np.uint32)), np.abs(g - o).max(), flush=True)

rays = np.zeros((n, 8), np.float32); rays[:, :6] = g
gp, gt = dev.trace_rays(rays)
op, ot = osc.trace_rays(rays.astype(np.float64), mode=bridge.F32_STACK)
hit = op >= 0
print("primary IDs equal:", np.array_equal(gp, op), "t equal on hits:", np.array_equal(gt[hit], ot[hit].astype(np.float32)),
      "hits", hit.mean(), flush=True)
op64, _ = osc.trace_rays(rays.astype(np.float64), mode=bridge.F64_DIJKSTRA)
print("vs f64 best-first mismatches:", int((gp != op64).sum()), "of", n, flush=True)

t = time.time()
pix, st = dev.render(spec)
print("render wall", time.time() - t, st.as_dict(), flush=True)
t = time.time()
pix, st = dev.render(spec)
print("render wall (2nd)", time.time() - t, "Mrays/s", st.rays / st.ms_render / 1e3, flush=True)
from PIL import Image
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
Image.fromarray(pix).save(os.path.join(ROOT, "gpurun_out/c1_gpu.png"))
gold = np.asarray(Image.open(os.path.join(ROOT, "tests/golden/samples_images/code_example.png")).convert("RGBA")).astype(int)
d = np.abs(pix.astype(int) - gold)[..., :3].max(-1)
print("vs golden: px>20:", int((d > 20).sum()), "limit", int(d.size * 0.01), flush=True)
spec.AntiAliasingFactor = 4
acc_g, st = dev.render_accum(spec)
acc_o, _, ctr = osc.render(spec, mode=bridge.F32_STACK, rgba=False)
dd = np.abs(acc_g.astype(np.float64) - acc_o).max(-1)
print("accum vs oracle f32 (4spp): max", dd.max(), "n>1e-3", int((dd > 1e-3).sum()), "n>1e-5", int((dd > 1e-5).sum()), flush=True)
print("gpu stats", st.as_dict()); print("orc ctr", ctr, flush=True)
