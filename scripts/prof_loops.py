"""Lane-utilisation of the shadow kernel's loops, from the instrumented build (make -C raytracer-go_amd/csrc prof).
Run on the GPU box: python scripts/prof_loops.py [rows]"""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
backend = importlib.import_module("raytracer-go_amd.backend")
backend.LIB_PATH = os.path.join(ROOT, "raytracer-go_amd/csrc/librtgo_prof.so")
scenes = importlib.import_module("raytracer-go_amd.scenes")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
spec, scene = scenes.config3(1920, rows, 64 if rows < 1080 else 16, 16, asset_dir=os.path.join(ROOT, "tests/golden"))
dev = backend.DeviceScene(scene)
out = (ctypes.c_ulonglong * (12 + 129 + 64 + 4))()
L = backend.lib()
acc, st = dev.render_accum(spec)
L.rt_debug_prof(out, 0)
v = list(out)
# points 0/1 also count the primary rays' traversal (k_extend): 3 % of the rays at config 3
rounds = v[4]
print("shadow rays", st.rays_shadow, "ms_shadow %.1f" % st.ms_shadow, "wave rounds", rounds, "lanes/round %.1f" % (v[5] / max(1, rounds)))
for name, i in (("block visits", 0), ("shape tests", 1), ("leaf slabs", 3)):
    if not v[2 * i]:
        continue
    print("%-12s: %.1f per wave round, %.1f per ray, lane utilisation %.2f" %
          (name, v[2 * i] / rounds, v[2 * i + 1] / st.rays_shadow, v[2 * i + 1] / (64.0 * v[2 * i])))
it = v[8]
if it:
    print("shadow-ray walk: %.1f iterations per wave round; per iteration %.1f lanes only hold leaves, %.1f lanes have nothing to do; %.2f shape loops per wave round"
          % (it / rounds, v[9] / it, v[10] / it, v[11] / rounds))
hw, hi = v[12:12 + 65], v[12 + 65:12 + 129]
tot = sum(hw)
if tot:
    cum = 0
    print("walk iterations by walking lanes (cumulative share of iterations):")
    for w in range(65):
        cum += hw[w]
        if w % 4 == 0 or w == 64:
            print("  <= %2d lanes: %.3f" % (w, cum / tot))
    n = sum(hi)
    print("wave rounds by iterations:", " ".join("%d:%.3f" % (i, hi[i] / n) for i in range(64) if hi[i] / n >= 0.002))
    sw = v[12 + 129:12 + 129 + 64]
    reach = [sum(hi[i + 1:]) for i in range(64)]  # wave rounds that run iteration index i (0-based): those of more than i iterations
    print("iteration index: share of all iterations, walking lanes of the waves still going")
    total_it = sum(reach[:63]) or 1
    for i in range(0, 63, 3):
        if reach[i]:
            print("  it %2d: %.3f of iterations, %.1f lanes walking" % (i, reach[i] / total_it, sw[i] / reach[i]))
f = v[12 + 129 + 64:12 + 129 + 68]
if f[0]:
    print("two consecutive rounds of a wavefront fused (a lane starts its second ray when its first is done): %.1f iterations now, "
          "%.1f fused (max over lanes of the two rays' visits), %.1f at best (perfect balance)" % (f[1] / f[0], f[2] / f[0], f[3] / f[0]))
