"""Design-time: a slab (two parallel planes) around every leaf's shapes as a second bound for the shadow-ray walk
(scripts/sim_groupwalk.cpp sim_leaf_slab; CPU only, the oracle's fp32 mirror of the device tree)."""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
so = "/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", "-w", "-o", so, os.path.join(ROOT, "scripts", "sim_groupwalk.cpp")])
rt = importlib.import_module("raytracer-go_amd"); scenes = importlib.import_module("raytracer-go_amd.scenes"); abi = rt._abi
L = C.CDLL(so)
L.orc_scene_create.restype = C.c_void_p; L.orc_scene_create.argtypes = [C.c_void_p]
L.orc_set_leaf_size.argtypes = [C.c_void_p, C.c_int]; L.orc_set_tree.argtypes = [C.c_void_p, C.c_int]
L.sim_leaf_slab.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spec, scene = (scenes.config3 if cfg == 3 else scenes.config5)(200, 112, 1, 16)
flat = abi.FlatScene(scene)
h = L.orc_scene_create(C.addressof(flat.desc)); L.orc_set_tree(h, 1); L.orc_set_leaf_size(h, int(os.environ.get('SIM_LEAF', '20')))
s = abi.image_spec(spec)
for mode, name in enumerate(["mean normal", "best of 10 diagonals", "mean normal, inner too", "mean normal + narrowest in-plane direction"]):
    o = np.zeros(16)
    L.sim_leaf_slab(h, C.addressof(s), 0x5EED, mode, o.ctypes.data)
    for cls, b in (("occluded", 0), ("others", 6)):
        n = max(o[b], 1)
        print(f"{name:22s} {cls:9s}: {o[b]:.0f} rays, {o[b + 1] / n:.2f} visits, {o[b + 2] / n:.2f} leaf boxes met, {o[b + 3] / n:.2f} slabs met, "
              f"shape tests {o[b + 4] / n:.2f} -> {o[b + 5] / n:.2f}", flush=True)
    print("answers changed:", o[12])
L.sim_ball_beam_slab.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]
o = np.zeros(16)
hist = np.zeros(192)
L.sim_ball_beam_slab(h, C.addressof(s), 0x5EED, 0.25 * 1.0001 + 1e-5, o.ctypes.data, hist.ctypes.data)
print(f"pair beams: {o[0]:.0f} pairs, {o[1]:.0f} rays ({o[12] / o[1]:.3f} occluded, {o[11] / o[1]:.3f} in fully dark pairs); clear by boxes {o[3] / o[1]:.3f} of the rays, "
      f"with leaf slabs {o[5] / o[1]:.3f}; still not clear though unoccluded {o[10] / o[1]:.3f}; beam visits per pair {o[7] / o[0]:.2f} -> {o[8] / o[0]:.2f}, "
      f"leaf boxes met per pair in the slab walk {o[9] / o[0]:.2f}; clear pairs with an occluded ray: {o[6]:.0f}")
print("beam walks with slabs, by visits: clear pairs | others | rays of clear pairs (cumulative share of all rays)")
cum = 0
for b in range(64):
    cum += hist[128 + b]
    if hist[b] or hist[64 + b]:
        print(f"  {b:2d}: {hist[b]:7.0f} {hist[64 + b]:7.0f}  {cum / o[1]:.3f}")
