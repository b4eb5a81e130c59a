"""Design-time: how many of the shadow rays that the pair beams leave over would beams of 2 / 4 / 8 repetitions settle?
(scripts/sim_groupwalk.cpp sim_sub_beam; CPU only, the oracle's fp32 mirror of the device tree.)"""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
so = "/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", "-w", "-o", so, os.path.join(ROOT, "scripts", "sim_groupwalk.cpp")])
rt = importlib.import_module("raytracer-go_amd"); scenes = importlib.import_module("raytracer-go_amd.scenes"); abi = rt._abi
L = C.CDLL(so)
L.orc_scene_create.restype = C.c_void_p; L.orc_scene_create.argtypes = [C.c_void_p]
L.orc_set_leaf_size.argtypes = [C.c_void_p, C.c_int]; L.orc_set_tree.argtypes = [C.c_void_p, C.c_int]
L.sim_sub_beam.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_int, C.c_void_p]
spec, scene = scenes.config3(200, 112, 1, 16)
flat = abi.FlatScene(scene)
h = L.orc_scene_create(C.addressof(flat.desc)); L.orc_set_tree(h, 1); L.orc_set_leaf_size(h, 6)
s = abi.image_spec(spec)
for group in (8, 4, 2, 1):
    o = np.zeros(16)
    L.sim_sub_beam(h, C.addressof(s), 0x5EED, 0.25 * 1.0001 + 1e-5, group, o.ctypes.data)
    print(f"group {group}: non-clear pairs {o[0]:.0f} with {o[1]:.0f} rays ({o[9] / o[1]:.2f} occluded, {o[11] / o[1]:.2f} in all-dark pairs); "
          f"{o[2]:.0f} sub-beams, {o[3] / o[2]:.3f} clear -> {o[4] / o[1]:.3f} of the rays settled, "
          f"beam visits {o[5] / o[2]:.1f} each ({o[10] / max(1, o[11] / group):.1f} in dark pairs); rays: {o[7] / o[1]:.1f} visits + {o[13] / o[1]:.1f} shape tests on average, "
          f"settled ones {o[6] / max(1, o[4]):.1f} + {o[12] / max(1, o[4]):.1f}; violations {o[8]:.0f}", flush=True)
