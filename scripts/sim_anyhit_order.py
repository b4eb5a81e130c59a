"""Design-time: the order in which a shadow ray's walk enters the children it meets (scripts/sim_groupwalk.cpp
sim_anyhit_order; CPU only, the oracle's fp32 mirror of the device tree)."""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
so = "/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", "-w", "-o", so, os.path.join(ROOT, "scripts", "sim_groupwalk.cpp")])
rt = importlib.import_module("raytracer-go_amd"); scenes = importlib.import_module("raytracer-go_amd.scenes"); abi = rt._abi
L = C.CDLL(so)
L.orc_scene_create.restype = C.c_void_p; L.orc_scene_create.argtypes = [C.c_void_p]
L.orc_set_leaf_size.argtypes = [C.c_void_p, C.c_int]; L.orc_set_tree.argtypes = [C.c_void_p, C.c_int]
L.sim_anyhit_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
spec, scene = scenes.config3(200, 112, 1, 16)
flat = abi.FlatScene(scene)
h = L.orc_scene_create(C.addressof(flat.desc)); L.orc_set_tree(h, 1); L.orc_set_leaf_size(h, 6)
s = abi.image_spec(spec)
names = ["octant order (device)", "longest chord first", "nearest entry first", "most shapes first", "smallest box first"]
for mode, name in enumerate(names):
    o = np.zeros(8)
    L.sim_anyhit_order(h, C.addressof(s), 0x5EED, mode, o.ctypes.data)
    print(f"{name:24s}: {o[0]:.0f} rays, {o[1] / o[0]:.3f} occluded: {o[2] / o[1]:.2f} visits + {o[3] / o[1]:.2f} shape tests; "
          f"others {o[4] / (o[0] - o[1]):.2f} + {o[5] / (o[0] - o[1]):.2f}; all {(o[2] + o[4]) / o[0]:.2f} + {(o[3] + o[5]) / o[0]:.2f}", flush=True)
