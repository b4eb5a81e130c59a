"""Design-time: leaves of up to 31 shapes cut into groups of G consecutive shapes with a box and two slabs of their own
(scripts/sim_groupwalk.cpp sim_sub_leaf; CPU only).  python scripts/sim_sub_leaf.py [leaf size] [G]"""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
so = "/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", "-w", "-o", so, os.path.join(ROOT, "scripts", "sim_groupwalk.cpp")])
rt = importlib.import_module("raytracer-go_amd"); scenes = importlib.import_module("raytracer-go_amd.scenes"); abi = rt._abi
L = C.CDLL(so)
L.orc_scene_create.restype = C.c_void_p; L.orc_scene_create.argtypes = [C.c_void_p]
L.orc_set_leaf_size.argtypes = [C.c_void_p, C.c_int]; L.orc_set_tree.argtypes = [C.c_void_p, C.c_int]
L.sim_sub_leaf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
spec, scene = scenes.config3(200, 112, 1, 16)
flat = abi.FlatScene(scene)
s = abi.image_spec(spec)
for leaf in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,31").split(",")]:
    h = L.orc_scene_create(C.addressof(flat.desc)); L.orc_set_tree(h, 1); L.orc_set_leaf_size(h, leaf)
    for G in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,6,8").split(",")]:
        o = np.zeros(20)
        L.sim_sub_leaf(h, C.addressof(s), 0x5EED, G, o.ctypes.data)
        for cls, b in (("occluded", 0), ("others", 8)):
            n = max(o[b], 1)
            print(f"leaf {leaf:2d} groups of {G}: {cls:9s} {o[b + 1] / n:6.2f} visits, {o[b + 2] / n:5.2f} leaf boxes met, {o[b + 3] / n:5.2f} leaves met, "
                  f"{o[b + 4] / n:5.2f} groups tested, {o[b + 5] / n:5.2f} met, shape tests {o[b + 6] / n:6.2f} -> {o[b + 7] / n:6.2f}", flush=True)
        print("answers changed:", o[16])
