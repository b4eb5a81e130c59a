"""Drives scripts/sim_groupwalk.cpp (design-time CPU simulation of a shared shadow walk per (hit, light) pair)."""
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", "-o", so,
                       os.path.join(ROOT, "scripts", "sim_groupwalk.cpp")])
rt = importlib.import_module("raytracer-go_amd")
scenes = importlib.import_module("raytracer-go_amd.scenes")
abi = rt._abi
L = C.CDLL(so)
L.orc_scene_create.restype = C.c_void_p
L.orc_scene_create.argtypes = [C.c_void_p]
L.orc_set_leaf_size.argtypes = [C.c_void_p, C.c_int]
L.orc_set_tree.argtypes = [C.c_void_p, C.c_int]
L.orc_last_error.restype = C.c_char_p
names = ["pairs", "rays", "base_inner", "base_leaf", "grp_visits", "grp_touched", "ray_leafbox", "ray_leafpass", "ray_shape",
         "wave_base_iters", "wave_grp_visits", "wave_grp_touched", "waves", "occluded_rays", "fully_lit_pairs", "fully_dark_pairs", "empty_walk_pairs"]


class Out(C.Structure):
    _fields_ = [(n, C.c_double) for n in names]


L.sim_groupwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
import numpy as np
W, H = int(sys.argv[1]), int(sys.argv[2])
spec, scene = scenes.config3(W, H, 1, 16, asset_dir=os.path.join(ROOT, "tests", "golden"))
flat = abi.FlatScene(scene)
h = L.orc_scene_create(C.addressof(flat.desc))
L.orc_set_tree(h, 1)
L.orc_set_leaf_size(h, 6)
s = abi.image_spec(spec)
for sub, tight in ((0, 1),):
    o = Out()
    rec = np.zeros((400000, 8), np.float32)
    nrec = C.c_int64()
    rc = L.sim_groupwalk(h, C.addressof(s), 0x5EED, sub, tight, C.addressof(o), rec.ctypes.data, rec.shape[0], C.byref(nrec))
    np.save("/tmp/sim_rec.npy", rec[:nrec.value])
    if rc:
        print("error", L.orc_last_error())
        break
    d = {n: getattr(o, n) for n in names}
    r, p = d["rays"], d["pairs"]
    print(f"sub={sub} tight={tight} pairs={p:.0f} rays/pair={r/p:.2f} occluded={d['occluded_rays']/r:.3f} "
          f"lit pairs={d['fully_lit_pairs']/p:.3f} dark pairs={d['fully_dark_pairs']/p:.3f} empty-walk pairs={d['empty_walk_pairs']/p:.3f}")
    print(f"  per ray today: visits={d['base_inner']/r:.2f} shape tests={d['base_leaf']/r:.2f}; wave iters (max of 4 pairs' rays)={d['wave_base_iters']/d['waves']:.2f}")
    print(f"  beam: visits/pair={d['grp_visits']/p:.2f} touched leaves/pair={d['grp_touched']/p:.2f} per wave max visits={d['wave_grp_visits']/d['waves']:.2f} touched={d['wave_grp_touched']/d['waves']:.2f}")
    print(f"  per ray: leaf box tests={d['ray_leafbox']/r:.2f} passed={d['ray_leafpass']/r:.2f} shape tests={d['ray_shape']/r:.2f}")
