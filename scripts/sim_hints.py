"""Drives sim_hints() of scripts/sim_groupwalk.cpp: occluder hints carried from one repetition of a pair to the next."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", "-w", "-o", so,
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
L.sim_hints.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
W, H = int(sys.argv[1]), int(sys.argv[2])
cfg = sys.argv[3] if len(sys.argv) > 3 else "3"
gold = os.path.join(ROOT, "tests", "golden")
spec, scene = scenes.config3(W, H, 1, 16, asset_dir=gold) if cfg == "3" else scenes.config5(W, H, 1, 16, asset_dir=gold)
flat = abi.FlatScene(scene)
h = L.orc_scene_create(C.addressof(flat.desc))
L.orc_set_tree(h, 1)
L.orc_set_leaf_size(h, 6)
s = abi.image_spec(spec)
for up in (0, 1, 2, 3):
    out = np.zeros(8)
    rc = L.sim_hints(h, C.addressof(s), 0x5EED, up, out.ctypes.data)
    if rc:
        print("error", L.orc_last_error())
        break
    r = out[0]
    today = out[1] * 190 + out[2] * 93
    new = out[3] * 190 + out[4] * 93
    print(f"up={up}: rays {r:.0f}; today visits {out[1]/r:.2f} shapes {out[2]/r:.2f}; hinted visits {out[3]/r:.2f} shapes {out[4]/r:.2f}; "
          f"rays with hint {out[5]/r:.3f}, hint held a hit {out[6]/max(1,out[5]):.3f}; model cost new/today {new/today:.3f}")
