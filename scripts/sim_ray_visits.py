"""Design-time (CPU): block visits of every shadow ray of config 3 at 480x270x1, pair by pair, through
scripts/sim_groupwalk.cpp sim_ray_visits -> /tmp/ray_visits.npy (read by sim_lane_util.py and sim_refill.py)."""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT='/root/repo'; sys.path.insert(0,ROOT)
so="/tmp/libsim_groupwalk.so"
subprocess.check_call(["g++","-O2","-std=c++17","-fPIC","-ffp-contract=off","-pthread","-shared","-w","-o",so,os.path.join(ROOT,"scripts","sim_groupwalk.cpp")])
rt=importlib.import_module("raytracer-go_amd"); scenes=importlib.import_module("raytracer-go_amd.scenes"); abi=rt._abi
L=C.CDLL(so)
L.orc_scene_create.restype=C.c_void_p; L.orc_scene_create.argtypes=[C.c_void_p]
L.orc_set_leaf_size.argtypes=[C.c_void_p,C.c_int]; L.orc_set_tree.argtypes=[C.c_void_p,C.c_int]
L.sim_ray_visits.restype=C.c_int64
L.sim_ray_visits.argtypes=[C.c_void_p,C.c_void_p,C.c_uint64,C.c_void_p,C.c_int64,C.c_int]
spec,scene=scenes.config3(480,270,1,16)
flat=abi.FlatScene(scene)
h=L.orc_scene_create(C.addressof(flat.desc)); L.orc_set_tree(h,1); L.orc_set_leaf_size(h,6)
s=abi.image_spec(spec)
out=np.zeros((600000,19),np.int32)
n=L.sim_ray_visits(h,C.addressof(s),0x5EED,out.ctypes.data,out.shape[0],19)
out=out[:n]; np.save("/tmp/ray_visits.npy",out)
print("pairs",n)
