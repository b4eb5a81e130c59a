"""ctypes binding of librtgo_hip.so (include/rtgo.h).  The HIP library is the only compute
path: if it is missing, or no gfx950 device is visible, every call raises — there is no CPU
fallback in the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# RTGO_LIB: another build of the same library (A/B runs of compile-time knobs: scripts/sweep_variants.sh)
LIB_PATH = os.environ.get("RTGO_LIB") or os.path.join(_CSRC, "librtgo_hip.so")

# every extern "C" symbol include/rtgo.h declares
SYMBOLS = ("rt_last_error", "rt_init", "rt_scene_create", "rt_scene_destroy", "rt_scene_get_info",
           "rt_render", "rt_render_accum_device", "rt_render_accum", "rt_resolve_device",
           "rt_generate_image", "rt_trace_rays", "rt_camera_rays", "rt_scatter_probe",
           "rt_render_accum_device_static", "rt_multi_create", "rt_multi_destroy", "rt_multi_render", "rt_generate_image_multi",
           "rt_partition_share", "rt_multi_get_info", "rt_build_info")


class RtgoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtgo error {code}: {message}")
        self.code = code


def build(force=False):
    """Compile librtgo_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("rtgo_hip.hip", "device_math.hpp", "octree.cpp", "octree.hpp")]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "rtgo.h"))
    libs = [LIB_PATH, os.path.join(_CSRC, "librtgo_spill.so")]  # the second is a test build (tiny LDS stack)
    stale = any(not os.path.exists(l) or any(os.path.getmtime(s) > os.path.getmtime(l) for s in srcs) for l in libs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtgoError(_abi.RT_ERR_DEVICE, f"{LIB_PATH} is not built (run __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_init.argtypes = [C.c_int]
        L.rt_scene_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_scene_destroy.restype = None
        L.rt_scene_get_info.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_render_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_render_accum_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_render_accum_device_static.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_resolve_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rt_generate_image.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.rt_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p]
        L.rt_partition_share.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_multi_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.rt_multi_destroy.argtypes = [C.c_void_p]
        L.rt_multi_destroy.restype = None
        L.rt_multi_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.rt_generate_image_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.rt_scatter_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rt_multi_get_info.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_build_info.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != _abi.RT_OK:
        raise RtgoError(rc, lib().rt_last_error().decode(errors="replace"))


def init(device=0):
    _check(lib().rt_init(device))


def build_info():
    """The loaded library's source hash and kernel knobs (rt_build_info) as a dict of strings."""
    return dict(kv.split("=", 1) for kv in lib().rt_build_info().decode().split())


class DeviceScene:
    """A Scene resident in HBM: flattened tree + shape / material / texture / light tables."""

    def __init__(self, scene, device=0):
        self._h = C.c_void_p()
        _check(lib().rt_init(device))  # first: the runtime's start-up then runs under the flattening below (rtgo.h)
        self._flat = _abi.FlatScene(scene)
        _check(lib().rt_scene_create(C.addressof(self._flat.desc), device, C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = _abi.RtSceneInfo()
        _check(lib().rt_scene_get_info(self._h, C.addressof(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def render(self, spec, seed=0x5EED, part=None):
        """GenerateImage body -> (HxWx4 uint8, RtStats)."""
        s = _abi.image_spec(spec)
        p = part if part is not None else _abi.partition(seed=seed)
        out = np.zeros((spec.Height, spec.Width, 4), dtype=np.uint8)
        st = _abi.RtStats()
        _check(lib().rt_render(self._h, C.addressof(s), C.addressof(p), out.ctypes.data, C.addressof(st)))
        return out, st

    def render_accum(self, spec, seed=0x5EED, part=None):
        """Un-normalised fp32 RGB sums -> (HxWx3 float32, RtStats)."""
        s = _abi.image_spec(spec)
        p = part if part is not None else _abi.partition(seed=seed)
        out = np.zeros((spec.Height, spec.Width, 3), dtype=np.float32)
        st = _abi.RtStats()
        _check(lib().rt_render_accum(self._h, C.addressof(s), C.addressof(p), out.ctypes.data, C.addressof(st)))
        return out, st

    def render_accum_device(self, spec, accum_ptr, stream=None, seed=0x5EED, part=None, want_stats=True):
        """Adds this rank's share into a caller-owned device buffer (e.g. tensor.data_ptr())."""
        s = _abi.image_spec(spec)
        p = part if part is not None else _abi.partition(seed=seed)
        st = _abi.RtStats()
        _check(lib().rt_render_accum_device(self._h, C.addressof(s), C.addressof(p), C.c_void_p(accum_ptr),
                                            C.c_void_p(stream or 0), C.addressof(st) if want_stats else None))
        return st

    def render_accum_device_static(self, spec, accum_ptr, stream=None, seed=0x5EED, part=None, want_stats=True):
        """render_accum_device through the entry point that allocates nothing (static 72 MiB workspace per GPU)."""
        s = _abi.image_spec(spec)
        p = part if part is not None else _abi.partition(seed=seed)
        st = _abi.RtStats()
        _check(lib().rt_render_accum_device_static(self._h, C.addressof(s), C.addressof(p), C.c_void_p(accum_ptr),
                                                   C.c_void_p(stream or 0), C.addressof(st) if want_stats else None))
        return st

    def trace_rays(self, rays, any_hit=False, uniform_tmin=False):
        """rays (n, 8) float32 {o, d, tMin, tMax} -> (shape index or -1 [any_hit: 0/1], t).
        uniform_tmin (with any_hit): every ray uses ray 0's tMin — the walk as the render path runs it."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        _check(lib().rt_trace_rays(self._h, rays.ctypes.data, n, (2 if uniform_tmin else 1) if any_hit else 0, prim.ctypes.data, t.ctypes.data))
        return prim, t

    def scatter_probe(self, spec, rays6, keys, seed=0x5EED):
        """material.scatter of the nearest hit of each ray: rays6 (n, 6) {o, d}, keys (n, 3) {pixel, sample, bounce}
        -> (n, 20) float32 {shouldTrace, attenuation, scattered o, scattered d, colour, t, p, normal}."""
        s = _abi.image_spec(spec)
        rays6 = np.ascontiguousarray(rays6, dtype=np.float32)
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        out = np.zeros((rays6.shape[0], 20), dtype=np.float32)
        _check(lib().rt_scatter_probe(self._h, C.addressof(s), seed, rays6.ctypes.data, keys.ctypes.data, rays6.shape[0], out.ctypes.data))
        return out

    def camera_rays(self, spec, pixels, seed=0x5EED):
        s = _abi.image_spec(spec)
        pixels = np.ascontiguousarray(pixels, dtype=np.int32)
        out = np.empty((pixels.shape[0], 6), dtype=np.float32)
        _check(lib().rt_camera_rays(self._h, C.addressof(s), seed, pixels.ctypes.data, pixels.shape[0], out.ctypes.data))
        return out


def partition_share(spec, part):
    """What one rank of a partition renders: (sample_begin, sample_end, n_tiles, tile_size).  Needs no device."""
    s = _abi.image_spec(spec)
    out = _abi.RtShare()
    _check(lib().rt_partition_share(C.addressof(s), C.addressof(part), C.addressof(out)))
    return out.sample_begin, out.sample_end, out.n_tiles, out.tile_size


class MultiDeviceScene:
    """A Scene replicated on several GPUs of one node (rt_multi): render() is one blocking call that splits the
    frame over the devices and reduces the shares with RCCL onto the first one."""

    def __init__(self, scene, devices):
        self._h = C.c_void_p()
        for d in devices:  # first: the runtime's start-up then runs under the flattening below (rtgo.h)
            _check(lib().rt_init(d))
        self._flat = _abi.FlatScene(scene)
        self.devices = (C.c_int32 * len(devices))(*devices)
        _check(lib().rt_multi_create(C.addressof(self._flat.desc), self.devices, len(devices), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = _abi.RtMultiInfo()
        _check(lib().rt_multi_get_info(self._h, C.addressof(i)))
        return {n: getattr(i, n) for n, _ in i._fields_ if not n.startswith("_")}

    def render(self, spec, mode=_abi.RT_PART_TILES, seed=0x5EED, out=None):
        s = _abi.image_spec(spec)
        if out is None:
            out = np.zeros((spec.Height, spec.Width, 4), dtype=np.uint8)
        st = _abi.RtStats()
        _check(lib().rt_multi_render(self._h, C.addressof(s), mode, seed, out.ctypes.data, C.addressof(st)))
        return out, st


def resolve_device(accum_ptr, width, height, spp, rgba_ptr, stream=None):
    _check(lib().rt_resolve_device(C.c_void_p(accum_ptr), width, height, spp, C.c_void_p(rgba_ptr), C.c_void_p(stream or 0)))
