"""ctypes mirror of include/rtgo.h and the scene flattener.

The flattener is the Python twin of the type switch a Go cgo shim performs inside
GenerateImage: the reference's Shape / Material / Light / texture
interfaces are sealed (shape.go:24-30, material.go:10, light.go:9-15, texture.go:12-14),
so a switch over {Sphere, TrianglePlane} x {4 materials} x {3 lights} x {2 textures, nil}
is total.
"""
import ctypes as C

import numpy as np

RT_OK = 0
RT_ERR_INVALID, RT_ERR_DEVICE, RT_ERR_ALGORITHM, RT_ERR_LIMIT = -1, -2, -3, -4
RT_TEX_CHECKERS, RT_TEX_IMAGE = 1, 2
RT_MAT_STANDARD, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_PHONGBLINN = 1, 2, 3, 4
RT_SHAPE_SPHERE, RT_SHAPE_TRIANGLE = 1, 2
RT_LIGHT_AMBIENT, RT_LIGHT_POINT, RT_LIGHT_SPOT = 1, 2, 3
RT_PART_ALL, RT_PART_TILES, RT_PART_SAMPLES = 0, 1, 2


class RtVec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("color1", RtVec3), ("color2", RtVec3),
                ("checkers_width", C.c_double), ("checkers_height", C.c_double),
                ("rgba", C.c_void_p)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("color", RtVec3),
                ("specular_color", RtVec3), ("param", C.c_double)]


class RtShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("single_sided", C.c_int32),
                ("_pad", C.c_int32), ("a", RtVec3), ("b", RtVec3), ("c", RtVec3)]


class RtLight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("color", RtVec3), ("position", RtVec3),
                ("look_at", RtVec3), ("intensity", C.c_double), ("specular_intensity", C.c_double),
                ("angle_deg", C.c_double), ("decay", C.c_double)]


class RtSceneDesc(C.Structure):
    _fields_ = [("camera_look_from", RtVec3), ("camera_look_at", RtVec3), ("camera_up", RtVec3),
                ("camera_aperture", C.c_double), ("camera_fov_deg", C.c_double),
                ("shapes", C.c_void_p), ("n_shapes", C.c_int64),
                ("materials", C.c_void_p), ("n_materials", C.c_int64),
                ("textures", C.c_void_p), ("n_textures", C.c_int64),
                ("lights", C.c_void_p), ("n_lights", C.c_int64)]


class RtImageSpec(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("anti_aliasing_factor", C.c_int32),
                ("ray_tracing_max_depth", C.c_int32),
                ("soft_shadow_monte_carlo_repetitions", C.c_int32), ("worker_count", C.c_int32),
                ("bvh_traversal_algorithm", C.c_int32), ("_pad", C.c_int32)]


class RtPartition(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("mode", C.c_int32), ("rank", C.c_int32),
                ("world_size", C.c_int32), ("tile_size", C.c_int32),
                ("max_paths_per_batch", C.c_int64)]


class RtStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_continuation", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("rays_shadow_reference_equivalent", C.c_uint64),
                ("phong_hits", C.c_uint64), ("paths", C.c_uint64),
                ("ms_build", C.c_double), ("ms_render", C.c_double),
                ("ms_extend", C.c_double), ("ms_shadow", C.c_double), ("ms_shade", C.c_double),
                ("rays_shadow_beam", C.c_uint64), ("ms_shadow_busy", C.c_double), ("beam_walks", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}

    @property
    def rays(self):
        return self.rays_primary + self.rays_continuation + self.rays_shadow


class RtShare(C.Structure):
    _fields_ = [("sample_begin", C.c_int32), ("sample_end", C.c_int32), ("tile_size", C.c_int32), ("_pad", C.c_int32),
                ("n_tiles", C.c_int64)]


class RtMultiInfo(C.Structure):
    _fields_ = [("n_shares", C.c_int32), ("n_distinct_devices", C.c_int32), ("rccl_ranks", C.c_int32), ("_pad", C.c_int32)]


class RtSceneInfo(C.Structure):
    _fields_ = [("n_shapes", C.c_int64), ("n_inner_nodes", C.c_int64),
                ("n_child_records", C.c_int64), ("tree_depth", C.c_int32),
                ("stack_bound", C.c_int32), ("device_bytes", C.c_int64)]


SHAPE_DTYPE = np.dtype(RtShape)
assert SHAPE_DTYPE.itemsize == 88 and C.sizeof(RtTexture) == 88 and C.sizeof(RtLight) == 112
assert C.sizeof(RtMaterial) == 64


def _v(dst, src):
    dst.x, dst.y, dst.z = float(src.X), float(src.Y), float(src.Z)


class FlatScene:
    """rt_scene_desc plus the buffers that keep its pointers alive."""

    def __init__(self, scene):
        from . import api  # local import: api imports this module

        textures, tex_keep, tex_index = [], [], {}
        materials, mat_index = [], {}

        def texture_id(t):
            if t is None:
                return -1
            key = id(t)
            if key in tex_index:
                return tex_index[key]
            rt = RtTexture()
            if isinstance(t, api.CheckersTexture):
                rt.kind = RT_TEX_CHECKERS
                _v(rt.color1, t.ColorFrac1)
                _v(rt.color2, t.ColorFrac2)
                rt.checkers_width, rt.checkers_height = float(t.CheckersWidth), float(t.CheckersHeight)
            elif isinstance(t, api.ImageTexture):
                rt.kind = RT_TEX_IMAGE
                pix = np.ascontiguousarray(t.Img.Pix, dtype=np.uint8)
                if pix.ndim != 3 or pix.shape[2] != 4:
                    raise TypeError("ImageTexture.Img.Pix must be HxWx4 uint8")
                rt.height, rt.width = int(pix.shape[0]), int(pix.shape[1])
                rt.rgba = pix.ctypes.data
                tex_keep.append(pix)
            else:
                raise TypeError(f"unknown texture type {type(t).__name__}")
            tex_index[key] = len(textures)
            tex_keep.append(t)
            textures.append(rt)
            return tex_index[key]

        def material_id(m):
            key = id(m)
            if key in mat_index:
                return mat_index[key]
            rm = RtMaterial()
            rm.texture = -1
            if isinstance(m, api.Standard):
                rm.kind = RT_MAT_STANDARD
                _v(rm.color, m.ColorFrac)
                rm.texture = texture_id(m.Texture)
            elif isinstance(m, api.Metal):
                rm.kind = RT_MAT_METAL
                _v(rm.color, m.Albedo)
                rm.param = float(m.Fuzz)
            elif isinstance(m, api.Dielectric):
                rm.kind = RT_MAT_DIELECTRIC
                rm.param = float(m.RefractiveIndex)
            elif isinstance(m, api.PhongBlinn):
                rm.kind = RT_MAT_PHONGBLINN
                _v(rm.color, m.ColorFrac)
                _v(rm.specular_color, m.SpecularColorFrac)
                rm.param = float(m.SpecHardness)
                rm.texture = texture_id(m.Texture)
            else:
                raise TypeError(f"unknown material type {type(m).__name__}")
            mat_index[key] = len(materials)
            tex_keep.append(m)
            materials.append(rm)
            return mat_index[key]

        # one table for all shapes: meshes write their records straight into their slice of it
        total = sum(len(s) if isinstance(s, api.TriangleMesh) else 1 for s in scene.Shapes)
        table = np.zeros(total, dtype=SHAPE_DTYPE)
        at = 0
        for s in scene.Shapes:
            if isinstance(s, api.TriangleMesh):
                s.to_records(material_id(s.Mat), out=table[at:at + len(s)])
                at += len(s)
                continue
            r = table[at]
            if isinstance(s, api.Sphere):
                r["kind"] = RT_SHAPE_SPHERE
                r["a"] = (s.Center.X, s.Center.Y, s.Center.Z)
                r["b"] = (s.Radius, 0.0, 0.0)
            elif isinstance(s, api.TrianglePlane):
                r["kind"] = RT_SHAPE_TRIANGLE
                r["single_sided"] = 1 if s.SingleSided else 0
                r["a"] = (s.PointA.X, s.PointA.Y, s.PointA.Z)
                r["b"] = (s.PointB.X, s.PointB.Y, s.PointB.Z)
                r["c"] = (s.PointC.X, s.PointC.Y, s.PointC.Z)
            else:
                raise TypeError(f"unknown shape type {type(s).__name__}")
            r["material"] = material_id(s.Mat)
            at += 1
        self.shapes = table

        lights = []
        for l in scene.Lights:
            rl = RtLight()
            _v(rl.color, l.ColorFrac)
            rl.intensity = float(l.LightIntensity)
            if isinstance(l, api.AmbientLight):
                rl.kind = RT_LIGHT_AMBIENT
            elif isinstance(l, api.PointLight):
                rl.kind = RT_LIGHT_POINT
            elif isinstance(l, api.SpotLight):
                rl.kind = RT_LIGHT_SPOT
                _v(rl.look_at, l.LookAt)
                rl.angle_deg = float(l.Angle)
            else:
                raise TypeError(f"unknown light type {type(l).__name__}")
            if rl.kind != RT_LIGHT_AMBIENT:
                _v(rl.position, l.Position)
                rl.specular_intensity = float(l.SpecularLightIntensity)
                rl.decay = float(l.InverseSquareLawDecayFactor)
            lights.append(rl)

        self._materials = (RtMaterial * max(1, len(materials)))(*materials)
        self._textures = (RtTexture * max(1, len(textures)))(*textures)
        self._lights = (RtLight * max(1, len(lights)))(*lights)
        self._keep = tex_keep

        d = RtSceneDesc()
        _v(d.camera_look_from, scene.CameraLookFrom)
        _v(d.camera_look_at, scene.CameraLookAt)
        _v(d.camera_up, scene.CameraUp)
        d.camera_aperture = float(scene.CameraAperature)
        d.camera_fov_deg = float(scene.CameraFov)
        d.shapes, d.n_shapes = self.shapes.ctypes.data, len(self.shapes)
        d.materials, d.n_materials = C.addressof(self._materials), len(materials)
        d.textures, d.n_textures = C.addressof(self._textures), len(textures)
        d.lights, d.n_lights = C.addressof(self._lights), len(lights)
        self.desc = d


def image_spec(spec):
    s = RtImageSpec()
    s.width, s.height = int(spec.Width), int(spec.Height)
    s.anti_aliasing_factor = int(spec.AntiAliasingFactor)
    s.ray_tracing_max_depth = int(spec.RayTracingMaxDepth)
    s.soft_shadow_monte_carlo_repetitions = int(spec.SoftShadowMonteCarloRepetitions)
    s.worker_count = int(spec.WorkerCount)
    s.bvh_traversal_algorithm = int(spec.BvhTraversalAlgorithm)
    return s


def partition(seed=0x5EED, mode=RT_PART_ALL, rank=0, world_size=1, tile_size=0, max_paths=0):
    p = RtPartition()
    p.seed, p.mode, p.rank, p.world_size = int(seed), int(mode), int(rank), int(world_size)
    p.tile_size, p.max_paths_per_batch = int(tile_size), int(max_paths)
    return p
