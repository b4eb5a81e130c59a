"""Benchmark / parity scenes of BASELINE.json's five configs.

There are no STL models in the image and no network, so the Stanford Bunny and Dragon are
replaced by SEEDED PROCEDURAL STAND-INS with matched triangle counts (a noise-displaced
icosphere of 81,920 triangles and a noise-displaced (2,3) torus-knot tube of 871,200
triangles).  If RTGO_BUNNY_STL / RTGO_DRAGON_STL name binary STL files on the machine, those
are loaded instead (stl.py), rescaled to the same bounding size.

Everything is built from the public API types of api.py, exactly as a Go user would build a
Scene (example.go:9-224 is the model).
"""
import math
import os

import numpy as np

from . import api
from .api import (AmbientLight, CheckersTexture, Dielectric, ImageSpec, ImageTexture, Metal, PhongBlinn,
                  PointLight, Scene, Sphere, SpotLight, Standard, TriangleMesh, TrianglePlane, Vec)


# ---- procedural meshes -----------------------------------------------------------------
def _sphere_noise(p, rng, octaves=6, amp=0.12):
    """Smooth seeded displacement field on the unit sphere."""
    d = np.zeros(p.shape[:-1])
    for k in range(octaves):
        f = rng.normal(size=3) * (1.5 + 1.7 * k)
        d += (amp / (1 + k)) * np.sin(p @ f + rng.uniform(0, 2 * math.pi))
    return d


def icosphere_mesh(level=6, seed=7, amp=0.12):
    """Noise-displaced icosphere: 20 * 4**level triangles (level 6 = 81,920), wound outward."""
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    tri = v[f]  # (20, 3, 3)
    for _ in range(level):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1),
                              np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)], 0)
    rng = np.random.default_rng(seed)
    r = 1.0 + _sphere_noise(tri, rng, amp=amp)
    return np.ascontiguousarray(tri * r[..., None])


def torus_knot_mesh(nu=660, nv=660, seed=11, p=2, q=3, tube=0.42, amp=0.10):
    """Noise-displaced (p,q) torus-knot tube: 2*nu*nv triangles (660x660 = 871,200), wound outward."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0, 2 * math.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * math.pi, nv, endpoint=False)

    def curve(t):
        r = 2 + np.cos(q * t)
        return np.stack([r * np.cos(p * t), r * np.sin(p * t), -np.sin(q * t)], -1)

    c = curve(u)
    h = 1e-4
    tan = curve(u + h) - curve(u - h)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    acc = curve(u + h) - 2 * c + curve(u - h)
    nrm = acc - (acc * tan).sum(1, keepdims=True) * tan
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bin_ = np.cross(tan, nrm)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    disp = np.zeros_like(uu)
    for k in range(8):  # integer frequencies: seamless in u and v
        m, n = int(rng.integers(1, 40)), int(rng.integers(0, 9))
        disp += (amp / (1 + 0.6 * k)) * np.sin(m * uu + n * vv + rng.uniform(0, 2 * math.pi))
    rad = tube * (1.0 + disp)
    pts = c[:, None, :] + rad[..., None] * (np.cos(vv)[..., None] * nrm[:, None, :] + np.sin(vv)[..., None] * bin_[:, None, :])
    p00 = pts
    p10 = np.roll(pts, -1, axis=0)
    p01 = np.roll(pts, -1, axis=1)
    p11 = np.roll(p10, -1, axis=1)
    t1 = np.stack([p00, p10, p11], 2).reshape(-1, 3, 3)
    t2 = np.stack([p00, p11, p01], 2).reshape(-1, 3, 3)
    tri = np.concatenate([t1, t2], 0)
    # make the winding outward (normal away from the tube axis) regardless of frame handedness
    ctr = np.repeat(c[:, None, :], nv, 1).reshape(-1, 3)
    ctr = np.concatenate([ctr, ctr], 0)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    flip = ((tri.mean(1) - ctr) * n).sum(1) < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return np.ascontiguousarray(tri)


def _fit(tri, center, size):
    """Uniformly rescale a triangle soup so its largest extent is `size`, centred at `center`
    in x/z and resting on y = center[1]."""
    lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    s = size / (hi - lo).max()
    out = (tri - (lo + hi) / 2) * s
    out[..., 1] += (hi[1] - lo[1]) * s / 2
    return out + np.asarray(center, dtype=np.float64)


def bunny_vertices():
    path = os.environ.get("RTGO_BUNNY_STL")
    if path:
        from . import stl
        return stl.load_binary_stl(path)
    return icosphere_mesh(6)


def dragon_vertices(nu=660, nv=660):
    path = os.environ.get("RTGO_DRAGON_STL")
    if path:
        from . import stl
        return stl.load_binary_stl(path)
    tri = torus_knot_mesh(nu, nv)
    # stand the knot up (its axis along z -> along y is not needed; tilt it a little for depth)
    m = TriangleMesh(tri)
    m.Rotate(Vec(20, 35, 0))
    return m.vertices


def _room(half=6.0, height=8.0, depth_front=-10.0, depth_back=6.0, mats=None):
    """5-face open box (floor, ceiling, back, left, right) = 10 triangles, normals facing inward
    (PhongBlinn never flips normals, material.go:98)."""
    x0, x1, y0, y1, z0, z1 = -half, half, 0.0, height, depth_front, depth_back
    floor, ceil_, back, left, right = mats
    V = Vec
    return [
        # floor, normal +y: (B-A) x (C-A) with A=(x0,y0,z0), B=(x0,y0,z1), C=(x1,y0,z0) = +y
        TrianglePlane(V(x0, y0, z0), V(x0, y0, z1), V(x1, y0, z0), False, floor),
        TrianglePlane(V(x1, y0, z1), V(x1, y0, z0), V(x0, y0, z1), False, floor),
        # ceiling, normal -y
        TrianglePlane(V(x0, y1, z0), V(x1, y1, z0), V(x0, y1, z1), False, ceil_),
        TrianglePlane(V(x1, y1, z1), V(x0, y1, z1), V(x1, y1, z0), False, ceil_),
        # back wall z = z1, normal -z
        TrianglePlane(V(x0, y0, z1), V(x0, y1, z1), V(x1, y0, z1), False, back),
        TrianglePlane(V(x1, y1, z1), V(x1, y0, z1), V(x0, y1, z1), False, back),
        # left wall x = x0 (image right), normal +x
        TrianglePlane(V(x0, y0, z0), V(x0, y1, z0), V(x0, y0, z1), False, left),
        TrianglePlane(V(x0, y1, z1), V(x0, y0, z1), V(x0, y1, z0), False, left),
        # right wall x = x1, normal -x
        TrianglePlane(V(x1, y0, z0), V(x1, y0, z1), V(x1, y1, z0), False, right),
        TrianglePlane(V(x1, y1, z1), V(x1, y1, z0), V(x1, y0, z1), False, right),
    ]


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")  # holds samples_textures/ like the reference's repository root


def _tile_texture(asset_dir):
    """The one image asset the reference ships (samples_textures/Tiles075_1K_Color.jpg, kept under the package's
    assets/); falls back to a checkerboard when the directory is not there."""
    asset_dir = asset_dir or ASSET_DIR
    if asset_dir:
        path = os.path.join(asset_dir, "samples_textures", "Tiles075_1K_Color.jpg")
        if os.path.exists(path):
            with open(path, "rb") as f:
                return ImageTexture(Img=api.LoadRGBAImage(f))
    return CheckersTexture(Vec(0.8, 0.8, 0.8), Vec(0.3, 0.3, 0.35), 16.0, 16.0)


# ---- the five configs ------------------------------------------------------------------
def config1(width=640, height=380, asset_dir=None):
    """README code-example scene (example.go:9-224); 640x380 is the size the golden image pins."""
    return api.ExampleRegression(width, height, asset_dir)


def config2(width=1920, height=1080, spp=16, reps=1):
    """Bunny (~70k triangles; stand-in 81,920), double-sided PhongBlinn, 1 PointLight + weak
    ambient, 2-triangle floor, aperture 0."""
    bunny = TriangleMesh(_fit(bunny_vertices(), (0.0, 0.0, 0.0), 4.0), False,
                         PhongBlinn(ColorFrac=Vec(0.85, 0.75, 0.6), SpecularColorFrac=Vec(1, 1, 1), SpecHardness=24))
    fl = PhongBlinn(ColorFrac=Vec(0.6, 0.6, 0.6), SpecularColorFrac=Vec(0.2, 0.2, 0.2), SpecHardness=4,
                    Texture=CheckersTexture(Vec(0.9, 0.9, 0.9), Vec(0.25, 0.3, 0.35), 40.0, 40.0))
    R = 40.0
    shapes = [TrianglePlane(Vec(-R, 0, -R), Vec(-R, 0, R), Vec(R, 0, -R), False, fl),
              TrianglePlane(Vec(R, 0, R), Vec(R, 0, -R), Vec(-R, 0, R), False, fl), bunny]
    lights = [AmbientLight(Vec(1, 1, 1), 0.08),
              PointLight(ColorFrac=Vec(1.0, 0.95, 0.9), Position=Vec(5, 9, -6), LightIntensity=90,
                         SpecularLightIntensity=40, InverseSquareLawDecayFactor=0.6)]
    spec = ImageSpec(width, height, spp, 16, reps, 16, api.Dijkstra)
    scene = Scene(Vec(0, 3.2, -8.5), Vec(0, 1.8, 0), Vec(0, 1, 0), Vec(0, 1.8, 0), 0.0, 45.0, shapes, lights)
    return spec, scene


def config3(width=1920, height=1080, spp=64, reps=16, asset_dir=None, nu=660, nv=660):
    """Dragon (~870k triangles; stand-in 871,200) in a 10-triangle box room, 3 SpotLights +
    ambient, soft shadows (reps 16), aperture blur 0.05, depth 16 — the headline config."""
    dragon = TriangleMesh(_fit(dragon_vertices(nu, nv), (0.0, 0.35, 0.5), 6.5), False,
                          PhongBlinn(ColorFrac=Vec(0.55, 0.75, 0.6), SpecularColorFrac=Vec(1, 1, 1), SpecHardness=40))
    tiles = _tile_texture(asset_dir)
    floor = PhongBlinn(ColorFrac=Vec(0.7, 0.7, 0.7), SpecularColorFrac=Vec(0.3, 0.3, 0.3), SpecHardness=8, Texture=tiles)
    wall = PhongBlinn(ColorFrac=Vec(0.75, 0.72, 0.68), SpecularColorFrac=Vec(0.1, 0.1, 0.1), SpecHardness=4)
    back = PhongBlinn(ColorFrac=Vec(0.7, 0.7, 0.7), SpecularColorFrac=Vec(0.1, 0.1, 0.1), SpecHardness=4, Texture=tiles)
    shapes = _room(mats=(floor, wall, back, wall, wall)) + [dragon]
    lights = [
        AmbientLight(Vec(1, 1, 1), 0.06),
        SpotLight(ColorFrac=Vec(1.0, 0.55, 0.45), Position=Vec(4.5, 7.2, -5.0), LightIntensity=110, SpecularLightIntensity=60,
                  LookAt=Vec(0, 2.5, 0.5), Angle=38, InverseSquareLawDecayFactor=0.8),
        SpotLight(ColorFrac=Vec(0.45, 0.65, 1.0), Position=Vec(-4.8, 7.0, -3.5), LightIntensity=110, SpecularLightIntensity=60,
                  LookAt=Vec(0, 2.5, 0.5), Angle=38, InverseSquareLawDecayFactor=0.8),
        SpotLight(ColorFrac=Vec(0.75, 1.0, 0.6), Position=Vec(0.0, 7.5, 4.5), LightIntensity=90, SpecularLightIntensity=40,
                  LookAt=Vec(0, 2.0, 0.0), Angle=45, InverseSquareLawDecayFactor=0.8),
    ]
    spec = ImageSpec(width, height, spp, 16, reps, 16, api.Dijkstra)
    scene = Scene(Vec(0, 4.0, -9.0), Vec(0, 3.0, 0.5), Vec(0, 1, 0), Vec(0, 3.0, 0.5), 0.05, 55.0, shapes, lights)
    return spec, scene


def config4(asset_dir=None):
    """Config 3's scene at 3840x2160, 1024 spp: sample-partitioned across the GPUs."""
    return config3(3840, 2160, 1024, 16, asset_dir)


def config5(width=3840, height=2160, spp=64, reps=16, asset_dir=None, nu=660, nv=660):
    """Bunny + Dragon + 6 Dielectric spheres (RI 1.52) on a Metal floor, 4 SpotLights; rendered
    tile-split (64x64 tiles dealt round-robin to the GPUs)."""
    dragon = TriangleMesh(_fit(dragon_vertices(nu, nv), (-1.6, 0.3, 1.2), 5.0), False,
                          PhongBlinn(ColorFrac=Vec(0.55, 0.75, 0.6), SpecularColorFrac=Vec(1, 1, 1), SpecHardness=40))
    bunny = TriangleMesh(_fit(bunny_vertices(), (3.2, 0.0, -0.5), 2.6), False,
                         PhongBlinn(ColorFrac=Vec(0.85, 0.7, 0.55), SpecularColorFrac=Vec(1, 1, 1), SpecHardness=24))
    metal = Metal(Albedo=Vec(0.85, 0.85, 0.9), Fuzz=0.03)
    wall = PhongBlinn(ColorFrac=Vec(0.7, 0.7, 0.72), SpecularColorFrac=Vec(0.1, 0.1, 0.1), SpecHardness=4)
    back = PhongBlinn(ColorFrac=Vec(0.7, 0.7, 0.7), SpecularColorFrac=Vec(0.1, 0.1, 0.1), SpecHardness=4,
                      Texture=_tile_texture(asset_dir))
    shapes = _room(mats=(metal, wall, back, wall, wall))
    # Metal.scatter starts the reflected ray AT the hit point (material.go:49, no epsilon): a two-sided mirror
    # triangle is then hit again at t = +-rounding noise.  The reference's own mirrors are SingleSided for that
    # reason (example.go:141-152), and so is this floor: a ray leaving it sees its back face.
    for tri in shapes[:2]:
        tri.SingleSided = True
    glass = Dielectric(RefractiveIndex=1.52)
    for k in range(6):
        a = 2 * math.pi * k / 6
        shapes.append(Sphere(Vec(3.4 * math.cos(a) + 0.4, 0.55, 3.0 * math.sin(a) - 2.2), 0.55, glass))
    shapes += [dragon, bunny]
    lights = [AmbientLight(Vec(1, 1, 1), 0.05)]
    for k, col in enumerate([(1.0, 0.6, 0.5), (0.5, 0.7, 1.0), (0.8, 1.0, 0.6), (1.0, 0.95, 0.8)]):
        a = 2 * math.pi * (k + 0.5) / 4
        lights.append(SpotLight(ColorFrac=Vec(*col), Position=Vec(5.0 * math.cos(a), 7.3, 4.5 * math.sin(a) - 1.0),
                                LightIntensity=95, SpecularLightIntensity=50, LookAt=Vec(0, 1.5, 0), Angle=40,
                                InverseSquareLawDecayFactor=0.8))
    spec = ImageSpec(width, height, spp, 16, reps, 16, api.Dijkstra)
    scene = Scene(Vec(0, 4.2, -9.2), Vec(0, 2.2, 0.0), Vec(0, 1, 0), Vec(0, 2.2, 0), 0.03, 55.0, shapes, lights)
    return spec, scene
