"""Host-side mirror of the Go `raytracer` package API (the drop-in surface).

Go is not in this image, so the host side above the C-ABI is mirrored here with the same
exported names, field names, argument meaning and error behaviour as the reference:

    ImageSpec, Scene, GenerateImage                 raytracer/raytracer.go:18-107
    Shape: Sphere, TrianglePlane (+Rotate/Scale/Translate)   raytracer/shape.go:18-45,80-89,165-181,240-263
    Material: Standard, Metal, Dielectric, PhongBlinn        raytracer/material.go:9-32
    Light: AmbientLight, PointLight, SpotLight               raytracer/light.go:8-39
    CheckersTexture, ImageTexture, LoadRGBAImage             raytracer/texture.go:16-25,55-83
    ExampleRegression                                        raytracer/example.go:9-224

GenerateImage renders on the MI355X through librtgo_hip.so; there is no CPU fallback —
a missing library or device raises.
"""
import math
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np

from . import _abi

# raytracer.go:20-23
Dijkstra = 0
DepthFirstSearch = 1


@dataclass
class Vec:
    """gonum r3.Vec."""
    X: float = 0.0
    Y: float = 0.0
    Z: float = 0.0


@dataclass
class ImageSpec:  # raytracer.go:25-33
    Width: int = 0
    Height: int = 0
    AntiAliasingFactor: int = 0
    RayTracingMaxDepth: int = 0
    SoftShadowMonteCarloRepetitions: int = 0
    WorkerCount: int = 0  # goroutine count in the reference; accepted and ignored here
    BvhTraversalAlgorithm: int = Dijkstra


@dataclass
class Scene:  # raytracer.go:35-46
    CameraLookFrom: Vec = field(default_factory=Vec)
    CameraLookAt: Vec = field(default_factory=Vec)
    CameraUp: Vec = field(default_factory=Vec)
    CameraFocusPoint: Vec = field(default_factory=Vec)  # never read by the reference (raytracer.go:59,67)
    CameraAperature: float = 0.0
    CameraFov: float = 0.0
    Shapes: List[Any] = field(default_factory=list)
    Lights: List[Any] = field(default_factory=list)


# ---- textures ------------------------------------------------------------------------
@dataclass
class CheckersTexture:  # texture.go:16-21
    ColorFrac1: Vec = field(default_factory=Vec)
    ColorFrac2: Vec = field(default_factory=Vec)
    CheckersWidth: float = 0.0
    CheckersHeight: float = 0.0


class RGBAImage:
    """image.RGBA: Pix is an HxWx4 uint8 array, row 0 = top."""

    def __init__(self, pix):
        self.Pix = np.ascontiguousarray(pix, dtype=np.uint8)

    @property
    def Width(self):
        return self.Pix.shape[1]

    @property
    def Height(self):
        return self.Pix.shape[0]

    def save_container(self, path):
        """Write the decoded pixels as the raw "RGBA8 <w> <h>\\n" + bytes container that the C++ mirror's
        LoadRGBAImage (include/raytracer.hpp) reads: image.Decode is Go stdlib and has no C++ twin here."""
        with open(path, "wb") as f:
            f.write(f"RGBA8 {self.Width} {self.Height}\n".encode())
            f.write(self.Pix.tobytes())


@dataclass
class ImageTexture:  # texture.go:23-25
    Img: Optional[RGBAImage] = None


def LoadRGBAImage(file) -> RGBAImage:
    """texture.go:55-83.  image.Decode + the /255.99 quantiser: every channel's 16-bit value
    c16 (alpha-premultiplied, = c8*257 for opaque pixels) becomes floor(min(255, c16/255.99)),
    which maps 254 -> 255 and is the identity otherwise."""
    from PIL import Image

    img = Image.open(file)
    img.load()
    a = np.asarray(img.convert("RGBA"), dtype=np.float64)
    alpha16 = a[..., 3:4] * 257.0
    rgb16 = np.floor(a[..., :3] * 257.0 * alpha16 / 65535.0)  # color.NRGBA.RGBA() premultiplies
    c16 = np.concatenate([rgb16, alpha16], axis=-1)
    return RGBAImage(np.floor(np.minimum(255.0, c16 / 255.99)).astype(np.uint8))


# ---- materials -----------------------------------------------------------------------
@dataclass
class Standard:  # material.go:13-16
    ColorFrac: Vec = field(default_factory=Vec)
    Texture: Any = None


@dataclass
class Metal:  # material.go:18-21
    Albedo: Vec = field(default_factory=Vec)
    Fuzz: float = 0.0


@dataclass
class Dielectric:  # material.go:23-25
    RefractiveIndex: float = 0.0


@dataclass
class PhongBlinn:  # material.go:27-32
    ColorFrac: Vec = field(default_factory=Vec)
    SpecularColorFrac: Vec = field(default_factory=Vec)
    SpecHardness: float = 0.0
    Texture: Any = None


# ---- shapes --------------------------------------------------------------------------
def _rotate_point(p: Vec, rv: Vec) -> Vec:  # shape.go:240-263 (Z, then X, then Y; degrees)
    k = math.pi / 180.0
    x = p.X * math.cos(k * rv.Z) - p.Y * math.sin(k * rv.Z)
    y = p.X * math.sin(k * rv.Z) + p.Y * math.cos(k * rv.Z)
    z = p.Z
    y, z = (y * math.cos(k * rv.X) - z * math.sin(k * rv.X),
            y * math.sin(k * rv.X) + z * math.cos(k * rv.X))
    x, z = (x * math.cos(k * rv.Y) + z * math.sin(k * rv.Y),
            -1 * x * math.sin(k * rv.Y) + z * math.cos(k * rv.Y))
    return Vec(x, y, z)


@dataclass
class Sphere:  # shape.go:33-37
    Center: Vec = field(default_factory=Vec)
    Radius: float = 0.0
    Mat: Any = None

    def Translate(self, tv: Vec):  # shape.go:80-82
        self.Center = Vec(tv.X + self.Center.X, tv.Y + self.Center.Y, tv.Z + self.Center.Z)

    def Scale(self, c: float):  # shape.go:84-86: radius only
        self.Radius *= c

    def Rotate(self, rv: Vec):  # shape.go:88-89: no-op
        pass


@dataclass
class TrianglePlane:  # shape.go:39-45
    PointA: Vec = field(default_factory=Vec)
    PointB: Vec = field(default_factory=Vec)
    PointC: Vec = field(default_factory=Vec)
    SingleSided: bool = False
    Mat: Any = None

    def Translate(self, tv: Vec):  # shape.go:165-169
        for n in ("PointA", "PointB", "PointC"):
            p = getattr(self, n)
            setattr(self, n, Vec(tv.X + p.X, tv.Y + p.Y, tv.Z + p.Z))

    def Scale(self, c: float):  # shape.go:171-175
        for n in ("PointA", "PointB", "PointC"):
            p = getattr(self, n)
            setattr(self, n, Vec(c * p.X, c * p.Y, c * p.Z))

    def Rotate(self, rv: Vec):  # shape.go:177-181
        for n in ("PointA", "PointB", "PointC"):
            setattr(self, n, _rotate_point(getattr(self, n), rv))


class TriangleMesh:
    """Bulk form of N TrianglePlane shapes sharing one material, kept as an (N, 3, 3) float64
    array so a 870k-triangle model does not become 870k Python objects.  It expands, in order,
    to N consecutive entries of Scene.Shapes; Rotate/Scale/Translate apply the per-triangle
    rules of shape.go:165-181 to every vertex."""

    def __init__(self, vertices, SingleSided=False, Mat=None):
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        if v.ndim != 3 or v.shape[1:] != (3, 3):
            raise ValueError("vertices must have shape (N, 3, 3)")
        self.vertices, self.SingleSided, self.Mat = v, bool(SingleSided), Mat

    def __len__(self):
        return self.vertices.shape[0]

    def Translate(self, tv: Vec):
        self.vertices = np.array([tv.X, tv.Y, tv.Z]) + self.vertices

    def Scale(self, c: float):
        self.vertices = c * self.vertices

    def Rotate(self, rv: Vec):
        k = math.pi / 180.0
        x, y, z = self.vertices[..., 0], self.vertices[..., 1], self.vertices[..., 2]
        x, y = (x * math.cos(k * rv.Z) - y * math.sin(k * rv.Z), x * math.sin(k * rv.Z) + y * math.cos(k * rv.Z))
        y, z = (y * math.cos(k * rv.X) - z * math.sin(k * rv.X), y * math.sin(k * rv.X) + z * math.cos(k * rv.X))
        x, z = (x * math.cos(k * rv.Y) + z * math.sin(k * rv.Y), -1 * x * math.sin(k * rv.Y) + z * math.cos(k * rv.Y))
        self.vertices = np.ascontiguousarray(np.stack([x, y, z], axis=-1))

    def to_records(self, material_index, out=None):
        """The mesh as rt_shape records (into `out`, a slice of the scene's table, when given)."""
        n = len(self)
        rec = np.empty(n, dtype=_abi.SHAPE_DTYPE) if out is None else out
        if n == 0:  # an empty STL / mesh: no records (a reshape(0, -1) cannot infer its width)
            return rec
        # a record is 4 int32 {kind, material, single_sided, pad} and 9 doubles {a, b, c}: one strided copy each
        rec.view(np.int32).reshape(n, -1)[:, :4] = (_abi.RT_SHAPE_TRIANGLE, material_index, 1 if self.SingleSided else 0, 0)
        rec.view(np.float64).reshape(n, -1)[:, 2:] = self.vertices.reshape(n, 9)
        return rec


# ---- lights --------------------------------------------------------------------------
@dataclass
class AmbientLight:  # light.go:18-21
    ColorFrac: Vec = field(default_factory=Vec)
    LightIntensity: float = 0.0


@dataclass
class PointLight:  # light.go:23-29
    ColorFrac: Vec = field(default_factory=Vec)
    Position: Vec = field(default_factory=Vec)
    LightIntensity: float = 0.0
    SpecularLightIntensity: float = 0.0
    InverseSquareLawDecayFactor: float = 0.0


@dataclass
class SpotLight:  # light.go:31-39
    ColorFrac: Vec = field(default_factory=Vec)
    Position: Vec = field(default_factory=Vec)
    LightIntensity: float = 0.0
    SpecularLightIntensity: float = 0.0
    LookAt: Vec = field(default_factory=Vec)
    Angle: float = 0.0  # half-angle, degrees
    InverseSquareLawDecayFactor: float = 0.0


# ---- GenerateImage -------------------------------------------------------------------
def GenerateImage(imageSpec: ImageSpec, scene: Scene, seed: int = 0x5EED) -> RGBAImage:
    """raytracer.go:58-107 on the MI355X.  Prints the reference's three kinds of progress
    message (accelerationstructures.go:28,63; raytracer.go:105).  An unknown
    BvhTraversalAlgorithm raises, as the reference panics (accelerationstructures.go:73)."""
    from . import backend

    print("Building BoundingVolumeHierarchy")
    with backend.DeviceScene(scene) as dev:
        print("Finished building BoundingVolumeHierarchy")
        pix, stats = dev.render(imageSpec, seed=seed)
    print(f"Finished ray tracing in {stats.ms_render:.3f}ms")
    return RGBAImage(pix)


# ---- ExampleRegression ---------------------------------------------------------------
def ExampleRegression(width: int, height: int, repoBaseDir: str):
    """example.go:9-224, field for field.  repoBaseDir must hold
    samples_textures/Tiles075_1K_Color.jpg; a missing file raises (the reference panics)."""
    import os

    floorRadius = 100.0
    centerPiecesRadius = 2.0
    backMirrorRadius = 4 * centerPiecesRadius
    backMirrorBorder = centerPiecesRadius / 2

    cameraLookFrom = Vec(0, 3 * centerPiecesRadius, -5)
    cameraLookAt = Vec(0, 2 * centerPiecesRadius, 0)
    cameraUp = Vec(0, 1, 0)

    texturePlane = CheckersTexture(Vec(0, 1, 0), Vec(0, 0, 1), 100.0, 100.0)
    textureLeftSphere = CheckersTexture(Vec(0, 0, 0), Vec(1, 1, 1), 10.0, 10.0)
    with open(os.path.join(repoBaseDir, "samples_textures/Tiles075_1K_Color.jpg"), "rb") as f:
        textureRightSphere = ImageTexture(Img=LoadRGBAImage(f))

    r, R, b = centerPiecesRadius, backMirrorRadius, backMirrorBorder
    frame = Vec(150 / 255.0, 111 / 255.0, 51 / 255.0)
    white = Vec(1, 1, 1)
    shapes = [
        Sphere(Vec(4 * r, r, 0), r, Standard(Texture=textureLeftSphere)),
        Sphere(Vec(2 * r, r, 0), r, Dielectric(RefractiveIndex=1.52)),
        Sphere(Vec(0, r, 0), r, PhongBlinn(SpecHardness=1, SpecularColorFrac=Vec(1, 1, 1), ColorFrac=Vec(1, 1, 1))),
        Sphere(Vec(-2 * r, r, 0), r, Metal(Albedo=Vec(1.0, 1.0, 1.0), Fuzz=0)),
        Sphere(Vec(-4 * r, r, 0), r, PhongBlinn(SpecHardness=1, SpecularColorFrac=Vec(1, 1, 1), Texture=textureRightSphere)),
        # floor
        TrianglePlane(Vec(-floorRadius, 0, -floorRadius), Vec(-floorRadius, 0, floorRadius), Vec(floorRadius, 0, -floorRadius), True,
                      PhongBlinn(ColorFrac=Vec(0, 0, 0), SpecularColorFrac=Vec(1, 1, 1), SpecHardness=1, Texture=texturePlane)),
        TrianglePlane(Vec(floorRadius, 0, floorRadius), Vec(floorRadius, 0, -floorRadius), Vec(-floorRadius, 0, floorRadius), True,
                      PhongBlinn(ColorFrac=Vec(0, 0, 0), SpecularColorFrac=Vec(1, 1, 1), SpecHardness=1, Texture=texturePlane)),
        # back mirror
        TrianglePlane(Vec(R, R, R), Vec(R, 0, R), Vec(-R, R, R), True, Standard(ColorFrac=frame)),
        TrianglePlane(Vec(-R, 0, R), Vec(-R, R, R), Vec(R, 0, R), True, Standard(ColorFrac=frame)),
        TrianglePlane(Vec(R - b, R - b, R - b), Vec(R - b, b, R - b), Vec(-(R - b), R - b, R - b), True,
                      Metal(Albedo=white, Fuzz=0)),
        TrianglePlane(Vec(-(R - b), b, R - b), Vec(-(R - b), R - b, R - b), Vec(R - b, b, R - b), True,
                      Metal(Albedo=white, Fuzz=0)),
    ]
    lights = [
        AmbientLight(ColorFrac=Vec(255 / 255.0, 0 / 255.0, 0 / 255.0), LightIntensity=0.2),
        SpotLight(ColorFrac=Vec(171 / 255.0, 137 / 255.0, 255 / 255.0), LightIntensity=100, SpecularLightIntensity=100,
                  Position=Vec(6 * r, 5 * r, -3 * r), LookAt=Vec(0, 0, 0), Angle=30, InverseSquareLawDecayFactor=1.0),
        PointLight(ColorFrac=Vec(67 / 255.0, 163 / 255.0, 241 / 255.0), LightIntensity=100, SpecularLightIntensity=10,
                   Position=Vec(-4 * r, r, 3 * r), InverseSquareLawDecayFactor=0.5),
    ]
    imageSpec = ImageSpec(Width=width, Height=height, AntiAliasingFactor=32, RayTracingMaxDepth=16,
                          SoftShadowMonteCarloRepetitions=16, WorkerCount=16, BvhTraversalAlgorithm=Dijkstra)
    scene = Scene(CameraLookFrom=cameraLookFrom, CameraLookAt=cameraLookAt, CameraUp=cameraUp,
                  CameraFocusPoint=cameraLookAt, CameraAperature=0.015, CameraFov=60.0,
                  Shapes=shapes, Lights=lights)
    return imageSpec, scene
