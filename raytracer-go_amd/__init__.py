"""raytracer-go_amd — MI355X (gfx950) render backend behind the Go `raytracer` package API.

The directory name has a hyphen (it mirrors the reference repo's name), so import it with
    rt = importlib.import_module("raytracer-go_amd")
"""
from . import _abi  # noqa: F401
from .api import (  # noqa: F401
    AmbientLight, CheckersTexture, DepthFirstSearch, Dielectric, Dijkstra, ExampleRegression,
    GenerateImage, ImageSpec, ImageTexture, LoadRGBAImage, Metal, PhongBlinn, PointLight,
    RGBAImage, Scene, Sphere, SpotLight, Standard, TriangleMesh, TrianglePlane, Vec,
)
