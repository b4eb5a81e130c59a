"""Host-side logic (no GPU): the Go-API mirror, the scene flattener, the C-ABI surface."""
import ctypes
import io
import math
import os
import re

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(backend):
    """The C-ABI library loads without a GPU and exports every function include/rtgo.h declares."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "rtgo.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr))
    assert declared == set(backend.SYMBOLS), declared ^ set(backend.SYMBOLS)
    backend.build()
    lib = ctypes.CDLL(backend.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_abi_struct_layout(rt):
    a = rt._abi
    assert ctypes.sizeof(a.RtVec3) == 24 and ctypes.sizeof(a.RtShape) == 88 and ctypes.sizeof(a.RtMaterial) == 64
    assert ctypes.sizeof(a.RtTexture) == 88 and ctypes.sizeof(a.RtLight) == 112 and ctypes.sizeof(a.RtImageSpec) == 32
    assert ctypes.sizeof(a.RtPartition) == 32 and ctypes.sizeof(a.RtSceneInfo) == 40 and ctypes.sizeof(a.RtShare) == 24


def test_no_gpu_fails_loudly(backend):
    """No CPU fallback: without a device the product raises instead of rendering."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(backend.RtgoError):
        backend.init(0)


def test_flatten_example_scene(rt, c1):
    spec, scene = c1
    flat = rt._abi.FlatScene(scene)
    assert flat.desc.n_shapes == 11 and flat.desc.n_lights == 3 and flat.desc.n_textures == 3
    s = flat.shapes
    assert list(s["kind"][:5]) == [1] * 5 and list(s["kind"][5:]) == [2] * 6
    assert s["b"]["x"][0] == 2.0 and s["a"]["x"][0] == 8.0
    assert all(s["single_sided"][5:] == 1)
    # the two floor triangles hold equal-valued but distinct PhongBlinn structs -> distinct material rows are fine,
    # shared objects are deduplicated
    kinds = [flat._materials[i].kind for i in range(flat.desc.n_materials)]
    assert kinds.count(rt._abi.RT_MAT_PHONGBLINN) >= 3 and rt._abi.RT_MAT_DIELECTRIC in kinds
    assert flat._lights[1].kind == rt._abi.RT_LIGHT_SPOT and flat._lights[1].angle_deg == 30


def test_empty_mesh_flattens_to_no_records(rt, c1):
    """A mesh of zero triangles (an empty STL) adds nothing to the shape table."""
    spec, scene = c1
    empty = rt.TriangleMesh(np.zeros((0, 3, 3)), False, scene.Shapes[0].Mat)
    assert len(empty.to_records(0)) == 0
    sc = rt.Scene(scene.CameraLookFrom, scene.CameraLookAt, scene.CameraUp, scene.CameraFocusPoint, scene.CameraAperature,
                  scene.CameraFov, [scene.Shapes[0], empty, scene.Shapes[5]], scene.Lights)
    flat = rt._abi.FlatScene(sc)
    assert flat.desc.n_shapes == 2 and list(flat.shapes["kind"]) == [1, 2]


def test_build_info_names_the_sources_and_knobs(backend):
    """rt_build_info: what bench.py compares with the build profiles/*/bounds.json was counted on."""
    backend.build()
    info = backend.build_info()
    assert set(info) >= {"src", "shadow_waves", "beam_waves", "extend_waves", "lds_stack", "pend_cap"}
    assert re.fullmatch(r"[0-9a-f]{12}", info["src"]) and int(info["shadow_waves"]) >= 1


def test_bench_without_a_gpu_exits_with_a_message():
    """bench.py --gpus N, with or without an outside launcher: no device, no line (and never a CPU fallback)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env_extra in ({}, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(env_extra)
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode != 0 and "MI355X" in p.stderr and not p.stdout.strip()


def test_shape_transforms(rt):
    """shape.go:80-89,165-181,240-263: Z then X then Y rotation in degrees; Sphere.Rotate is a no-op,
    Sphere.Scale scales the radius only."""
    V = rt.Vec
    t = rt.TrianglePlane(V(1, 0, 0), V(0, 1, 0), V(0, 0, 1))
    t.Rotate(V(0, 0, 90))
    assert np.allclose((t.PointA.X, t.PointA.Y, t.PointA.Z), (0, 1, 0), atol=1e-12)
    assert np.allclose((t.PointB.X, t.PointB.Y, t.PointB.Z), (-1, 0, 0), atol=1e-12)
    t = rt.TrianglePlane(V(1, 0, 0), V(0, 1, 0), V(0, 0, 1))
    t.Rotate(V(90, 0, 0))
    assert np.allclose((t.PointB.X, t.PointB.Y, t.PointB.Z), (0, 0, 1), atol=1e-12)
    t = rt.TrianglePlane(V(1, 0, 0), V(0, 1, 0), V(0, 0, 1))
    t.Rotate(V(0, 90, 0))
    assert np.allclose((t.PointA.X, t.PointA.Y, t.PointA.Z), (0, 0, -1), atol=1e-12)
    t.Scale(2.0)
    t.Translate(V(1, 2, 3))
    assert np.allclose((t.PointA.X, t.PointA.Y, t.PointA.Z), (1, 2, 1), atol=1e-12)
    s = rt.Sphere(V(1, 1, 1), 2.0)
    s.Rotate(V(10, 20, 30)); s.Scale(1.5); s.Translate(V(1, 0, -1))
    assert (s.Center.X, s.Center.Y, s.Center.Z, s.Radius) == (2, 1, 0, 3.0)
    # the bulk mesh applies the same per-vertex rule
    tri = rt.TrianglePlane(V(1, 2, 3), V(-1, 0.5, 2), V(0, 1, -4))
    mesh = rt.TriangleMesh(np.array([[[1, 2, 3], [-1, 0.5, 2], [0, 1, -4]]], dtype=float))
    for obj in (tri, mesh):
        obj.Rotate(V(25, -40, 70)); obj.Scale(0.3); obj.Translate(V(0.1, 0.2, 0.3))
    assert np.allclose(mesh.vertices[0], [[p.X, p.Y, p.Z] for p in (tri.PointA, tri.PointB, tri.PointC)], atol=1e-14)


def test_load_rgba_image_quantiser(rt):
    """texture.go:66-80: floor(min(255, c16/255.99)) is the identity on 8-bit values except 254 -> 255."""
    from PIL import Image
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    buf = io.BytesIO()
    Image.fromarray(np.stack([ramp, ramp, ramp], -1)).save(buf, format="PNG")
    buf.seek(0)
    pix = rt.LoadRGBAImage(buf).Pix
    expect = ramp.astype(int).copy()
    expect[expect == 254] = 255
    assert np.array_equal(pix[..., 0], expect) and (pix[..., 3] == 255).all()


def test_stl_round_trip(tmp_path):
    import importlib
    stl = importlib.import_module("raytracer-go_amd.stl")
    rng = np.random.default_rng(0)
    v = rng.normal(size=(100, 3, 3)).astype(np.float32).astype(np.float64)
    p = tmp_path / "m.stl"
    stl.save_binary_stl(str(p), v)
    assert os.path.getsize(p) == 84 + 50 * 100
    assert np.array_equal(stl.load_binary_stl(str(p)), v)
    with open(p, "r+b") as f:
        f.truncate(84 + 50 * 99 + 10)
    with pytest.raises(ValueError):
        stl.load_binary_stl(str(p))


def test_stand_in_meshes(scenes):
    m = scenes.icosphere_mesh(3)
    assert m.shape == (20 * 4 ** 3, 3, 3)
    n = np.cross(m[:, 1] - m[:, 0], m[:, 2] - m[:, 0])
    assert ((m.mean(1) * n).sum(1) > 0).all()  # wound outward
    k = scenes.torus_knot_mesh(40, 24)
    assert k.shape == (2 * 40 * 24, 3, 3)
    assert np.array_equal(scenes.torus_knot_mesh(40, 24), k)  # seeded: reproducible
    spec, scene = scenes.config3(nu=30, nv=20)
    assert len(scene.Shapes) == 11 and len(scene.Shapes[-1]) == 1200 and len(scene.Lights) == 4


def test_sample_split_covers_all_samples():
    import importlib
    d = importlib.import_module("raytracer-go_amd.dist")
    for spp in (1, 7, 64, 1024):
        for world in (1, 2, 3, 8):
            r = [d.sample_range(spp, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == spp and all(r[i][1] == r[i + 1][0] for i in range(world - 1))


def test_partition_shares_cover_the_frame(backend, rt):
    """rt_partition_share (the share arithmetic of rt_render_accum_device and rt_multi_render, no device needed):
    over the ranks of a world the sample ranges tile [0, spp) in order and the tile counts add up to the tile grid."""
    import importlib
    abi = rt._abi
    d = importlib.import_module("raytracer-go_amd.dist")
    for (W, H, tile) in ((640, 380, 64), (1920, 1080, 64), (100, 37, 8), (3840, 2160, 128)):
        n_tiles = -(-W // tile) * -(-H // tile)
        for spp in (1, 7, 64, 1024):
            spec = rt.ImageSpec(W, H, spp, 16, 16, 1, 0)
            for world in (1, 2, 3, 8):
                sh = [backend.partition_share(spec, abi.partition(mode=abi.RT_PART_SAMPLES, rank=r, world_size=world, tile_size=tile)) for r in range(world)]
                assert [x[:2] for x in sh] == [d.sample_range(spp, r, world) for r in range(world)]
                assert sh[0][0] == 0 and sh[-1][1] == spp and all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
                assert all(x[2] == n_tiles and x[3] == tile for x in sh)
                th = [backend.partition_share(spec, abi.partition(mode=abi.RT_PART_TILES, rank=r, world_size=world, tile_size=tile)) for r in range(world)]
                assert sum(x[2] for x in th) == n_tiles and all(x[:2] == (0, spp) for x in th)
                assert [x[2] for x in th] == [len([k for k in range(n_tiles) if d.tile_owner(k, world) == r]) for r in range(world)]
    with pytest.raises(backend.RtgoError):
        backend.partition_share(rt.ImageSpec(64, 64, 1, 1, 1, 1, 0), abi.partition(mode=abi.RT_PART_TILES, rank=2, world_size=2))


def test_tree_builders_under_sanitizers(tmp_path):
    """tests/cpp/builder_check.cpp: both host builders + the flattener on random and degenerate scenes,
    compiled with AddressSanitizer and UBSan (host code only).  Checks that every shape sits in exactly
    one leaf, every box encloses what is below it, indices are in range and the exact stack bound holds."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "builder_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                           "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "builder_check.cpp"),
                           os.path.join(root, "raytracer-go_amd", "csrc", "octree.cpp"), "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all builder invariants hold" in p.stdout
