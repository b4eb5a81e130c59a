"""include/raytracer.hpp — the C++ twin of the Go package API above the C-ABI.

CPU: the header compiles with g++, its ExampleRegression + transforms flatten to the same C-ABI
scene description as the Python mirror's (which test_host_api.py checks against example.go).
GPU: examples/example_regression (the main.go flow in C++) renders the same bytes as the Python
GenerateImage for the same scene, through the same librtgo_hip.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fmt(x):
    return repr(float(x))


def _vec(v):
    return [float(v.x), float(v.y), float(v.z)]


def _python_dump(rt, spec, scene):
    abi = rt._abi
    flat = abi.FlatScene(scene)
    d = flat.desc
    s = abi.image_spec(spec)
    out = [["spec", s.width, s.height, s.anti_aliasing_factor, s.ray_tracing_max_depth,
            s.soft_shadow_monte_carlo_repetitions, s.worker_count, s.bvh_traversal_algorithm]]
    out.append(["camera"] + _vec(d.camera_look_from) + _vec(d.camera_look_at) + _vec(d.camera_up)
               + [d.camera_aperture, d.camera_fov_deg])
    mats = C.cast(d.materials, C.POINTER(abi.RtMaterial))
    texs = C.cast(d.textures, C.POINTER(abi.RtTexture))
    for sh in flat.shapes:
        m = mats[int(sh["material"])]
        row = ["shape", int(sh["kind"]), int(sh["single_sided"])] + list(sh["a"]) + list(sh["b"]) + list(sh["c"])
        row += ["mat", m.kind] + _vec(m.color) + _vec(m.specular_color) + [m.param]
        if m.texture >= 0:
            t = texs[m.texture]
            h = 0
            if t.kind == abi.RT_TEX_IMAGE:
                pix = np.ctypeslib.as_array(C.cast(t.rgba, C.POINTER(C.c_uint8)), shape=(t.width * t.height * 4,))
                for b in pix.tolist():  # same FNV-style polynomial as tests/cpp/flat_dump.cpp, mod 2^64
                    h = (h * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
            row += ["tex", t.kind, t.width, t.height] + _vec(t.color1) + _vec(t.color2) + [t.checkers_width, t.checkers_height, h]
        out.append(row)
    lights = C.cast(d.lights, C.POINTER(abi.RtLight))
    for i in range(d.n_lights):
        l = lights[i]
        out.append(["light", l.kind] + _vec(l.color) + _vec(l.position) + _vec(l.look_at)
                   + [l.intensity, l.specular_intensity, l.angle_deg, l.decay])
    return out


def _parse(text):
    rows = []
    for line in text.strip().splitlines():
        rows.append(line.split())
    return rows


@pytest.fixture(scope="module")
def asset_dir(tmp_path_factory, rt, gold_dir):
    """The C++ LoadRGBAImage reads pre-decoded pixels: write them next to a copy of the layout."""
    d = tmp_path_factory.mktemp("assets")
    os.makedirs(d / "samples_textures")
    with open(os.path.join(gold_dir, "samples_textures", "Tiles075_1K_Color.jpg"), "rb") as f:
        rt.LoadRGBAImage(f).save_container(d / "samples_textures" / "Tiles075_1K_Color.rgba")
    return str(d)


def test_header_flattening_matches_python_mirror(rt, gold_dir, asset_dir, tmp_path):
    exe = tmp_path / "flat_dump"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "flat_dump.cpp"), "-o", str(exe)])
    got = _parse(subprocess.check_output([str(exe), asset_dir], text=True))

    spec, scene = rt.ExampleRegression(640, 380, gold_dir)
    extra = rt.TrianglePlane(rt.Vec(1, 2, 3), rt.Vec(4, 5, 7), rt.Vec(-1, 0, 2), False, rt.Metal(rt.Vec(0.5, 0.25, 1), 0.125))
    extra.Rotate(rt.Vec(10, 20, 30))
    extra.Scale(1.5)
    extra.Translate(rt.Vec(0.5, -1, 2))
    scene.Shapes.append(extra)
    scene.Shapes[0].Scale(1.25)
    scene.Shapes[0].Translate(rt.Vec(0, 0.5, 0))
    want = _python_dump(rt, spec, scene)

    tables = got.pop()
    assert tables[0] == "tables"
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert len(g) == len(w), (g, w)
        for a, b in zip(g, w):
            if isinstance(b, str):
                assert a == b
            elif isinstance(b, int):
                assert int(a) == b, (g, w)
            else:
                assert float(a) == float(b), (g, w)  # bit-exact: same float64 arithmetic on both sides
    # value-identical neighbours share rows: the two floor triangles, the two frame and the two mirror triangles
    assert int(tables[1]) <= len(scene.Shapes) - 3


def test_missing_texture_file_is_an_error(tmp_path):
    """example.go:36-44 panics when the texture cannot be opened; the C++ mirror throws and the example exits 1."""
    exe = os.path.join(ROOT, "examples", "example_regression")
    if not os.path.exists(exe):
        pytest.skip("examples/example_regression not built (run __graft_entry__.build())")
    p = subprocess.run([exe, "16", "16", str(tmp_path), str(tmp_path / "o.rgba")], capture_output=True, text=True)
    assert p.returncode == 1 and "panic: open" in p.stderr


@pytest.mark.gpu
def test_cpp_example_renders_the_same_bytes_as_python(rt, gold_dir, asset_dir, tmp_path):
    exe = os.path.join(ROOT, "examples", "example_regression")
    assert os.path.exists(exe), "examples/example_regression is built by __graft_entry__.build()"
    out = tmp_path / "frame.rgba"
    p = subprocess.run([exe, "160", "95", asset_dir, str(out), "4"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Finished building BoundingVolumeHierarchy" in p.stdout
    got = np.fromfile(out, dtype=np.uint8).reshape(95, 160, 4)
    spec, scene = rt.ExampleRegression(160, 95, gold_dir)
    spec.AntiAliasingFactor = 4
    want = rt.GenerateImage(spec, scene).Pix
    assert (got == want).all()


def _write_shim_scene(rt, spec, scene, path):
    """The flattened scene as the C structs of include/rtgo.h (tests/cpp/shim_sequence.c documents the layout)."""
    abi = rt._abi
    flat = abi.FlatScene(scene)
    d = flat.desc
    texs = C.cast(d.textures, C.POINTER(abi.RtTexture))
    with open(path, "wb") as f:
        f.write(np.array([d.n_shapes, d.n_materials, d.n_textures, d.n_lights], dtype=np.int64).tobytes())
        f.write(bytes(abi.image_spec(spec)))
        for v in (d.camera_look_from, d.camera_look_at, d.camera_up):
            f.write(bytes(v))
        f.write(np.array([d.camera_aperture, d.camera_fov_deg], dtype=np.float64).tobytes())
        f.write(flat.shapes.tobytes())
        f.write(C.string_at(d.materials, C.sizeof(abi.RtMaterial) * d.n_materials))
        pixels = []
        for i in range(d.n_textures):
            t = abi.RtTexture.from_buffer_copy(C.string_at(C.addressof(texs[i]), C.sizeof(abi.RtTexture)))
            if t.kind == abi.RT_TEX_IMAGE:
                pixels.append(C.string_at(t.rgba, t.width * t.height * 4))
            t.rgba = None
            f.write(bytes(t))
        f.write(C.string_at(d.lights, C.sizeof(abi.RtLight) * d.n_lights))
        for p in pixels:
            f.write(p)


@pytest.mark.gpu
def test_shim_call_sequence_in_c_renders_the_same_bytes_as_python(rt, gold_dir, tmp_path):
    """INTEGRATION.md's cgo shim cannot be compiled here (no Go): examples/shim_sequence performs its exact C-side call
    sequence — C-allocated tables, rt_scene_create(desc, 0, &s), rt_render(s, &spec, NULL, pix, &stats),
    rt_scene_destroy — on the flattened ExampleRegression and must produce the bytes of the Python twin."""
    exe = os.path.join(ROOT, "examples", "shim_sequence")
    assert os.path.exists(exe), "examples/shim_sequence is built by __graft_entry__.build()"
    spec, scene = rt.ExampleRegression(160, 95, gold_dir)
    spec.AntiAliasingFactor = 4
    _write_shim_scene(rt, spec, scene, tmp_path / "scene.bin")
    p = subprocess.run([exe, str(tmp_path / "scene.bin"), str(tmp_path / "frame.rgba")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Finished building BoundingVolumeHierarchy" in p.stdout and "Finished ray tracing in" in p.stdout
    got = np.fromfile(tmp_path / "frame.rgba", dtype=np.uint8).reshape(95, 160, 4)
    want = rt.GenerateImage(spec, scene).Pix
    assert (got == want).all()
    # the reference's panic on an unknown algorithm id comes back through rt_last_error (accelerationstructures.go:73)
    spec.BvhTraversalAlgorithm = 7
    _write_shim_scene(rt, spec, scene, tmp_path / "bad.bin")
    p = subprocess.run([exe, str(tmp_path / "bad.bin"), str(tmp_path / "bad.rgba")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "No trace algorithm found for 7" in p.stderr
