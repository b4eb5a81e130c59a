"""Pins the CPU oracle to the reference's own regression artefact (the only fixture the
reference holds for this path): samples_images/code_example.png with the tolerance of
raytracer_test.go:32-80, and the flat-colour known answers of that image."""
import copy
import os

import numpy as np
import pytest

from conftest import compare_images


@pytest.fixture(scope="module")
def oracle_c1(bridge, c1):
    spec, scene = c1
    osc = bridge.OracleScene(scene)
    acc, pix, ctr = osc.render(spec, mode=bridge.F64_DIJKSTRA)
    return osc, spec, acc, pix, ctr


def test_oracle_matches_reference_golden(oracle_c1, golden_rgba):
    _, _, _, pix, _ = oracle_c1
    different, allowed = compare_images(pix, golden_rgba)
    assert different < allowed, f"{different} pixels differ, reference allows < {allowed}"


def test_oracle_flat_colour_known_answers(oracle_c1, golden_rgba):
    """background / ambient-only PhongBlinn uint8(0.2*255.99)=51 / Standard frame uint8(150/255*255.99)=150:
    jitter-independent away from edges, so nearly every golden pixel of that exact colour is exact here too."""
    _, _, _, pix, _ = oracle_c1
    for colour, count in (((0, 0, 0), 62209), ((51, 0, 0), 34104), ((150, 111, 51), 16572)):
        gm = (golden_rgba[..., :3] == colour).all(-1)
        om = (pix[..., :3] == colour).all(-1)
        assert gm.sum() == count
        assert (gm & om).sum() >= 0.97 * count, (colour, int((gm & om).sum()))
        assert abs(int(om.sum()) - count) <= 0.01 * count
    mean = pix[..., :3].reshape(-1, 3).mean(0)
    assert np.allclose(mean, (55.8, 27.9, 33.2), atol=0.6)


def test_oracle_ray_budget(oracle_c1):
    """7.78 M primary rays, <= 32 shadow rays per PhongBlinn hit (2 positional lights x 16 reps)."""
    _, spec, _, _, ctr = oracle_c1
    assert ctr["rays_primary"] == 640 * 380 * 32
    assert ctr["rays_shadow"] == ctr["phong_hits"] * 2 * spec.SoftShadowMonteCarloRepetitions


def test_dfs_equals_dijkstra(bridge, c1):
    """The reference test compares its two traversals (raytracer_test.go:28-29); with the same Philox
    samples they must agree exactly."""
    spec, scene = c1
    s = copy.copy(spec)
    s.Width, s.Height, s.AntiAliasingFactor = 160, 95, 4
    with bridge.OracleScene(scene) as osc:
        a, pa, _ = osc.render(s, mode=bridge.F64_DIJKSTRA)
        b, pb, _ = osc.render(s, mode=bridge.F64_DFS)
    assert np.array_equal(pa, pb)
    assert np.abs(a - b).max() == 0.0


def test_f32_mode_tracks_f64(bridge, c1):
    spec, scene = c1
    s = copy.copy(spec)
    s.Width, s.Height, s.AntiAliasingFactor = 160, 95, 4
    with bridge.OracleScene(scene) as osc:
        a, pa, _ = osc.render(s, mode=bridge.F64_DIJKSTRA)
        b, pb, _ = osc.render(s, mode=bridge.F32_STACK)
    d = np.abs(pa.astype(int) - pb.astype(int))[..., :3].max(-1)
    assert (d > 2).mean() < 0.005  # float32 flips a texel / checker / Schlick decision on a few pixels


def test_trace_modes_agree_on_ids(bridge, c1):
    spec, scene = c1
    rng = np.random.default_rng(3)
    n = 4000
    px = np.stack([rng.integers(0, 640, n), rng.integers(0, 380, n), rng.integers(0, 32, n)], 1)
    with bridge.OracleScene(scene) as osc:
        rays = np.zeros((n, 8))
        rays[:, :6] = osc.camera_rays(spec, px)
        p0, t0 = osc.trace_rays(rays, mode=bridge.F64_DIJKSTRA)
        p1, t1 = osc.trace_rays(rays, mode=bridge.F64_DFS)
        p2, t2 = osc.trace_rays(rays, mode=bridge.F32_STACK)
    assert np.array_equal(p0, p1)
    assert (p0 != p2).mean() < 1e-3
    hit = (p0 >= 0) & (p0 == p2)
    assert np.allclose(t0[hit], t2[hit], rtol=1e-4)


def test_sphere_far_root_quirk(bridge, rt):
    """shape.go:64: a ray starting inside a sphere never hits it."""
    scene = rt.Scene(CameraLookFrom=rt.Vec(0, 0, -5), CameraLookAt=rt.Vec(0, 0, 0), CameraUp=rt.Vec(0, 1, 0), CameraFov=60,
                     Shapes=[rt.Sphere(rt.Vec(0, 0, 0), 1.0, rt.Standard(ColorFrac=rt.Vec(1, 0, 0)))], Lights=[])
    rays = np.array([[0, 0, 0, 0, 0, 1, 0, 0], [0, 0, -3, 0, 0, 1, 0, 0]], dtype=np.float64)
    with bridge.OracleScene(scene) as osc:
        for mode in (0, 1, 2):
            p, t = osc.trace_rays(rays, mode=mode)
            assert p[0] == -1 and p[1] == 0 and abs(t[1] - 2.0) < 1e-6


def test_scatter_known_answers(bridge, rt):
    """Hand-computed single-ray answers for M1/M2/M3 (material.go:34-70)."""
    V = rt.Vec
    spec = rt.ImageSpec(8, 8, 1, 4, 1, 1, 0)

    def one(mat, shape=None):
        sh = shape or rt.TrianglePlane(V(-1, -1, 0), V(1, -1, 0), V(0, 1, 0), False, mat)
        scene = rt.Scene(CameraLookFrom=V(0, 0, -5), CameraLookAt=V(0, 0, 0), CameraUp=V(0, 1, 0), CameraFov=60, Shapes=[sh], Lights=[])
        with bridge.OracleScene(scene) as osc:
            rc, out = osc.scatter_probe(spec, [0, 0, -2, 0, 0, 1])
        assert rc == 0
        return out

    out = one(rt.Standard(ColorFrac=V(0.25, 0.5, 0.75)))
    assert out[0] == 0 and np.allclose(out[10:13], (0.25, 0.5, 0.75)) and abs(out[13] - 2.0) < 1e-12
    # triangle normal = Unit((B-A) x (C-A)) = +z; the ray travels along +z so reflected.n = -1 < 0: no trace
    out = one(rt.Metal(Albedo=V(1, 1, 1), Fuzz=0))
    assert out[0] == 0 and np.allclose(out[17:20], (0, 0, 1))
    # flipped winding: normal -z, mirror reflection back along -z
    out = one(None, rt.TrianglePlane(V(-1, -1, 0), V(0, 1, 0), V(1, -1, 0), False, rt.Metal(Albedo=V(0.5, 0.6, 0.7), Fuzz=0)))
    assert out[0] == 1 and np.allclose(out[1:4], (0.5, 0.6, 0.7)) and np.allclose(out[7:10], (0, 0, -1)) and np.allclose(out[4:7], (0, 0, 0))
    # Dielectric at normal incidence from outside (d.n < 0): ratio 1/RI; straight through or straight back, origin offset 1e-5*dir
    out = one(None, rt.TrianglePlane(V(-1, -1, 0), V(0, 1, 0), V(1, -1, 0), False, rt.Dielectric(RefractiveIndex=1.5)))
    assert out[0] == 1 and np.allclose(out[1:4], (1, 1, 1)) and abs(abs(out[9]) - 1) < 1e-9 and abs(out[6] - 1e-5 * out[9]) < 1e-12


def test_oracle_reproduces_committed_vectors(bridge, c1, gold_dir):
    """tests/golden/c1_vectors.npz (SURVEY 8c i-iii: primary hit table + full-frame hash, mirror-bounce
    secondary hits, per-material scatter answers) was written by scripts/make_golden_vectors.py from the
    float64 best-first mode; the oracle must still produce exactly that, whatever changes in its fp32
    mirror of the device."""
    import importlib.util
    import os

    spec_ = importlib.util.spec_from_file_location("make_golden_vectors", os.path.join(os.path.dirname(gold_dir), "..", "scripts", "make_golden_vectors.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    want = np.load(os.path.join(gold_dir, "c1_vectors.npz"))
    spec, scene = c1
    with bridge.OracleScene(scene) as osc:
        got = mod.build(spec, osc)
    assert str(got["full_ids_sha256"]) == str(want["full_ids_sha256"])
    for k in ("pixels", "prim", "metal_index", "metal_secondary_prim"):
        assert np.array_equal(got[k], want[k]), k
    for k in ("rays", "t", "metal_secondary_rays", "metal_secondary_t", "scatter_probe_rays", "scatter_probe_out"):
        assert np.allclose(got[k], want[k], rtol=1e-12, atol=1e-12), k
    # reference semantics visible in the answers (material.go): Metal keeps the hit point as origin, Dielectric
    # steps 1e-5 along the new direction, Standard / PhongBlinn do not continue
    out, shapes = want["scatter_probe_out"], want["scatter_probe_rays"][:, 6].astype(int)
    for row, shape in zip(out, shapes):
        if shape in (3, 9, 10):      # Metal
            assert row[0] == 1.0 and np.array_equal(row[4:7], row[14:17]) and np.allclose(row[1:4], 1.0)
        elif shape == 1:             # Dielectric
            assert row[0] == 1.0 and np.allclose(row[4:7], row[14:17] + 1e-5 * row[7:10], atol=1e-15)
        else:                        # Standard, PhongBlinn: terminal colour
            assert row[0] == 0.0 and (row[10:13] >= 0).all() and (row[10:13] <= 1.0001).all()


def test_unit_sphere_sampler_is_uniform_in_the_ball(bridge):
    """randomInUnitSphere (material.go:134-143): cube rejection fed by one Philox block + xorshift128 retries.
    Device and oracle share that stream definition, so its distribution is pinned here on its own: moments
    and cell probabilities of the uniform unit ball."""
    n = 1_000_000
    p = bridge.sample_unit_sphere(n)
    r2 = (p * p).sum(1)
    assert r2.max() < 1.0
    tol = 5.0 / np.sqrt(n)  # ~5 sigma for O(1)-variance statistics
    assert np.abs(p.mean(0)).max() < tol
    assert abs(r2.mean() - 0.6) < tol and np.abs((p * p).mean(0) - 0.2).max() < tol
    assert np.abs(np.cov(p.T) - np.eye(3) * 0.2).max() < tol
    for radius in (0.25, 0.5, 0.75, 0.9):
        assert abs((r2 < radius * radius).mean() - radius ** 3) < tol
    octant = (p[:, 0] > 0) * 1 + (p[:, 1] > 0) * 2 + (p[:, 2] > 0) * 4
    assert np.abs(np.bincount(octant, minlength=8) / n - 0.125).max() < tol
    z = p[:, 2] / np.sqrt(r2)  # direction: cos(theta) uniform on [-1, 1]
    assert np.abs(np.histogram(z, bins=10, range=(-1, 1))[0] / n - 0.1).max() < tol


def test_oracle_under_sanitizers(tmp_path, gold_dir):
    """tests/cpp/oracle_check.cpp links the oracle source with AddressSanitizer and UBSan and drives every
    shape / material / light / texture kind through all three modes and both fp32 tree mirrors: a checker
    with undefined behaviour would be a poor checker.  The harness also asserts DFS == best-first and
    that the fp32 mirrors stay within tolerance of fp64."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(gold_dir))
    exe = tmp_path / "oracle_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "oracle_check.cpp"),
                           os.path.join(root, "oracle", "rtgo_oracle.cpp"), "-pthread", "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "oracle self-check ok" in p.stdout


def _stress_scene(seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_slabs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_slabs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make_scene(seed)


@pytest.mark.parametrize("seed", [2, 3, 10002, 10050, 20018])
def test_f32_walk_finds_what_all_shapes_give_far_from_the_origin(bridge, seed):
    """The float32 walk (F32_STACK: the device's box test fma(plane, 1 / d, -(o / d)) on the device's boxes) against NO tree
    at all (F32_ALL_SHAPES: the same shape tests over every shape), 5,000 to 10^5 units from the world's origin, on
    curved, exactly flat and far-camera scenes of scripts/stress_slabs.py: every nearest t and every occlusion answer
    equal.  That is what the boxes' pad (2^-19 of the scene's largest coordinate, fill_f32_boxes / flatten_octree) is
    for: the box test rounds o / d before it subtracts and is an ulp of the COORDINATES off, the shape tests subtract a
    vertex from the origin first and are not; without the pad the walk loses 1 hit in 1,000 at these distances."""
    spec, scene, kind, scale, off = _stress_scene(seed)
    rng = np.random.default_rng(seed)
    n_rays = 6000
    px = np.stack([rng.integers(0, spec.Width, n_rays), rng.integers(0, spec.Height, n_rays), rng.integers(0, spec.AntiAliasingFactor, n_rays)], 1).astype(np.int32)
    with bridge.OracleScene(scene) as osc:
        rays = np.zeros((n_rays, 8))
        rays[:, :6] = osc.camera_rays(spec, px, mode=bridge.F32_STACK).astype(np.float32)
        rays[:, 7] = 3.0e38
        ids, t = osc.trace_rays(rays, mode=bridge.F32_STACK)
        ids_all, t_all = osc.trace_rays(rays, mode=bridge.F32_ALL_SHAPES)
        assert (ids >= 0).sum() > n_rays // 4
        assert np.array_equal(ids >= 0, ids_all >= 0) and np.array_equal(t[ids >= 0], t_all[ids >= 0])  # (equal t from two shapes: either may be named)
        hit = ids >= 0
        p = (rays[hit, :3].astype(np.float32) + t[hit, None].astype(np.float32) * rays[hit, 3:6].astype(np.float32)).astype(np.float32)
        lights = [l for l in scene.Lights if hasattr(l, "Position")]
        L = np.array([[l.Position.X, l.Position.Y, l.Position.Z] for l in lights], np.float32)[rng.integers(0, len(lights), p.shape[0])]
        j = rng.normal(size=p.shape); j *= (rng.uniform(0, 1, (p.shape[0], 1)) ** (1 / 3)) / np.linalg.norm(j, axis=1, keepdims=True)
        d = ((L + np.float32(0.25) * j.astype(np.float32)).astype(np.float32) - p).astype(np.float32)
        dist = np.sqrt((d * d).sum(1, dtype=np.float32)).astype(np.float32)
        ok = dist > 0
        sh = np.zeros((int(ok.sum()), 8))
        sh[:, :3], sh[:, 3:6], sh[:, 6], sh[:, 7] = p[ok], d[ok] * (np.float32(1) / dist[ok])[:, None], np.float32(0.01), dist[ok]
        occ, _ = osc.trace_rays(sh, mode=bridge.F32_STACK, any_hit=True)
        occ_all, _ = osc.trace_rays(sh, mode=bridge.F32_ALL_SHAPES, any_hit=True)
        assert 0 < occ.sum() and np.array_equal(occ, occ_all), int((occ != occ_all).sum())
