"""Round-2 parity cases (VERDICT r01 "Next round" 1 and 8): the reference-faithful float64 best-first oracle
(F64_DIJKSTRA: accelerationstructures.go:81-124, material.go:73-132, every shadow ray traced) against the
device at the full mesh sizes of configs 2, 3 and 5; config 4 (Dragon scene, sample-partitioned over 8 ranks);
the device's material.scatter against the committed known answers; the binary-STL path on the GPU."""
import copy
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _f64_check(backend, bridge, spec, scene, name, n_rays=60000, id_rate=2e-4, px_frac=0.985):
    """Primitive IDs of camera rays (device fp32 vs the reference's float64 best-first traversal) and the radiance
    of a whole frame (fraction of pixels whose mean colour agrees within 1/255).  The two differ where float32
    and float64 disagree about a silhouette, a texel, or the visibility of a grazing shadow ray."""
    rng = np.random.default_rng(4)
    px = np.stack([rng.integers(0, spec.Width, n_rays), rng.integers(0, spec.Height, n_rays),
                   rng.integers(0, spec.AntiAliasingFactor, n_rays)], 1).astype(np.int32)
    with backend.DeviceScene(scene) as dev, bridge.OracleScene(scene) as osc:
        rays = np.zeros((n_rays, 8), np.float32)
        rays[:, :6] = dev.camera_rays(spec, px)
        ids, t = dev.trace_rays(rays)
        rays64 = np.zeros((n_rays, 8))
        rays64[:, :6] = osc.camera_rays(spec, px, mode=bridge.F64_DIJKSTRA)
        ids64, t64 = osc.trace_rays(rays64, mode=bridge.F64_DIJKSTRA)
        acc, st = dev.render_accum(spec)
        ref, _, ctr = osc.render(spec, mode=bridge.F64_DIJKSTRA, rgba=False)
    differ = ids != ids64
    same = ~differ & (ids >= 0)
    t_err = np.abs(t[same] / t64[same] - 1).max()
    spp = spec.AntiAliasingFactor
    d = np.abs(acc.astype(np.float64) - ref).max(-1) / spp
    within = (d <= 1 / 255).mean()
    print(f"{name}: shape ids differ on {differ.sum()} of {n_rays} rays, max relative t error {t_err:.2e}; "
          f"{100 * within:.3f} % of {d.size} pixels within 1/255 (max {d.max():.3f}); "
          f"paths {st.paths}, phong hits {st.phong_hits} (float64 {ctr['phong_hits']})")
    assert differ.mean() <= id_rate and t_err < 1e-4
    assert st.rays_primary == ctr["rays_primary"] == spec.Width * spec.Height * spp
    # the reference traces every shadow ray (light.go:124); the device's reference-equivalent count is that number
    assert abs(int(st.rays_shadow_reference_equivalent) - int(ctr["rays_shadow"])) <= 2e-3 * ctr["rays_shadow"]
    assert abs(int(st.phong_hits) - int(ctr["phong_hits"])) <= 2e-3 * ctr["phong_hits"]
    assert within >= px_frac
    return st


def test_config2_against_float64_reference(backend, bridge, scenes):
    """config 2, full Bunny stand-in (81,920 triangles + 2 floor triangles), soft shadows."""
    spec, scene = scenes.config2(160, 90, 2, 16)
    _f64_check(backend, bridge, spec, scene, "config 2")


def test_config3_against_float64_reference(backend, bridge, scenes, gold_dir):
    """config 3, full Dragon stand-in (871,210 shapes): the reference's insertion octree walked best-first in
    float64 against the device's SAH tree walked with a stack in float32 (quantised blocks for the shadow rays)."""
    spec, scene = scenes.config3(128, 72, 2, 16, asset_dir=gold_dir)
    st = _f64_check(backend, bridge, spec, scene, "config 3")
    assert st.rays_shadow < st.rays_shadow_reference_equivalent


def test_config5_against_float64_reference(backend, bridge, scenes, gold_dir):
    """config 5 with its full meshes (Dragon + Bunny stand-ins, 953,136 shapes), glass spheres and the Metal floor:
    deep paths, so one flipped Schlick / fuzz decision moves a pixel; the bar on pixels is looser."""
    spec, scene = scenes.config5(128, 72, 2, 8, asset_dir=gold_dir)
    st = _f64_check(backend, bridge, spec, scene, "config 5", px_frac=0.95)
    assert st.rays_continuation > 0.2 * st.rays_primary


def test_config4_sample_partition_of_the_dragon_scene(backend, bridge, scenes, rt, gold_dir):
    """config 4: the Dragon scene (all 871,210 shapes) rendered as 8 sample ranges (RT_PART_SAMPLES, world 8), the
    shares summed as the RCCL reduce would: equals the unpartitioned frame up to float32 summation order, every
    (pixel, sample) is rendered exactly once, and the frame still matches the float32 oracle."""
    abi = rt._abi
    spec, scene = scenes.config3(96, 54, 16, 16, asset_dir=gold_dir)
    world = 8
    with backend.DeviceScene(scene) as dev, bridge.OracleScene(scene) as osc:
        assert dev.info()["n_shapes"] == 871210
        whole, st = dev.render_accum(spec)
        parts = [dev.render_accum(spec, part=abi.partition(mode=abi.RT_PART_SAMPLES, rank=r, world_size=world)) for r in range(world)]
        ref, _, ctr = osc.render(spec, mode=bridge.F32_STACK, rgba=False)
    total = np.zeros_like(whole, dtype=np.float64)
    for acc, _ in parts:
        total += acc
    assert sum(p[1].paths for p in parts) == st.paths == spec.Width * spec.Height * spec.AntiAliasingFactor
    assert all(p[1].paths == spec.Width * spec.Height * 2 for p in parts)
    assert sum(p[1].rays_shadow for p in parts) == st.rays_shadow == ctr["rays_shadow"]
    # the same samples, summed in another order: float32 rounding of sums of at most 16 terms <= 1 each
    assert np.abs(total - whole).max() <= 16 * 2 ** -22
    d = np.abs(total - ref).max(-1)
    assert (d > 1e-3).mean() <= 2e-3, f"{(d > 1e-3).sum()} of {d.size} pixels off, max {d.max()}"


def test_device_scatter_matches_committed_answers(backend, c1, gold_dir):
    """material.scatter on the device (rt_scatter_probe -> k_scatter_probe: scatter_specular of k_shade, phong_term /
    shadow rays of k_phong_pairs / k_shadow / k_phong) against the committed float64 answers of one ray per shape of
    the README scene (tests/golden/c1_vectors.npz, written by scripts/make_golden_vectors.py; material.go:34-132):
    Standard + checkers, Dielectric, PhongBlinn, Metal sphere, PhongBlinn + image texture, floor, frame, mirrors.
    No oracle at run time."""
    v = np.load(os.path.join(gold_dir, "c1_vectors.npz"))
    spec, scene = c1
    probes, want = v["scatter_probe_rays"], v["scatter_probe_out"]
    keys = np.tile(np.array([[7, 3, 1]], np.uint32), (probes.shape[0], 1))  # pixel 7, sample 3, bounce 1 as in the script
    with backend.DeviceScene(scene) as dev:
        got = dev.scatter_probe(spec, probes[:, :6], keys).astype(np.float64)
        ids, _ = dev.trace_rays(np.concatenate([probes[:, :6], np.zeros((probes.shape[0], 2))], 1))
    assert np.array_equal(ids, probes[:, 6].astype(np.int32))
    kinds = set()
    for row, exp, shape in zip(got, want, probes[:, 6].astype(int)):
        assert row[0] == exp[0], (shape, row[0], exp[0])                               # shouldTrace
        assert np.allclose(row[13:20], exp[13:20], rtol=2e-5, atol=2e-5), shape          # t, p, normal
        assert np.allclose(row[1:4], exp[1:4], rtol=1e-6, atol=1e-6), shape              # attenuation
        if exp[0] == 1.0 or shape in (3, 9, 10):                                         # Metal / Dielectric: the scattered ray
            assert np.allclose(row[4:10], exp[4:10], rtol=5e-5, atol=5e-5), (shape, row[4:10], exp[4:10])
            kinds.add("specular")
        else:                                                                            # Standard / PhongBlinn: terminal colour
            assert np.allclose(row[10:13], exp[10:13], rtol=1e-4, atol=1e-4), (shape, row[10:13], exp[10:13])
            kinds.add("terminal")
    assert kinds == {"specular", "terminal"} and got.shape[0] >= 8


def test_binary_stl_path_on_the_device(backend, scenes, rt, tmp_path, monkeypatch):
    """SURVEY 8f row 2 (go.mod:6 lists an STL reader): the Bunny stand-in written as a binary STL, named by
    RTGO_BUNNY_STL, read back by stl.py and rendered as config 2 — the same shape IDs, t and frame as the
    in-memory mesh narrowed to the float32 an STL file holds."""
    stl = __import__("importlib").import_module("raytracer-go_amd.stl")
    tri = scenes.icosphere_mesh(6)
    path = tmp_path / "bunny_standin.stl"
    stl.save_binary_stl(str(path), tri)
    assert os.path.getsize(path) == 84 + 50 * tri.shape[0]
    spec, scene_mem = scenes.config2(160, 90, 2, 4)
    for s in scene_mem.Shapes:  # the in-memory mesh with float32 vertices, through the same _fit
        if isinstance(s, rt.TriangleMesh):
            s.vertices = scenes._fit(tri.astype(np.float32).astype(np.float64), (0.0, 0.0, 0.0), 4.0)
    monkeypatch.setenv("RTGO_BUNNY_STL", str(path))
    _, scene_stl = scenes.config2(160, 90, 2, 4)
    monkeypatch.delenv("RTGO_BUNNY_STL")
    rng = np.random.default_rng(9)
    px = np.stack([rng.integers(0, spec.Width, 20000), rng.integers(0, spec.Height, 20000), rng.integers(0, 2, 20000)], 1).astype(np.int32)
    out = []
    for scene in (scene_mem, scene_stl):
        with backend.DeviceScene(scene) as dev:
            assert dev.info()["n_shapes"] == 81922
            rays = np.zeros((px.shape[0], 8), np.float32)
            rays[:, :6] = dev.camera_rays(spec, px)
            ids, t = dev.trace_rays(rays)
            acc, st = dev.render_accum(spec)
            out.append((ids, t, acc, st.rays_shadow))
    assert (out[0][0] >= 2).mean() > 0.1  # the mesh is on screen
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]


def test_multi_device_frame_behind_one_call(backend, rt, c1, hip_buffers):
    """rt_multi_*: one blocking call renders a frame on a list of devices (one host thread and one scene replica per
    share, RCCL reduce onto the first device, resolve there).  One share must equal rt_render byte for byte; a tile
    split reassembles bit-exactly, a sample split up to the order of the float32 sums (at most one step of the
    RGBA8 quantiser).  With a single GPU the shares sit on device 0 and are summed there; with two or more visible
    the same frames go over RCCL."""
    import ctypes

    abi = rt._abi
    spec, scene = rt.ExampleRegression(320, 190, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    spec.AntiAliasingFactor = 8
    with backend.DeviceScene(scene) as dev:
        want, st1 = dev.render(spec)
    n_dev = ctypes.c_int()
    assert hip_buffers.hip.hipGetDeviceCount(ctypes.byref(n_dev)) == 0 and n_dev.value >= 1
    lists = [[0], [0, 0], [0, 0, 0]] + ([[0, 1]] if n_dev.value >= 2 else []) + ([list(range(min(8, n_dev.value)))] if n_dev.value >= 3 else [])
    for devices in lists:
        with backend.MultiDeviceScene(scene, devices) as multi:
            for mode in (abi.RT_PART_TILES, abi.RT_PART_SAMPLES):
                got, st = multi.render(spec, mode=mode)
                assert st.paths == st1.paths and st.rays_shadow == st1.rays_shadow and st.rays_continuation == st1.rays_continuation
                if mode == abi.RT_PART_TILES or len(devices) == 1:
                    assert np.array_equal(got, want), (devices, mode)
                else:
                    d = np.abs(got.astype(int) - want.astype(int))
                    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (devices, d.max(), (d > 0).mean())
    with pytest.raises(backend.RtgoError):
        backend.MultiDeviceScene(scene, [0, 99])


def test_render_lanes_do_not_change_the_frame(backend, rt, c1, monkeypatch):
    """A render call deals its pixel batches out over up to four host threads and streams (rtgo_hip.hip Lane); a pixel's
    batches stay on one lane in the order of the single-lane loop, so the accumulation buffer and the counters are
    bit-identical for RTGO_LANES=1, 2 and 4: with many small batches, with more than 64 samples (several sample rounds
    per pixel), and with a single batch (one lane has nothing to do)."""
    abi = rt._abi
    spec, scene = c1
    spec = copy.copy(spec)
    spec.Width, spec.Height = 200, 120
    cases = [(8, 16, 8 * 16 * 16 * 3), (72, 8, 64 * 8 * 8 * 5), (4, 64, 0)]  # (spp, tile, paths per batch)
    with backend.DeviceScene(scene) as dev:
        for spp, tile, max_paths in cases:
            spec.AntiAliasingFactor = spp
            out = {}
            for lanes in ("1", "2", "4"):
                monkeypatch.setenv("RTGO_LANES", lanes)
                out[lanes] = dev.render_accum(spec, part=abi.partition(tile_size=tile, max_paths=max_paths))
            monkeypatch.delenv("RTGO_LANES")
            a1, s1 = out["1"]
            for lanes in ("2", "4"):
                a2, s2 = out[lanes]
                assert np.array_equal(a1, a2), (spp, tile, lanes, np.abs(a1 - a2).max())
                for k in ("paths", "rays_primary", "rays_continuation", "rays_shadow", "rays_shadow_beam", "phong_hits"):
                    assert getattr(s1, k) == getattr(s2, k), (k, lanes)
            assert s1.paths == spec.Width * spec.Height * spp


def test_pair_beams_decide_the_same_rays(backend, rt, scenes, c1, gold_dir, monkeypatch):
    """k_pair_beam proves all the soft-shadow repetitions of a (hit, light) pair unoccluded with one conservative walk
    (device_math.hpp Beam) and such pairs' rays are never generated.  With RTGO_BEAM=0 every ray is traced: the
    accumulation buffer and every counter must be identical, and the beams must decide a good part of the rays
    (README scene: spheres over a floor; config 3: the mesh in its room; config 2 with point-like reps = 1)."""
    spec1, scene1 = c1
    spec1 = copy.copy(spec1)
    spec1.Width, spec1.Height, spec1.AntiAliasingFactor = 200, 120, 4
    cases = [("config 1", spec1, scene1, 0.05)]
    spec3, scene3 = scenes.config3(160, 90, 4, 16, asset_dir=gold_dir)
    cases.append(("config 3", spec3, scene3, 0.2))
    spec2, scene2 = scenes.config2(160, 90, 2, 1)
    cases.append(("config 2, one repetition", spec2, scene2, 0.05))
    for name, spec, scene, least in cases:
        with backend.DeviceScene(scene) as dev:
            monkeypatch.setenv("RTGO_BEAM", "1")  # forced: by default scenes of fewer than 256 shapes go without
            got, st = dev.render_accum(spec)
            monkeypatch.setenv("RTGO_BEAM", "0")
            want, st0 = dev.render_accum(spec)
            monkeypatch.delenv("RTGO_BEAM")
        print(f"{name}: {st.rays_shadow_beam} of {st.rays_shadow} shadow rays decided by beams")
        assert st0.rays_shadow_beam == 0
        assert np.array_equal(got, want), (name, np.abs(got - want).max())
        for k in ("paths", "rays_primary", "rays_continuation", "rays_shadow", "phong_hits", "rays_shadow_reference_equivalent"):
            assert getattr(st, k) == getattr(st0, k), (name, k)
        assert st.rays_shadow_beam >= least * st.rays_shadow, name


def test_pair_beams_on_random_scenes(backend, rt, monkeypatch):
    """Beams and the resumed walks against plain per-ray tracing (RTGO_BEAM=0) on seeded random scenes: triangle
    soup + spheres around the origin or 1000 units away from it, point and spot lights in the open, among the shapes
    and right above a surface, 1 to 130 repetitions, a mirror so that later bounces shade too.  The accumulation
    buffers and the counters must be identical."""
    api = rt
    total_beam = 0
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        off = np.array([0.0, 0.0, 0.0]) if seed % 2 == 0 else np.array([1000.0, -500.0, 250.0])
        n_tri = 400 + 300 * seed
        c = rng.uniform(-3, 3, (n_tri, 1, 3)) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 2.0, 0.0])
        tri = c + rng.normal(0, 0.08 + 0.05 * seed, (n_tri, 3, 3)) + off
        mat = api.PhongBlinn(ColorFrac=api.Vec(0.8, 0.7, 0.6), SpecularColorFrac=api.Vec(1, 1, 1), SpecHardness=16)
        floor = api.PhongBlinn(ColorFrac=api.Vec(0.6, 0.6, 0.6), SpecularColorFrac=api.Vec(0.3, 0.3, 0.3), SpecHardness=4)
        R = 12.0
        o = off
        V = lambda x, y, z: api.Vec(float(x + o[0]), float(y + o[1]), float(z + o[2]))
        shapes = [api.TrianglePlane(V(-R, 0, -R), V(-R, 0, R), V(R, 0, -R), False, floor),
                  api.TrianglePlane(V(R, 0, R), V(R, 0, -R), V(-R, 0, R), False, floor),
                  api.TrianglePlane(V(-R, 0, R), V(-R, 8, R), V(R, 0, R), True, api.Metal(api.Vec(0.9, 0.9, 0.9), 0.0)),
                  api.TriangleMesh(tri, False, mat)]
        for k in range(8):
            p = rng.uniform(-3, 3, 3)
            shapes.append(api.Sphere(V(p[0], abs(p[1]) + 0.5, p[2]), float(rng.uniform(0.2, 0.7)), mat))
        lights = [api.AmbientLight(api.Vec(1, 1, 1), 0.05),
                  api.PointLight(ColorFrac=api.Vec(1, 1, 1), Position=V(4, 7, -5), LightIntensity=60, SpecularLightIntensity=20,
                                 InverseSquareLawDecayFactor=0.5),
                  api.PointLight(ColorFrac=api.Vec(1, 0.8, 0.8), Position=V(0.3, 2.0, 0.2), LightIntensity=10, SpecularLightIntensity=5,
                                 InverseSquareLawDecayFactor=0.5),  # among the triangles
                  api.SpotLight(ColorFrac=api.Vec(0.8, 0.9, 1), Position=V(-5, 0.3, -4), LightIntensity=40, SpecularLightIntensity=10,
                                LookAt=V(0, 1, 0), Angle=35.0, InverseSquareLawDecayFactor=0.3)]  # 0.3 above the floor
        reps = (1, 5, 16, 70, 3, 130)[seed % 6]  # 70, 130: more repetitions than the 64 lanes of a pair's group
        spec = api.ImageSpec(160, 100, 3, 6, reps, 16, api.Dijkstra)
        scene = api.Scene(V(0, 3.5, -9), V(0, 1.5, 0), api.Vec(0, 1, 0), V(0, 1.5, 0), 0.02, 50.0, shapes, lights)
        with backend.DeviceScene(scene) as dev:
            monkeypatch.setenv("RTGO_BEAM", "1")
            got, st = dev.render_accum(spec)
            monkeypatch.setenv("RTGO_RESUME", "0")
            mid, st_mid = dev.render_accum(spec)
            monkeypatch.delenv("RTGO_RESUME")
            monkeypatch.setenv("RTGO_BEAM", "0")
            want, st0 = dev.render_accum(spec)
            monkeypatch.delenv("RTGO_BEAM")
        assert np.array_equal(got, want) and np.array_equal(mid, want), (seed, np.abs(got - want).max(), np.abs(mid - want).max())
        for k in ("paths", "rays_primary", "rays_continuation", "rays_shadow", "phong_hits"):
            assert getattr(st, k) == getattr(st0, k) == getattr(st_mid, k), (seed, k)
        assert st.rays_continuation > 0 and st0.rays_shadow_beam == 0
        total_beam += st.rays_shadow_beam
    assert total_beam > 0


def test_two_host_threads_create_and_render_at_once(backend, rt, c1):
    """The library's process-wide state (the runtime warm-up that the first rt_init of a device starts, the static
    workspace's address, the thread-local error text) under two callers at once: each thread creates its own scene of
    the reference's example and renders it — ctypes releases the GIL for the calls — and both get the bytes of a frame
    rendered alone; a failing call on one thread leaves its text on that thread only."""
    import threading
    spec, scene = c1
    spec = copy.copy(spec)
    spec.Width, spec.Height, spec.AntiAliasingFactor = 160, 96, 4
    with backend.DeviceScene(scene) as dev:
        alone, _ = dev.render(spec)
    frames, errors = {}, {}
    start = threading.Barrier(2)

    def work(i):
        try:
            start.wait(timeout=60)
            for _ in range(2):
                with backend.DeviceScene(scene) as dev:
                    frames[i], _ = dev.render(spec)
            if i == 1:  # an error on this thread: the other thread's calls go on
                bad = copy.copy(spec)
                bad.BvhTraversalAlgorithm = 7
                with backend.DeviceScene(scene) as dev:
                    with pytest.raises(Exception, match="No trace algorithm found for 7"):
                        dev.render(bad)
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors[i] = e

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=170)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    for i in range(2):
        assert np.array_equal(frames[i], alone), i


def test_full_size_frame_equals_the_plain_path(backend, scenes, monkeypatch):
    """BASELINE's headline frame at its full size (config 3: 871,210 shapes, 1920x1080, 64 spp, 16 repetitions): what
    the default path renders — pair beams, rays resumed from their beams, leaf slabs, four lanes — is bit for bit the
    accumulation buffer of the plain path (every shadow ray walked from the root, every shape of every leaf whose box it
    meets tested, one lane), with the same ray counts."""
    spec, scene = scenes.config3(1920, 1080, 64, 16)
    keys = ("paths", "rays_primary", "rays_continuation", "rays_shadow", "phong_hits")
    with backend.DeviceScene(scene) as dev:
        a, sa = dev.render_accum(spec)
        monkeypatch.setenv("RTGO_BEAM", "0")
        monkeypatch.setenv("RTGO_LANES", "1")
        monkeypatch.setenv("RTGO_SLAB", "0")
        b, sb = dev.render_accum(spec)
    assert sa.rays_shadow_beam > 0.4 * sa.rays_shadow and sb.rays_shadow_beam == 0
    assert sa.paths == 1920 * 1080 * 64
    for k in keys:
        assert getattr(sa, k) == getattr(sb, k), k
    assert np.array_equal(a, b), float(np.abs(a - b).max())
