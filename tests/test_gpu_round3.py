"""Round-3 cases (VERDICT r02 "Next round" 1, 5, 6, 7): the multi-GPU call's error path, bench.py's launchers, a
scene far from the origin against the reference-faithful float64 traversal, the RMSE gate of the headline config."""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multi_share_failure_returns_an_error(backend, rt, c1, monkeypatch):
    """A share that fails must fail the whole rt_multi_render call — no rank may be left waiting in the reduce
    (the RCCL route hung when one share returned before ncclReduce).  The shares render first, all host
    threads are joined, and only then is the collective enqueued; here share 1 of 3 is made to fail."""
    spec, scene = copy.copy(c1[0]), c1[1]
    spec.Width, spec.Height, spec.AntiAliasingFactor = 160, 95, 2
    abi = rt._abi
    with backend.MultiDeviceScene(scene, [0, 0, 0]) as m:
        info = m.info()
        assert info == {"n_shares": 3, "n_distinct_devices": 1, "rccl_ranks": 0}
        good, st = m.render(spec, mode=abi.RT_PART_TILES)
        monkeypatch.setenv("RTGO_FAIL_SHARE", "1")
        with pytest.raises(backend.RtgoError) as e:
            m.render(spec, mode=abi.RT_PART_TILES)
        assert e.value.code == abi.RT_ERR_DEVICE and "share 1" in str(e.value) and "injected" in str(e.value)
        monkeypatch.delenv("RTGO_FAIL_SHARE")
        again, st2 = m.render(spec, mode=abi.RT_PART_TILES)  # the handle is still usable, and the frame is the same
        assert (again == good).all() and st2.rays == st.rays


def _bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", *args], capture_output=True, text=True, timeout=300)
    return p


def test_bench_launchers_agree_on_one_gpu():
    """`python bench.py --gpus N` needs no outside launcher: without WORLD_SIZE it drives rt_multi_* in-process.  At
    N = 1 that path and the single-device path must report the same rays and, within noise, the same frame time."""
    common = ("--gpus", "1", "--width", "640", "--height", "360", "--spp", "16", "--steps", "3", "--warmup", "1")
    a = _bench(*common)
    b = _bench(*common, "--launcher", "rt_multi")
    assert a.returncode == 0, a.stderr[-2000:]
    assert b.returncode == 0, b.stderr[-2000:]
    ja, jb = json.loads(a.stdout.strip().splitlines()[-1]), json.loads(b.stdout.strip().splitlines()[-1])
    assert ja["launcher"] == "single" and jb["launcher"] == "rt_multi" and jb["rccl_ranks"] == 0
    assert ja["rays_per_frame"] == jb["rays_per_frame"]
    assert ja["n_gpus"] == jb["n_gpus"] == 1
    # value counts traversals launched: rays walked one by one + one per pair-beam walk; the rays a beam
    # walk settled are only in value_rays_decided
    for j in (ja, jb):
        rp = j["rays_per_frame"]
        launched = rp["primary"] + rp["continuation"] + rp["shadow"] - rp["of_which_decided_by_pair_beams"] + rp["pair_beam_walks"]
        assert 0 < rp["pair_beam_walks"] <= rp["shadow"]
        assert abs(j["value"] * 1e6 * j["sec_per_frame"] - launched) <= 1e-6 * launched
        assert j["value_rays_decided"] >= j["value"]
    print("single", ja["ms_per_step"], "rt_multi", jb["ms_per_step"])
    assert abs(ja["ms_per_step"] - jb["ms_per_step"]) <= 0.15 * ja["ms_per_step"] + 1.0  # + the 0.9 MB frame's copy to the host


def test_bench_refuses_more_gpus_than_visible():
    """Fewer visible devices than --gpus: a message and a non-zero exit, never shares that double up on a device."""
    p = _bench("--gpus", "64", "--steps", "1")
    assert p.returncode != 0 and "visible" in p.stderr and not p.stdout.strip()
    env = dict(os.environ, WORLD_SIZE="64", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "visible" in p.stderr and not p.stdout.strip()


def test_bench_tile_mode_config5_small():
    """`--config 5 --mode tiles`: the other partition north_star names, through the same command."""
    p = _bench("--gpus", "1", "--config", "5", "--mode", "tiles", "--width", "640", "--height", "360", "--spp", "4", "--steps", "1", "--warmup", "0")
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["config"]["workload"].startswith("config5") and j["rays_per_frame"]["continuation"] > 0
    assert j["roofline"]["stale"] is True and j["roofline"]["frac"] is None  # the committed counts are the headline workload's


@pytest.mark.parametrize("offset", [(1e3, -5e2, 2.5e2), (1e4, 1e4, -1e4)])
def test_translated_scene_against_float64_reference(backend, bridge, scenes, rt, offset):
    """ADVICE r01-3 / VERDICT r02 weak 1b: the device's slab test is fma(box, 1/d, -(o/d)) with o/d rounded first,
    which loses bits when |o| >> the box (accelerationstructures.go:356-417 subtracts before it divides).  Config 2's
    mesh moved far from the origin, 60,000 camera rays: shape ids against the float64 best-first traversal."""
    spec, scene = scenes.config2(160, 90, 2, 4)
    off = rt.Vec(*offset)
    f32 = lambda x: float(np.float32(x))
    for s in scene.Shapes:
        s.Translate(off)
        # both sides get the geometry the device computes on: coordinates narrowed to float32 (rtgo.h), so that what
        # is compared is the traversal's arithmetic and not the rounding of the vertices at this distance from 0
        if isinstance(s, rt.TriangleMesh):
            s.vertices = s.vertices.astype(np.float32).astype(np.float64)
        else:
            for n in ("PointA", "PointB", "PointC"):
                v = getattr(s, n)
                setattr(s, n, rt.Vec(f32(v.X), f32(v.Y), f32(v.Z)))
    for l in scene.Lights:
        if hasattr(l, "Position"):
            l.Position = rt.Vec(l.Position.X + off.X, l.Position.Y + off.Y, l.Position.Z + off.Z)
    for name in ("CameraLookFrom", "CameraLookAt", "CameraFocusPoint"):
        v = getattr(scene, name)
        setattr(scene, name, rt.Vec(v.X + off.X, v.Y + off.Y, v.Z + off.Z))
    n_rays = 60000
    rng = np.random.default_rng(9)
    px = np.stack([rng.integers(0, spec.Width, n_rays), rng.integers(0, spec.Height, n_rays),
                   rng.integers(0, spec.AntiAliasingFactor, n_rays)], 1).astype(np.int32)
    with backend.DeviceScene(scene) as dev, bridge.OracleScene(scene) as osc:
        rays = np.zeros((n_rays, 8), np.float32)
        rays[:, :6] = dev.camera_rays(spec, px)
        ids, t = dev.trace_rays(rays)
        # the SAME float32 rays through the float64 traversal: what differs is the traversal's arithmetic alone
        ids64, t64 = osc.trace_rays(rays.astype(np.float64), mode=bridge.F64_DIJKSTRA)
        # shadow-style any-hit rays from the hit points towards the light: the quantised walk, same comparison
        hit = ids >= 0
        p = rays[hit, :3] + t[hit, None] * rays[hit, 3:6]
        lp = np.array([l.Position.X for l in scene.Lights if hasattr(l, "Position")][:1] + [l.Position.Y for l in scene.Lights if hasattr(l, "Position")][:1]
                      + [l.Position.Z for l in scene.Lights if hasattr(l, "Position")][:1], np.float32)
        d = lp[None, :] - p
        dist = np.linalg.norm(d, axis=1).astype(np.float32)
        sh = np.zeros((hit.sum(), 8), np.float32)
        sh[:, :3], sh[:, 3:6], sh[:, 6], sh[:, 7] = p, d / dist[:, None], 0.01, dist
        occ, _ = dev.trace_rays(sh, any_hit=True)
        # doesReachLight (light.go:141-160): nearest hit from tMin = 0.01, occluded iff it lies before the light
        near64, tn64 = osc.trace_rays(sh.astype(np.float64), mode=bridge.F64_DIJKSTRA)
        occ64 = ((near64 >= 0) & (tn64 <= sh[:, 7])).astype(np.int32)
    differ = ids != ids64
    # the failure a too-tight slab test produces: a hit the device walks past — it then misses, or finds a farther shape
    both = (ids >= 0) & (ids64 >= 0)
    lost = ((ids < 0) & (ids64 >= 0)) | (both & differ & (t > t64 * (1 + 1e-3)))
    occ_differ = occ != occ64
    lost_occ = (occ == 0) & (occ64 == 1)
    print(f"offset {offset}: shape ids differ on {differ.sum()} of {n_rays} rays ({lost.sum()} of them lost nearer hits); "
          f"any-hit differs on {occ_differ.sum()} of {occ.size} ({lost_occ.sum()} lost occluders, {((occ == 1) & (occ64 == 0)).sum()} gained)")
    scale = max(abs(x) for x in offset)
    # float32 spacing at the offset (6e-5 at 1e3, 1e-3 at 1e4) against triangles of ~0.03: rays within that of an edge flip
    limit = 2e-4 if scale <= 1e3 else 2e-3
    assert differ.mean() <= limit, differ.mean()
    assert lost.mean() <= limit / 4, lost.mean()
    assert occ_differ.mean() <= 2 * limit and lost_occ.mean() <= limit, (occ_differ.mean(), lost_occ.mean())


@pytest.mark.parametrize("name", ["c3", "c5"])
def test_rmse_against_converged_reference_full_meshes(backend, scenes, gold_dir, name):
    """north_star: "Stochastic configs are checked by RMSE against a converged reference image" (the reference's own
    reason: raytracer_test.go:35-38, its renderer is not reproducible).  The headline scene (config 3, all 871,210
    shapes) and config 5 (953,136 shapes, glass and a Metal floor) at 96x54 against a committed 1024-spp image of the
    float64 best-first oracle — the reference's insertion octree, every shadow ray traced, a seed and a sampler state
    the device never uses (scripts/make_converged_fixture.py).  A bias shared through the sampler spec would not
    average out: the RMSE must fall as 1/sqrt(spp) and the mean must agree."""
    path = os.path.join(gold_dir, f"{name}_converged_96x54.npy")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    ref = np.load(path).astype(np.float64)
    make = scenes.config3 if name == "c3" else scenes.config5
    spec, scene = make(96, 54, 16, 16, asset_dir=gold_dir)
    with backend.DeviceScene(scene) as dev:
        rmse = {}
        for spp in (16, 256):
            s = copy.copy(spec)
            s.AntiAliasingFactor = spp
            acc, _ = dev.render_accum(s, seed=1234)
            img = acc.astype(np.float64) / spp
            rmse[spp] = float(np.sqrt(((np.minimum(img, 1) - np.minimum(ref, 1)) ** 2).mean()))
        bias = float(np.abs((img - ref).mean()))
    # the converged image's own noise (1024 spp) is half of the 256-spp render's, so the ratio is sqrt((1/256 + 1/1024) / (1/16 + 1/1024)) = 0.28
    print(name, "rmse", rmse, "ratio", rmse[256] / rmse[16], "mean bias", bias)
    assert rmse[256] < 0.45 * rmse[16]
    assert bias < 2e-3


def test_headline_frame_size_against_float64_reference(backend, bridge, scenes, gold_dir):
    """the float64 reference-faithful checks ran at 128x72; at the headline's frame size the only
    check was the device against itself.  Here: config 3 at 1920x1080, 1 spp, the middle 540 rows — 1.04 M paths, their
    50 M reference-equivalent shadow rays — device against the oracle's F64_DIJKSTRA mode (the reference's insertion
    octree walked best-first in float64, every shadow ray traced): fraction of pixels whose colour agrees within
    1/255, PhongBlinn hits and reference-equivalent rays of those rows."""
    spec, scene = scenes.config3(1920, 1080, 1, 16, asset_dir=gold_dir)
    r0, r1 = 270, 810
    with backend.DeviceScene(scene) as dev, bridge.OracleScene(scene) as osc:
        acc, st = dev.render_accum(spec)
        ref, _, ctr = osc.render(spec, mode=bridge.F64_DIJKSTRA, rows=(r0, r1), rgba=False)
        # the same rows through the float32 mirror: ray for ray what the device does
        ref32, _, ctr32 = osc.render(spec, mode=bridge.F32_STACK, rows=(r0, r1), rgba=False)
    d = np.abs(acc[r0:r1].astype(np.float64) - ref[r0:r1]).max(-1)
    within = (d <= 1 / 255).mean()
    d32 = np.abs(acc[r0:r1].astype(np.float64) - ref32[r0:r1]).max(-1)
    print(f"1920x1080 rows {r0}..{r1}: {100 * within:.3f} % of {d.size} pixels within 1/255 of the float64 frame (max {d.max():.3f}); "
          f"against the float32 mirror: {100 * (d32 <= 1e-3).mean():.4f} % within 1e-3 (max {d32.max():.2e}); "
          f"float64 phong hits {ctr['phong_hits']}, shadow rays {ctr['rays_shadow']}")
    assert ctr["rays_primary"] == (r1 - r0) * spec.Width
    assert within >= 0.985
    assert (d32 <= 1e-3).mean() >= 0.999


def test_leaf_slabs_change_no_answer(backend, rt, scenes, gold_dir, monkeypatch):
    """Every leaf has a slab besides its box (device_math.hpp "leaf slabs"): a ray tests a leaf's shapes only where it
    is inside the box AND between the slab's planes, and a pair's beam goes on past a leaf whose slab it misses.  Both
    only ever say "cannot hit", so with RTGO_SLAB=0 (no slabs: every shape of every leaf met is tested) the
    accumulation buffer and every counter but the beams' share must be identical: on config 3 and config 5 (curved
    meshes, glass, a mirror), and on seeded scenes built to strain the margins — smooth bumpy sheets (thin slabs) seen at
    grazing angles, at the origin and 1000 / 10^4 units away, with spheres, slivers and zero-area triangles among the
    leaves; explicit any-hit rays (rt_trace_rays) through the big mesh answer alike as well.  The same scenes with
    RTGO_VOXEL=0: pairs that their cell's voxel beam (one conservative walk per light and grid cell, at scene creation)
    would settle get a beam walk of their own instead, and nothing else changes."""
    api = rt
    cases = []
    spec3, scene3 = scenes.config3(240, 135, 4, 16, asset_dir=gold_dir)
    cases.append(("config 3", spec3, scene3))
    spec5, scene5 = scenes.config5(240, 135, 2, 16)
    cases.append(("config 5", spec5, scene5))
    for seed, off in enumerate(([0.0, 0.0, 0.0], [1000.0, -500.0, 250.0], [1e4, 1e4, -1e4])):
        rng = np.random.default_rng(700 + seed)
        o = np.array(off)
        n = 96
        u, v = np.meshgrid(np.linspace(-3, 3, n), np.linspace(-3, 3, n), indexing="ij")
        y = 1.2 + 0.25 * np.sin(1.7 * u) * np.cos(1.3 * v) + 0.02 * rng.normal(size=u.shape)
        P = np.stack([u, y, v], axis=-1) + o
        a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
        tri = np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 3, 3), np.stack([d, c, b], axis=-2).reshape(-1, 3, 3)])
        tri = np.concatenate([tri, np.repeat(P[5, 5][None, None, :], 3, axis=1) + np.zeros((4, 3, 3)),  # zero-area triangles
                              (P[10, 10] + np.array([[0, 0.5, 0], [1e-4, 0.5, 0], [0, 1.5, 1e-4]]))[None]])  # a sliver
        mat = api.PhongBlinn(ColorFrac=api.Vec(0.8, 0.7, 0.6), SpecularColorFrac=api.Vec(1, 1, 1), SpecHardness=16)
        floor = api.PhongBlinn(ColorFrac=api.Vec(0.6, 0.6, 0.6), SpecularColorFrac=api.Vec(0.3, 0.3, 0.3), SpecHardness=4)
        V = lambda x, y_, z: api.Vec(float(x + o[0]), float(y_ + o[1]), float(z + o[2]))
        R = 9.0
        shapes = [api.TrianglePlane(V(-R, 0, -R), V(-R, 0, R), V(R, 0, -R), False, floor),
                  api.TrianglePlane(V(R, 0, R), V(R, 0, -R), V(-R, 0, R), False, floor),
                  api.TriangleMesh(tri, False, mat)]
        for k in range(6):
            p = rng.uniform(-2.5, 2.5, 3)
            shapes.append(api.Sphere(V(p[0], 1.3 + 0.1 * k, p[2]), float(rng.uniform(0.05, 0.3)), mat))
        lights = [api.AmbientLight(api.Vec(1, 1, 1), 0.05),
                  api.PointLight(ColorFrac=api.Vec(1, 1, 1), Position=V(6, 2.0, -5), LightIntensity=60, SpecularLightIntensity=20,
                                 InverseSquareLawDecayFactor=0.5),  # low: its rays graze the sheet
                  api.SpotLight(ColorFrac=api.Vec(0.8, 0.9, 1), Position=V(-1, 6, 1), LightIntensity=40, SpecularLightIntensity=10,
                                LookAt=V(0, 1, 0), Angle=40.0, InverseSquareLawDecayFactor=0.3)]
        spec = api.ImageSpec(200, 120, 3, 4, 16, 16, api.Dijkstra)
        scene = api.Scene(V(0, 2.2, -7), V(0, 1.2, 0), api.Vec(0, 1, 0), V(0, 1.2, 0), 0.02, 50.0, shapes, lights)
        cases.append((f"sheet at {off}", spec, scene))
    keys = ("paths", "rays_primary", "rays_continuation", "rays_shadow", "phong_hits", "rays_shadow_reference_equivalent")
    for name, spec, scene in cases:
        with backend.DeviceScene(scene) as dev:
            monkeypatch.setenv("RTGO_BEAM", "1")
            got, st = dev.render_accum(spec)
            monkeypatch.setenv("RTGO_SLAB", "0")
            want, st0 = dev.render_accum(spec)
            monkeypatch.setenv("RTGO_BEAM", "0")
            plain, st1 = dev.render_accum(spec)
            monkeypatch.delenv("RTGO_SLAB")
            slab_only, st2 = dev.render_accum(spec)  # slabs in the rays' shape loop, no beams
            monkeypatch.setenv("RTGO_BEAM", "1")
            monkeypatch.setenv("RTGO_VOXEL", "0")
            no_vox, st3 = dev.render_accum(spec)  # pair beams, but no pair settled by its cell's voxel beam
            monkeypatch.delenv("RTGO_VOXEL")
            monkeypatch.delenv("RTGO_BEAM")
            if name == "config 3":  # 200,000 explicit any-hit rays through the big mesh, with and without slabs
                rng = np.random.default_rng(5)
                o = rng.uniform(-2.0, 2.0, (200_000, 3)) + np.array([0.0, 1.0, 0.0])
                d = rng.normal(size=(200_000, 3))
                d /= np.linalg.norm(d, axis=1, keepdims=True)
                rays = np.concatenate([o, d, np.full((len(o), 1), 0.01), np.full((len(o), 1), 6.0)], axis=1).astype(np.float32)
                ids_a, _ = dev.trace_rays(rays, any_hit=True, uniform_tmin=True)
                monkeypatch.setenv("RTGO_SLAB", "0")
                ids_b, _ = dev.trace_rays(rays, any_hit=True, uniform_tmin=True)
                monkeypatch.delenv("RTGO_SLAB")
                assert np.array_equal(ids_a > 0, ids_b > 0), int(((ids_a > 0) != (ids_b > 0)).sum())
                assert 0.02 < (ids_a > 0).mean() < 0.98, float((ids_a > 0).mean())
        print(f"{name}: beams decide {st.rays_shadow_beam} of {st.rays_shadow} shadow rays with slabs, {st0.rays_shadow_beam} without")
        assert np.array_equal(got, want) and np.array_equal(got, plain) and np.array_equal(got, slab_only) and np.array_equal(got, no_vox), \
            (name, float(np.abs(got - want).max()), float(np.abs(got - plain).max()), float(np.abs(got - slab_only).max()), float(np.abs(got - no_vox).max()))
        for k in keys:
            assert getattr(st, k) == getattr(st0, k) == getattr(st1, k) == getattr(st2, k) == getattr(st3, k), (name, k)
        # the voxel beams settle pairs without a beam walk of their own: fewer walks, the same or more rays decided
        print(f"{name}: {st.beam_walks} beam walks with the voxel beams, {st3.beam_walks} without")
        assert st.beam_walks <= st3.beam_walks and st.rays_shadow_beam >= st3.rays_shadow_beam
        if name in ("config 3", "config 5"):
            assert st.beam_walks < 0.9 * st3.beam_walks, name
        assert st.rays_shadow_beam >= st0.rays_shadow_beam and st1.rays_shadow_beam == 0 and st2.rays_shadow_beam == 0


def test_conservative_bounds_far_from_the_origin_and_at_small_scales():
    """scripts/stress_slabs.py: seeded sheets, shells and crossing planes with spheres, 0.06 to 600 units across and up
    to 10^5 units from the origin (where float32 coordinates are multiples of 2^-7 and a ray's direction has components
    that are exactly 0), rendered with slabs, pair beams and voxel beams and without any of them: bit-identical.
    The scenes include the ones that found three flaws in round 3: a ray along a face of a leaf's box ((plane - o) * inf
    = NaN beside an infinity), the beam's bounds and the resumed rays' stack filter in their pre-rounded form
    fma(plane, 1 / c, -(P / c)), which far from the origin err by more than a leaf is wide."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_slabs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_slabs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = [seed for seed in (0, 2, 3, 5, 7, 10, 11, 15, 19, 22, 31, 34, 39, 43, 51, 55, 59, 64, 77, 90) if not mod.check(seed)]
    assert not bad, bad


def test_conservative_bounds_on_pathological_meshes():
    """The same script's seeds from 10000 on: exactly flat, axis-aligned sheets and walls (boxes and slabs of zero thickness, rays
    through the shared faces of leaves), triangles with repeated or collinear vertices and exact duplicates, needles and
    scene-sized triangles through a fine mesh, sheets folded on themselves; at the same offsets and scales.  Seed 10050 found
    that the boxes of the walk were not conservative against the triangle test far from the origin (the pre-rounded o / d of
    the slab test): a ray through the shared face of two leaves, 5,000 units out, missed both boxes from the root and hit the
    triangle when its walk started below them.  Every box of the flat tree is now grown by 2^-19 of the scene's largest
    coordinate (octree.cpp flatten_octree; the oracle's float32 mode likewise), and a beam's radius by the rounding of
    light + jitter."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_slabs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_slabs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = [seed for seed in (10050, 10000, 10003, 10005, 10010, 10013, 10022, 10027, 10034, 10040, 10047, 10002, 10019, 10049, 10063, 10095) if not mod.check(seed)]
    assert not bad, bad


def test_answers_do_not_depend_on_the_tree():
    """scripts/stress_slabs.py check_trees: 40,000 camera rays and the shadow-style any-hit rays from their hit points, through the
    SAH tree with leaves of 20 shapes, with leaves of 4, and through the reference's octree.  The three trees put different
    boxes between a ray and a shape; a walk whose box tests are conservative against the shapes' own tests finds the
    minimum (nearest hit) or the OR (any hit) of those tests over ALL shapes whatever the boxes, so every t and every
    occlusion answer must be equal, bit for bit — also 5,000 and 10^5 units from the origin, where before the boxes were
    padded 1 to 2 answers in 1,000 differed between the trees (0 at the origin)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_slabs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_slabs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = [seed for seed in (2, 3, 7, 10, 15, 22, 27, 10002, 10003, 10006, 10050, 10095, 20003, 20010, 20018) if not mod.check_trees(seed)]
    assert not bad, bad


@pytest.mark.parametrize("seed", [3, 10002, 10050, 20018, 20003])
def test_device_walks_find_what_all_shapes_give(backend, bridge, seed):
    """The device's two walks (nearest hit over the 32-byte records, any hit over the quantised blocks with slabs and shared shape
    loop) against the oracle's float32 shape tests over EVERY shape, no tree (F32_ALL_SHAPES), on stress scenes 5,000 to 10^5
    units from the origin: every t and every occlusion answer equal, bit for bit."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("stress_slabs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_slabs.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    spec, scene, kind, scale, off = mod.make_scene(seed)
    rng = np.random.default_rng(seed)
    n_rays = 8000
    px = np.stack([rng.integers(0, spec.Width, n_rays), rng.integers(0, spec.Height, n_rays), rng.integers(0, spec.AntiAliasingFactor, n_rays)], 1).astype(np.int32)
    with backend.DeviceScene(scene) as dev, bridge.OracleScene(scene) as osc:
        rays = np.zeros((n_rays, 8), np.float32)
        rays[:, :6] = dev.camera_rays(spec, px)
        rays[:, 7] = 3.0e38
        ids, t = dev.trace_rays(rays)
        ids_all, t_all = osc.trace_rays(rays.astype(np.float64), mode=bridge.F32_ALL_SHAPES)
        hit = ids >= 0
        assert hit.sum() > n_rays // 4 and np.array_equal(hit, ids_all >= 0) and np.array_equal(t[hit], t_all[hit].astype(np.float32))
        p = (rays[hit, :3] + t[hit, None] * rays[hit, 3:6]).astype(np.float32)
        lights = [l for l in scene.Lights if hasattr(l, "Position")]
        L = np.array([[l.Position.X, l.Position.Y, l.Position.Z] for l in lights], np.float32)[rng.integers(0, len(lights), p.shape[0])]
        j = rng.normal(size=p.shape); j *= (rng.uniform(0, 1, (p.shape[0], 1)) ** (1 / 3)) / np.linalg.norm(j, axis=1, keepdims=True)
        d = ((L + np.float32(0.25) * j.astype(np.float32)).astype(np.float32) - p).astype(np.float32)
        dist = np.sqrt((d * d).sum(1, dtype=np.float32)).astype(np.float32)
        ok = dist > 0
        sh = np.zeros((int(ok.sum()), 8), np.float32)
        sh[:, :3], sh[:, 3:6], sh[:, 6], sh[:, 7] = p[ok], d[ok] * (np.float32(1) / dist[ok])[:, None], 0.01, dist[ok]
        occ, _ = dev.trace_rays(sh, any_hit=True, uniform_tmin=True)
        occ_all, _ = osc.trace_rays(sh.astype(np.float64), mode=bridge.F32_ALL_SHAPES, any_hit=True)
        assert 0 < occ.sum() and np.array_equal(occ, occ_all), int((occ != occ_all).sum())


@pytest.mark.parametrize("seed,n_lights", [(0, 17), (1, 40), (5, 130)])
def test_many_positional_lights(backend, bridge, rt, seed, n_lights):
    """17, 40 and 130 point and spot lights on top of a stress scene's three (the (hit, light) pair word gives the light 5 to 8 bits,
    the voxel beams are dropped once their cells would take more than 32 MiB, a round of k_phong_pairs takes four lights):
    ray counters equal to the oracle's float32 mirror, accumulation buffer within its usual bound, and the frame with
    slabs and beams bit-identical to the frame without."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("stress_slabs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_slabs.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    spec, scene, kind, scale, off = mod.make_scene(seed)
    rng = np.random.default_rng(seed)
    V = lambda x, y, z: rt.Vec(float(x * scale + off[0]), float(y * scale + off[1]), float(z * scale + off[2]))
    for q in range(n_lights):
        p = rng.uniform(-5, 5, 3)
        if q % 2:
            scene.Lights.append(rt.PointLight(ColorFrac=rt.Vec(1, 1, 1), Position=V(p[0], 3 + abs(p[1]), p[2]), LightIntensity=3 * scale * scale, SpecularLightIntensity=1,
                                              InverseSquareLawDecayFactor=0.5))
        else:
            scene.Lights.append(rt.SpotLight(ColorFrac=rt.Vec(1, 1, 1), Position=V(p[0], 3 + abs(p[1]), p[2]), LightIntensity=3 * scale * scale, SpecularLightIntensity=1,
                                             LookAt=V(0, 1, 0), Angle=30.0, InverseSquareLawDecayFactor=0.5))
    spec = rt.ImageSpec(64, 40, 2, 2, 4, 16, rt.Dijkstra)
    with backend.DeviceScene(scene) as dev, bridge.OracleScene(scene) as osc:
        os.environ["RTGO_BEAM"] = "1"
        try:
            acc, st = dev.render_accum(spec)
            os.environ["RTGO_SLAB"] = "0"; os.environ["RTGO_BEAM"] = "0"
            plain, st0 = dev.render_accum(spec)
        finally:
            for k in ("RTGO_SLAB", "RTGO_BEAM"):
                os.environ.pop(k, None)
        ref, _, ctr = osc.render(spec, mode=bridge.F32_STACK, rgba=False)
    assert np.array_equal(acc, plain) and st.rays_shadow == st0.rays_shadow
    assert st.rays_primary == ctr["rays_primary"] and st.rays_continuation == ctr["rays_continuation"]
    assert st.rays_shadow == ctr["rays_shadow"] and st.phong_hits == ctr["phong_hits"]
    d = np.abs(acc.astype(np.float64) - ref).max(-1)
    assert (d > 1e-3).mean() <= 2e-3, f"{(d > 1e-3).sum()} pixels off, max {d.max()}"
