"""Stress of the conservative bounds (leaf slabs, pair beams, voxel beams): seeded random scenes rendered with everything
on and with RTGO_SLAB=0 RTGO_BEAM=0 (every shadow ray walked, every shape of every leaf met tested); the accumulation
buffers must be bit-identical.  Run on the GPU box: python scripts/stress_slabs.py [scenes [first seed]]
(seeds from 10000 on: flat axis-aligned sheets, degenerate and duplicate triangles, needles and scene-sized triangles in
a fine mesh, folded sheets; from 20000 on: lights inside their own jitter's reach of a surface, on a vertex, on the floor,
eight lights with cones of 1 to 179 degrees, a sphere around everything, a light 300 scene sizes away, a scene without height, a camera 20 scene sizes away, 1,500 spheres instead of the mesh)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
api = importlib.import_module("raytracer-go_amd")
backend = importlib.import_module("raytracer-go_amd.backend")


def make_scene(seed):
    rng = np.random.default_rng(9000 + seed)
    off = np.array([[0, 0, 0], [300, -200, 100], [5e3, 5e3, -5e3], [1e5, -1e5, 1e5]][seed % 4], dtype=float)
    scale = [1.0, 0.01, 100.0][seed % 3]  # scene size: centimetres to hundreds of units
    kind = seed % 5
    n = 64 + 16 * (seed % 4)
    u, v = np.meshgrid(np.linspace(-3, 3, n), np.linspace(-3, 3, n), indexing="ij")
    if kind == 0:    # bumpy sheet
        y = 1.2 + 0.3 * np.sin(2.1 * u + seed) * np.cos(1.7 * v)
        P = np.stack([u, y, v], -1)
    elif kind == 1:  # tilted, rippled plane (thin slabs at 45 degrees)
        y = 1.5 + 0.7 * u + 0.02 * np.sin(9 * v)
        P = np.stack([u, y, v], -1)
    elif kind == 2:  # cylinder shell (curved: thick slabs for wide leaves)
        a = (u + 3) / 6 * 2 * np.pi
        P = np.stack([1.5 * np.cos(a), 1.6 + 1.5 * np.sin(a) * 0.6, v], -1)
    elif kind == 3:  # noisy sheet (normals all over)
        y = 1.0 + 0.15 * rng.normal(size=u.shape)
        P = np.stack([u, y, v], -1)
    else:            # two crossing sheets
        y = 1.2 + 0.4 * np.sin(1.3 * u)
        P = np.stack([u, y, v], -1)
    a_, b_, c_, d_ = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    tri = np.concatenate([np.stack([a_, b_, c_], -2).reshape(-1, 3, 3), np.stack([d_, c_, b_], -2).reshape(-1, 3, 3)])
    if kind == 4:
        Q = np.stack([u, 1.2 + 0.5 * v / 3, v], -1)
        a_, b_, c_, d_ = Q[:-1, :-1], Q[1:, :-1], Q[:-1, 1:], Q[1:, 1:]
        tri = np.concatenate([tri, np.stack([a_, b_, c_], -2).reshape(-1, 3, 3), np.stack([d_, c_, b_], -2).reshape(-1, 3, 3)])
    if seed >= 10000:  # pathological inputs (seeds from 10000 on; the scenes of the seeds below stay as they were)
        kind = 5 + (seed // 4) % 4  # (seed % 4 picks the offset)
        if kind == 5:    # exactly flat, axis-aligned sheets and walls: boxes and slabs of zero thickness
            Pf = np.stack([u, np.full_like(u, 1.25), v], -1)
            Pw = np.stack([np.full_like(u, 0.5), 1.5 + 0.4 * u, v], -1)
            tri = np.concatenate([np.concatenate([np.stack([X[:-1, :-1], X[1:, :-1], X[:-1, 1:]], -2).reshape(-1, 3, 3),
                                                  np.stack([X[1:, 1:], X[:-1, 1:], X[1:, :-1]], -2).reshape(-1, 3, 3)]) for X in (Pf, Pw)])
        elif kind == 6:  # degenerate triangles among the others: repeated vertices, collinear vertices, exact duplicates
            t = tri.copy()
            idx = rng.permutation(len(t))
            q = len(t) // 8
            t[idx[:q], 1] = t[idx[:q], 0]                                 # two vertices equal
            t[idx[q:2 * q], 2] = 2 * t[idx[q:2 * q], 1] - t[idx[q:2 * q], 0]  # collinear
            t[idx[2 * q:3 * q]] = t[idx[3 * q:4 * q]]                     # the same triangle twice
            t[idx[4 * q:4 * q + 8]] = t[idx[4 * q:4 * q + 8], :1]         # a point
            tri = t
        elif kind == 7:  # sizes mixed in one tree: needles and slivers through a fine mesh, a few triangles as large as the scene
            n_big = 24
            c = rng.uniform(-2.5, 2.5, (n_big, 1, 3)) * np.array([1.0, 0.3, 1.0]) + np.array([0.0, 1.6, 0.0])
            d = rng.normal(0, 1, (n_big, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
            w = rng.normal(0, 1, (n_big, 1, 3)) * 1e-3
            needles = np.concatenate([c - 2.5 * d, c + 2.5 * d, c + w], 1)
            big = rng.uniform(-3, 3, (6, 3, 3)) * np.array([1.0, 0.2, 1.0]) + np.array([0.0, 2.4, 0.0])
            tri = np.concatenate([tri, needles, big])
        else:            # a sheet folded back on itself many times: normals cancel in many leaves (no slab), coplanar layers
            y = 1.2 + 0.05 * np.abs(((u * 8) % 2) - 1)
            P2 = np.stack([u, y, v], -1)
            tri = np.concatenate([tri, np.stack([P2[:-1, :-1], P2[1:, :-1], P2[:-1, 1:]], -2).reshape(-1, 3, 3),
                                  np.stack([P2[1:, 1:], P2[1:, :-1], P2[:-1, 1:]], -2).reshape(-1, 3, 3)])
    elif seed % 7 == 6:  # a soup instead: random triangles of all sizes and orientations (no surface for a slab to follow)
        n_tri = 600 + 37 * (seed % 11)
        c = rng.uniform(-3, 3, (n_tri, 1, 3)) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 2.0, 0.0])
        tri = c + rng.normal(0, 0.05 + 0.04 * (seed % 5), (n_tri, 3, 3))
    tri = tri * scale + off
    V = lambda x, y_, z: api.Vec(float(x * scale + off[0]), float(y_ * scale + off[1]), float(z * scale + off[2]))
    mat = api.PhongBlinn(ColorFrac=api.Vec(0.8, 0.7, 0.6), SpecularColorFrac=api.Vec(1, 1, 1), SpecHardness=16)
    floor = api.PhongBlinn(ColorFrac=api.Vec(0.6, 0.6, 0.6), SpecularColorFrac=api.Vec(0.3, 0.3, 0.3), SpecHardness=4)
    R = 9.0
    shapes = [api.TrianglePlane(V(-R, 0, -R), V(-R, 0, R), V(R, 0, -R), False, floor),
              api.TrianglePlane(V(R, 0, R), V(R, 0, -R), V(-R, 0, R), False, floor),
              api.TriangleMesh(tri, bool(seed & 1), mat)]
    for k in range(5):
        p = rng.uniform(-2.5, 2.5, 3)
        shapes.append(api.Sphere(V(p[0], 1.3 + 0.2 * k, p[2]), float(rng.uniform(0.05, 0.4)) * scale, mat))
    # (the soft-shadow jitter is 0.25 units whatever the scene's size: at scale 0.01 the light's ball dwarfs the scene)
    lights = [api.AmbientLight(api.Vec(1, 1, 1), 0.05),
              api.PointLight(ColorFrac=api.Vec(1, 1, 1), Position=V(6, 2.0 + 3 * (seed % 2), -5), LightIntensity=60 * scale * scale, SpecularLightIntensity=20,
                             InverseSquareLawDecayFactor=0.5),
              api.SpotLight(ColorFrac=api.Vec(0.8, 0.9, 1), Position=V(-1, 6, 1), LightIntensity=40 * scale * scale, SpecularLightIntensity=10,
                            LookAt=V(0, 1, 0), Angle=25.0 + 10 * (seed % 3), InverseSquareLawDecayFactor=0.3)]
    if seed % 5 == 3:  # a light among the shapes, and a mirror so that later bounces shade too
        lights.append(api.PointLight(ColorFrac=api.Vec(1, 0.8, 0.8), Position=V(0.3, 1.25, 0.2), LightIntensity=10 * scale * scale, SpecularLightIntensity=5,
                                     InverseSquareLawDecayFactor=0.5))
        shapes.append(api.TrianglePlane(V(-R, 0, R), V(-R, 8, R), V(R, 0, R), True, api.Metal(api.Vec(0.9, 0.9, 0.9), 0.0)))
    far_camera = False
    if seed >= 20000:  # lights and extents at their limits (seeds from 20000 on; the mesh is the one of seed % 5)
        var = (seed // 4) % 8
        kind = 10 + var
        if var == 0:    # a light closer to the surface than its own jitter: the hit point is inside the ball of light positions
            k = int(rng.integers(len(tri)))
            c = tri[k].mean(0)
            lights.append(api.PointLight(ColorFrac=api.Vec(1, 1, 0.9), Position=api.Vec(float(c[0]), float(c[1] + 0.12), float(c[2])), LightIntensity=2 * scale * scale,
                                         SpecularLightIntensity=1, InverseSquareLawDecayFactor=0.5))
        elif var == 1:  # a light exactly on a vertex of the mesh, and one exactly on the floor
            v0 = tri[int(rng.integers(len(tri)))][0]
            lights.append(api.PointLight(ColorFrac=api.Vec(1, 1, 0.9), Position=api.Vec(float(v0[0]), float(v0[1]), float(v0[2])), LightIntensity=2 * scale * scale,
                                         SpecularLightIntensity=1, InverseSquareLawDecayFactor=0.5))
            lights.append(api.PointLight(ColorFrac=api.Vec(0.9, 1, 1), Position=V(1.0, 0.0, -1.0), LightIntensity=2 * scale * scale, SpecularLightIntensity=1,
                                         InverseSquareLawDecayFactor=0.5))
        elif var == 2:  # eight lights, cones from 1 to 179 degrees
            for q, ang in enumerate((1.0, 5.0, 89.0, 90.0, 91.0, 179.0)):
                lights.append(api.SpotLight(ColorFrac=api.Vec(0.5, 0.6, 0.7), Position=V(-2.5 + q, 4.0 + 0.3 * q, -2.0 + 0.7 * q), LightIntensity=20 * scale * scale,
                                            SpecularLightIntensity=5, LookAt=V(0.2 * q, 1.0, 0.1 * q), Angle=ang, InverseSquareLawDecayFactor=0.3))
        elif var == 3:  # everything inside one huge sphere, the mesh tiny against the scene's box
            shapes.append(api.Sphere(V(0, 1, 0), 60.0 * scale, floor))
        elif var == 4:  # a light far outside, 300 scene sizes away (beams as long as the coordinates allow)
            lights.append(api.PointLight(ColorFrac=api.Vec(1, 1, 1), Position=V(900.0, 1500.0, -1200.0), LightIntensity=4e6 * scale * scale, SpecularLightIntensity=5,
                                         InverseSquareLawDecayFactor=0.5))
        elif var == 7:  # spheres instead of the mesh: 1,500 of them, 0.01 to 0.3 units, many overlapping (leaves without a slab, mixed leaves)
            c = rng.uniform(-3, 3, (1500, 3)) * np.array([1.0, 0.35, 1.0]) + np.array([0.0, 1.6, 0.0])
            r = 0.01 + 0.29 * rng.uniform(0, 1, 1500) ** 3
            shapes = shapes[:2] + [api.Sphere(V(c[q, 0], c[q, 1], c[q, 2]), float(r[q]) * scale, mat) for q in range(1500)] + \
                     [api.TrianglePlane(V(-1 + 0.1 * q, 0.5, -1), V(-1 + 0.1 * q, 2.5, 1), V(-0.9 + 0.1 * q, 0.5, 1), False, mat) for q in range(20)]
        elif var == 6:  # the camera 20 scene sizes away behind a long lens: primary rays start far outside the shapes' box
            far_camera = True
            # (no spheres: Sphere.hit's b^2 - a c, kept in the reference's form, cancels in float32 when |o - centre|^2 nears 2^24
            # r^2; from 700 units a 0.05-unit sphere "hits" rays that pass it by.  From 3,000 scene sizes the triangle test goes
            # the same way — a triangle seen edge-on has a determinant below its own rounding (the reference's parallelism check
            # is |det| < 0, shape.go:132) and reports hits hundreds of units in front of the scene, 7 rays in 40,000; from 100
            # scene sizes 1 ray in 9 million; which of them a walk meets depends on its leaves.  float32 limits of the shapes'
            # tests in the reference's expression order, not of the boxes: DESIGN section 5)
            shapes = [sh for sh in shapes if not isinstance(sh, api.Sphere)]
        else:           # only the floor and flat things on it: the scene's box has no height
            shapes = shapes[:2] + [api.TrianglePlane(V(-1, 0, -1), V(-1, 0, 1), V(1, 0, -1), False, mat), api.TrianglePlane(V(0.5, 0, 0.5), V(2, 0, 0.7), V(0.7, 0, 2), False, mat)]
            shapes += [api.TrianglePlane(V(0.01 * q - 2, 0, 0.013 * q), V(0.01 * q - 1.99, 0, 0.013 * q + 0.02), V(0.01 * q - 1.97, 0, 0.013 * q - 0.01), False, mat) for q in range(300)]
    spec = api.ImageSpec(160, 100, 2, 4, (16, 1, 5, 70, 16, 3)[seed % 6], 16, api.Dijkstra)
    scene = api.Scene(V(0, 2.2, -7), V(0, 1.2, 0), api.Vec(0, 1, 0), V(0, 1.2, 0), 0.02 * scale, 50.0, shapes, lights)
    if far_camera:
        scene = api.Scene(V(0, 1.2 + 20.0, -140.0), V(0, 1.2, 0), api.Vec(0, 1, 0), V(0, 1.2, 0), 0.02 * scale, 2.7, shapes, lights)
    return spec, scene, kind, scale, off


def check(seed, parts=False):
    """True if the scene of `seed` renders bit-identically with and without the conservative bounds."""
    spec, scene, kind, scale, off = make_scene(seed)
    with backend.DeviceScene(scene) as dev:
        os.environ["RTGO_BEAM"] = "1"
        got, st = dev.render_accum(spec)
        os.environ["RTGO_SLAB"] = "0"; os.environ["RTGO_BEAM"] = "0"
        want, st0 = dev.render_accum(spec)
        del os.environ["RTGO_SLAB"]; del os.environ["RTGO_BEAM"]
        if parts:  # which bound is it?
            found = {}
            for name, env in (("slabs only", {"RTGO_BEAM": "0"}), ("beams only", {"RTGO_BEAM": "1", "RTGO_SLAB": "0", "RTGO_VOXEL": "0"}),
                              ("beams + voxels, no slabs", {"RTGO_BEAM": "1", "RTGO_SLAB": "0"}), ("beams + slabs, no voxels", {"RTGO_BEAM": "1", "RTGO_VOXEL": "0"}),
                              ("beams only, rays from the root", {"RTGO_BEAM": "1", "RTGO_SLAB": "0", "RTGO_VOXEL": "0", "RTGO_RESUME": "0"})):
                os.environ.update(env)
                x, _ = dev.render_accum(spec)
                for k in env:
                    del os.environ[k]
                found[name] = int((x != want).any(axis=-1).sum())
            print("   pixels that differ:", found)
    same = bool(np.array_equal(got, want)) and st.rays_shadow == st0.rays_shadow
    print(f"seed {seed}: kind {kind} scale {scale} offset {off.tolist()} -> {'same' if same else 'DIFFERENT %g' % float(np.abs(got - want).max())}; "
          f"beams decide {st.rays_shadow_beam} of {st.rays_shadow}, {st.beam_walks} walks", flush=True)
    return same


def check_trees(seed, n_rays=40000):
    """True if shadow-style any-hit rays through the scene of `seed` answer alike whatever tree they walk: the SAH tree with
    leaves of 20 and of 4 shapes and the reference's octree put different boxes between a ray and a shape, and a walk whose
    box tests are conservative against the shapes' own tests gives the OR of those tests over all shapes either way."""
    spec, scene, kind, scale, off = make_scene(seed)
    rng = np.random.default_rng(seed)
    px = np.stack([rng.integers(0, spec.Width, n_rays), rng.integers(0, spec.Height, n_rays), rng.integers(0, spec.AntiAliasingFactor, n_rays)], 1).astype(np.int32)
    lights = [l for l in scene.Lights if hasattr(l, "Position")]
    answers, nearest, sh = [], [], None
    for env in ({}, {"RTGO_LEAF_SIZE": "4"}, {"RTGO_TREE": "octree"}):
        os.environ.update(env)
        try:
            with backend.DeviceScene(scene) as dev:
                if sh is None:  # the rays: from the first tree's nearest hits towards jittered light positions, as k_shadow makes them
                    rays = np.zeros((n_rays, 8), np.float32)
                    rays[:, :6] = dev.camera_rays(spec, px)
                    rays[:, 7] = 3.0e38
                ids, t = dev.trace_rays(rays)
                nearest.append(np.where(ids >= 0, t, np.float32(-1)))  # (two shapes at the same t: which one is named depends on the order)
                if sh is None:
                    hit = ids >= 0
                    p = (rays[hit, :3] + t[hit, None] * rays[hit, 3:6]).astype(np.float32)
                    L = np.array([[l.Position.X, l.Position.Y, l.Position.Z] for l in lights], np.float32)[rng.integers(0, len(lights), p.shape[0])]
                    j = rng.normal(size=p.shape); j *= (rng.uniform(0, 1, (p.shape[0], 1)) ** (1 / 3)) / np.linalg.norm(j, axis=1, keepdims=True)
                    shifted = (L + np.float32(0.25) * j.astype(np.float32)).astype(np.float32)
                    d = (shifted - p).astype(np.float32)
                    dist = np.sqrt((d * d).sum(1, dtype=np.float32)).astype(np.float32)
                    ok = dist > 0
                    sh = np.zeros((int(ok.sum()), 8), np.float32)
                    sh[:, :3], sh[:, 3:6], sh[:, 6], sh[:, 7] = p[ok], d[ok] * (np.float32(1) / dist[ok])[:, None], 0.01, dist[ok]
                occ, _ = dev.trace_rays(sh, any_hit=True, uniform_tmin=True)
                answers.append(occ.copy())
        finally:
            for k in env:
                del os.environ[k]
    n_diff = [int((a != answers[0]).sum()) for a in answers[1:]]
    t_diff = [int((a != nearest[0]).sum()) for a in nearest[1:]]
    same = not any(n_diff) and not any(t_diff)
    print(f"seed {seed}: kind {kind} scale {scale} offset {off.tolist()} -> {len(sh)} any-hit rays, {int(answers[0].sum())} occluded; "
          f"answers that differ from the default tree's: leaves of 4 {n_diff[0]}, octree {n_diff[1]}; nearest hits of {n_rays} camera rays at another t: {t_diff[0]}, {t_diff[1]}", flush=True)
    return same


if __name__ == "__main__":
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 10000: the pathological scenes
    only = os.environ.get("STRESS_ONLY")
    bad = 0
    for seed in range(first, first + n_scenes):
        if only and str(seed) not in only.split(","):
            continue
        if os.environ.get("STRESS_TREES"):
            bad += 0 if check_trees(seed) else 1
            continue
        bad += 0 if check(seed, parts=bool(os.environ.get("STRESS_PARTS"))) else 1
    print("scenes that differ:", bad)
    sys.exit(1 if bad else 0)
