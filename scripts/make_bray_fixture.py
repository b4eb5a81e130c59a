"""Measures the ALGORITHMIC traversal work per ray with the CPU oracle's float64 best-first
traversal (the reference's own, accelerationstructures.go:81-124): mean inner nodes expanded
(N_inner) and leaf shapes tested (N_leaf) per ray class, on a reduced-resolution pass over each
config.  B_ray = 224*N_inner + 40*N_leaf + 56 bytes.  Output: profiles/bray.json.
"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scenes = importlib.import_module("raytracer-go_amd.scenes")
from oracle import bridge

GOLD = os.path.join(ROOT, "tests", "golden")
out = {}
for name, (spec, scene), (w, h, spp) in (
        ("config1", scenes.config1(640, 380, GOLD), (320, 190, 2)),
        ("config2", scenes.config2(), (480, 270, 1)),
        ("config3", scenes.config3(asset_dir=GOLD), (480, 270, 1))):
    spec.Width, spec.Height, spec.AntiAliasingFactor = w, h, spp
    with bridge.OracleScene(scene) as osc:
        _, _, c = osc.render(spec, mode=bridge.F64_DIJKSTRA, rgba=False)
    row = {"sample": f"{w}x{h}x{spp}spp, oracle float64 best-first", "counters": c}
    for cls, rays in (("primary", "rays_primary"), ("continuation", "rays_continuation"), ("shadow", "rays_shadow")):
        n = c[rays]
        ni = c["inner_" + cls] / n if n else 0.0
        nl = c["leaf_" + cls] / n if n else 0.0
        row[cls] = {"n_inner": ni, "n_leaf": nl, "bytes_per_ray": 224 * ni + 40 * nl + 56}
    out[name] = row
    print(name, {k: row[k] for k in ("primary", "continuation", "shadow")}, flush=True)
with open(os.path.join(ROOT, "profiles", "bray.json"), "w") as f:
    json.dump(out, f, indent=1)
