"""Stage times of configs 2 and 5 on one lane (RTGO_LANES=1, so that the stages do not overlap): where the frame goes
when there are scattered rays (config 5: glass spheres, Metal floor) or few soft-shadow repetitions."""
import importlib, os, sys, time
os.environ["RTGO_LANES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
backend = importlib.import_module("raytracer-go_amd.backend")
scenes = importlib.import_module("raytracer-go_amd.scenes")
GOLD = os.path.join(ROOT, "tests", "golden")
for name, (spec, scene) in (("config 5 at 1080p 16 spp", scenes.config5(1920, 1080, 16, 16, asset_dir=GOLD)),
                            ("config 2 at 1080p 16 spp, reps 16", scenes.config2(1920, 1080, 16, 16)),
                            ("config 2 at 1080p 16 spp, reps 1", scenes.config2(1920, 1080, 16, 1))):
    with backend.DeviceScene(scene) as dev:
        dev.render_accum(spec)
        _, st = dev.render_accum(spec)
    print(f"{name}: render {st.ms_render:.1f} ms = extend {st.ms_extend:.1f} + shadow {st.ms_shadow:.1f} + rest {st.ms_shade:.1f}; "
          f"rays primary {st.rays_primary} continuation {st.rays_continuation} shadow {st.rays_shadow} (beams {st.rays_shadow_beam})", flush=True)
