// Test helper (CPU only): prints the C-ABI scene description that include/raytracer.hpp derives
// from its ExampleRegression, one resolved record per line, so tests/test_cpp_mirror.py can
// compare it with the Python mirror's flattening of the same scene.  No compute entry point is called.
#include <cstdio>

#include "raytracer.hpp"

static void v(const rt_vec3& a) { std::printf(" %.17g %.17g %.17g", a.x, a.y, a.z); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    try {
        auto [spec, scene] = raytracer::ExampleRegression(640, 380, argv[1]);
        scene.Shapes.push_back(raytracer::TrianglePlane{{1, 2, 3}, {4, 5, 7}, {-1, 0, 2}, false, raytracer::Metal{{0.5, 0.25, 1}, 0.125}});
        std::get<raytracer::TrianglePlane>(scene.Shapes.back()).Rotate({10, 20, 30});
        std::get<raytracer::TrianglePlane>(scene.Shapes.back()).Scale(1.5);
        std::get<raytracer::TrianglePlane>(scene.Shapes.back()).Translate({0.5, -1, 2});
        std::get<raytracer::Sphere>(scene.Shapes[0]).Scale(1.25);
        std::get<raytracer::Sphere>(scene.Shapes[0]).Translate({0, 0.5, 0});
        raytracer::detail::Flat f(scene);
        const rt_image_spec s = raytracer::detail::spec_of(spec);
        std::printf("spec %d %d %d %d %d %d %d\n", s.width, s.height, s.anti_aliasing_factor, s.ray_tracing_max_depth,
                    s.soft_shadow_monte_carlo_repetitions, s.worker_count, s.bvh_traversal_algorithm);
        std::printf("camera"); v(f.desc.camera_look_from); v(f.desc.camera_look_at); v(f.desc.camera_up);
        std::printf(" %.17g %.17g\n", f.desc.camera_aperture, f.desc.camera_fov_deg);
        for (int64_t i = 0; i < f.desc.n_shapes; ++i) {
            const rt_shape& sh = f.desc.shapes[i];
            const rt_material& m = f.desc.materials[sh.material];
            std::printf("shape %d %d", sh.kind, sh.single_sided); v(sh.a); v(sh.b); v(sh.c);
            std::printf(" mat %d", m.kind); v(m.color); v(m.specular_color); std::printf(" %.17g", m.param);
            if (m.texture >= 0) {
                const rt_texture& t = f.desc.textures[m.texture];
                unsigned long long sum = 0;
                if (t.kind == RT_TEX_IMAGE)
                    for (int64_t k = 0; k < (int64_t)t.width * t.height * 4; ++k) sum = sum * 1099511628211ull + t.rgba[k];
                std::printf(" tex %d %d %d", t.kind, t.width, t.height); v(t.color1); v(t.color2);
                std::printf(" %.17g %.17g %llu", t.checkers_width, t.checkers_height, sum);
            }
            std::printf("\n");
        }
        for (int64_t i = 0; i < f.desc.n_lights; ++i) {
            const rt_light& l = f.desc.lights[i];
            std::printf("light %d", l.kind); v(l.color); v(l.position); v(l.look_at);
            std::printf(" %.17g %.17g %.17g %.17g\n", l.intensity, l.specular_intensity, l.angle_deg, l.decay);
        }
        std::printf("tables %lld %lld\n", (long long)f.desc.n_materials, (long long)f.desc.n_textures);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "panic: %s\n", e.what());
        return 1;
    }
    return 0;
}
