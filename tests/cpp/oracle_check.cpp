// CPU-only: the oracle (oracle/rtgo_oracle.cpp) itself under AddressSanitizer / UBSan.  Builds a small scene
// that uses every shape, material, light and texture kind through the C-ABI structs, renders it in the three
// oracle modes and both fp32 tree mirrors, traces explicit rays, and checks basic agreement between the modes.
// A checker with undefined behaviour would be a poor checker.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rtgo.h"

struct orc_scene;
struct orc_counters { unsigned long long v[10]; };
extern "C" {
const char* orc_last_error(void);
orc_scene* orc_scene_create(const rt_scene_desc* d);
void orc_scene_destroy(orc_scene* s);
void orc_set_leaf_size(orc_scene* s, int leaf_size);
void orc_set_tree(orc_scene* s, int kind);
int orc_trace_rays(const orc_scene* s, int mode, const double* rays, int64_t n, int any_hit, int32_t* prim, double* t,
                   uint32_t* inner, uint32_t* leaf, int nthreads);
int orc_render(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int mode, int nthreads, int row0, int row1, int s0, int s1,
               double* accum_out, uint8_t* rgba_out, orc_counters* ctr);
}

int main() {
    std::vector<uint8_t> texels(8 * 8 * 4);
    for (size_t i = 0; i < texels.size(); ++i) texels[i] = (uint8_t)(i * 37 + 11);
    rt_texture tex[2] = {};
    tex[0].kind = RT_TEX_CHECKERS; tex[0].color1 = {1, 0, 0}; tex[0].color2 = {0, 0, 1}; tex[0].checkers_width = 9; tex[0].checkers_height = 5;
    tex[1].kind = RT_TEX_IMAGE; tex[1].width = 8; tex[1].height = 8; tex[1].rgba = texels.data();
    rt_material mat[6] = {};
    mat[0].kind = RT_MAT_STANDARD; mat[0].texture = 0;
    mat[1].kind = RT_MAT_METAL; mat[1].texture = -1; mat[1].color = {0.9, 0.9, 0.9}; mat[1].param = 0.1;
    mat[2].kind = RT_MAT_DIELECTRIC; mat[2].texture = -1; mat[2].param = 1.52;
    mat[3].kind = RT_MAT_PHONGBLINN; mat[3].texture = 1; mat[3].color = {0.8, 0.8, 0.8}; mat[3].specular_color = {1, 1, 1}; mat[3].param = 12;
    mat[4].kind = RT_MAT_PHONGBLINN; mat[4].texture = -1; mat[4].color = {0.3, 0.8, 0.4}; mat[4].specular_color = {1, 1, 1}; mat[4].param = 30;
    mat[5].kind = RT_MAT_STANDARD; mat[5].texture = -1; mat[5].color = {0.2, 0.4, 0.9};
    std::vector<rt_shape> shapes;
    auto sphere = [&](double x, double y, double z, double r, int m) { rt_shape s{}; s.kind = RT_SHAPE_SPHERE; s.material = m; s.a = {x, y, z}; s.b = {r, 0, 0}; shapes.push_back(s); };
    auto tri = [&](rt_vec3 a, rt_vec3 b, rt_vec3 c, int single, int m) { rt_shape s{}; s.kind = RT_SHAPE_TRIANGLE; s.material = m; s.single_sided = single; s.a = a; s.b = b; s.c = c; shapes.push_back(s); };
    tri({-20, 0, -20}, {-20, 0, 20}, {20, 0, -20}, 1, 3);
    tri({20, 0, 20}, {20, 0, -20}, {-20, 0, 20}, 1, 3);
    for (int i = 0; i < 5; ++i) sphere(-4.0 + 2.0 * i, 1.0, 0.5 * i, 1.0, i);
    for (int i = 0; i < 200; ++i) {  // a strip of small triangles so that both trees have several levels
        const double x = -5 + 0.05 * i, z = 3 + 0.01 * (i % 7);
        tri({x, 0.2, z}, {x + 0.05, 0.2, z}, {x, 0.6 + 0.002 * i, z + 0.03}, 0, 4 + (i & 1));
    }
    rt_light lights[3] = {};
    lights[0].kind = RT_LIGHT_AMBIENT; lights[0].color = {1, 1, 1}; lights[0].intensity = 0.1;
    lights[1].kind = RT_LIGHT_POINT; lights[1].color = {1, 0.9, 0.8}; lights[1].position = {3, 6, -4}; lights[1].intensity = 60; lights[1].specular_intensity = 20; lights[1].decay = 0.8;
    lights[2].kind = RT_LIGHT_SPOT; lights[2].color = {0.6, 0.7, 1}; lights[2].position = {-5, 7, -3}; lights[2].look_at = {0, 0, 1}; lights[2].intensity = 90;
    lights[2].specular_intensity = 30; lights[2].angle_deg = 25; lights[2].decay = 1.0;
    rt_scene_desc d{};
    d.camera_look_from = {0, 4, -10}; d.camera_look_at = {0, 1, 0}; d.camera_up = {0, 1, 0}; d.camera_aperture = 0.05; d.camera_fov_deg = 55;
    d.shapes = shapes.data(); d.n_shapes = (int64_t)shapes.size();
    d.materials = mat; d.n_materials = 6; d.textures = tex; d.n_textures = 2; d.lights = lights; d.n_lights = 3;
    rt_image_spec spec{};
    spec.width = 48; spec.height = 27; spec.anti_aliasing_factor = 2; spec.ray_tracing_max_depth = 6; spec.soft_shadow_monte_carlo_repetitions = 4;

    int fails = 0;
    orc_scene* s = orc_scene_create(&d);
    if (!s) { std::printf("create failed: %s\n", orc_last_error()); return 1; }
    const size_t npx = (size_t)spec.width * spec.height;
    std::vector<double> acc[4];
    for (int pass = 0; pass < 4; ++pass) {  // modes 0, 1, 2 (SAH mirror), 2 (octree mirror)
        const int mode = pass < 3 ? pass : 2;
        if (pass == 3) { orc_set_tree(s, 0); orc_set_leaf_size(s, 4); }
        acc[pass].assign(npx * 3, 0.0);
        std::vector<uint8_t> rgba(npx * 4);
        orc_counters c{};
        if (orc_render(s, &spec, 0x5EED, mode, 2, 0, spec.height, 0, spec.anti_aliasing_factor, acc[pass].data(), rgba.data(), &c) != 0) {
            std::printf("render mode %d failed: %s\n", mode, orc_last_error());
            ++fails;
        }
        for (size_t i = 3; i < rgba.size(); i += 4) if (rgba[i] != 255) { ++fails; break; }
    }
    auto frac_diff = [&](const std::vector<double>& a, const std::vector<double>& b, double tol) {
        size_t n = 0;
        for (size_t i = 0; i < a.size(); ++i) n += std::fabs(a[i] - b[i]) > tol;
        return (double)n / (double)a.size();
    };
    if (frac_diff(acc[0], acc[1], 1e-12) > 0.0) { std::printf("DFS differs from best-first\n"); ++fails; }
    if (frac_diff(acc[0], acc[2], 1e-3) > 0.02) { std::printf("fp32 SAH mirror differs from fp64: %.4f\n", frac_diff(acc[0], acc[2], 1e-3)); ++fails; }
    if (frac_diff(acc[2], acc[3], 1e-6) > 0.005) { std::printf("fp32 mirrors of the two trees differ: %.4f\n", frac_diff(acc[2], acc[3], 1e-6)); ++fails; }
    // explicit rays, nearest hit in every mode and any-hit in the fp32 mode
    std::vector<double> rays;
    for (int i = 0; i < 500; ++i) {
        const double ang = 0.0125 * i;
        const double o[3] = {12 * std::cos(ang), 3 + std::sin(3 * ang), 12 * std::sin(ang)};
        const double dd[3] = {-o[0], 0.8 - o[1], 1 - o[2]};
        const double l = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
        const double r[8] = {o[0], o[1], o[2], dd[0] / l, dd[1] / l, dd[2] / l, 0.0, 1e30};
        rays.insert(rays.end(), r, r + 8);
    }
    std::vector<int32_t> prim[3];
    std::vector<double> t(500);
    std::vector<uint32_t> ni(500), nl(500);
    for (int mode = 0; mode < 3; ++mode) {
        prim[mode].assign(500, -2);
        if (orc_trace_rays(s, mode, rays.data(), 500, 0, prim[mode].data(), t.data(), ni.data(), nl.data(), 2) != 0) ++fails;
    }
    int differ = 0, hits = 0;
    for (int i = 0; i < 500; ++i) { differ += prim[0][i] != prim[1][i] || prim[0][i] != prim[2][i]; hits += prim[0][i] >= 0; }
    if (differ > 2 || hits < 100) { std::printf("trace modes differ on %d rays (%d hits)\n", differ, hits); ++fails; }
    for (int i = 0; i < 500; ++i) { rays[8 * i + 6] = 0.01; rays[8 * i + 7] = 30.0; }
    std::vector<int32_t> occ(500);
    if (orc_trace_rays(s, 2, rays.data(), 500, 1, occ.data(), t.data(), ni.data(), nl.data(), 2) != 0) ++fails;
    orc_scene_destroy(s);
    std::printf(fails ? "%d FAILURES\n" : "oracle self-check ok (%d of 500 probe rays hit)\n", fails ? fails : hits);
    return fails ? 1 : 0;
}
