// Design-time simulation (CPU, not part of the product or of the oracle's checks): how much tree walking
// would a (hit, light) pair's soft-shadow repetitions share if the pair walked the device tree ONCE with a
// conservative beam (interval arithmetic over the repetitions' directions) instead of once per repetition?
// Built on the oracle's fp32 mirror of the device tree:  g++ -O2 -shared -fPIC scripts/sim_groupwalk.cpp
// (includes oracle/rtgo_oracle.cpp as a translation unit); driven by scripts/sim_groupwalk.py.
#include "../oracle/rtgo_oracle.cpp"
#include <unordered_map>

namespace {

struct Beam {
    float o[3];
    float ilo[3], ihi[3];   // bounds of 1/d over the rays (valid when !straddle)
    bool straddle[3];
    float tmin, tmax;
    int sign;
};

inline bool beam_box(const Beam& b, const float* bmin, const float* bmax) {
    float tn = b.tmin, tf = b.tmax;
    for (int a = 0; a < 3; ++a) {
        if (b.straddle[a]) continue;  // directions of both signs on this axis: no bound from it
        const float x0 = bmin[a] - b.o[a], x1 = bmax[a] - b.o[a];
        const float c0 = x0 * b.ilo[a], c1 = x0 * b.ihi[a], c2 = x1 * b.ilo[a], c3 = x1 * b.ihi[a];
        // per ray: entry = min(t(b0), t(b1)), exit = max(...); over the interval: lower bound of entry = the smaller
        // of the two planes' minima ... conservative: entry_lo = min over rays of min(plane0, plane1)
        const float p0lo = fminf(c0, c1), p0hi = fmaxf(c0, c1), p1lo = fminf(c2, c3), p1hi = fmaxf(c2, c3);
        const float entry_lo = fminf(p0lo, p1lo);
        const float exit_hi = fmaxf(p0hi, p1hi);
        // tighter: entry of a ray is the min of its two plane values, so entry_lo = min(p0lo, p1lo) is exact for the
        // hull; exit_hi likewise
        tn = fmaxf(tn, entry_lo);
        tf = fminf(tf, exit_hi);
    }
    return tn <= tf;
}
// tighter variant: for same-sign direction intervals the near plane is known per axis
inline bool beam_box_tight(const Beam& b, const float* bmin, const float* bmax) {
    float tn = b.tmin, tf = b.tmax;
    for (int a = 0; a < 3; ++a) {
        if (b.straddle[a]) {
            continue;
        }
        const bool neg = (b.sign >> a) & 1;
        const float xn = (neg ? bmax[a] : bmin[a]) - b.o[a];  // near plane offset
        const float xf = (neg ? bmin[a] : bmax[a]) - b.o[a];  // far plane offset
        // t = x * inv; inv in [ilo, ihi] (same sign as direction).  entry lower bound = min(xn*ilo, xn*ihi)
        const float e = fminf(xn * b.ilo[a], xn * b.ihi[a]);
        const float x = fmaxf(xf * b.ilo[a], xf * b.ihi[a]);
        tn = fmaxf(tn, e);
        tf = fminf(tf, x);
    }
    return tn <= tf;
}

struct SimOut {
    double pairs, rays;
    double base_inner, base_leaf;          // per-ray walk: block visits, shape tests (early exit)
    double grp_visits, grp_touched;        // beam walk: block visits, beam-touched leaves per pair
    double ray_leafbox, ray_leafpass, ray_shape;  // per ray: own-ray leaf box tests, those passed, shape tests
    double wave_base_iters, wave_grp_visits, wave_grp_touched, waves;  // max over the 4 pairs of a wavefront
    double occluded_rays, fully_lit_pairs, fully_dark_pairs, empty_walk_pairs;
};

}  // namespace

extern "C" int sim_groupwalk(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int subn, int tight, SimOut* out, float* rec, int64_t rec_cap, int64_t* n_rec) {
    int64_t nr = 0;
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    SimOut o{};
    Node* root = effective(sc.wide_root);
    const int leaf_size = sc.leaf_size;
    struct PairCost { double base_max, visits, touched; };
    std::vector<std::vector<PairCost>> per_light(T.lights.size());
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> l2p = sub(L.pos, hit.p);
                const V3<float> ldir = unit(l2p);
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                std::vector<Ray<float>> rays; std::vector<float> dist;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    rays.push_back({hit.p, scale(1.0f / d, ld)});
                    dist.push_back(d);
                }
                if (rays.empty()) continue;
                o.pairs += 1; o.rays += (double)rays.size();
                // baseline
                double bmax = 0; int nocc = 0; double pair_inner = 0, pair_leaf = 0;
                std::vector<char> base_occ(rays.size());
                for (size_t k = 0; k < rays.size(); ++k) {
                    TraceCounters tc; Hit<float> hh;
                    const bool occ = trace_stack_f32(sc, rays[k], 0.01f, dist[k], true, hh, &tc);
                    base_occ[k] = occ;
                    o.base_inner += (double)tc.inner; o.base_leaf += (double)tc.leaf; pair_inner += (double)tc.inner; pair_leaf += (double)tc.leaf;
                    bmax = std::max(bmax, (double)tc.inner);
                    nocc += occ;
                }
                o.occluded_rays += nocc;
                if (nocc == 0) o.fully_lit_pairs += 1;
                if (nocc == (int)rays.size()) o.fully_dark_pairs += 1;
                // beam walk(s): sub-beams of `sub` rays (0 = all)
                double visits = 0, touched = 0;
                const size_t step = subn > 0 ? (size_t)subn : rays.size();
                std::vector<char> occ(rays.size(), 0);
                for (size_t b0 = 0; b0 < rays.size(); b0 += step) {
                    const size_t b1 = std::min(rays.size(), b0 + step);
                    Beam B{};
                    B.o[0] = hit.p.x; B.o[1] = hit.p.y; B.o[2] = hit.p.z;
                    B.tmin = 0.01f; B.tmax = 0;
                    float dlo[3] = {1e30f, 1e30f, 1e30f}, dhi[3] = {-1e30f, -1e30f, -1e30f};
                    V3<float> mean{0, 0, 0};
                    for (size_t k = b0; k < b1; ++k) {
                        const float d[3] = {rays[k].d.x, rays[k].d.y, rays[k].d.z};
                        for (int a = 0; a < 3; ++a) { dlo[a] = std::min(dlo[a], d[a]); dhi[a] = std::max(dhi[a], d[a]); }
                        B.tmax = std::max(B.tmax, dist[k]);
                        mean = add(mean, rays[k].d);
                    }
                    B.sign = (mean.x < 0 ? 1 : 0) | (mean.y < 0 ? 2 : 0) | (mean.z < 0 ? 4 : 0);
                    for (int a = 0; a < 3; ++a) {
                        B.straddle[a] = !(dlo[a] > 0.0f || dhi[a] < 0.0f);
                        if (!B.straddle[a]) { B.ilo[a] = 1.0f / dhi[a]; B.ihi[a] = 1.0f / dlo[a]; }
                    }
                    std::vector<StackRay> srs;
                    for (size_t k = b0; k < b1; ++k) srs.push_back(make_stack_ray(rays[k]));
                    size_t undecided = b1 - b0;
                    std::vector<Node*> stack;
                    Node* cur = root;
                    std::vector<int> group;
                    if (root->as_leaf) { cur = nullptr; }
                    while (cur && undecided > 0) {
                        visits += 1;
                        Node* pending[8]; int np = 0;
                        for (int k = 0; k < 8; ++k) {
                            Node* e = cur->slot[k ^ B.sign];
                            if (!e) continue;
                            const bool hitb = tight ? beam_box_tight(B, e->fmin, e->fmax) : beam_box(B, e->fmin, e->fmax);
                            if (!hitb) continue;
                            if (e->as_leaf) {
                                touched += 1;
                                group.clear();
                                subtree_shapes(e, &group);
                                for (size_t k2 = b0; k2 < b1; ++k2) {
                                    if (occ[k2]) continue;
                                    o.ray_leafbox += 1;
                                    if (!slab_f32(srs[k2 - b0], e->fmin, e->fmax, 0.01f, dist[k2])) continue;
                                    o.ray_leafpass += 1;
                                    for (int shape : group) {
                                        o.ray_shape += 1;
                                        const Hit<float> shh = T.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE
                                                                   ? triangle_occludes_f32(T.shapes[(size_t)shape], shape, rays[k2], 0.01f, dist[k2])
                                                                   : shape_hit(T, shape, rays[k2], 0.01f, dist[k2]);
                                        if (shh.t > 0.0f && shh.t <= dist[k2]) { occ[k2] = 1; --undecided; break; }
                                    }
                                }
                            } else pending[np++] = e;
                        }
                        for (int q = np - 1; q >= 1; --q) stack.push_back(pending[q]);
                        if (np > 0) { cur = pending[0]; continue; }
                        if (stack.empty()) break;
                        cur = stack.back(); stack.pop_back();
                    }
                }
                for (size_t k = 0; k < rays.size(); ++k)
                    if ((bool)occ[k] != (bool)base_occ[k]) { g_err = "beam walk disagrees with per-ray walk"; return -1; }
                o.grp_visits += visits; o.grp_touched += touched;
                if (touched == 0) o.empty_walk_pairs += 1;
                per_light[li].push_back({bmax, visits, touched});
                if (rec && nr < rec_cap) {
                    float* q = rec + nr * 8;
                    q[0] = (float)rays.size(); q[1] = (float)nocc; q[2] = (float)pair_inner; q[3] = (float)pair_leaf;
                    q[4] = (float)visits; q[5] = (float)touched; q[6] = (float)bmax; q[7] = (float)li;
                    ++nr;
                }
            }
        }
    for (auto& v : per_light)
        for (size_t q = 0; q < v.size(); q += 4) {
            double a = 0, b = 0, c = 0;
            for (size_t k = q; k < std::min(v.size(), q + 4); ++k) { a = std::max(a, v[k].base_max); b = std::max(b, v[k].visits); c = std::max(c, v[k].touched); }
            o.wave_base_iters += a; o.wave_grp_visits += b; o.wave_grp_touched += c; o.waves += 1;
        }
    *out = o;
    if (n_rec) *n_rec = nr;
    return 0;
}

// ---- second experiment: occluder hints -------------------------------------------------------------------
// The repetitions of a pair run one after another; a repetition that found an occluder leaves a hint (the
// ancestor block `up` levels above the leaf that held it); the next repetitions walk that subtree first and
// only walk from the root when it holds no hit.
namespace {
struct WalkResult { bool hit; Node* leaf; double inner, shapes; };
WalkResult walk_from(const Scene& sc, Node* start, const Ray<float>& r, float tmin, float tmax) {
    const StackRay sr = make_stack_ray(r);
    WalkResult w{false, nullptr, 0, 0};
    std::vector<Node*> stack;
    std::vector<int> group;
    const int leaf_size = sc.leaf_size;
    Node* cur = start;
    for (;;) {
        w.inner += 1;
        Node* pending[8]; int np = 0;
        Node* leaves[8]; int nl = 0;
        for (int k = 0; k < 8; ++k) {
            Node* e = cur->slot[k ^ sr.sign];
            if (!e) continue;
            if (!slab_f32(sr, e->fmin, e->fmax, tmin, tmax)) continue;
            if (e->as_leaf) leaves[nl++] = e; else pending[np++] = e;
        }
        for (int q = 0; q < nl; ++q) {
            group.clear();
            subtree_shapes(leaves[q], &group);
            for (int shape : group) {
                w.shapes += 1;
                const Hit<float> sh = sc.t32.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE
                                          ? triangle_occludes_f32(sc.t32.shapes[(size_t)shape], shape, r, tmin, tmax)
                                          : shape_hit(sc.t32, shape, r, tmin, tmax);
                if (sh.t > 0.0f && sh.t <= tmax) { w.hit = true; w.leaf = leaves[q]; return w; }
            }
        }
        for (int i = np - 1; i >= 1; --i) stack.push_back(pending[i]);
        if (np > 0) { cur = pending[0]; continue; }
        if (stack.empty()) break;
        cur = stack.back();
        stack.pop_back();
    }
    return w;
}
}  // namespace

// out: [0] rays, [1] today's visits, [2] today's shape tests, [3] hinted visits, [4] hinted shape tests,
//      [5] rays that had a hint, [6] hints that held a hit
extern "C" int sim_hints(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int up, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    const int leaf_size = sc.leaf_size;
    // parent links of the block tree
    std::unordered_map<Node*, Node*> parent;
    {
        std::vector<Node*> st{root};
        while (!st.empty()) {
            Node* n = st.back(); st.pop_back();
            for (Node* e : n->slot) if (e) { parent[e] = n; if (!e->as_leaf) st.push_back(e); }
        }
    }
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> l2p = sub(L.pos, hit.p);
                const V3<float> ldir = unit(l2p);
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                Node* hint = nullptr;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    const Ray<float> ray{hit.p, scale(1.0f / d, ld)};
                    out[0] += 1;
                    const WalkResult base = walk_from(sc, root, ray, 0.01f, d);
                    out[1] += base.inner; out[2] += base.shapes;
                    bool done = false;
                    if (hint) {
                        out[5] += 1;
                        const WalkResult w = walk_from(sc, hint, ray, 0.01f, d);
                        out[3] += w.inner; out[4] += w.shapes;
                        if (w.hit) { out[6] += 1; done = true; if (!base.hit) { g_err = "hint found a hit the full walk did not"; return -1; } }
                    }
                    if (!done) {
                        out[3] += base.inner; out[4] += base.shapes;
                        if (base.hit) {
                            Node* a = parent[base.leaf];
                            for (int u = 0; u < up && a != root; ++u) a = parent[a];
                            hint = a;
                        }
                    }
                }
            }
        }
    return 0;
}

// ---- third experiment: what do shadow rays spend their shape tests on? --------------------------------------
// out: [0] rays, [1] shape tests, [2] of them on the first `n_big` shapes of the scene (config 3: the 10 room triangles),
//      [3] leaves entered, [4] leaves entered that hold one of those shapes, [5] block visits
extern "C" int sim_big_shapes(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int n_big, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    const int leaf_size = sc.leaf_size;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    std::vector<int> group;
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Material<float>& m = T.materials[T.shapes[hit.shape].material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                if (saturate(dot(hit.normal, ldir)) == 0.0f && std::pow(saturate(dot(hit.normal, unit(add(ldir, r.d)))), m.param) == 0.0f) continue;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    const Ray<float> ray{hit.p, scale(1.0f / d, ld)};
                    const StackRay sr = make_stack_ray(ray);
                    out[0] += 1;
                    std::vector<Node*> stack;
                    Node* cur = root;
                    bool done = false;
                    while (!done) {
                        out[5] += 1;
                        Node* pending[8]; int np = 0;
                        for (int q = 0; q < 8 && !done; ++q) {
                            Node* e = cur->slot[q ^ sr.sign];
                            if (!e || !slab_f32(sr, e->fmin, e->fmax, 0.01f, d)) continue;
                            if (!e->as_leaf) { pending[np++] = e; continue; }
                            group.clear();
                            subtree_shapes(e, &group);
                            out[3] += 1;
                            bool has_big = false;
                            for (int shape : group) has_big = has_big || shape < n_big;
                            if (has_big) out[4] += 1;
                            for (int shape : group) {
                                out[1] += 1;
                                if (shape < n_big) out[2] += 1;
                                const Hit<float> sh = triangle_occludes_f32(T.shapes[(size_t)shape], shape, ray, 0.01f, d);
                                if (sh.t > 0.0f && sh.t <= d) { done = true; break; }
                            }
                        }
                        if (done) break;
                        for (int q = np - 1; q >= 1; --q) stack.push_back(pending[q]);
                        if (np > 0) { cur = pending[0]; continue; }
                        if (stack.empty()) break;
                        cur = stack.back(); stack.pop_back();
                    }
                }
            }
        }
    return 0;
}

// Per-ray block visits of every shadow ray, pair by pair (design-time: how uneven are the 64 walks of a wavefront,
// and would a lane that walks 2 or 4 repetitions of its pair one after the other even them out?).
// out rows: {light, pixel, n_rays, visits[0..reps)} as int32, reps <= 64.
extern "C" int64_t sim_ray_visits(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int32_t* out, int64_t cap_rows, int row_len) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    int64_t n = 0;
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                if (n >= cap_rows) return -1;
                int32_t* o = out + n * row_len;
                o[0] = lpos; o[1] = row * W + i;
                int nr = 0;
                for (int k = 0; k < reps && k < row_len - 3; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    o[3 + k] = -1;
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    TraceCounters tc; Hit<float> hh;
                    (void)trace_stack_f32(sc, Ray<float>{hit.p, scale(1.0f / d, ld)}, 0.01f, d, true, hh, &tc);
                    o[3 + k] = (int32_t)tc.inner;
                    ++nr;
                }
                o[2] = nr;
                if (nr > 0) ++n;
            }
        }
    return n;
}

// ---- a pair's shadow rays proven unoccluded by ONE conservative walk -----------------------------------------
// Every jittered light position lies in the cube of half-width R around the light, so every shadow ray of a (hit,
// light) pair lies in the "beam" { p + s (D + j) : s in [s0, 1], |j_a| <= R }, D = light - p.  Per axis the beam at
// parameter s spans [p_a + s (D_a - R), p_a + s (D_a + R)]; it meets a box's slab on that axis iff
// s (D_a - R) <= hi_a - p_a and s (D_a + R) >= lo_a - p_a: two half-lines in s.  If the intersection over the three
// axes and [s0, 1] is empty for a box, no ray of the pair meets it.  A walk that meets no LEAF box proves all the
// pair's rays unoccluded; it stops at the first leaf box it meets (then the pair's rays are traced as today).
// out: [0] pairs, [1] clear pairs, [2] rays of clear pairs, [3] rays, [4] beam visits of clear pairs, [5] beam visits of
// the others, [6] per-ray visits of clear pairs' rays, [7] per-ray visits of all rays, [8] clear pairs with an occluded ray (must be 0)
extern "C" int sim_ball_beam(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, double R, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 18; ++i) out[i] = 0;
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                int n_rays = 0, n_occ = 0; double ray_visits = 0;
                std::vector<Ray<float>> prays; std::vector<float> pdist; std::vector<char> pocc;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    TraceCounters tc; Hit<float> hh;
                    const bool oc = trace_stack_f32(sc, Ray<float>{hit.p, scale(1.0f / d, ld)}, 0.01f, d, true, hh, &tc);
                    n_occ += oc ? 1 : 0;
                    prays.push_back(Ray<float>{hit.p, scale(1.0f / d, ld)}); pdist.push_back(d); pocc.push_back(oc);
                    ray_visits += (double)tc.inner;
                    ++n_rays;
                }
                if (n_rays == 0) continue;
                // the beam
                const double p[3] = {hit.p.x, hit.p.y, hit.p.z};
                const double D[3] = {L.pos.x - p[0], L.pos.y - p[1], L.pos.z - p[2]};
                const double len = std::sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
                const double s0 = 0.01 / (len + R * 1.7320508) * 0.999, s1 = 1.0001;
                const int sign = (D[0] < 0 ? 1 : 0) | (D[1] < 0 ? 2 : 0) | (D[2] < 0 ? 4 : 0);
                auto meets = [&](const float* lo, const float* hi) {
                    double a = s0, b = s1;
                    for (int ax = 0; ax < 3; ++ax) {
                        const double c1 = D[ax] - R, c2 = D[ax] + R;       // s c1 <= hi - p ; s c2 >= lo - p
                        const double h1 = (double)hi[ax] - p[ax], l1 = (double)lo[ax] - p[ax];
                        if (c1 > 0) b = std::min(b, h1 / c1); else if (c1 < 0) a = std::max(a, h1 / c1); else if (h1 < 0) return false;
                        if (c2 > 0) a = std::max(a, l1 / c2); else if (c2 < 0) b = std::min(b, l1 / c2); else if (l1 > 0) return false;
                    }
                    return a <= b;
                };
                double visits = 0; bool clear = true;
                double prefix = 0; bool single = true;  // visits at the top of the walk with exactly one touched child, an inner one
                std::vector<Node*> stack;
                Node* cur = root;
                if (root->as_leaf) { clear = !meets(root->fmin, root->fmax); cur = nullptr; }
                while (cur && clear) {
                    visits += 1;
                    Node* pending[8]; int np = 0;
                    for (int k = 0; k < 8 && clear; ++k) {
                        Node* e = cur->slot[k ^ sign];
                        if (!e || !meets(e->fmin, e->fmax)) continue;
                        if (e->as_leaf) {
                            if (!std::getenv("SIM_SHAPE_BOXES")) { clear = false; continue; }
                            // the leaf's box is met: is any of its SHAPES' own boxes?
                            std::vector<int> group;
                            subtree_shapes(e, &group);
                            for (int shape : group) {
                                V3<double> lo, hi;
                                shape_bounds(sc.t64.shapes[(size_t)shape], lo, hi);
                                const float flo[3] = {(float)lo.x - 1e-6f, (float)lo.y - 1e-6f, (float)lo.z - 1e-6f}, fhi[3] = {(float)hi.x + 1e-6f, (float)hi.y + 1e-6f, (float)hi.z + 1e-6f};
                                out[12] += 1;
                                if (meets(flo, fhi)) { clear = false; break; }
                            }
                        } else pending[np++] = e;
                    }
                    if (single && clear && np == 1) prefix += 1; else single = false;
                    if (!clear) break;
                    for (int q = np - 1; q >= 1; --q) stack.push_back(pending[q]);
                    if (np > 0) { cur = pending[0]; continue; }
                    if (stack.empty()) break;
                    cur = stack.back(); stack.pop_back();
                }
                out[0] += 1; out[3] += n_rays; out[7] += ray_visits;
                if (clear) { out[1] += 1; out[2] += n_rays; out[4] += visits; out[6] += ray_visits; if (n_occ) out[8] += 1; }
                else {
                    out[5] += visits; out[9] += prefix; out[10] += ray_visits; out[11] += n_rays;
                    // hand-over: the rays resume from the beam's state (the block where it met a leaf + its stack)
                    out[13] += (double)stack.size();
                    for (size_t k = 0; k < prays.size(); ++k) {
                        bool found = false; double v = 0, sh = 0;
                        std::vector<Node*> starts{cur};
                        for (size_t q = stack.size(); q-- > 0;) starts.push_back(stack[q]);
                        const StackRay srk = make_stack_ray(prays[k]);
                        for (Node* st0 : starts) {
                            if (st0 != cur && !slab_f32(srk, st0->fmin, st0->fmax, 0.01f, pdist[k])) out[17] += 1;  // an inherited block the ray misses
                            const WalkResult w = walk_from(sc, st0, prays[k], 0.01f, pdist[k]);
                            v += w.inner; sh += w.shapes;
                            if (w.hit) { found = true; break; }
                        }
                        if (found != (bool)pocc[k]) out[16] += 1;
                        out[14] += v; out[15] += sh;
                    }
                }
            }
        }
    return 0;
}

// ---- in which order should an any-hit walk enter the children it meets? ---------------------------------------
// mode 0: ray order by octant (slot k ^ sign, what the device does); 1: longest chord (tf - tn) first; 2: nearest entry
// first; 3: most shapes first; 4: smallest box first.  Leaves are tested where they are met (as on the device).  From the
// root, every shadow ray.  out: [0] rays, [1] occluded rays, [2] block visits of occluded rays, [3] their shape tests,
// [4] block visits of the others, [5] their shape tests
extern "C" int sim_anyhit_order(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int mode, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (root->as_leaf) return 0;
    struct Cand { Node* e; float key; };
    auto walk = [&](const Ray<float>& r, float tmin, float tmax, double* visits, double* shapes) {
        const StackRay sr = make_stack_ray(r);
        std::vector<Node*> stack;
        std::vector<int> group;
        Node* cur = root;
        for (;;) {
            *visits += 1;
            Cand pending[8]; int np = 0;
            for (int k = 0; k < 8; ++k) {
                Node* e = cur->slot[k ^ sr.sign];
                if (!e) continue;
                float tn = tmin, tf = tmax;
                for (int a = 0; a < 3; ++a) {
                    const float lo = (sr.sign >> a) & 1 ? e->fmax[a] : e->fmin[a], hi = (sr.sign >> a) & 1 ? e->fmin[a] : e->fmax[a];
                    const float t0 = lo * sr.inv[a] - sr.oi[a], t1 = hi * sr.inv[a] - sr.oi[a];
                    if (t0 == t0) tn = std::max(tn, t0);
                    if (t1 == t1) tf = std::min(tf, t1);
                }
                if (!(tn <= tf)) continue;
                if (e->as_leaf) {
                    group.clear();
                    subtree_shapes(e, &group);
                    for (int shape : group) {
                        *shapes += 1;
                        const Hit<float> sh = T.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE ? triangle_occludes_f32(T.shapes[(size_t)shape], shape, r, tmin, tmax)
                                                                                                : shape_hit(T, shape, r, tmin, tmax);
                        if (sh.t > 0.0f && sh.t <= tmax) return true;
                    }
                    continue;
                }
                float key = (float)np;
                if (mode == 1) key = -(tf - tn);
                else if (mode == 2) key = tn;
                else if (mode == 3) key = -(float)e->count;
                else if (mode == 4) {
                    const float dx = e->fmax[0] - e->fmin[0], dy = e->fmax[1] - e->fmin[1], dz = e->fmax[2] - e->fmin[2];
                    key = dx * dy + dy * dz + dz * dx;
                }
                pending[np++] = Cand{e, key};
            }
            std::stable_sort(pending, pending + np, [](const Cand& a, const Cand& b) { return a.key < b.key; });
            for (int i = np - 1; i >= 1; --i) stack.push_back(pending[i].e);
            if (np > 0) { cur = pending[0].e; continue; }
            if (stack.empty()) return false;
            cur = stack.back();
            stack.pop_back();
        }
    };
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    double v = 0, st = 0;
                    const bool oc = walk(Ray<float>{hit.p, scale(1.0f / d, ld)}, 0.01f, d, &v, &st);
                    out[0] += 1;
                    if (oc) { out[1] += 1; out[2] += v; out[3] += st; } else { out[4] += v; out[5] += st; }
                }
            }
        }
    return 0;
}

// ---- sub-beams: the pairs the ball beam does not prove clear, tried again in groups of `group` repetitions ----------
// A group's beam is the box of ITS jittered light positions (not the whole ball): { p + s (D' + j) }, j within the
// group's per-axis range.  out: [0] non-clear pairs, [1] their rays, [2] sub-beams walked, [3] clear sub-beams,
// [4] rays of clear sub-beams, [5] block visits of all sub-beam walks, [6] per-ray visits of the rays in clear sub-beams,
// [7] per-ray visits of all rays of non-clear pairs, [8] clear sub-beams holding an occluded ray (must be 0),
// [9] occluded rays, [10] sub-beam visits in pairs whose rays are all occluded, [11] rays of such pairs,
// [12] per-ray shape tests of the rays in clear sub-beams, [13] per-ray shape tests of all rays of non-clear pairs
extern "C" int sim_sub_beam(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, double R, int group, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 16; ++i) out[i] = 0;
    if (root->as_leaf) return 0;
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                struct PR { V3<float> target; bool occ; double visits, shapes; };
                std::vector<PR> prays;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    TraceCounters tc; Hit<float> hh;
                    const bool oc = trace_stack_f32(sc, Ray<float>{hit.p, scale(1.0f / d, ld)}, 0.01f, d, true, hh, &tc);
                    prays.push_back(PR{shifted, oc, (double)tc.inner, (double)tc.leaf});
                }
                if (prays.empty()) continue;
                const double p[3] = {hit.p.x, hit.p.y, hit.p.z};
                // walk of the beam whose directions span [dlo, dhi] per axis; true = met no leaf box
                auto beam_clear = [&](const double* dlo, const double* dhi, double* visits) {
                    double len2 = 0;
                    for (int a = 0; a < 3; ++a) { const double far = std::max(std::fabs(dlo[a]), std::fabs(dhi[a])); len2 += far * far; }
                    const double s0 = 0.01 / std::sqrt(len2) * 0.999, s1 = 1.0001;
                    const double mid[3] = {(dlo[0] + dhi[0]) * 0.5, (dlo[1] + dhi[1]) * 0.5, (dlo[2] + dhi[2]) * 0.5};
                    const int sign = (mid[0] < 0 ? 1 : 0) | (mid[1] < 0 ? 2 : 0) | (mid[2] < 0 ? 4 : 0);
                    auto meets = [&](const float* lo, const float* hi) {
                        double a = s0, b = s1;
                        for (int ax = 0; ax < 3; ++ax) {
                            const double c1 = dlo[ax], c2 = dhi[ax];
                            const double h1 = (double)hi[ax] - p[ax], l1 = (double)lo[ax] - p[ax];
                            if (c1 > 0) b = std::min(b, h1 / c1); else if (c1 < 0) a = std::max(a, h1 / c1); else if (h1 < 0) return false;
                            if (c2 > 0) a = std::max(a, l1 / c2); else if (c2 < 0) b = std::min(b, l1 / c2); else if (l1 > 0) return false;
                        }
                        return a <= b;
                    };
                    std::vector<Node*> stack;
                    Node* cur = root;
                    while (cur) {
                        *visits += 1;
                        Node* pending[8]; int np = 0;
                        for (int k = 0; k < 8; ++k) {
                            Node* e = cur->slot[k ^ sign];
                            if (!e || !meets(e->fmin, e->fmax)) continue;
                            if (e->as_leaf) return false;
                            pending[np++] = e;
                        }
                        for (int q = np - 1; q >= 1; --q) stack.push_back(pending[q]);
                        if (np > 0) { cur = pending[0]; continue; }
                        if (stack.empty()) break;
                        cur = stack.back(); stack.pop_back();
                    }
                    return true;
                };
                {
                    double dlo[3], dhi[3], v = 0;
                    const double Lp[3] = {L.pos.x, L.pos.y, L.pos.z};
                    for (int a = 0; a < 3; ++a) { dlo[a] = Lp[a] - p[a] - R; dhi[a] = Lp[a] - p[a] + R; }
                    if (beam_clear(dlo, dhi, &v)) continue;  // the ball beam settles this pair
                }
                out[0] += 1; out[1] += (double)prays.size();
                int n_occ = 0; double all_visits = 0;
                for (const PR& q : prays) { n_occ += q.occ ? 1 : 0; all_visits += q.visits; out[13] += q.shapes; }
                out[7] += all_visits; out[9] += n_occ;
                const bool dark = n_occ == (int)prays.size();
                if (dark) out[11] += (double)prays.size();
                for (size_t g0 = 0; g0 < prays.size(); g0 += (size_t)group) {
                    const size_t g1 = std::min(prays.size(), g0 + (size_t)group);
                    double dlo[3] = {1e300, 1e300, 1e300}, dhi[3] = {-1e300, -1e300, -1e300};
                    for (size_t k = g0; k < g1; ++k) {
                        const double t[3] = {prays[k].target.x, prays[k].target.y, prays[k].target.z};
                        for (int a = 0; a < 3; ++a) { dlo[a] = std::min(dlo[a], t[a] - p[a]); dhi[a] = std::max(dhi[a], t[a] - p[a]); }
                    }
                    for (int a = 0; a < 3; ++a) { dlo[a] -= 1e-5; dhi[a] += 1e-5; }
                    double v = 0;
                    const bool clear = beam_clear(dlo, dhi, &v);
                    out[2] += 1; out[5] += v;
                    if (dark) out[10] += v;
                    if (!clear) continue;
                    out[3] += 1;
                    for (size_t k = g0; k < g1; ++k) {
                        out[4] += 1; out[6] += prays[k].visits; out[12] += prays[k].shapes;
                        if (prays[k].occ) out[8] += 1;
                    }
                }
            }
        }
    return 0;
}

// ---- would the occluder of a pair's CENTRAL ray (p -> the light's centre) decide its other rays? ---------------
// out: [0] rays of pairs that are not clear by sim_ball_beam's criterion (approximated: pairs with an occluded ray or
// whose central ray is occluded are surely not clear; lit pairs are left out), [1] occluded rays among them,
// [2] occluded rays that the central ray's triangle also occludes, [3] ... that some triangle of that triangle's leaf
// occludes, [4] pairs considered, [5] pairs whose central ray is occluded, [6] visits of the central rays
extern "C" int sim_central(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    for (int i = 0; i < 7; ++i) out[i] = 0;
    // leaf of every shape: the as_leaf node that holds it
    std::unordered_map<int, Node*> leaf_of;
    {
        std::vector<Node*> todo{effective(sc.wide_root)};
        while (!todo.empty()) {
            Node* x = todo.back(); todo.pop_back();
            if (x->as_leaf || x->leaf) { std::vector<int> g; subtree_shapes(x, &g); for (int q : g) leaf_of[q] = x; continue; }
            for (Node* e : x->slot) if (e) todo.push_back(e);
        }
    }
    auto occludes = [&](int shape, const Ray<float>& ray, float dist) {
        const Hit<float> h = T.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE ? triangle_occludes_f32(T.shapes[(size_t)shape], shape, ray, 0.01f, dist)
                                                                                : shape_hit(T, shape, ray, 0.01f, dist);
        return h.t > 0.0f && h.t <= dist;
    };
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Material<float>& m = T.materials[T.shapes[hit.shape].material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                // central ray
                const V3<float> cd = sub(L.pos, hit.p);
                const float cdist = norm(cd);
                TraceCounters tc; Hit<float> ch;
                const bool cocc = trace_stack_f32(sc, Ray<float>{hit.p, scale(1.0f / cdist, cd)}, 0.01f, cdist, true, ch, &tc);
                std::vector<int> leafshapes;
                if (cocc) { auto it = leaf_of.find(ch.shape); if (it != leaf_of.end()) subtree_shapes(it->second, &leafshapes); }
                int n_rays = 0, n_occ = 0, by_tri = 0, by_leaf = 0;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    const Ray<float> ray{hit.p, scale(1.0f / d, ld)};
                    Hit<float> hh;
                    const bool occ = trace_stack_f32(sc, ray, 0.01f, d, true, hh, nullptr);
                    ++n_rays;
                    if (!occ) continue;
                    ++n_occ;
                    if (cocc && occludes(ch.shape, ray, d)) { ++by_tri; ++by_leaf; continue; }
                    for (int q : leafshapes) if (occludes(q, ray, d)) { ++by_leaf; break; }
                }
                if (n_occ == 0 && !cocc) continue;
                out[0] += n_rays; out[1] += n_occ; out[2] += by_tri; out[3] += by_leaf; out[4] += 1; out[5] += cocc; out[6] += (double)tc.inner;
            }
        }
    return 0;
}

// ---- the beam walk continued past the first K leaves it meets ------------------------------------------------
// Design question: k_pair_beam stops at the first leaf box the beam meets and the pair's rays resume there.  If it
// went on, collecting the first K leaves met, the rays would test those leaves (own box test, then shapes) and resume
// later in the walk — or not walk at all when the beam's walk ends with at most K leaves.
// out: [0] non-clear pairs, [1] their rays, [2] beam visits, [3] pairs whose walk ended (no ray walk at all),
//      [4] ray block visits (resumed, inherited blocks filtered by their boxes), [5] ray leaf-box tests of collected leaves,
//      [6] ray shape tests, [7] mismatches against the per-ray walk (must be 0), [8] handed-over stack entries
extern "C" int sim_beam_k(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, double R, int K, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 9; ++i) out[i] = 0;
    if (root->as_leaf) return 0;
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Material<float>& m = T.materials[T.shapes[hit.shape].material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                std::vector<Ray<float>> prays; std::vector<float> pdist; std::vector<char> pocc;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    Hit<float> hh;
                    pocc.push_back(trace_stack_f32(sc, Ray<float>{hit.p, scale(1.0f / d, ld)}, 0.01f, d, true, hh, nullptr));
                    prays.push_back(Ray<float>{hit.p, scale(1.0f / d, ld)}); pdist.push_back(d);
                }
                if (prays.empty()) continue;
                const double p[3] = {hit.p.x, hit.p.y, hit.p.z};
                const double D[3] = {L.pos.x - p[0], L.pos.y - p[1], L.pos.z - p[2]};
                const double len = std::sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
                const double s0 = 0.01 / (len + R * 1.7320508) * 0.999, s1 = 1.0001;
                const int sign = (D[0] < 0 ? 1 : 0) | (D[1] < 0 ? 2 : 0) | (D[2] < 0 ? 4 : 0);
                auto meets = [&](const float* lo, const float* hi) {
                    double a = s0, b = s1;
                    for (int ax = 0; ax < 3; ++ax) {
                        const double c1 = D[ax] - R, c2 = D[ax] + R;
                        const double h1 = (double)hi[ax] - p[ax], l1 = (double)lo[ax] - p[ax];
                        if (c1 > 0) b = std::min(b, h1 / c1); else if (c1 < 0) a = std::max(a, h1 / c1); else if (h1 < 0) return false;
                        if (c2 > 0) a = std::max(a, l1 / c2); else if (c2 < 0) b = std::min(b, l1 / c2); else if (l1 > 0) return false;
                    }
                    return a <= b;
                };
                double visits = 0;
                std::vector<Node*> stack, leaves;
                Node* cur = root;
                bool ended = false;
                for (;;) {
                    visits += 1;
                    Node* pending[8]; int np = 0;
                    std::vector<Node*> here;
                    for (int k = 0; k < 8; ++k) {
                        Node* e = cur->slot[k ^ sign];
                        if (!e || !meets(e->fmin, e->fmax)) continue;
                        if (e->as_leaf) here.push_back(e); else pending[np++] = e;
                    }
                    if ((int)(leaves.size() + here.size()) > K) break;  // hand over with `cur` to be visited again by the rays
                    leaves.insert(leaves.end(), here.begin(), here.end());
                    for (int q = np - 1; q >= 1; --q) stack.push_back(pending[q]);
                    if (np > 0) { cur = pending[0]; continue; }
                    if (stack.empty()) { ended = true; break; }
                    cur = stack.back(); stack.pop_back();
                }
                if (ended && leaves.empty()) continue;  // a clear pair
                out[0] += 1; out[1] += (double)prays.size(); out[2] += visits; out[3] += ended ? 1 : 0; out[8] += ended ? 0 : (double)stack.size();
                std::vector<int> group;
                for (size_t k = 0; k < prays.size(); ++k) {
                    const StackRay srk = make_stack_ray(prays[k]);
                    bool found = false;
                    for (Node* lf : leaves) {
                        out[5] += 1;
                        if (!slab_f32(srk, lf->fmin, lf->fmax, 0.01f, pdist[k])) continue;
                        group.clear(); subtree_shapes(lf, &group);
                        for (int shape : group) {
                            out[6] += 1;
                            const Hit<float> shh = T.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE ? triangle_occludes_f32(T.shapes[(size_t)shape], shape, prays[k], 0.01f, pdist[k])
                                                                                                     : shape_hit(T, shape, prays[k], 0.01f, pdist[k]);
                            if (shh.t > 0.0f && shh.t <= pdist[k]) { found = true; break; }
                        }
                        if (found) break;
                    }
                    if (!found && !ended) {
                        std::vector<Node*> starts{cur};
                        for (size_t q = stack.size(); q-- > 0;) starts.push_back(stack[q]);
                        for (Node* st0 : starts) {
                            if (st0 != cur && !slab_f32(srk, st0->fmin, st0->fmax, 0.01f, pdist[k])) continue;  // filtered at set-up
                            const WalkResult w = walk_from(sc, st0, prays[k], 0.01f, pdist[k]);
                            out[4] += w.inner; out[6] += w.shapes;
                            if (w.hit) { found = true; break; }
                        }
                    }
                    if (found != (bool)pocc[k]) out[7] += 1;
                }
            }
        }
    return 0;
}

// ---- leaf slabs: a second bound per leaf, two parallel planes around its shapes ---------------------------------
// A leaf's axis-aligned box is loose around a tilted patch of surface; a shadow ray that meets the box but not the patch
// pays for every shape of the leaf.  Here every leaf also gets a slab {x : dmin <= n . x <= dmax}; a ray tests its
// shapes only if, inside the box's [tn, tf], it also crosses the slab.  mode 0: n = the leaf's mean triangle normal;
// mode 1: n = the best of 10 fixed diagonals (4 body, 6 face).  out (occluded rays / others): [0]/[6] rays,
// [1]/[7] block visits, [2]/[8] leaf boxes met, [3]/[9] of which the slab is met too, [4]/[10] shape tests without the
// slab, [5]/[11] shape tests with it; [12] rays whose answer the slab changed (must be 0).
extern "C" int sim_leaf_slab(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int mode, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 16; ++i) out[i] = 0;
    if (root->as_leaf) return 0;
    struct Slab { double n[3], dmin, dmax; bool any; double n2[3], d2min, d2max; };
    std::unordered_map<Node*, Slab> slabs;
    const int inner_max = std::getenv("SIM_INNER_MAX") ? std::atoi(std::getenv("SIM_INNER_MAX")) : 64;  // mode 2: inner children with at most this many shapes get a slab too
    static const double dirs[10][3] = {{1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}, {1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}};
    auto slab_of = [&](Node* lf) -> const Slab& {
        auto it = slabs.find(lf);
        if (it != slabs.end()) return it->second;
        Slab sl{};
        sl.any = true;
        std::vector<int> g;
        subtree_shapes(lf, &g);
        std::vector<std::array<double, 3>> pts;
        double nsum[3] = {0, 0, 0};
        for (int shape : g) {
            const Shape<float>& sh = T.shapes[(size_t)shape];
            if (sh.kind != RT_SHAPE_TRIANGLE) { sl.any = false; break; }
            const double a[3] = {sh.a.x, sh.a.y, sh.a.z}, b[3] = {sh.b.x, sh.b.y, sh.b.z}, c[3] = {sh.c.x, sh.c.y, sh.c.z};
            const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            if (n[0] * nsum[0] + n[1] * nsum[1] + n[2] * nsum[2] < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
            for (int k = 0; k < 3; ++k) nsum[k] += n[k];
            pts.push_back({a[0], a[1], a[2]}); pts.push_back({b[0], b[1], b[2]}); pts.push_back({c[0], c[1], c[2]});
        }
        if (sl.any) {
            auto span = [&](const double* n, double* lo, double* hi) {
                *lo = 1e300; *hi = -1e300;
                for (auto& p : pts) { const double v = n[0] * p[0] + n[1] * p[1] + n[2] * p[2]; *lo = std::min(*lo, v); *hi = std::max(*hi, v); }
            };
            if (mode != 1) {
                const double len = std::sqrt(nsum[0] * nsum[0] + nsum[1] * nsum[1] + nsum[2] * nsum[2]);
                if (len > 0) { for (int k = 0; k < 3; ++k) sl.n[k] = nsum[k] / len; span(sl.n, &sl.dmin, &sl.dmax); } else sl.any = false;
            } else {
                double best = 1e300;
                for (auto& d : dirs) {
                    const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    double lo, hi; span(d, &lo, &hi);
                    if ((hi - lo) / len < best) { best = (hi - lo) / len; for (int k = 0; k < 3; ++k) sl.n[k] = d[k]; sl.dmin = lo; sl.dmax = hi; }
                }
            }
            const double pad = 1e-5 * (std::fabs(sl.dmin) + std::fabs(sl.dmax) + 1e-3);
            sl.dmin -= pad; sl.dmax += pad;
            if (sl.any && mode == 3) {  // the in-plane direction (of 16) along which the leaf is narrowest
                double u[3] = {0, 0, 0};
                const int ax = std::fabs(sl.n[0]) < std::fabs(sl.n[1]) ? (std::fabs(sl.n[0]) < std::fabs(sl.n[2]) ? 0 : 2) : (std::fabs(sl.n[1]) < std::fabs(sl.n[2]) ? 1 : 2);
                double e[3] = {0, 0, 0}; e[ax] = 1;
                u[0] = sl.n[1] * e[2] - sl.n[2] * e[1]; u[1] = sl.n[2] * e[0] - sl.n[0] * e[2]; u[2] = sl.n[0] * e[1] - sl.n[1] * e[0];
                const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                for (int k = 0; k < 3; ++k) u[k] /= ul;
                const double v[3] = {sl.n[1] * u[2] - sl.n[2] * u[1], sl.n[2] * u[0] - sl.n[0] * u[2], sl.n[0] * u[1] - sl.n[1] * u[0]};
                double best = 1e300;
                for (int q = 0; q < 16; ++q) {
                    const double c = std::cos(q * M_PI / 16), s_ = std::sin(q * M_PI / 16);
                    const double d[3] = {c * u[0] + s_ * v[0], c * u[1] + s_ * v[1], c * u[2] + s_ * v[2]};
                    double lo, hi; span(d, &lo, &hi);
                    if (hi - lo < best) { best = hi - lo; for (int k = 0; k < 3; ++k) sl.n2[k] = d[k]; sl.d2min = lo - pad; sl.d2max = hi + pad; }
                }
            }
        }
        return slabs.emplace(lf, sl).first->second;
    };
    // returns 0 = unoccluded, 1 = occluded; counts into c[0..4] = visits, leaf boxes, slabs passed, shapes plain, shapes with slab
    auto walk = [&](const Ray<float>& r, float tmin, float tmax, double* c, bool use_slab) {
        const StackRay sr = make_stack_ray(r);
        std::vector<Node*> stack;
        std::vector<int> group;
        Node* cur = root;
        for (;;) {
            c[0] += 1;
            Node* pending[8]; int np = 0;
            for (int k = 0; k < 8; ++k) {
                Node* e = cur->slot[k ^ sr.sign];
                if (!e) continue;
                float tn = tmin, tf = tmax;
                for (int a = 0; a < 3; ++a) {
                    const float lo = (sr.sign >> a) & 1 ? e->fmax[a] : e->fmin[a], hi = (sr.sign >> a) & 1 ? e->fmin[a] : e->fmax[a];
                    const float t0 = lo * sr.inv[a] - sr.oi[a], t1 = hi * sr.inv[a] - sr.oi[a];
                    if (t0 == t0) tn = std::max(tn, t0);
                    if (t1 == t1) tf = std::min(tf, t1);
                }
                if (!(tn <= tf)) continue;
                if (!e->as_leaf && use_slab && mode >= 2 && e->count <= inner_max) {
                    c[4] += 1;
                    const Slab& sl = slab_of(e);
                    if (sl.any) {
                        const double so = sl.n[0] * r.p.x + sl.n[1] * r.p.y + sl.n[2] * r.p.z, sd = sl.n[0] * r.d.x + sl.n[1] * r.d.y + sl.n[2] * r.d.z;
                        const double a0 = so + (double)tn * sd, a1 = so + (double)tf * sd;
                        const double m = 1e-5 * (std::fabs(a0) + std::fabs(a1));
                        if (std::max(a0, a1) + m < sl.dmin || std::min(a0, a1) - m > sl.dmax) continue;
                    }
                }
                if (e->as_leaf) {
                    c[1] += 1;
                    if (use_slab) {
                        const Slab& sl = slab_of(e);
                        if (sl.any) {
                            const double so = sl.n[0] * r.p.x + sl.n[1] * r.p.y + sl.n[2] * r.p.z, sd = sl.n[0] * r.d.x + sl.n[1] * r.d.y + sl.n[2] * r.d.z;
                            const double a0 = so + (double)tn * sd, a1 = so + (double)tf * sd;
                            const double m = 1e-5 * (std::fabs(a0) + std::fabs(a1));
                            if (std::max(a0, a1) + m < sl.dmin || std::min(a0, a1) - m > sl.dmax) continue;
                            if (mode == 3) {
                                // inside the first slab the ray runs from t0s to t1s; the second slab over that piece
                                double t0s = tn, t1s = tf;
                                if (sd != 0 && !std::getenv("SIM_NOCLIP")) { const double ta = (sl.dmin - so) / sd, tb = (sl.dmax - so) / sd; t0s = std::max(t0s, std::min(ta, tb)); t1s = std::min(t1s, std::max(ta, tb)); }
                                const double so2 = sl.n2[0] * r.p.x + sl.n2[1] * r.p.y + sl.n2[2] * r.p.z, sd2 = sl.n2[0] * r.d.x + sl.n2[1] * r.d.y + sl.n2[2] * r.d.z;
                                const double b0 = so2 + t0s * sd2, b1 = so2 + t1s * sd2;
                                if (std::max(b0, b1) + m < sl.d2min || std::min(b0, b1) - m > sl.d2max) continue;
                            }
                        }
                    }
                    c[2] += 1;
                    group.clear();
                    subtree_shapes(e, &group);
                    for (int shape : group) {
                        c[3] += 1;
                        const Hit<float> sh = T.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE ? triangle_occludes_f32(T.shapes[(size_t)shape], shape, r, tmin, tmax)
                                                                                                : shape_hit(T, shape, r, tmin, tmax);
                        if (sh.t > 0.0f && sh.t <= tmax) return true;
                    }
                    continue;
                }
                pending[np++] = e;
            }
            for (int i = np - 1; i >= 1; --i) stack.push_back(pending[i]);
            if (np > 0) { cur = pending[0]; continue; }
            if (stack.empty()) return false;
            cur = stack.back();
            stack.pop_back();
        }
    };
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    const Ray<float> sray{hit.p, scale(1.0f / d, ld)};
                    double c0[5] = {0, 0, 0, 0, 0}, c1[5] = {0, 0, 0, 0, 0};
                    const bool oc = walk(sray, 0.01f, d, c0, false);
                    const bool oc1 = walk(sray, 0.01f, d, c1, true);
                    if (oc != oc1) out[12] += 1;
                    double* o = oc ? out : out + 6;
                    o[0] += 1; o[1] += c1[0]; o[2] += c1[1]; o[3] += c1[2]; o[4] += c0[3]; o[5] += c1[3];
                }
            }
        }
    return 0;
}

// ---- pair beams with leaf slabs: a beam that meets a leaf's box must also meet the leaf's slab (sim_leaf_slab, mean
// normal) to count as meeting the leaf.  out: [0] pairs, [1] rays, [2] pairs clear by boxes alone, [3] their rays,
// [4] pairs clear with slabs, [5] their rays, [6] clear pairs (with slabs) holding an occluded ray (must be 0),
// [7] beam visits without slabs, [8] beam visits with slabs (walks go on past leaves whose slab is missed),
// [9] leaf boxes met by beams in the slab walk, [10] rays of pairs that are not clear with slabs but whose rays are all unoccluded,
// [11] rays of pairs with all rays occluded, [12] occluded rays
extern "C" int sim_ball_beam_slab(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, double R, double* out, double* hist) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 16; ++i) out[i] = 0;
    if (root->as_leaf) return 0;
    struct Slab { double n[3], dmin, dmax; bool any; };
    std::unordered_map<Node*, Slab> slabs;
    auto slab_of = [&](Node* lf) -> const Slab& {
        auto it = slabs.find(lf);
        if (it != slabs.end()) return it->second;
        Slab sl{};
        sl.any = true;
        std::vector<int> g;
        subtree_shapes(lf, &g);
        std::vector<std::array<double, 3>> pts;
        double nsum[3] = {0, 0, 0};
        for (int shape : g) {
            const Shape<float>& sh = T.shapes[(size_t)shape];
            if (sh.kind != RT_SHAPE_TRIANGLE) { sl.any = false; break; }
            const double a[3] = {sh.a.x, sh.a.y, sh.a.z}, b[3] = {sh.b.x, sh.b.y, sh.b.z}, c[3] = {sh.c.x, sh.c.y, sh.c.z};
            const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            if (n[0] * nsum[0] + n[1] * nsum[1] + n[2] * nsum[2] < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
            for (int k = 0; k < 3; ++k) nsum[k] += n[k];
            pts.push_back({a[0], a[1], a[2]}); pts.push_back({b[0], b[1], b[2]}); pts.push_back({c[0], c[1], c[2]});
        }
        if (sl.any) {
            const double len = std::sqrt(nsum[0] * nsum[0] + nsum[1] * nsum[1] + nsum[2] * nsum[2]);
            if (len > 0) {
                for (int k = 0; k < 3; ++k) sl.n[k] = nsum[k] / len;
                sl.dmin = 1e300; sl.dmax = -1e300;
                for (auto& p : pts) { const double v = sl.n[0] * p[0] + sl.n[1] * p[1] + sl.n[2] * p[2]; sl.dmin = std::min(sl.dmin, v); sl.dmax = std::max(sl.dmax, v); }
                const double pad = 1e-5 * (std::fabs(sl.dmin) + std::fabs(sl.dmax) + 1e-3);
                sl.dmin -= pad; sl.dmax += pad;
            } else sl.any = false;
        }
        return slabs.emplace(lf, sl).first->second;
    };
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                int n_rays = 0, n_occ = 0;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    Hit<float> hh;
                    const bool oc = trace_stack_f32(sc, Ray<float>{hit.p, scale(1.0f / d, ld)}, 0.01f, d, true, hh, nullptr);
                    n_occ += oc ? 1 : 0;
                    ++n_rays;
                }
                if (n_rays == 0) continue;
                const double p[3] = {hit.p.x, hit.p.y, hit.p.z};
                const double D[3] = {L.pos.x - p[0], L.pos.y - p[1], L.pos.z - p[2]};
                const double len = std::sqrt(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
                const double s0 = 0.01 / (len + R * 1.7320508) * 0.999, s1 = 1.0001;
                const int sign = (D[0] < 0 ? 1 : 0) | (D[1] < 0 ? 2 : 0) | (D[2] < 0 ? 4 : 0);
                auto meets = [&](const float* lo, const float* hi, double* a_out, double* b_out) {
                    double a = s0, b = s1;
                    for (int ax = 0; ax < 3; ++ax) {
                        const double c1 = D[ax] - R, c2 = D[ax] + R;
                        const double h1 = (double)hi[ax] - p[ax], l1 = (double)lo[ax] - p[ax];
                        if (c1 > 0) b = std::min(b, h1 / c1); else if (c1 < 0) a = std::max(a, h1 / c1); else if (h1 < 0) return false;
                        if (c2 > 0) a = std::max(a, l1 / c2); else if (c2 < 0) b = std::min(b, l1 / c2); else if (l1 > 0) return false;
                    }
                    *a_out = a; *b_out = b;
                    return a <= b;
                };
                for (int pass = 0; pass < 2; ++pass) {
                    double visits = 0; bool clear = true;
                    std::vector<Node*> stack;
                    Node* cur = root;
                    while (cur && clear) {
                        visits += 1;
                        Node* pending[8]; int np = 0;
                        for (int k = 0; k < 8 && clear; ++k) {
                            Node* e = cur->slot[k ^ sign];
                            double a, b;
                            if (!e || !meets(e->fmin, e->fmax, &a, &b)) continue;
                            if (e->as_leaf) {
                                if (pass == 0) { clear = false; continue; }
                                out[9] += 1;
                                const Slab& sl = slab_of(e);
                                if (!sl.any) { clear = false; continue; }
                                const double np_ = sl.n[0] * p[0] + sl.n[1] * p[1] + sl.n[2] * p[2], nD = sl.n[0] * D[0] + sl.n[1] * D[1] + sl.n[2] * D[2];
                                const double n1 = std::getenv("SIM_L1") ? R * (std::fabs(sl.n[0]) + std::fabs(sl.n[1]) + std::fabs(sl.n[2])) : R * 1.00001;  // the jittered light positions lie in the BALL of radius R: |n . j| <= R |n|_2
                                const double lo = std::min(np_ + a * (nD - n1), np_ + b * (nD - n1)), hi = std::max(np_ + a * (nD + n1), np_ + b * (nD + n1));
                                if (hi < sl.dmin || lo > sl.dmax) continue;  // the beam misses the slab inside the box
                                clear = false;
                            } else pending[np++] = e;
                        }
                        if (!clear) break;
                        for (int q = np - 1; q >= 1; --q) stack.push_back(pending[q]);
                        if (np > 0) { cur = pending[0]; continue; }
                        if (stack.empty()) break;
                        cur = stack.back(); stack.pop_back();
                    }
                    if (pass == 0) { out[7] += visits; if (clear) { out[2] += 1; out[3] += n_rays; } }
                    else {
                        if (hist) { const int b = std::min(63, (int)visits); hist[(clear ? 0 : 64) + b] += 1; hist[128 + b] += clear ? n_rays : 0; }
                        out[8] += visits;
                        if (clear) { out[4] += 1; out[5] += n_rays; if (n_occ) out[6] += 1; }
                        else if (n_occ == 0) out[10] += n_rays;
                    }
                }
                out[0] += 1; out[1] += n_rays; out[12] += n_occ;
                if (n_occ == n_rays) out[11] += n_rays;
            }
        }
    return 0;
}

// ---- sub-leaves: a leaf of up to 31 shapes as groups of G consecutive shapes, each with a box and two slabs of its own ----
// out (occluded rays / others, stride 8): [0] rays, [1] block visits, [2] leaf boxes met, [3] leaves whose slabs are met,
// [4] sub-leaves tested, [5] sub-leaves met, [6] shape tests with leaf slabs only, [7] shape tests with sub-leaves; [16] answers changed
extern "C" int sim_sub_leaf(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int G, double* out) {
    const Scene& sc = s->sc;
    const int W = spec->width, H = spec->height, reps = spec->soft_shadow_monte_carlo_repetitions;
    const CameraF64 c64 = new_camera(sc, W, H);
    const Camera<float> cf = narrow_camera<float>(c64);
    const Tables<float>& T = sc.t32;
    Node* root = effective(sc.wide_root);
    for (int i = 0; i < 20; ++i) out[i] = 0;
    if (root->as_leaf) return 0;
    struct Bound { double lo[3], hi[3], n[3], dmin, dmax, n2[3], d2min, d2max; bool any; std::vector<int> shapes; };
    struct LeafB { Bound leaf; std::vector<Bound> subs; };
    std::unordered_map<Node*, LeafB> cache;
    auto make_bound = [&](const std::vector<int>& g) {
        Bound b{};
        b.shapes = g; b.any = true;
        for (int k = 0; k < 3; ++k) { b.lo[k] = 1e300; b.hi[k] = -1e300; }
        std::vector<std::array<double, 3>> pts;
        double ns[3] = {0, 0, 0};
        for (int shape : g) {
            const Shape<float>& sh = T.shapes[(size_t)shape];
            if (sh.kind != RT_SHAPE_TRIANGLE) { b.any = false; continue; }
            const double a[3] = {sh.a.x, sh.a.y, sh.a.z}, bb[3] = {sh.b.x, sh.b.y, sh.b.z}, c[3] = {sh.c.x, sh.c.y, sh.c.z};
            const double e1[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            if (n[0] * ns[0] + n[1] * ns[1] + n[2] * ns[2] < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
            for (int k = 0; k < 3; ++k) ns[k] += n[k];
            pts.push_back({a[0], a[1], a[2]}); pts.push_back({bb[0], bb[1], bb[2]}); pts.push_back({c[0], c[1], c[2]});
        }
        for (auto& p : pts) for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], p[k]); b.hi[k] = std::max(b.hi[k], p[k]); }
        const double len = std::sqrt(ns[0] * ns[0] + ns[1] * ns[1] + ns[2] * ns[2]);
        if (!b.any || !(len > 0)) { b.any = false; return b; }
        for (int k = 0; k < 3; ++k) b.n[k] = ns[k] / len;
        auto span = [&](const double* d, double* lo, double* hi) {
            *lo = 1e300; *hi = -1e300;
            for (auto& p : pts) { const double v = d[0] * p[0] + d[1] * p[1] + d[2] * p[2]; *lo = std::min(*lo, v); *hi = std::max(*hi, v); }
        };
        span(b.n, &b.dmin, &b.dmax);
        const int ax = std::fabs(b.n[0]) < std::fabs(b.n[1]) ? (std::fabs(b.n[0]) < std::fabs(b.n[2]) ? 0 : 2) : (std::fabs(b.n[1]) < std::fabs(b.n[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[ax] = 1;
        double u[3] = {b.n[1] * e[2] - b.n[2] * e[1], b.n[2] * e[0] - b.n[0] * e[2], b.n[0] * e[1] - b.n[1] * e[0]};
        const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int k = 0; k < 3; ++k) u[k] /= ul;
        const double v[3] = {b.n[1] * u[2] - b.n[2] * u[1], b.n[2] * u[0] - b.n[0] * u[2], b.n[0] * u[1] - b.n[1] * u[0]};
        double best = 1e300;
        for (int q = 0; q < 16; ++q) {
            const double c = std::cos(q * M_PI / 16), s_ = std::sin(q * M_PI / 16);
            const double d[3] = {c * u[0] + s_ * v[0], c * u[1] + s_ * v[1], c * u[2] + s_ * v[2]};
            double lo, hi; span(d, &lo, &hi);
            if (hi - lo < best) { best = hi - lo; for (int k = 0; k < 3; ++k) b.n2[k] = d[k]; b.d2min = lo; b.d2max = hi; }
        }
        const double pad = 1.5e-5 * ((b.hi[0] - b.lo[0]) + (b.hi[1] - b.lo[1]) + (b.hi[2] - b.lo[2])) + 1e-12;
        b.dmin -= pad; b.dmax += pad; b.d2min -= pad; b.d2max += pad;
        return b;
    };
    auto bounds_of = [&](Node* lf) -> const LeafB& {
        auto it = cache.find(lf);
        if (it != cache.end()) return it->second;
        LeafB lb;
        std::vector<int> g;
        subtree_shapes(lf, &g);
        lb.leaf = make_bound(g);
        for (size_t k = 0; k < g.size(); k += (size_t)G) lb.subs.push_back(make_bound(std::vector<int>(g.begin() + (long)k, g.begin() + (long)std::min(g.size(), k + (size_t)G))));
        return cache.emplace(lf, std::move(lb)).first->second;
    };
    // the device's test: the ray's interval inside the box (never culls by itself), first slab at its ends, second slab on the piece inside the first
    auto misses = [&](const Bound& b, const Ray<float>& r, double tmin, double tmax, bool own_box) {
        if (!b.any) return false;
        double tn = tmin, tf = tmax;
        if (own_box)
            for (int a = 0; a < 3; ++a) {
                const double o = a == 0 ? r.p.x : a == 1 ? r.p.y : r.p.z, d = a == 0 ? r.d.x : a == 1 ? r.d.y : r.d.z;
                if (d == 0) continue;
                const double t0 = (b.lo[a] - o) / d, t1 = (b.hi[a] - o) / d;
                tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1));
            }
        tn -= 6e-5 * std::fabs(tn); tf += 6e-5 * std::fabs(tf);
        const double so = b.n[0] * r.p.x + b.n[1] * r.p.y + b.n[2] * r.p.z, sd = b.n[0] * r.d.x + b.n[1] * r.d.y + b.n[2] * r.d.z;
        const double a0 = so + tn * sd, a1 = so + tf * sd, m = 1.5e-5 * (std::fabs(so) + std::fabs(tf * sd));
        if (std::max(a0, a1) + m < b.dmin || std::min(a0, a1) - m > b.dmax) return true;
        double t0s = tn, t1s = tf;
        if (sd != 0) { const double ta = (b.dmin - m - so) / sd, tb = (b.dmax + m - so) / sd; t0s = std::max(t0s, std::min(ta, tb)); t1s = std::min(t1s, std::max(ta, tb)); }
        const double so2 = b.n2[0] * r.p.x + b.n2[1] * r.p.y + b.n2[2] * r.p.z, sd2 = b.n2[0] * r.d.x + b.n2[1] * r.d.y + b.n2[2] * r.d.z;
        const double b0 = so2 + t0s * sd2, b1 = so2 + t1s * sd2;
        return std::max(b0, b1) + m < b.d2min || std::min(b0, b1) - m > b.d2max;
    };
    auto walk = [&](const Ray<float>& r, float tmin, float tmax, double* c, bool use_sub) {
        const StackRay sr = make_stack_ray(r);
        std::vector<Node*> stack;
        Node* cur = root;
        for (;;) {
            c[0] += 1;
            Node* pending[8]; int np = 0;
            for (int k = 0; k < 8; ++k) {
                Node* e = cur->slot[k ^ sr.sign];
                if (!e) continue;
                float tn = tmin, tf = tmax;
                for (int a = 0; a < 3; ++a) {
                    const float lo = (sr.sign >> a) & 1 ? e->fmax[a] : e->fmin[a], hi = (sr.sign >> a) & 1 ? e->fmin[a] : e->fmax[a];
                    const float t0 = lo * sr.inv[a] - sr.oi[a], t1 = hi * sr.inv[a] - sr.oi[a];
                    if (t0 == t0) tn = std::max(tn, t0);
                    if (t1 == t1) tf = std::min(tf, t1);
                }
                if (!(tn <= tf)) continue;
                if (e->as_leaf) {
                    c[1] += 1;
                    const LeafB& lb = bounds_of(e);
                    if (misses(lb.leaf, r, tmin, tmax, true)) continue;
                    c[2] += 1;
                    bool hit = false;
                    for (const Bound& sb : lb.subs) {
                        if (use_sub && lb.subs.size() > 1) {
                            c[3] += 1;
                            if (misses(sb, r, tmin, tmax, true)) continue;
                            c[4] += 1;
                        }
                        for (int shape : sb.shapes) {  // (the device tests every shape of every surviving group at once)
                            c[5] += 1;
                            const Hit<float> sh = T.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE ? triangle_occludes_f32(T.shapes[(size_t)shape], shape, r, tmin, tmax)
                                                                                                    : shape_hit(T, shape, r, tmin, tmax);
                            if (sh.t > 0.0f && sh.t <= tmax) hit = true;
                        }
                    }
                    if (hit) return true;
                    continue;
                }
                pending[np++] = e;
            }
            for (int i = np - 1; i >= 1; --i) stack.push_back(pending[i]);
            if (np > 0) { cur = pending[0]; continue; }
            if (stack.empty()) return false;
            cur = stack.back();
            stack.pop_back();
        }
    };
    for (int row = 0; row < H; ++row)
        for (int i = 0; i < W; ++i) {
            const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), 0u};
            const Ray<float> r = camera_ray(cf, W, H, i, H - 1 - row, key);
            Hit<float> hit;
            if (!trace_stack_f32(sc, r, 0.0f, FLT_MAX, false, hit, nullptr)) continue;
            const Shape<float>& sh = T.shapes[hit.shape];
            const Material<float>& m = T.materials[sh.material];
            if (m.kind != RT_MAT_PHONGBLINN) continue;
            int lpos = -1;
            for (size_t li = 0; li < T.lights.size(); ++li) {
                const Light<float>& L = T.lights[li];
                if (L.kind == RT_LIGHT_AMBIENT) continue;
                ++lpos;
                const V3<float> ldir = unit(sub(L.pos, hit.p));
                const float inten = saturate(dot(hit.normal, ldir));
                const V3<float> h = unit(add(ldir, r.d));
                const float si = std::pow(saturate(dot(hit.normal, h)), m.param);
                if (inten == 0.0f && (si == 0.0f || (m.spec.x == 0 && m.spec.y == 0 && m.spec.z == 0))) continue;
                for (int k = 0; k < reps; ++k) {
                    const V3<float> jit = scale(0.25f, random_in_unit_sphere<float>(key, 0, P_SHADOW, (uint32_t)(lpos * reps + k)));
                    const V3<float> shifted = add(L.pos, jit);
                    if (L.kind == RT_LIGHT_SPOT && !(dot(L.look_dir, unit(sub(hit.p, shifted))) >= L.cos_angle)) continue;
                    const V3<float> ld = sub(shifted, hit.p);
                    const float d = norm(ld);
                    const Ray<float> sray{hit.p, scale(1.0f / d, ld)};
                    double c0[6] = {0, 0, 0, 0, 0, 0}, c1[6] = {0, 0, 0, 0, 0, 0};
                    const bool oc = walk(sray, 0.01f, d, c0, false);
                    const bool oc1 = walk(sray, 0.01f, d, c1, true);
                    if (oc != oc1) out[16] += 1;
                    double* o = oc ? out : out + 8;
                    o[0] += 1; o[1] += c1[0]; o[2] += c1[1]; o[3] += c1[2]; o[4] += c1[3]; o[5] += c1[4]; o[6] += c0[5]; o[7] += c1[5];
                }
            }
        }
    return 0;
}
