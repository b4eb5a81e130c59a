// =====================================================================================
// rtgo_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A CPU restatement of the btleedev/raytracer-go hot path (camera ray-gen -> octree
// traversal + ray/sphere + ray/triangle -> material shading -> Monte-Carlo shadow rays
// -> pixel accumulate -> RGBA8).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this; the product (raytracer-go_amd/, librtgo_hip.so) never
// does.  Every function cites the reference file:line it restates (paths relative to
// /root/reference).
//
// Pinning: the Go toolchain and gonum are absent here, so the reference cannot run.  The
// oracle is pinned by the reference's own regression artefact: its float64 / best-first
// mode renders ExampleRegression(640,380) and is compared with samples_images/
// code_example.png using the reference's own tolerance (raytracer_test.go:32-80), plus
// the flat-colour known answers of that image (tests/test_oracle_golden.py).  Below image
// level the reference holds no vectors: intersection / material / light unit results are
// "parity unpinned" other than through that image.
//
// Third-party arithmetic restated here (absent from /root/reference): gonum.org/v1/gonum
// v0.9.1 spatial/r3 (go.mod:8): Add/Sub/Scale/Dot/Cross component-wise, Norm =
// hypot(x, hypot(y, z)), Unit(p) = Scale(1/Norm(p), p) with Unit(0) = NaN vector; Go
// container/heap (Push = append+up, Pop = swap(0,n-1)+down+remove last).
//
// Modes
//   0  F64_DIJKSTRA  float64, best-first heap traversal   (accelerationstructures.go:81-124)
//   1  F64_DFS       float64, recursive DFS               (accelerationstructures.go:127-158)
//   2  F32_STACK     float32 arithmetic in the operation order of the HIP kernels, ordered
//                    stack traversal with single-child chains collapsed — used only to
//                    compare the device path bit-for-bit (primitive IDs, t, camera rays).
//
// Randomness: the reference draws from Go's unseeded, mutex-shared global math/rand and
// is not reproducible run to run (raytracer_test.go:35-36).  The oracle (and the device
// path) use counter-based Philox4x32-10 keyed by seed, counter = (pixel, sample,
// bounce*256+purpose, sub*256+try); each uniform is a 24-bit fraction so that float32 and
// float64 modes see the very same samples.
// =====================================================================================
#include "../include/rtgo.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011) and the counter layout shared with the device path
// ------------------------------------------------------------------------------------
struct U4 { uint32_t v[4]; };

inline U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{{c0, c1, c2, c3}};
}

enum Purpose : uint32_t { P_AA = 0, P_LENS = 1, P_FUZZ = 2, P_SCHLICK = 3, P_SHADOW = 4 };
constexpr int kMaxTries = 64;  // rejection loops (camera.go:46-55, material.go:134-143) are capped

struct RngKey {
    uint32_t k0, k1, pixel, sample;
};
inline U4 draw(const RngKey& k, uint32_t bounce, uint32_t purpose, uint32_t sub, uint32_t tr) {
    return philox(k.pixel, k.sample, bounce * 256u + purpose, sub * 256u + tr, k.k0, k.k1);
}
template <class R> inline R u01(uint32_t w) { return (R)(w >> 8) * (R)(1.0 / 16777216.0); }

// splitmix64 for the octree centroid jitter (accelerationstructures.go:267)
struct SplitMix {
    uint64_t s;
    double next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
};
constexpr uint64_t kJitterSeed = 0x6A09E667F3BCC909ull;

// ------------------------------------------------------------------------------------
// r3.Vec restated
// ------------------------------------------------------------------------------------
template <class R> struct V3 { R x, y, z; };
template <class R> inline V3<R> add(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> inline V3<R> sub(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> inline V3<R> scale(R f, V3<R> a) { return {f * a.x, f * a.y, f * a.z}; }
template <class R> inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> inline V3<R> cross(V3<R> p, V3<R> q) {
    return {p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x};
}
inline double norm(V3<double> p) { return std::hypot(p.x, std::hypot(p.y, p.z)); }  // gonum r3.Norm
inline float norm(V3<float> p) { return sqrtf(p.x * p.x + p.y * p.y + p.z * p.z); }  // device form
template <class R> inline V3<R> unit(V3<R> p) {
    if (p.x == 0 && p.y == 0 && p.z == 0) {
        const R n = std::numeric_limits<R>::quiet_NaN();
        return {n, n, n};
    }
    return scale((R)1 / norm(p), p);
}
template <class R> inline V3<R> narrow(const rt_vec3& v) { return {(R)v.x, (R)v.y, (R)v.z}; }

constexpr double kMaxF64 = DBL_MAX;                       // math.MaxFloat64
constexpr double kMinInt64 = -9223372036854775808.0;      // float64(math.MinInt64)

// ------------------------------------------------------------------------------------
// Scene tables (narrowed per arithmetic type)
// ------------------------------------------------------------------------------------
template <class R> struct Shape {
    int kind, material, single_sided;
    V3<R> a, b, c;  // sphere: a = center, b.x = radius
};
template <class R> struct Material {
    int kind, texture;
    V3<R> color, spec;
    R param;
};
template <class R> struct Texture {
    int kind, w, h;
    V3<R> c1, c2;
    R cw, ch;
    const uint8_t* rgba;
};
template <class R> struct Light {
    int kind;
    V3<R> color, pos, look_dir;  // look_dir = Unit(LookAt - Position) (light.go:127)
    R intensity, spec_intensity, angle_deg, cos_angle, decay;
};
template <class R> struct Camera {  // camera.go:9-16
    V3<R> u, v, origin, lower_left, horizontal, vertical;
    R lens_radius;
};

template <class R> struct Tables {
    std::vector<Shape<R>> shapes;
    std::vector<Material<R>> materials;
    std::vector<Texture<R>> textures;
    std::vector<Light<R>> lights;
};

// ------------------------------------------------------------------------------------
// Octree (accelerationstructures.go:11-65, 189-353), always built in float64
// ------------------------------------------------------------------------------------
struct Node {
    int node_id = 0;
    V3<double> pmin{}, pmax{};
    bool leaf = true;
    int shape = -1;  // index into shapes, -1 = nil
    Node* children[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool has_children = false;
    // F32_STACK helpers (filled after the build)
    float fmin[3], fmax[3];
    int count = 0;  // shapes in the subtree
    bool as_leaf = false;  // the device flattener makes ONE leaf of this subtree (mark_leaves)
    Node* slot[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // device block rooted here
};

struct Scene {
    rt_scene_desc desc;  // pointers inside are NOT retained
    std::vector<uint8_t> tex_store;
    Tables<double> t64;
    Tables<float> t32;
    rt_vec3 cam_from, cam_at, cam_up;
    double aperture, fov;
    std::vector<std::unique_ptr<Node>> pool;
    Node* root = nullptr;      // the reference's octree (all fp64 modes; F32_STACK when tree_kind == 0)
    Node* wide_root = nullptr; // F32_STACK, tree_kind == 1: the device's binned-SAH tree in 8-slot node form
    int tree_kind = 1;         // device default (RTGO_TREE): 1 = sah, 0 = octree
    int leaf_size = 4;  // F32_STACK: a subtree with at most this many shapes is one leaf (device flattener rule)
    int n_nodes_alloc = 0;
};

// shape.go:91-93 and :183-207
void shape_bounds(const Shape<double>& s, V3<double>& lo, V3<double>& hi) {
    if (s.kind == RT_SHAPE_SPHERE) {
        const double r = s.b.x;
        lo = sub(s.a, V3<double>{r, r, r});
        hi = add(s.a, V3<double>{r, r, r});
    } else {
        lo = {kMaxF64, kMaxF64, kMaxF64};
        hi = {kMinInt64, kMinInt64, kMinInt64};
        for (const V3<double>* p : {&s.a, &s.b, &s.c}) {
            lo.x = std::fmin(lo.x, p->x); lo.y = std::fmin(lo.y, p->y); lo.z = std::fmin(lo.z, p->z);
            hi.x = std::fmax(hi.x, p->x); hi.y = std::fmax(hi.y, p->y); hi.z = std::fmax(hi.z, p->z);
        }
    }
}
// shape.go:95-97 and :209-211
V3<double> shape_centroid(const Shape<double>& s) {
    if (s.kind == RT_SHAPE_SPHERE) return s.a;
    return scale(1 / 3.0, add(s.a, add(s.b, s.c)));
}

struct Builder {
    Scene& sc;
    SplitMix rng{kJitterSeed};
    int node_counter = 1;  // accelerationstructures.go:56

    Node* new_node(int id, V3<double> lo, V3<double> hi) {
        sc.pool.emplace_back(new Node());
        Node* n = sc.pool.back().get();
        n->node_id = id; n->pmin = lo; n->pmax = hi;
        return n;
    }
    // accelerationstructures.go:266-279
    int quadrant(int shape, const V3<double>& lo, const V3<double>& hi) {
        V3<double> c = shape_centroid(sc.t64.shapes[shape]);
        const double jx = rng.next(), jy = rng.next(), jz = rng.next();
        c = add(c, scale(1e-10, V3<double>{jx, jy, jz}));
        int idx = 0;
        if (c.x > lo.x + (hi.x - lo.x) / 2) idx += 1;
        if (c.y > lo.y + (hi.y - lo.y) / 2) idx += 2;
        if (c.z > lo.z + (hi.z - lo.z) / 2) idx += 4;
        return idx;
    }
    // accelerationstructures.go:282-353
    void split(Node* n) {
        const V3<double> lo = n->pmin, hi = n->pmax;
        const double hx = (hi.x - lo.x) / 2, hy = (hi.y - lo.y) / 2, hz = (hi.z - lo.z) / 2;
        node_counter += 8;
        for (int i = 0; i < 8; ++i) {
            V3<double> clo, chi;
            clo.x = (i & 1) ? lo.x + hx : lo.x;  chi.x = (i & 1) ? hi.x : lo.x + hx;
            clo.y = (i & 2) ? lo.y + hy : lo.y;  chi.y = (i & 2) ? hi.y : lo.y + hy;
            clo.z = (i & 4) ? lo.z + hz : lo.z;  chi.z = (i & 4) ? hi.z : lo.z + hz;
            n->children[i] = new_node(node_counter + 8 - i, clo, chi);
        }
        n->has_children = true;
    }
    // accelerationstructures.go:226-255 (iterative on the descent, recursive on the split)
    void add_shape(Node* cur, int shape) {
        for (int level = 0;; ++level) {
            // the reference never ends on two centroids its jitter cannot separate (the device's octree builder refuses
            // such a scene the same way, csrc/octree.cpp build_octree)
            if (level > 1300) throw std::invalid_argument("centroids that the insertion octree cannot separate");
            if (cur->leaf) {
                if (cur->shape < 0) { cur->shape = shape; return; }
                cur->leaf = false;
                split(cur);
                const int removed = cur->shape;
                cur->shape = -1;
                add_shape(cur, removed);
                // fallthrough: now add `shape` to the same (non-leaf) node
            }
            cur = cur->children[quadrant(shape, cur->pmin, cur->pmax)];
        }
    }
    // accelerationstructures.go:189-211
    void refit(Node* n, V3<double>& lo, V3<double>& hi) {
        lo = {kMaxF64, kMaxF64, kMaxF64};
        hi = {kMinInt64, kMinInt64, kMinInt64};
        if (n->leaf) {
            if (n->shape >= 0) shape_bounds(sc.t64.shapes[n->shape], lo, hi);
        } else {
            for (Node* c : n->children) {
                V3<double> clo, chi;
                refit(c, clo, chi);
                lo.x = std::fmin(lo.x, clo.x); lo.y = std::fmin(lo.y, clo.y); lo.z = std::fmin(lo.z, clo.z);
                hi.x = std::fmax(hi.x, chi.x); hi.y = std::fmax(hi.y, chi.y); hi.z = std::fmax(hi.z, chi.z);
            }
        }
        n->pmin = lo; n->pmax = hi;
    }
    // accelerationstructures.go:213-224
    void prune(Node* n) {
        if (!n->has_children) return;
        for (int i = 0; i < 8; ++i) {
            Node* v = n->children[i];
            if (v->pmin.x == kMaxF64 && v->pmin.y == kMaxF64 && v->pmin.z == kMaxF64 &&
                v->pmax.x == kMinInt64 && v->pmax.y == kMinInt64 && v->pmax.z == kMinInt64) {
                n->children[i] = nullptr;
            } else {
                prune(v);
            }
        }
    }
    // accelerationstructures.go:27-65
    void build() {
        V3<double> lo{kMaxF64, kMaxF64, kMaxF64}, hi{kMinInt64, kMinInt64, kMinInt64};
        for (const auto& s : sc.t64.shapes) {
            V3<double> a, b;
            shape_bounds(s, a, b);
            lo.x = std::fmin(lo.x, a.x); lo.y = std::fmin(lo.y, a.y); lo.z = std::fmin(lo.z, a.z);
            hi.x = std::fmax(hi.x, b.x); hi.y = std::fmax(hi.y, b.y); hi.z = std::fmax(hi.z, b.z);
        }
        lo = sub(lo, scale(1e-10, V3<double>{1, 1, 1}));
        hi = add(hi, scale(1e-10, V3<double>{1, 1, 1}));
        sc.root = new_node(0, lo, hi);
        for (int i = 0; i < (int)sc.t64.shapes.size(); ++i) add_shape(sc.root, i);
        V3<double> a, b;
        refit(sc.root, a, b);
        prune(sc.root);
    }
};

// float32 boxes rounded outward, for the F32_STACK mode (same rule as the device flattener)
inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}
// pad: the device flattener grows every box by 2^-19 of the scene's largest coordinate before it rounds it outward, so that
// the float32 slab test with its pre-rounded o / d is conservative against the shapes' own tests (octree.cpp flatten_octree)
void fill_f32_boxes(Node* n, double pad) {
    n->fmin[0] = round_down(n->pmin.x - pad); n->fmin[1] = round_down(n->pmin.y - pad); n->fmin[2] = round_down(n->pmin.z - pad);
    n->fmax[0] = round_up(n->pmax.x + pad);   n->fmax[1] = round_up(n->pmax.y + pad);   n->fmax[2] = round_up(n->pmax.z + pad);
    n->count = (n->leaf && n->shape >= 0) ? 1 : 0;
    if (n->has_children)
        for (Node* c : n->children) if (c) { fill_f32_boxes(c, pad); n->count += c->count; }
}
// shapes of a subtree, depth first in octant order (the order of a device leaf's shapes)
void subtree_shapes(const Node* n, std::vector<int>* out) {
    if (n->leaf) { if (n->shape >= 0) out->push_back(n->shape); return; }
    for (const Node* c : n->children) if (c) subtree_shapes(c, out);
}
// Follow single-child chains: after the refit a node with one non-nil child has exactly that
// child's box, so the device flattener drops it; the F32_STACK traversal does the same.
inline Node* effective(Node* n) {
    while (!n->leaf) {
        Node* only = nullptr; int cnt = 0;
        for (Node* c : n->children) if (c) { only = c; ++cnt; }
        if (cnt != 1) break;
        n = only;
    }
    return n;
}
// 2^-19 of the largest coordinate a ray of the render can start at: the tree's box (the box of the first node that is not
// a single-child chain link) and the camera's lens (flatten_octree's origin_reach)
inline double box_pad(Node* root, const Scene& sc) {
    const Node* e = effective(root);
    double m = std::fmax(std::fmax(std::fabs(sc.cam_from.x), std::fabs(sc.cam_from.y)), std::fabs(sc.cam_from.z)) + std::fabs(sc.aperture) * 0.5;
    if (!std::isfinite(m)) m = 0.0;
    for (double v : {e->pmin.x, e->pmin.y, e->pmin.z, e->pmax.x, e->pmax.y, e->pmax.z}) m = std::fmax(m, std::fabs(v));
    return std::isfinite(m) ? m * 1.9073486328125e-06 : 0.0;
}
void fill_f32_boxes(Node* n, const Scene& sc) { fill_f32_boxes(n, box_pad(n, sc)); }

// The device flattener's block of an effective inner node x (csrc/octree.cpp block_slots), restated:
// slot o starts as the chain-skipped child of octant o; then, while some inner child (more than
// leaf_size shapes) can be replaced by its own children within 8 slots, the one with the largest box
// area (first in slot order on ties) is, and its children take free slots by greedy matching on
// score = sum over axes of (+/-)(child centre - block centre), + where the slot's octant bit is set
// (largest first; ties to the lower child octant, then the lower slot).
inline double node_area(const Node* n) {
    const double dx = n->pmax.x - n->pmin.x, dy = n->pmax.y - n->pmin.y, dz = n->pmax.z - n->pmin.z;
    return 2.0 * ((dx * dy + dy * dz) + dz * dx);
}
// The device flattener's leaf rule (csrc/octree.cpp is_leaf), restated: at most leaf_size shapes, and the box either
// small (less than 1/64 of the area of the scene's box) or, for at most 6 shapes, tight (at most 1.5 x the area of the
// largest single-shape box in it); a subtree of few but disparate large shapes stays a block of smaller leaves.
double max_shape_area(const Node* n) {
    if (n->leaf) return n->shape >= 0 ? node_area(n) : 0.0;
    double m = 0;
    for (const Node* c : n->children) if (c) m = std::max(m, max_shape_area(c));
    return m;
}
void mark_leaves(Node* n, int leaf_size, double root_area) {
    n->as_leaf = false;
    if (n->count <= leaf_size) {
        if (n->count <= 1 || n->leaf) n->as_leaf = true;
        else {
            const double a = node_area(n);
            n->as_leaf = a * 64.0 < root_area || (n->count <= 6 && a <= 1.5 * max_shape_area(n));  // tight boxes only for leaves of at most 6 shapes
        }
    }
    if (!n->leaf)
        for (Node* c : n->children) if (c) mark_leaves(c, leaf_size, root_area);
}
void build_block(Node* x, bool collapse) {
    int n = 0;
    for (int o = 0; o < 8; ++o) {
        x->slot[o] = x->children[o] ? effective(x->children[o]) : nullptr;
        if (x->slot[o]) ++n;
    }
    const V3<double> m{(x->pmin.x + x->pmax.x) * 0.5, (x->pmin.y + x->pmax.y) * 0.5, (x->pmin.z + x->pmax.z) * 0.5};
    while (collapse) {
        int pick = -1, pick_k = 0;
        double pick_area = 0;
        for (int s = 0; s < 8; ++s) {
            Node* e = x->slot[s];
            if (!e || e->as_leaf) continue;
            int k = 0;
            for (Node* c : e->children) if (c) ++k;
            if (n - 1 + k > 8) continue;
            const double a = node_area(e);
            if (pick < 0 || a > pick_area) { pick = s; pick_area = a; pick_k = k; }
        }
        if (pick < 0) break;
        Node* e = x->slot[pick];
        x->slot[pick] = nullptr;
        std::vector<Node*> kids;
        for (Node* c : e->children) if (c) kids.push_back(effective(c));
        std::vector<char> placed(kids.size(), 0);
        for (size_t round = 0; round < kids.size(); ++round) {
            int bi = -1, bs = -1;
            double bscore = 0;
            for (size_t i = 0; i < kids.size(); ++i) {
                if (placed[i]) continue;
                const Node* c = kids[i];
                const V3<double> ctr{(c->pmin.x + c->pmax.x) * 0.5, (c->pmin.y + c->pmax.y) * 0.5, (c->pmin.z + c->pmax.z) * 0.5};
                for (int sl = 0; sl < 8; ++sl) {
                    if (x->slot[sl]) continue;
                    const double sx = (sl & 1) ? 1.0 : -1.0, sy = (sl & 2) ? 1.0 : -1.0, sz = (sl & 4) ? 1.0 : -1.0;
                    const double score = (sx * (ctr.x - m.x) + sy * (ctr.y - m.y)) + sz * (ctr.z - m.z);
                    if (bi < 0 || score > bscore) { bi = (int)i; bs = sl; bscore = score; }
                }
            }
            x->slot[bs] = kids[(size_t)bi];
            placed[(size_t)bi] = 1;
        }
        n += pick_k - 1;
    }
}
void build_blocks(Node* root, int leaf_size) {
    const bool collapse = std::getenv("RTGO_NO_COLLAPSE") == nullptr;
    mark_leaves(root, leaf_size, node_area(effective(root)));
    std::vector<Node*> todo{effective(root)};
    while (!todo.empty()) {
        Node* x = todo.back();
        todo.pop_back();
        if (x->leaf || x->as_leaf) continue;
        build_block(x, collapse);
        for (Node* e : x->slot) if (e && !e->as_leaf) todo.push_back(e);
    }
}

// The device's default tree (csrc/octree.cpp build_sah), restated.  Spec:
//   split(range): boxes and centroid bounds (centroid = centre of the shape's box) of the range; at most
//     leaf_size shapes -> leaf.  Otherwise, for axis x, y, z with a positive centroid extent: 16 equal
//     bins over the centroid bounds (bin = int((c - lo) * (16 / extent)), clamped to 15); for every
//     boundary k = 0..14 with shapes on both sides, cost = area(left boxes) * n_left + area(right boxes)
//     * n_right; the strictly smallest cost wins (first axis, first boundary on ties).  The range is
//     split stably (order kept on both sides) by bin <= k; with no valid boundary, in the middle.
//   widen(node): children = {left, right}; while fewer than 8 and some child is not a leaf, the inner
//     child with the largest box area (first on ties) is replaced by its left child and its right child
//     is appended.  Children take slots by the same greedy matching as build_block.  A leaf's shapes
//     become single-shape children in slots 0.. in range order.
struct SahShape { V3<double> lo, hi, c; int shape; };
struct SahSplit { V3<double> lo, hi; int left = -1, right = -1, b = 0, e = 0; };
inline double box_area(const V3<double>& lo, const V3<double>& hi) {
    const double dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return 2.0 * ((dx * dy + dy * dz) + dz * dx);
}
inline double axis_of(const V3<double>& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline void grow(V3<double>& lo, V3<double>& hi, const V3<double>& l2, const V3<double>& h2) {
    lo.x = std::fmin(lo.x, l2.x); lo.y = std::fmin(lo.y, l2.y); lo.z = std::fmin(lo.z, l2.z);
    hi.x = std::fmax(hi.x, h2.x); hi.y = std::fmax(hi.y, h2.y); hi.z = std::fmax(hi.z, h2.z);
}
struct SahTree {
    Scene& sc;
    int leaf_size;
    std::vector<SahShape> shp;
    std::vector<SahSplit> nodes;
    static constexpr int kBins = 16;
    static int bin_of(double c, double lo, double scale) {
        const int k = (int)((c - lo) * scale);
        return k >= kBins ? kBins - 1 : k;
    }
    static constexpr int kMaxSahDepth = 48;  // deeper levels halve the range (device builder rule)
    int split(int b, int e, int depth = 0) {
        const V3<double> big{DBL_MAX, DBL_MAX, DBL_MAX}, small{-DBL_MAX, -DBL_MAX, -DBL_MAX};
        SahSplit nd;
        nd.b = b; nd.e = e; nd.lo = big; nd.hi = small;
        V3<double> clo = big, chi = small;
        for (int i = b; i < e; ++i) { grow(nd.lo, nd.hi, shp[i].lo, shp[i].hi); grow(clo, chi, shp[i].c, shp[i].c); }
        const int me = (int)nodes.size();
        nodes.push_back(nd);
        if (e - b <= leaf_size) return me;
        int best_axis = -1, best_k = 0;
        double best_cost = DBL_MAX;
        for (int a = 0; a < 3 && depth < kMaxSahDepth; ++a) {
            const double ext = axis_of(chi, a) - axis_of(clo, a);
            if (!(ext > 0)) continue;
            const double scale = kBins / ext;
            V3<double> blo[kBins], bhi[kBins];
            int bn[kBins];
            for (int k = 0; k < kBins; ++k) { blo[k] = big; bhi[k] = small; bn[k] = 0; }
            for (int i = b; i < e; ++i) {
                const int k = bin_of(axis_of(shp[i].c, a), axis_of(clo, a), scale);
                bn[k]++;
                grow(blo[k], bhi[k], shp[i].lo, shp[i].hi);
            }
            double right_area[kBins]; int right_n[kBins];
            V3<double> lo = big, hi = small;
            int cnt = 0;
            for (int k = kBins - 1; k >= 1; --k) {
                grow(lo, hi, blo[k], bhi[k]);
                cnt += bn[k];
                right_area[k] = cnt ? box_area(lo, hi) : 0;
                right_n[k] = cnt;
            }
            lo = big; hi = small; cnt = 0;
            for (int k = 0; k + 1 < kBins; ++k) {
                grow(lo, hi, blo[k], bhi[k]);
                cnt += bn[k];
                if (cnt == 0 || right_n[k + 1] == 0) continue;
                const double cost = box_area(lo, hi) * cnt + right_area[k + 1] * right_n[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_k = k; }
            }
        }
        int mid = b + (e - b) / 2;
        if (best_axis >= 0) {
            const double lo = axis_of(clo, best_axis), scale = kBins / (axis_of(chi, best_axis) - lo);
            std::vector<SahShape> left, right;
            for (int i = b; i < e; ++i)
                (bin_of(axis_of(shp[i].c, best_axis), lo, scale) <= best_k ? left : right).push_back(shp[i]);
            mid = b + (int)left.size();
            std::copy(left.begin(), left.end(), shp.begin() + b);
            std::copy(right.begin(), right.end(), shp.begin() + mid);
        }
        const int l = split(b, mid, depth + 1);
        const int r = split(mid, e, depth + 1);
        nodes[(size_t)me].left = l; nodes[(size_t)me].right = r;
        return me;
    }
    Node* make(const V3<double>& lo, const V3<double>& hi) {
        sc.pool.emplace_back(new Node());
        Node* n = sc.pool.back().get();
        n->pmin = lo; n->pmax = hi;
        return n;
    }
    Node* widen(int at) {
        const SahSplit nd = nodes[(size_t)at];
        Node* out = make(nd.lo, nd.hi);
        out->leaf = false; out->has_children = true;
        if (nd.left < 0) {
            for (int i = nd.b; i < nd.e; ++i) {
                Node* c = make(shp[i].lo, shp[i].hi);
                c->leaf = true; c->shape = shp[i].shape;
                out->children[i - nd.b] = c;
            }
            return out;
        }
        std::vector<int> kids{nd.left, nd.right};
        while (kids.size() < 8) {
            int pick = -1;
            double pick_area = -1;
            for (size_t i = 0; i < kids.size(); ++i) {
                const SahSplit& k = nodes[(size_t)kids[i]];
                if (k.left < 0) continue;
                const double a = box_area(k.lo, k.hi);
                if (a > pick_area) { pick_area = a; pick = (int)i; }
            }
            if (pick < 0) break;
            const SahSplit k = nodes[(size_t)kids[(size_t)pick]];
            kids[(size_t)pick] = k.left;
            kids.push_back(k.right);
        }
        const V3<double> m{(nd.lo.x + nd.hi.x) * 0.5, (nd.lo.y + nd.hi.y) * 0.5, (nd.lo.z + nd.hi.z) * 0.5};
        int slot_of[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        std::vector<char> placed(kids.size(), 0);
        for (size_t round = 0; round < kids.size(); ++round) {
            int bi = -1, bs = -1;
            double bscore = 0;
            for (size_t i = 0; i < kids.size(); ++i) {
                if (placed[i]) continue;
                const SahSplit& k = nodes[(size_t)kids[i]];
                const V3<double> ctr{(k.lo.x + k.hi.x) * 0.5, (k.lo.y + k.hi.y) * 0.5, (k.lo.z + k.hi.z) * 0.5};
                for (int sl = 0; sl < 8; ++sl) {
                    if (slot_of[sl] >= 0) continue;
                    double score = 0;
                    score += ((sl & 1) ? 1.0 : -1.0) * (ctr.x - m.x);
                    score += ((sl & 2) ? 1.0 : -1.0) * (ctr.y - m.y);
                    score += ((sl & 4) ? 1.0 : -1.0) * (ctr.z - m.z);
                    if (bi < 0 || score > bscore) { bi = (int)i; bs = sl; bscore = score; }
                }
            }
            slot_of[bs] = kids[(size_t)bi];
            placed[(size_t)bi] = 1;
        }
        for (int sl = 0; sl < 8; ++sl)
            if (slot_of[sl] >= 0) out->children[sl] = widen(slot_of[sl]);
        return out;
    }
};
void fill_f32_boxes(Node* n, const Scene& sc);
Node* build_wide_tree(Scene& sc, int leaf_size) {
    const int n = (int)sc.t64.shapes.size();
    if (n == 0) return nullptr;
    SahTree t{sc, leaf_size < 1 ? 1 : (leaf_size > 8 ? 8 : leaf_size), {}, {}};
    t.shp.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        SahShape& s = t.shp[(size_t)i];
        shape_bounds(sc.t64.shapes[(size_t)i], s.lo, s.hi);
        s.c = {(s.lo.x + s.hi.x) * 0.5, (s.lo.y + s.hi.y) * 0.5, (s.lo.z + s.hi.z) * 0.5};
        s.shape = i;
    }
    t.split(0, n);
    Node* root = t.widen(0);
    fill_f32_boxes(root, sc);
    return root;
}

// ------------------------------------------------------------------------------------
// Rays and hits (ray.go:7-14, shape.go:10-16)
// ------------------------------------------------------------------------------------
template <class R> struct Ray { V3<R> p, d; };
template <class R> inline V3<R> point_at(const Ray<R>& r, R t) { return add(r.p, scale(t, r.d)); }
template <class R> struct Hit {
    R t;
    V3<R> p, normal;
    int shape;  // -1 = none
};

// shape.go:47-78.  The "second root" is a copy of the first (:64) so it can never add a hit.
template <class R> inline Hit<R> sphere_hit(const Shape<R>& s, int idx, const Ray<R>& r, R tmin, R tmax) {
    const V3<R> oc = sub(r.p, s.a);
    const R a = dot(r.d, r.d);
    const R b = dot(oc, r.d);
    const R c = dot(oc, oc) - s.b.x * s.b.x;
    const R disc = b * b - a * c;
    if (disc > 0) {
        const R first = (-b - std::sqrt(b * b - a * c)) / a;
        if (first > tmin && first <= tmax) {
            const V3<R> p = point_at(r, first);
            return {first, p, scale((R)1 / s.b.x, sub(point_at(r, first), s.a)), idx};
        }
    }
    return {(R)-1, {}, {}, -1};
}
// shape.go:117-163 (Moller-Trumbore); normal never flipped; |det| < 0 check is dead (:132)
template <class R> inline Hit<R> triangle_hit(const Shape<R>& tr, int idx, const Ray<R>& r, R tmin, R tmax) {
    const V3<R> dir = r.d;
    const V3<R> e1 = sub(tr.b, tr.a);
    const V3<R> e2 = sub(tr.c, tr.a);
    const V3<R> pvec = cross(dir, e2);
    const R det = dot(e1, pvec);
    if (tr.single_sided) { if (det < 0) return {(R)-1, {}, {}, -1}; }
    const R inv = (R)1 / det;
    const V3<R> tvec = sub(r.p, tr.a);
    const R u = dot(tvec, pvec) * inv;
    if (u < 0 || u > 1) return {(R)-1, {}, {}, -1};
    const V3<R> qvec = cross(tvec, e1);
    const R v = dot(dir, qvec) * inv;
    if (v < 0 || u + v > 1) return {(R)-1, {}, {}, -1};
    const R t = dot(e2, qvec) * inv;
    if (t < tmin || t > tmax) return {(R)-1, {}, {}, -1};
    return {t, point_at(r, t), unit(cross(e1, e2)), idx};
}
// Device shadow-ray form (device_math.hpp triangle_occludes): fused, division-free, a yes/no answer.
inline float dotf(const V3<float>& a, const V3<float>& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
inline V3<float> crossf(const V3<float>& a, const V3<float>& b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
inline float flip(float x, uint32_t sb) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u ^= sb;
    std::memcpy(&x, &u, 4);
    return x;
}
inline Hit<float> triangle_occludes_f32(const Shape<float>& tr, int idx, const Ray<float>& r, float tmin, float tmax) {
    const V3<float> e1 = sub(tr.b, tr.a), e2 = sub(tr.c, tr.a);
    const V3<float> pvec = crossf(r.d, e2);
    const float det = dotf(e1, pvec);
    const V3<float> tvec = sub(r.p, tr.a);
    const V3<float> qvec = crossf(tvec, e1);
    uint32_t sb;
    std::memcpy(&sb, &det, 4);
    sb &= 0x80000000u;
    const float ad = flip(det, sb);
    const float us = flip(dotf(tvec, pvec), sb), vs = flip(dotf(r.d, qvec), sb), ts = flip(dotf(e2, qvec), sb);
    // the device never forms t = ts / ad: t > 0 is ts > 0, the interval is compared against ad
    const bool hit = ad > 0.0f && !(tr.single_sided && sb) && us >= 0.0f && vs >= 0.0f && us + vs <= ad &&
                     ts >= tmin * ad && ts <= tmax * ad && ts > 0.0f;
    if (!hit) return {-1.0f, {}, {}, -1};
    return {tmax, {}, {}, idx};  // any accepted value: an occluder's t is not used (light.go:141-160 as any-hit)
}
template <class R> inline Hit<R> shape_hit(const Tables<R>& T, int idx, const Ray<R>& r, R tmin, R tmax) {
    const Shape<R>& s = T.shapes[idx];
    return s.kind == RT_SHAPE_SPHERE ? sphere_hit(s, idx, r, tmin, tmax) : triangle_hit(s, idx, r, tmin, tmax);
}

// accelerationstructures.go:356-417, verbatim semantics (no tFar < 0 reject; -0.0 is "positive")
inline bool hit_bounding_box(const Ray<double>& r, const V3<double>& pmin, const V3<double>& pmax, double& tnear) {
    if (r.p.x >= pmin.x && r.p.y >= pmin.y && r.p.z >= pmin.z &&
        r.p.x <= pmax.x && r.p.y <= pmax.y && r.p.z <= pmax.z) { tnear = 0; return true; }
    const double ix = 1 / r.d.x, iy = 1 / r.d.y, iz = 1 / r.d.z;
    V3<double> b0 = pmin, b1 = pmax;
    if (r.d.x < 0) { b0.x = pmax.x; b1.x = pmin.x; }
    if (r.d.y < 0) { b0.y = pmax.y; b1.y = pmin.y; }
    if (r.d.z < 0) { b0.z = pmax.z; b1.z = pmin.z; }
    double tmin = (b0.x - r.p.x) * ix, tmax = (b1.x - r.p.x) * ix;
    const double tymin = (b0.y - r.p.y) * iy, tymax = (b1.y - r.p.y) * iy;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    const double tzmin = (b0.z - r.p.z) * iz, tzmax = (b1.z - r.p.z) * iz;
    if (tmin > tzmax || tmax < tzmin) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    tnear = tmin;
    return true;
}

struct TraceCounters { uint64_t inner = 0, leaf = 0; };

// bvhpriorityqueue.go + Go container/heap
struct HeapItem { Node* node; double t; };
struct MinHeap {
    std::vector<HeapItem> a;
    static bool less(const HeapItem& x, const HeapItem& y) {  // bvhpriorityqueue.go:22-28
        if (std::fabs(x.t - y.t) < 1e-9) return x.node->node_id < y.node->node_id;
        return x.t < y.t;
    }
    void up(int j) {
        for (;;) {
            const int i = (j - 1) / 2;
            if (i == j || !less(a[j], a[i])) break;
            std::swap(a[i], a[j]);
            j = i;
        }
    }
    void down(int i0, int n) {
        int i = i0;
        for (;;) {
            const int j1 = 2 * i + 1;
            if (j1 >= n || j1 < 0) break;
            int j = j1;
            const int j2 = j1 + 1;
            if (j2 < n && less(a[j2], a[j1])) j = j2;
            if (!less(a[j], a[i])) break;
            std::swap(a[i], a[j]);
            i = j;
        }
    }
    void push(HeapItem it) { a.push_back(it); up((int)a.size() - 1); }
    HeapItem pop() {
        const int n = (int)a.size() - 1;
        std::swap(a[0], a[n]);
        down(0, n);
        HeapItem it = a.back();
        a.pop_back();
        return it;
    }
};

// accelerationstructures.go:81-124
bool trace_dijkstra(const Scene& sc, const Ray<double>& r, double tmin, Hit<double>& out, TraceCounters* tc) {
    thread_local MinHeap heap;
    heap.a.clear();
    heap.a.push_back({sc.root, 0});
    Hit<double> hr{kMaxF64, {}, {}, -1};
    while (!heap.a.empty()) {
        const HeapItem item = heap.pop();
        Node* node = item.node;
        if (item.t > hr.t) break;
        if (node->leaf) {
            if (node->shape >= 0) {
                if (tc) tc->leaf++;
                const Hit<double> sh = shape_hit(sc.t64, node->shape, r, tmin, hr.t);
                if (sh.t > 0.0 && sh.t < hr.t) hr = sh;
            }
        } else if (node->has_children) {
            if (tc) tc->inner++;
            for (Node* v : node->children) {
                if (!v) continue;
                double tnear;
                if (hit_bounding_box(r, v->pmin, v->pmax, tnear)) heap.push({v, tnear});
            }
        }
    }
    out = hr;
    return hr.t != kMaxF64;
}

// accelerationstructures.go:127-158.  Returns (hit, record); record may be "nil" (valid=false).
struct DfsResult { bool hit; bool valid; Hit<double> rec; };
DfsResult trace_dfs_node(const Scene& sc, const Ray<double>& r, double tmin, double tmax, Node* node, TraceCounters* tc) {
    double tn;
    if (!hit_bounding_box(r, node->pmin, node->pmax, tn)) return {false, true, {-1, {}, {}, -1}};
    if (node->leaf) {
        if (node->shape < 0) return {false, false, {}};
        if (tc) tc->leaf++;
        const Hit<double> hr = shape_hit(sc.t64, node->shape, r, tmin, tmax);
        return {hr.t > 0.0, true, hr};
    }
    if (tc) tc->inner++;
    Hit<double> best{tmax, {}, {}, -1};
    if (node->has_children) {
        for (Node* v : node->children) {
            if (!v) continue;
            const DfsResult rr = trace_dfs_node(sc, r, tmin, tmax, v, tc);
            if (rr.hit && rr.rec.t > tmin && rr.rec.t < best.t) best = rr.rec;
        }
    }
    return {best.t != kMaxF64, true, best};
}
bool trace_dfs(const Scene& sc, const Ray<double>& r, double tmin, Hit<double>& out, TraceCounters* tc) {
    const DfsResult rr = trace_dfs_node(sc, r, tmin, kMaxF64, sc.root, tc);
    if (rr.hit) out = rr.rec; else out = {kMaxF64, {}, {}, -1};
    return rr.hit;
}

// ------------------------------------------------------------------------------------
// F32_STACK: the device traversal restated on the pointer tree.
//   - child boxes tested with the FMA slab form t = fma(b, 1/d, -(o * 1/d)), IEEE
//     minNum/maxNum folding, clipped to [0 (or tMin), best];
//   - per node: every present child's box is tested against the best t at node entry, in ray
//     order k = 0..7, octant o = k ^ signmask; then the hit leaf children are intersected in
//     that order; then hit inner children are pushed far-to-near and the nearest is entered;
//   - nearest hit accepted with the reference's rule t > 0 && t < best (strict);
//   - any_hit: stop at the first accepted t <= tmax.
// ------------------------------------------------------------------------------------
struct StackRay {
    Ray<float> r;
    float inv[3], oi[3];
    int sign;
};
inline StackRay make_stack_ray(const Ray<float>& r) {
    StackRay s;
    s.r = r;
    s.inv[0] = 1.0f / r.d.x; s.inv[1] = 1.0f / r.d.y; s.inv[2] = 1.0f / r.d.z;
    s.oi[0] = r.p.x * s.inv[0]; s.oi[1] = r.p.y * s.inv[1]; s.oi[2] = r.p.z * s.inv[2];
    s.sign = (r.d.x < 0 ? 1 : 0) | (r.d.y < 0 ? 2 : 0) | (r.d.z < 0 ? 4 : 0);
    return s;
}
inline bool slab_f32(const StackRay& s, const float* bmin, const float* bmax, float t0, float t1) {
    float tn = t0, tf = t1;
    for (int a = 0; a < 3; ++a) {
        const float lo = fmaf(bmin[a], s.inv[a], -s.oi[a]);
        const float hi = fmaf(bmax[a], s.inv[a], -s.oi[a]);
        tn = fmaxf(tn, fminf(lo, hi));
        tf = fminf(tf, fmaxf(lo, hi));
    }
    return tn <= tf;
}
bool trace_stack_f32(const Scene& sc, const Ray<float>& r, float tmin, float tmax, bool any_hit,
                     Hit<float>& out, TraceCounters* tc) {
    const StackRay sr = make_stack_ray(r);
    Hit<float> hr{any_hit ? tmax : FLT_MAX, {}, {}, -1};
    bool found = false;
    std::vector<Node*> stack;
    Node* root = effective(sc.tree_kind == 1 && sc.wide_root ? sc.wide_root : sc.root);
    Node fake;  // a root that is itself one leaf gets a synthetic one-child parent, as in the device flattener
    if (root->as_leaf || root->count == 0) {
        if (root->count == 0) { out = hr; return false; }
        fake.leaf = false; fake.has_children = true; fake.slot[0] = root;
        root = &fake;
    }
    std::vector<int> group;
    Node* cur = root;
    const float t0 = any_hit ? tmin : 0.0f;  // device: shadow rays clip the slab test to [tmin, tmax]
    for (;;) {
        if (tc) tc->inner++;
        Node* pending[8]; int np = 0;
        Node* leaves[8]; int nl = 0;
        // pass 1: every present child's box against the best t at node entry, in ray order
        const float entry_best = hr.t;
        for (int k = 0; k < 8; ++k) {
            Node* e = cur->slot[k ^ sr.sign];  // chain-skipped already: same box as the un-collapsed child after the refit
            if (!e) continue;
            if (!slab_f32(sr, e->fmin, e->fmax, t0, entry_best)) continue;
            if (e->as_leaf) leaves[nl++] = e; else pending[np++] = e;
        }
        // pass 2: hit leaf children in ray order; a leaf's shapes in their stored (depth-first) order
        for (int q = 0; q < nl; ++q) {
            group.clear();
            subtree_shapes(leaves[q], &group);
            for (int shape : group) {
                if (tc) tc->leaf++;
                const Hit<float> sh = (any_hit && sc.t32.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE)
                                          ? triangle_occludes_f32(sc.t32.shapes[(size_t)shape], shape, r, tmin, hr.t)
                                          : shape_hit(sc.t32, shape, r, tmin, hr.t);
                if (any_hit) {
                    if (sh.t > 0.0f && sh.t <= hr.t) { out = sh; return true; }
                } else if (sh.t > 0.0f && sh.t < hr.t) { hr = sh; found = true; }
            }
        }
        for (int i = np - 1; i >= 1; --i) stack.push_back(pending[i]);
        if (np > 0) { cur = pending[0]; continue; }
        if (stack.empty()) break;
        cur = stack.back();
        stack.pop_back();
    }
    out = hr;
    return found;
}

// ------------------------------------------------------------------------------------
// Mode-dispatched trace:  traceFunction(r, tMin)  (accelerationstructures.go:67-75)
// ------------------------------------------------------------------------------------
enum Mode { F64_DIJKSTRA = 0, F64_DFS = 1, F32_STACK = 2 };

template <class R> struct Tracer;
template <> struct Tracer<double> {
    const Scene& sc; int mode; mutable TraceCounters* tc;
    const Tables<double>& T() const { return sc.t64; }
    bool trace(const Ray<double>& r, double tmin, Hit<double>& h) const {
        return mode == F64_DFS ? trace_dfs(sc, r, tmin, h, tc) : trace_dijkstra(sc, r, tmin, h, tc);
    }
    // light.go:141-160
    bool reaches_light(const V3<double>& origin, const V3<double>& lightpos) const {
        const V3<double> ld = sub(lightpos, origin);
        const Ray<double> r{origin, unit(ld)};
        Hit<double> h;
        if (!trace(r, 0.01, h)) return true;
        const double to_light = dot(ld, ld);
        const V3<double> hp = sub(h.p, origin);
        return to_light < dot(hp, hp);
    }
    // light.go:122-131 (spot) — acos form
    bool in_cone(const Light<double>& L, const V3<double>& point, const V3<double>& shifted) const {
        const V3<double> a = L.look_dir;
        const V3<double> b = unit(sub(point, shifted));
        const double la = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
        const double lb = std::sqrt(b.x * b.x + b.y * b.y + b.z * b.z);
        const double ang = std::acos(dot(a, b) / (la * lb)) * 180 / M_PI;  // light.go:134-139
        return ang <= L.angle_deg;
    }
};
template <> struct Tracer<float> {
    const Scene& sc; int mode; mutable TraceCounters* tc;
    const Tables<float>& T() const { return sc.t32; }
    bool trace(const Ray<float>& r, float tmin, Hit<float>& h) const {
        return trace_stack_f32(sc, r, tmin, FLT_MAX, false, h, tc);
    }
    // device form of light.go:141-160: any-hit with tMax = |L' - p|
    bool reaches_light(const V3<float>& origin, const V3<float>& lightpos) const {
        const V3<float> ld = sub(lightpos, origin);
        const float dist = norm(ld);
        const Ray<float> r{origin, scale(1.0f / dist, ld)};
        Hit<float> h;
        return !trace_stack_f32(sc, r, 0.01f, dist, true, h, tc);
    }
    // device form of light.go:122-131: cos(angle) >= cos(Angle)
    bool in_cone(const Light<float>& L, const V3<float>& point, const V3<float>& shifted) const {
        const V3<float> b = unit(sub(point, shifted));
        return dot(L.look_dir, b) >= L.cos_angle;
    }
};

// ------------------------------------------------------------------------------------
// Textures (texture.go:27-53) and UV maps (shape.go:100-105, 213-227)
// ------------------------------------------------------------------------------------
template <class R> inline long floor_to_int(R x) {
    if (!(x == x)) return 0;
    const R f = std::floor(x);
    if (f >= (R)9.0e18) return LONG_MAX;
    if (f <= (R)-9.0e18) return LONG_MIN;
    return (long)f;
}
template <class R> V3<R> texture_color(const Texture<R>& t, R u, R v) {
    if (t.kind == RT_TEX_CHECKERS) {  // texture.go:27-36
        const long u2 = floor_to_int(u * t.cw), v2 = floor_to_int(v * t.ch);
        return ((u2 + v2) % 2 == 0) ? t.c1 : t.c2;
    }
    // texture.go:38-53: nearest texel, no clamp/wrap; out of bounds = zero colour = background
    const long u2 = floor_to_int(u * (R)t.w), v2 = floor_to_int(v * (R)t.h);
    uint32_t r = 0, g = 0, b = 0, a = 0;
    if (u2 >= 0 && u2 < t.w && v2 >= 0 && v2 < t.h) {
        const uint8_t* px = t.rgba + ((size_t)v2 * t.w + u2) * 4;
        r = px[0] * 257u; g = px[1] * 257u; b = px[2] * 257u; a = px[3] * 257u;  // color.RGBA.RGBA()
    }
    const R r256 = (R)r / (R)255.99, g256 = (R)g / (R)255.99, b256 = (R)b / (R)255.99;
    const R a1 = std::min((R)1.0, (R)a / ((R)255.99 * (R)255.99));
    return {((R)1.0 - a1) * (R)0.0 + (a1 * r256) / (R)255.99,
            ((R)1.0 - a1) * (R)0.0 + (a1 * g256) / (R)255.99,
            ((R)1.0 - a1) * (R)0.0 + (a1 * b256) / (R)255.99};
}
template <class R> void texture_map(const Shape<R>& s, const V3<R>& point, R& u, R& v) {
    if (s.kind == RT_SHAPE_SPHERE) {  // shape.go:100-105
        const V3<R> q = sub(point, s.a);
        const R theta = std::atan2((R)-1 * q.z, q.x);
        const R phi = std::acos((R)-1 * q.y / s.b.x);
        u = (theta + (R)M_PI) / ((R)2 * (R)M_PI);
        v = phi / (R)M_PI;
    } else {  // shape.go:213-227; named result v is still 0 when u is evaluated => u = 1 - w
        const V3<R> v0 = sub(s.b, s.a), v1 = sub(s.c, s.a), v2 = sub(point, s.a);
        const R d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
        const R denom = d00 * d11 - d01 * d01;
        const R w = (d00 * d21 - d01 * d20) / denom;
        u = (R)1.0 - (R)0 - w;
        v = (d11 * d20 - d01 * d21) / denom;
    }
}

// ------------------------------------------------------------------------------------
// Materials (material.go:34-172)
// ------------------------------------------------------------------------------------
// material.go:134-143.  Randomness spec: one Philox block per call; try 0 uses its first
// three words, every further try three words of the xorshift128 stream seeded with the block's four.
inline uint32_t xorshift128(U4& s) {
    const uint32_t t = s.v[0] ^ (s.v[0] << 11);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3];
    s.v[3] = s.v[3] ^ (s.v[3] >> 19) ^ t ^ (t >> 8);
    return s.v[3];
}
template <class R> inline V3<R> random_in_unit_sphere(const RngKey& k, uint32_t bounce, uint32_t purpose, uint32_t sub_) {
    U4 s = draw(k, bounce, purpose, sub_, 0);
    uint32_t w0 = s.v[0], w1 = s.v[1], w2 = s.v[2];
    for (int tr = 0; tr < kMaxTries; ++tr) {
        const V3<R> p = sub(scale((R)2, V3<R>{u01<R>(w0), u01<R>(w1), u01<R>(w2)}), V3<R>{1, 1, 1});
        if (p.x * p.x + p.y * p.y + p.z * p.z < (R)1.0) return p;
        w0 = xorshift128(s); w1 = xorshift128(s); w2 = xorshift128(s);
    }
    return {0, 0, 0};
}
template <class R> inline V3<R> reflected(const V3<R>& v, const V3<R>& n) {  // material.go:145-147
    return unit(sub(v, scale((R)2 * dot(v, n), n)));
}
template <class R> inline V3<R> refracted(const V3<R>& v, const V3<R>& n, R eta) {  // material.go:149-155
    const V3<R> uv = unit(v);
    const R cos_theta = std::min(dot(scale((R)-1, uv), n), (R)1.0);
    const V3<R> perp = scale(eta, add(uv, scale(cos_theta, n)));
    // precedence quirk kept: |1 - x^2 + y^2 + z^2|
    const V3<R> par = scale(-std::sqrt(std::fabs((R)1.0 - perp.x * perp.x + perp.y * perp.y + perp.z * perp.z)), n);
    return unit(add(perp, par));
}
template <class R> inline R schlick(R cosine, R ri) {  // material.go:157-161
    R r0 = ((R)1 - ri) / ((R)1 + ri);
    r0 = r0 * r0;
    return r0 + ((R)1 - r0) * std::pow((R)1 - cosine, (R)5);
}
template <class R> inline R saturate(R i) { return i > 1 ? (R)1 : (i < 0 ? (R)0 : i); }  // material.go:164-172

struct RayCounts {
    uint64_t primary = 0, continuation = 0, shadow = 0, phong_hits = 0;
    TraceCounters tc_primary, tc_continuation, tc_shadow;
};

template <class R> struct Scatter { bool should_trace; V3<R> attenuation; Ray<R> scattered; V3<R> color; };

// PhongBlinn.scatter, material.go:73-132
template <class R>
V3<R> phong_blinn(const Tracer<R>& trc, const Material<R>& m, const Ray<R>& r, const Hit<R>& hit, int reps,
                  const RngKey& key, uint32_t bounce, RayCounts* rc, TraceCounters* shadow_tc) {
    const Tables<R>& T = trc.T();
    V3<R> c{0, 0, 0};
    int lpos = 0;
    if (rc) rc->phong_hits++;
    for (const Light<R>& L : T.lights) {
        if (L.kind != RT_LIGHT_AMBIENT) {
            // contribution of one visible repetition (material.go:83-119); the light position used
            // here is NOT jittered, so it is the same for every repetition.
            const V3<R> l2p = sub(L.pos, hit.p);
            const V3<R> ldir = unit(l2p);
            const R d2 = l2p.x * l2p.x + l2p.y * l2p.y + l2p.z * l2p.z;
            R decay = L.decay * d2;
            if (decay <= 1) decay = 1;
            V3<R> mc = m.color;
            if (m.texture >= 0) {
                R u, v;
                texture_map(T.shapes[hit.shape], hit.p, u, v);
                mc = texture_color(T.textures[m.texture], u, v);
            }
            const R ndl = dot(hit.normal, ldir);
            const R inten = saturate(ndl);
            const V3<R> diff = scale(inten * L.intensity / decay, V3<R>{mc.x * L.color.x, mc.y * L.color.y, mc.z * L.color.z});
            const V3<R> h = unit(add(ldir, r.d));  // material.go:107: ray direction, not view direction
            const R ndh = dot(hit.normal, h);
            const R si = std::pow(saturate(ndh), m.param);
            const V3<R> spec = scale(si * L.spec_intensity / decay, V3<R>{m.spec.x * L.color.x, m.spec.y * L.color.y, m.spec.z * L.color.z});
            const V3<R> comb{std::min((R)1.0, diff.x + spec.x), std::min((R)1.0, diff.y + spec.y), std::min((R)1.0, diff.z + spec.z)};
            int visible = 0;
            // device rule (k_phong_pairs): a term that is exactly zero adds nothing whatever its shadow rays find,
            // so the fp32 mirror generates none for it; the reference-faithful fp64 modes trace them all
            const bool skip = !std::is_same<R, double>::value && comb.x == 0 && comb.y == 0 && comb.z == 0;
            for (int i = 0; i < reps && !skip; ++i) {
                const V3<R> jit = scale((R)0.25, random_in_unit_sphere<R>(key, bounce, P_SHADOW, (uint32_t)(lpos * reps + i)));
                const V3<R> shifted = add(L.pos, jit);
                bool vis;
                if (std::is_same<R, double>::value) {
                    // light.go:93-96 / 122-131: the reference traces first, then tests the cone
                    TraceCounters* save = trc.tc;
                    trc.tc = shadow_tc;
                    vis = trc.reaches_light(hit.p, shifted);
                    trc.tc = save;
                    if (rc) rc->shadow++;
                    if (L.kind == RT_LIGHT_SPOT) vis = trc.in_cone(L, hit.p, shifted) && vis;
                } else {
                    // device order: cone first, trace only inside the cone
                    vis = (L.kind != RT_LIGHT_SPOT) || trc.in_cone(L, hit.p, shifted);
                    if (vis) {
                        TraceCounters* save = trc.tc;
                        trc.tc = shadow_tc;
                        vis = trc.reaches_light(hit.p, shifted);
                        trc.tc = save;
                        if (rc) rc->shadow++;
                    }
                }
                if (vis) {
                    ++visible;
                    if (std::is_same<R, double>::value) c = add(c, scale((R)1 / (R)reps, comb));  // material.go:120
                }
            }
            if (!std::is_same<R, double>::value && visible > 0) {
                const R w = (R)visible / (R)reps;  // device form: one weighted add per light
                c = add(c, scale(w, comb));
            }
            ++lpos;
        } else {
            c = add(c, scale(L.intensity, L.color));  // material.go:125
        }
    }
    c.x = std::min((R)1.0, c.x); c.y = std::min((R)1.0, c.y); c.z = std::min((R)1.0, c.z);
    return c;
}

template <class R>
Scatter<R> scatter(const Tracer<R>& trc, const Ray<R>& r, const Hit<R>& hit, int reps, const RngKey& key,
                   uint32_t bounce, RayCounts* rc, TraceCounters* shadow_tc) {
    const Tables<R>& T = trc.T();
    const Shape<R>& sh = T.shapes[hit.shape];
    const Material<R>& m = T.materials[sh.material];
    switch (m.kind) {
    case RT_MAT_STANDARD: {  // material.go:34-41
        V3<R> col = m.color;
        if (m.texture >= 0) {
            R u, v;
            texture_map(sh, hit.p, u, v);
            col = texture_color(T.textures[m.texture], u, v);
        }
        return {false, {}, {hit.p, {}}, col};
    }
    case RT_MAT_METAL: {  // material.go:43-50
        const R fuzz = m.param < 1.0 ? m.param : (R)1.0;
        const V3<R> refl = reflected(r.d, hit.normal);
        const V3<R> rnd = random_in_unit_sphere<R>(key, bounce, P_FUZZ, 0);
        return {dot(refl, hit.normal) > 0, m.color, {hit.p, add(refl, scale(fuzz, rnd))}, {}};
    }
    case RT_MAT_DIELECTRIC: {  // material.go:52-70
        const R ratio = dot(r.d, hit.normal) > 0 ? m.param : (R)1.0 / m.param;
        const R cos_theta = std::min(dot(scale((R)-1, r.d), hit.normal), (R)1.0);
        const R sin_theta = std::sqrt((R)1 - cos_theta * cos_theta);
        const bool cannot = ratio * sin_theta > (R)1.0;
        V3<R> dir;
        if (cannot || schlick(cos_theta, ratio) > u01<R>(draw(key, bounce, P_SCHLICK, 0, 0).v[0]))
            dir = reflected(r.d, hit.normal);
        else
            dir = refracted(r.d, hit.normal, ratio);
        return {true, {1, 1, 1}, {add(hit.p, scale((R)0.00001, dir)), dir}, {}};
    }
    default:  // RT_MAT_PHONGBLINN
        return {false, {}, {}, phong_blinn(trc, m, r, hit, reps, key, bounce, rc, shadow_tc)};
    }
}

// ------------------------------------------------------------------------------------
// Camera (camera.go:18-55) and the pixel loop (raytracer.go:58-162)
// ------------------------------------------------------------------------------------
struct CameraF64 { V3<double> w, u, v, origin, ll, hor, ver; double lens; };
CameraF64 new_camera(const Scene& sc, int W, int H) {
    const V3<double> from = narrow<double>(sc.cam_from), at = narrow<double>(sc.cam_at), up = narrow<double>(sc.cam_up);
    const V3<double> d = sub(from, at);
    const double focus = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);  // raytracer.go:67
    const double aspect = (double)W / (double)H;                         // raytracer.go:65
    const double theta = sc.fov * M_PI / 180.0;
    const double hh = std::tan(theta / 2), hw = aspect * hh;
    CameraF64 c;
    c.w = unit(sub(from, at));
    c.u = unit(cross(up, c.w));
    c.v = cross(c.w, c.u);
    c.origin = from;
    c.ll = sub(sub(sub(from, scale(hw * focus, c.u)), scale(hh * focus, c.v)), scale(focus, c.w));
    c.hor = scale(2 * hw * focus, c.u);
    c.ver = scale(2 * hh * focus, c.v);
    c.lens = sc.aperture / 2;
    return c;
}
template <class R> Camera<R> narrow_camera(const CameraF64& c) {
    auto n = [](const V3<double>& v) { return V3<R>{(R)v.x, (R)v.y, (R)v.z}; };
    return {n(c.u), n(c.v), n(c.origin), n(c.ll), n(c.hor), n(c.ver), (R)c.lens};
}
// raytracer.go:114-116 + camera.go:37-55.  (i, j): j counted from the bottom.
template <class R> Ray<R> camera_ray(const Camera<R>& cam, int W, int H, int i, int j, const RngKey& key) {
    const U4 aa = draw(key, 0, P_AA, 0, 0);
    const R u = ((R)i + u01<R>(aa.v[0])) / (R)W;
    const R v = ((R)j + u01<R>(aa.v[1])) / (R)H;
    V3<R> disk{0, 0, 0};
    for (int tr = 0; tr < kMaxTries; ++tr) {
        const U4 w = draw(key, 0, P_LENS, 0, tr);
        const V3<R> p = sub(scale((R)2, V3<R>{u01<R>(w.v[0]), u01<R>(w.v[1]), 0}), V3<R>{1, 1, 0});
        if (dot(p, p) < (R)1.0) { disk = p; break; }
    }
    const V3<R> rd = scale(cam.lens_radius, disk);
    const V3<R> offset = add(scale(rd.x, cam.u), scale(rd.y, cam.v));
    return {add(cam.origin, offset),
            unit(sub(sub(add(add(cam.lower_left, scale(u, cam.horizontal)), scale(v, cam.vertical)), cam.origin), offset))};
}

// raytracer.go:135-162, iteratively: pixel += prod(attenuation) * terminal
template <class R>
V3<R> color(const Tracer<R>& trc_in, Ray<R> r, int max_depth, int reps, const RngKey& key, RayCounts* rc) {
    V3<R> thr{1, 1, 1};
    Tracer<R> trc = trc_in;
    for (int depth = 0;; ++depth) {
        Hit<R> hit;
        // The reference still traces a ray at depth == max_depth and then discards the hit
        // (raytracer.go:143-145); the device path does not launch that ray, and neither does F32_STACK.
        if (!std::is_same<R, double>::value && depth >= max_depth) return {0, 0, 0};
        if (rc) { if (depth == 0) rc->primary++; else rc->continuation++; }
        trc.tc = rc ? (depth == 0 ? &rc->tc_primary : &rc->tc_continuation) : nullptr;
        const bool got = trc.trace(r, (R)0.0, hit);
        if (!(got && depth < max_depth)) return {0, 0, 0};  // background consts, raytracer.go:14-16
        const Scatter<R> s = scatter(trc, r, hit, reps, key, (uint32_t)depth, rc, rc ? &rc->tc_shadow : nullptr);
        if (!s.should_trace) return {thr.x * s.color.x, thr.y * s.color.y, thr.z * s.color.z};
        thr = {thr.x * s.attenuation.x, thr.y * s.attenuation.y, thr.z * s.attenuation.z};
        r = s.scattered;
    }
}

template <class R> void narrow_tables(const Tables<double>& a, Tables<R>& b) {
    auto n = [](const V3<double>& v) { return V3<R>{(R)v.x, (R)v.y, (R)v.z}; };
    for (const auto& s : a.shapes) b.shapes.push_back({s.kind, s.material, s.single_sided, n(s.a), n(s.b), n(s.c)});
    for (const auto& m : a.materials) b.materials.push_back({m.kind, m.texture, n(m.color), n(m.spec), (R)m.param});
    for (const auto& t : a.textures) b.textures.push_back({t.kind, t.w, t.h, n(t.c1), n(t.c2), (R)t.cw, (R)t.ch, t.rgba});
    for (const auto& l : a.lights)
        b.lights.push_back({l.kind, n(l.color), n(l.pos), n(l.look_dir), (R)l.intensity, (R)l.spec_intensity,
                            (R)l.angle_deg, (R)l.cos_angle, (R)l.decay});
}

thread_local std::string g_err;

}  // namespace

// =====================================================================================
// extern "C" surface used by tests / bench (ctypes)
// =====================================================================================
extern "C" {

struct orc_scene { Scene sc; };

struct orc_counters {
    uint64_t rays_primary, rays_continuation, rays_shadow, phong_hits;
    uint64_t inner_primary, leaf_primary, inner_continuation, leaf_continuation, inner_shadow, leaf_shadow;
};

const char* orc_last_error(void) { return g_err.c_str(); }

orc_scene* orc_scene_create(const rt_scene_desc* d) {
    if (!d || d->n_shapes < 0) { g_err = "bad desc"; return nullptr; }
    auto* o = new orc_scene();
    Scene& sc = o->sc;
    sc.cam_from = d->camera_look_from; sc.cam_at = d->camera_look_at; sc.cam_up = d->camera_up;
    sc.aperture = d->camera_aperture; sc.fov = d->camera_fov_deg;
    size_t tex_bytes = 0;
    for (int64_t i = 0; i < d->n_textures; ++i)
        if (d->textures[i].kind == RT_TEX_IMAGE) tex_bytes += (size_t)d->textures[i].width * d->textures[i].height * 4;
    sc.tex_store.resize(tex_bytes);
    size_t off = 0;
    for (int64_t i = 0; i < d->n_textures; ++i) {
        const rt_texture& t = d->textures[i];
        const uint8_t* px = nullptr;
        if (t.kind == RT_TEX_IMAGE) {
            const size_t n = (size_t)t.width * t.height * 4;
            std::memcpy(sc.tex_store.data() + off, t.rgba, n);
            px = sc.tex_store.data() + off;
            off += n;
        }
        sc.t64.textures.push_back({t.kind, t.width, t.height, narrow<double>(t.color1), narrow<double>(t.color2),
                                   t.checkers_width, t.checkers_height, px});
    }
    for (int64_t i = 0; i < d->n_materials; ++i) {
        const rt_material& m = d->materials[i];
        sc.t64.materials.push_back({m.kind, m.texture, narrow<double>(m.color), narrow<double>(m.specular_color), m.param});
    }
    for (int64_t i = 0; i < d->n_shapes; ++i) {
        const rt_shape& s = d->shapes[i];
        sc.t64.shapes.push_back({s.kind, s.material, s.single_sided, narrow<double>(s.a), narrow<double>(s.b), narrow<double>(s.c)});
    }
    for (int64_t i = 0; i < d->n_lights; ++i) {
        const rt_light& l = d->lights[i];
        Light<double> L{};
        L.kind = l.kind; L.color = narrow<double>(l.color); L.pos = narrow<double>(l.position);
        L.look_dir = l.kind == RT_LIGHT_SPOT ? unit(sub(narrow<double>(l.look_at), narrow<double>(l.position))) : V3<double>{0, 0, 0};
        L.intensity = l.intensity; L.spec_intensity = l.specular_intensity;
        L.angle_deg = l.angle_deg; L.cos_angle = std::cos(l.angle_deg * M_PI / 180.0); L.decay = l.decay;
        sc.t64.lights.push_back(L);
    }
    narrow_tables(sc.t64, sc.t32);
    Builder b{sc};
    try {
        b.build();
    } catch (const std::exception& e) {
        g_err = e.what();
        delete o;
        return nullptr;
    }
    fill_f32_boxes(sc.root, sc);
    build_blocks(sc.root, sc.leaf_size);
    sc.wide_root = build_wide_tree(sc, sc.leaf_size);
    if (sc.wide_root) build_blocks(sc.wide_root, sc.leaf_size);
    return o;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
// F32_STACK only: shapes per leaf of the device tree being mirrored (RTGO_LEAF_SIZE, default 4)
void orc_set_leaf_size(orc_scene* s, int leaf_size) {
    s->sc.leaf_size = leaf_size < 1 ? 1 : leaf_size;
    build_blocks(s->sc.root, s->sc.leaf_size);
    s->sc.wide_root = build_wide_tree(s->sc, s->sc.leaf_size);  // the SAH leaves depend on leaf_size
    if (s->sc.wide_root) build_blocks(s->sc.wide_root, s->sc.leaf_size);
}
// F32_STACK only: which device tree to mirror (RTGO_TREE): 1 = binned SAH (default), 0 = the reference's octree
void orc_set_tree(orc_scene* s, int kind) { s->sc.tree_kind = kind; }

// out[0] = pointer-tree node count, out[1] = inner nodes (non-leaf, after prune), out[2] = max depth
void orc_tree_info(const orc_scene* s, int64_t* out) {
    int64_t inner = 0, leaves = 0; int maxd = 0;
    struct It { Node* n; int d; };
    std::vector<It> st{{s->sc.root, 0}};
    while (!st.empty()) {
        It it = st.back(); st.pop_back();
        maxd = std::max(maxd, it.d);
        if (it.n->leaf) { if (it.n->shape >= 0) ++leaves; continue; }
        ++inner;
        for (Node* c : it.n->children) if (c) st.push_back({c, it.d + 1});
    }
    out[0] = (int64_t)s->sc.pool.size(); out[1] = inner; out[2] = maxd; out[3] = leaves;
}

// rays: n * 8 doubles {o, d, tMin, tMax}.  prim_out: shape or -1 (any_hit: 1 occluded / 0).
int orc_trace_rays(const orc_scene* s, int mode, const double* rays, int64_t n, int any_hit,
                   int32_t* prim_out, double* t_out, uint32_t* inner_out, uint32_t* leaf_out, int nthreads) {
    // mode 3 (F32_ALL_SHAPES): no tree at all — the float32 shape tests of F32_STACK over EVERY shape, in index order.  What a
    // walk whose box tests are conservative against the shapes' own tests must find: the minimum t (any_hit: the OR).
    if (mode < 0 || mode > 3) { g_err = "bad mode"; return -1; }
    if (any_hit && mode != F32_STACK && mode != 3) { g_err = "any_hit is an F32_STACK probe"; return -1; }
    nthreads = std::max(1, nthreads);
    std::vector<std::thread> th;
    for (int w = 0; w < nthreads; ++w)
        th.emplace_back([=]() {
            for (int64_t i = w; i < n; i += nthreads) {
                const double* q = rays + i * 8;
                TraceCounters tc;
                int prim = -1; double t = 0;
                if (mode == 3) {
                    const Ray<float> r{{(float)q[0], (float)q[1], (float)q[2]}, {(float)q[3], (float)q[4], (float)q[5]}};
                    const float tmin = (float)q[6];
                    Hit<float> best{any_hit ? (float)q[7] : FLT_MAX, {}, {}, -1};
                    bool got = false;
                    for (int shape = 0; shape < (int)s->sc.t32.shapes.size() && !(any_hit && got); ++shape) {
                        tc.leaf++;
                        const Hit<float> sh = (any_hit && s->sc.t32.shapes[(size_t)shape].kind == RT_SHAPE_TRIANGLE)
                                                  ? triangle_occludes_f32(s->sc.t32.shapes[(size_t)shape], shape, r, tmin, best.t)
                                                  : shape_hit(s->sc.t32, shape, r, tmin, best.t);
                        if (any_hit) { if (sh.t > 0.0f && sh.t <= best.t) got = true; }
                        else if (sh.t > 0.0f && sh.t < best.t) { best = sh; got = true; }
                    }
                    prim = any_hit ? (got ? 1 : 0) : (got ? best.shape : -1);
                    t = best.t;
                } else if (mode == F32_STACK) {
                    const Ray<float> r{{(float)q[0], (float)q[1], (float)q[2]}, {(float)q[3], (float)q[4], (float)q[5]}};
                    Hit<float> h;
                    const bool got = trace_stack_f32(s->sc, r, (float)q[6], any_hit ? (float)q[7] : FLT_MAX, any_hit != 0, h, &tc);
                    prim = any_hit ? (got ? 1 : 0) : (got ? h.shape : -1);
                    t = h.t;
                } else {
                    const Ray<double> r{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
                    Hit<double> h;
                    const bool got = mode == F64_DFS ? trace_dfs(s->sc, r, q[6], h, &tc) : trace_dijkstra(s->sc, r, q[6], h, &tc);
                    prim = got ? h.shape : -1;
                    t = h.t;
                }
                prim_out[i] = prim;
                if (t_out) t_out[i] = t;
                if (inner_out) inner_out[i] = (uint32_t)tc.inner;
                if (leaf_out) leaf_out[i] = (uint32_t)tc.leaf;
            }
        });
    for (auto& t : th) t.join();
    return 0;
}

// pixels: n * 3 int32 {i, j (from bottom), sample}; out: n * 6 doubles {o, d}
int orc_camera_rays(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int mode,
                    const int32_t* pixels, int64_t n, double* out) {
    const int W = spec->width, H = spec->height;
    const CameraF64 c64 = new_camera(s->sc, W, H);
    const Camera<double> cd = narrow_camera<double>(c64);
    const Camera<float> cf = narrow_camera<float>(c64);
    for (int64_t q = 0; q < n; ++q) {
        const int i = pixels[q * 3], j = pixels[q * 3 + 1], smp = pixels[q * 3 + 2];
        const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)((H - 1 - j) * W + i), (uint32_t)smp};
        if (mode == F32_STACK) {
            const Ray<float> r = camera_ray(cf, W, H, i, j, key);
            out[q * 6 + 0] = r.p.x; out[q * 6 + 1] = r.p.y; out[q * 6 + 2] = r.p.z;
            out[q * 6 + 3] = r.d.x; out[q * 6 + 4] = r.d.y; out[q * 6 + 5] = r.d.z;
        } else {
            const Ray<double> r = camera_ray(cd, W, H, i, j, key);
            out[q * 6 + 0] = r.p.x; out[q * 6 + 1] = r.p.y; out[q * 6 + 2] = r.p.z;
            out[q * 6 + 3] = r.d.x; out[q * 6 + 4] = r.d.y; out[q * 6 + 5] = r.d.z;
        }
    }
    return 0;
}

// GenerateImage (raytracer.go:58-107) restricted to output rows [row0, row1) and samples
// [s0, s1) of each pixel.  accum_out: W*H*3 doubles, un-normalised sums, row 0 = top (ADDED
// into).  rgba_out (optional): quantised with spec->anti_aliasing_factor as the divisor.
int orc_render(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, int mode, int nthreads,
               int row0, int row1, int s0, int s1, double* accum_out, uint8_t* rgba_out, orc_counters* ctr) {
    if (mode < 0 || mode > 2) { g_err = "bad mode"; return -1; }
    if (spec->bvh_traversal_algorithm != 0 && spec->bvh_traversal_algorithm != 1) {
        g_err = "No trace algorithm found";  // accelerationstructures.go:73
        return -3;
    }
    const int W = spec->width, H = spec->height;
    const CameraF64 c64 = new_camera(s->sc, W, H);
    const Camera<double> cd = narrow_camera<double>(c64);
    const Camera<float> cf = narrow_camera<float>(c64);
    nthreads = std::max(1, nthreads);
    std::vector<RayCounts> counts(nthreads);
    std::atomic<int> next_row{row0};
    std::vector<std::thread> th;
    for (int w = 0; w < nthreads; ++w)
        th.emplace_back([&, w]() {
            RayCounts* rc = &counts[w];
            for (;;) {
                const int row = next_row.fetch_add(1);
                if (row >= row1) break;
                const int j = H - 1 - row;
                for (int i = 0; i < W; ++i) {
                    V3<double> px{0, 0, 0};
                    for (int smp = s0; smp < s1; ++smp) {  // raytracer.go:113-118
                        const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row * W + i), (uint32_t)smp};
                        if (mode == F32_STACK) {
                            const Tracer<float> trc{s->sc, mode, nullptr};
                            const V3<float> c = color(trc, camera_ray(cf, W, H, i, j, key), spec->ray_tracing_max_depth,
                                                      spec->soft_shadow_monte_carlo_repetitions, key, rc);
                            px = add(px, V3<double>{c.x, c.y, c.z});
                        } else {
                            const Tracer<double> trc{s->sc, mode, nullptr};
                            const V3<double> c = color(trc, camera_ray(cd, W, H, i, j, key), spec->ray_tracing_max_depth,
                                                       spec->soft_shadow_monte_carlo_repetitions, key, rc);
                            px = add(px, c);
                        }
                    }
                    double* a = accum_out + ((size_t)row * W + i) * 3;
                    a[0] += px.x; a[1] += px.y; a[2] += px.z;
                }
            }
        });
    for (auto& t : th) t.join();
    if (rgba_out) {  // raytracer.go:93-96, :119
        const double inv = 1.0 / (double)spec->anti_aliasing_factor;
        for (int row = row0; row < row1; ++row)
            for (int i = 0; i < W; ++i) {
                const double* a = accum_out + ((size_t)row * W + i) * 3;
                uint8_t* o = rgba_out + ((size_t)row * W + i) * 4;
                for (int ch = 0; ch < 3; ++ch) o[ch] = (uint8_t)std::fmin(255.0, (inv * a[ch]) * 255.99);
                o[3] = 255;
            }
    }
    if (ctr) {
        std::memset(ctr, 0, sizeof(*ctr));
        for (const RayCounts& c : counts) {
            ctr->rays_primary += c.primary; ctr->rays_continuation += c.continuation;
            ctr->rays_shadow += c.shadow; ctr->phong_hits += c.phong_hits;
            ctr->inner_primary += c.tc_primary.inner; ctr->leaf_primary += c.tc_primary.leaf;
            ctr->inner_continuation += c.tc_continuation.inner; ctr->leaf_continuation += c.tc_continuation.leaf;
            ctr->inner_shadow += c.tc_shadow.inner; ctr->leaf_shadow += c.tc_shadow.leaf;
        }
    }
    return 0;
}

// Known-answer probe for one hit: given a ray and the shape it is known to hit, evaluates
// material.scatter (material.go:34-132) in float64/best-first mode.
// out: {should_trace, att.xyz, scattered.p.xyz, scattered.d.xyz, color.xyz, t, p.xyz, n.xyz} = 20 doubles
int orc_scatter_probe(const orc_scene* s, const rt_image_spec* spec, uint64_t seed, const double* ray6,
                      uint32_t pixel, uint32_t sample, uint32_t bounce, double* out) {
    const Ray<double> r{{ray6[0], ray6[1], ray6[2]}, {ray6[3], ray6[4], ray6[5]}};
    Hit<double> hit;
    const Tracer<double> trc{s->sc, F64_DIJKSTRA, nullptr};
    for (int i = 0; i < 20; ++i) out[i] = 0;
    if (!trc.trace(r, 0.0, hit)) return 1;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), pixel, sample};
    const Scatter<double> sc = scatter(trc, r, hit, spec->soft_shadow_monte_carlo_repetitions, key, bounce, nullptr, nullptr);
    out[0] = sc.should_trace;
    out[1] = sc.attenuation.x; out[2] = sc.attenuation.y; out[3] = sc.attenuation.z;
    out[4] = sc.scattered.p.x; out[5] = sc.scattered.p.y; out[6] = sc.scattered.p.z;
    out[7] = sc.scattered.d.x; out[8] = sc.scattered.d.y; out[9] = sc.scattered.d.z;
    out[10] = sc.color.x; out[11] = sc.color.y; out[12] = sc.color.z;
    out[13] = hit.t;
    out[14] = hit.p.x; out[15] = hit.p.y; out[16] = hit.p.z;
    out[17] = hit.normal.x; out[18] = hit.normal.y; out[19] = hit.normal.z;
    return 0;
}

// n samples of random_in_unit_sphere (float64 form) for the distribution test: out = n * 3 doubles
void orc_sample_unit_sphere(uint64_t seed, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(i & 0xffffffff), (uint32_t)(i >> 32)};
        const V3<double> p = random_in_unit_sphere<double>(key, 3, P_SHADOW, (uint32_t)(i % 48));
        out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
    }
}

}  // extern "C"
