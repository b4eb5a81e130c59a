// CPU-only check of the host tree builders (raytracer-go_amd/csrc/octree.cpp), compiled by
// tests/test_host_api.py with -fsanitize=address,undefined.  For random and degenerate scenes and both
// builders it verifies the invariants the traversal relies on:
//   - every shape is in exactly one leaf;
//   - a leaf record's box encloses the bounds of the leaf's shapes, an inner record's box encloses
//     every record of its block (so no ray can miss a shape by culling a box);
//   - record / block indices are in range, blocks hold 1..8 records in slot order;
//   - walking the blocks depth-first never needs more stack than FlatTree::stack_bound;
//   - every quantised block (QBlock, the shadow-ray walk's copy of the tree) decodes, in exact arithmetic, to boxes
//     that enclose the child records' fp32 boxes, names the same leaf shapes and the blocks of the same inner children;
//   - the SAH tree of a scene long enough for the builder's threads (chunked sweeps and forked halves) is
//     byte-identical for 1, 3, 7 and 16 threads;
//   - a coordinate of -0.0 builds the tree that +0.0 builds (the SAH builder's min / max rely on it).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../raytracer-go_amd/csrc/octree.hpp"

using namespace rtgo;

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++fails; } } while (0)

static void bounds(const rt_shape& s, double lo[3], double hi[3]) {
    if (s.kind == RT_SHAPE_SPHERE) {
        const double c[3] = {s.a.x, s.a.y, s.a.z};
        for (int a = 0; a < 3; ++a) { lo[a] = c[a] - s.b.x; hi[a] = c[a] + s.b.x; }
        return;
    }
    const double p[3][3] = {{s.a.x, s.a.y, s.a.z}, {s.b.x, s.b.y, s.b.z}, {s.c.x, s.c.y, s.c.z}};
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::fmin(p[0][a], std::fmin(p[1][a], p[2][a]));
        hi[a] = std::fmax(p[0][a], std::fmax(p[1][a], p[2][a]));
    }
}

static int max_stack = 0;
static void walk(const FlatTree& t, uint32_t ref, int stack, const ChildRecord* parent, std::vector<int>& seen,
                 const std::vector<rt_shape>& shapes, int depth) {
    const uint32_t first = ref >> 8, present = ref & 0xffu;
    const int n = __builtin_popcount(present);
    CHECK(n >= 1 && n <= 8 && (size_t)first + n <= t.records.size(), "block %u has %d records", first, n);
    CHECK(depth <= t.depth, "block deeper (%d) than FlatTree::depth %d", depth, t.depth);
    int inner = 0;
    for (int k = 0; k < n; ++k) inner += (t.records[first + k].w1 & kLeafFlag) ? 0 : 1;
    const int here = stack + (inner > 0 ? inner - 1 : 0);
    if (here > max_stack) max_stack = here;
    for (int k = 0; k < n; ++k) {
        const ChildRecord& r = t.records[first + k];
        if (parent)
            for (int a = 0; a < 3; ++a)
                CHECK(r.bmin[a] >= parent->bmin[a] && r.bmax[a] <= parent->bmax[a], "record %u sticks out of its parent on axis %d", first + k, a);
        if (r.w1 & kLeafFlag) {
            const uint32_t pf = r.w0 >> 8, pc = r.w0 & 0xffu;
            CHECK(pc >= 1 && (size_t)pf + pc <= t.order.size(), "leaf range %u+%u", pf, pc);
            for (uint32_t q = pf; q < pf + pc && q < t.order.size(); ++q) {
                const int sh = t.order[q];
                CHECK(sh >= 0 && (size_t)sh < shapes.size(), "shape index %d", sh);
                seen[(size_t)sh]++;
                double lo[3], hi[3];
                bounds(shapes[(size_t)sh], lo, hi);
                for (int a = 0; a < 3; ++a)
                    CHECK((double)r.bmin[a] <= lo[a] - t.qpad && (double)r.bmax[a] >= hi[a] + t.qpad, "leaf box misses shape %d grown by the pad on axis %d", sh, a);
            }
        } else {
            walk(t, r.w0, here, &r, seen, shapes, depth + 1);
        }
    }
}

// Walks records and QBlocks side by side: block `ref` of the record tree must be block `blk` of the quantised tree.
static void walk_q(const FlatTree& t, uint32_t ref, uint32_t blk, size_t* visited) {
    ++*visited;
    CHECK((size_t)(blk + 1) * kQBlockWords <= t.qblocks.size(), "qblock %u out of range", blk);
    if ((size_t)(blk + 1) * kQBlockWords > t.qblocks.size()) return;
    const uint32_t* w = t.qblocks.data() + (size_t)blk * kQBlockWords;
    const uint32_t first = ref >> 8, present = ref & 0xffu;
    CHECK((w[3] >> 24) == present, "qblock %u: present mask %02x, records say %02x", blk, w[3] >> 24, present);
    float origin[3];
    std::memcpy(origin, w, 12);
    const uint8_t* meta = reinterpret_cast<const uint8_t*>(w + 6);
    const uint8_t* qlo = reinterpret_cast<const uint8_t*>(w + 8);
    const uint8_t* qhi = reinterpret_cast<const uint8_t*>(w + 14);
    uint32_t leafm = 0, shape = w[5], rank = 0;
    int k = 0;
    for (int o = 0; o < 8; ++o) {
        if (!((present >> o) & 1u)) { CHECK(meta[o] == 0, "qblock %u: absent slot %d has meta %u", blk, o, meta[o]); continue; }
        const ChildRecord& r = t.records[first + (size_t)k++];
        for (int a = 0; a < 3; ++a) {
            const int e = (int)((w[3] >> (8 * a)) & 0xffu);
            CHECK(e >= 1 && e <= 254, "qblock %u: exponent %d", blk, e);
            const double step = std::ldexp(1.0, e - 127);
            const double lo = (double)origin[a] + qlo[8 * a + o] * step, hi = (double)origin[a] + qhi[8 * a + o] * step;
            CHECK(lo <= (double)r.bmin[a] && hi >= (double)r.bmax[a], "qblock %u slot %d axis %d: [%g, %g] does not enclose [%g, %g]", blk, o, a, lo, hi,
                  (double)r.bmin[a], (double)r.bmax[a]);
            CHECK((double)r.bmin[a] - lo < step && hi - (double)r.bmax[a] < step, "qblock %u slot %d axis %d: more than one grid step of slack", blk, o, a);
        }
        if (r.w1 & kLeafFlag) {
            leafm |= 1u << o;
            CHECK((r.w0 >> 8) == shape && (r.w0 & 0xffu) == (meta[o] & 0x1fu) && (meta[o] >> 5) == 0, "qblock %u slot %d: leaf %u+%u, record says %u+%u", blk, o,
                  shape, meta[o] & 0x1fu, r.w0 >> 8, r.w0 & 0xffu);
            shape += meta[o] & 0x1fu;
        } else {
            CHECK((meta[o] & 0x1fu) == 0 && (uint32_t)(meta[o] >> 5) == rank, "qblock %u slot %d: inner rank %u, expected %u", blk, o, meta[o] >> 5, rank);
            walk_q(t, r.w0, (w[4] & 0xffffffu) + rank, visited);
            ++rank;
        }
    }
    CHECK((w[4] >> 24) == leafm, "qblock %u: leaf mask %02x, records say %02x", blk, w[4] >> 24, leafm);
}

static void check_scene(const char* name, const std::vector<rt_shape>& shapes, int leaf_size) {
    for (int kind = 0; kind < 2; ++kind) {
        const HostNodes nodes = kind == 0 ? build_sah(shapes.data(), (int64_t)shapes.size(), leaf_size)
                                                      : build_octree(shapes.data(), (int64_t)shapes.size());
        FlatTree t;
        const char* err = nullptr;
        CHECK(flatten_octree(nodes, leaf_size, &t, &err), "%s: flatten failed: %s", name, err ? err : "?");
        std::vector<int> seen(shapes.size(), 0);
        max_stack = 0;
        if (!shapes.empty()) walk(t, t.root_ref, 0, nullptr, seen, shapes, 1);
        for (size_t i = 0; i < shapes.size(); ++i) CHECK(seen[i] == 1, "%s (%s): shape %zu is in %d leaves", name, kind ? "octree" : "sah", i, seen[i]);
        CHECK(max_stack < t.stack_bound, "%s: walk needs %d stack entries, bound %d", name, max_stack, t.stack_bound);
        CHECK(t.order.size() == shapes.size(), "%s: order has %zu entries", name, t.order.size());
        size_t qvisited = 0;
        if (!shapes.empty()) walk_q(t, t.root_ref, 0, &qvisited);
        CHECK(t.qblocks.size() == (size_t)std::max<long long>(1, (long long)t.n_inner) * kQBlockWords && (shapes.empty() || qvisited == (size_t)t.n_inner),
              "%s: %zu quantised words, %zu blocks reached, %lld blocks", name, t.qblocks.size(), qvisited, (long long)t.n_inner);
        std::printf("%-28s %-6s leaf %d: %6zu shapes %6lld blocks %7zu records depth %2d stack %2d (bound %2d)\n", name, kind ? "octree" : "sah",
                    leaf_size, shapes.size(), (long long)t.n_inner, t.records.size(), t.depth, max_stack, t.stack_bound);
    }
}

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    auto sphere = [&](double x, double y, double z, double r) { rt_shape s{}; s.kind = RT_SHAPE_SPHERE; s.a = {x, y, z}; s.b = {r, 0, 0}; return s; };
    auto tri = [&](double x, double y, double z, double e) {
        rt_shape s{}; s.kind = RT_SHAPE_TRIANGLE;
        s.a = {x, y, z}; s.b = {x + e * u(rng), y + e * u(rng), z + e * u(rng)}; s.c = {x + e * u(rng), y + e * u(rng), z + e * u(rng)};
        return s;
    };
    for (int leaf : {1, 4, 6, 8}) {
        check_scene("empty", {}, leaf);
        check_scene("one sphere", {sphere(0, 0, 0, 1)}, leaf);
        std::vector<rt_shape> v;
        for (int i = 0; i < 5000; ++i) v.push_back(i % 3 ? tri(10 * u(rng), 10 * u(rng), 10 * u(rng), 0.3) : sphere(10 * u(rng), 10 * u(rng), 10 * u(rng), 0.2));
        check_scene("random mix", v, leaf);
        v.clear();
        for (int i = 0; i < 300; ++i) v.push_back(sphere(1.0, 2.0, 3.0, 0.5));  // identical centroids
        check_scene("coincident", v, leaf);
        v.clear();
        for (int i = 0; i < 1500; ++i) { const double x = std::pow(1.1, i / 10.0); v.push_back(sphere(x, 0.02 * x * (i % 7 - 3), 0, 0.04 * x)); }
        check_scene("exponential spacing", v, leaf);
        v.clear();
        for (int i = 0; i < 2000; ++i) v.push_back(tri(i * 0.01, 0, 0, 1e-9));  // a line of needle-thin triangles
        v.push_back(tri(0, 0, 0, 1e3));                                          // and one huge triangle over them
        check_scene("line + huge", v, leaf);
    }
    {
        std::vector<rt_shape> v;
        for (int i = 0; i < 300000; ++i) {  // two clusters of unequal size: uneven splits near the root
            const double off = i % 5 ? 0.0 : 40.0;
            v.push_back(i % 11 ? tri(off + 10 * u(rng), 10 * u(rng), 10 * u(rng), 0.05) : sphere(off + 10 * u(rng), 10 * u(rng), 10 * u(rng), 0.03));
        }
        FlatTree ref;
        for (const char* threads : {"1", "3", "7", "16"}) {
            setenv("RTGO_BUILD_THREADS", threads, 1);
            FlatTree t;
            const char* err = nullptr;
            CHECK(flatten_octree(build_sah(v.data(), (int64_t)v.size(), 6), 6, &t, &err), "flatten failed: %s", err ? err : "?");
            if (threads[0] == '1') { ref = t; continue; }
            CHECK(t.records.size() == ref.records.size() && t.order == ref.order && t.root_ref == ref.root_ref && t.stack_bound == ref.stack_bound &&
                      t.qblocks == ref.qblocks && t.block_first == ref.block_first && t.block_box == ref.block_box &&
                      std::memcmp(t.records.data(), ref.records.data(), t.records.size() * sizeof(ChildRecord)) == 0,
                  "SAH tree differs between 1 and %s threads", threads);
        }
        unsetenv("RTGO_BUILD_THREADS");
        check_scene("two clusters, threaded", v, 6);
    }
    {   // -0.0 among the coordinates: the same tree as with +0.0, whatever the threads (the builder's one-instruction
        // min / max return either argument of an equal pair)
        std::vector<rt_shape> plus, minus;
        for (int i = 0; i < 4000; ++i) {
            const double x = (i % 40) * 0.25 - 5.0, y = (i / 40) * 0.25 - 12.5;
            rt_shape t = tri(x, y, 0, 0.2);
            plus.push_back(t);
            for (rt_vec3* v : {&t.a, &t.b, &t.c}) { if (v->x == 0) v->x = -0.0; if (v->y == 0) v->y = -0.0; if (v->z == 0) v->z = -0.0; }
            minus.push_back(t);
        }
        FlatTree tp, tm;
        const char* err = nullptr;
        CHECK(flatten_octree(build_sah(plus.data(), (int64_t)plus.size(), 6), 6, &tp, &err) && flatten_octree(build_sah(minus.data(), (int64_t)minus.size(), 6), 6, &tm, &err), "flatten failed");
        CHECK(tp.order == tm.order && tp.qblocks == tm.qblocks && tp.records.size() == tm.records.size(), "-0.0 coordinates change the SAH tree");
    }
    {   // equal centroids where the reference's 1e-10 jitter is below one ulp: its insertion octree never ends
        // (accelerationstructures.go:226-279); build_octree refuses the scene, build_sah (no jitter, ranges halved
        // when nothing separates them) builds it
        std::vector<rt_shape> v{sphere(1e8, 1e8, 1e8, 0.5), sphere(1e8, 1e8, 1e8, 0.25), sphere(0, 0, 0, 1)};
        bool refused = false;
        try { (void)build_octree(v.data(), (int64_t)v.size()); } catch (const std::invalid_argument&) { refused = true; }
        CHECK(refused, "build_octree did not refuse inseparable centroids");
        FlatTree t;
        const char* err = nullptr;
        CHECK(flatten_octree(build_sah(v.data(), (int64_t)v.size(), 6), 6, &t, &err) && t.order.size() == 3, "build_sah failed on inseparable centroids");
    }
    {   // the builders' threads: an exception in a worker comes back on the starting thread, after every worker has ended
        std::atomic<int> done{0};
        bool thrown = false;
        try {
            ThreadGroup g;
            for (int i = 0; i < 6; ++i)
                g.run([&done, i] {
                    if (i == 3) throw std::runtime_error("worker 3");
                    ++done;
                });
            g.join();
        } catch (const std::runtime_error& e) {
            thrown = std::string(e.what()) == "worker 3";
        }
        CHECK(thrown && done.load() == 5, "ThreadGroup: thrown %d, workers done %d", (int)thrown, done.load());
    }
    std::printf(fails ? "%d FAILURES\n" : "all builder invariants hold\n", fails);
    return fails ? 1 : 0;
}
