// See octree.hpp.  Host code only (no HIP): float64 throughout, like the reference.
#include "octree.hpp"

#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <deque>
#include <unordered_map>
#include <atomic>
#include <memory>
#include <thread>
#include <condition_variable>
#include <chrono>
#include <cstdio>

namespace rtgo {
namespace {

// The reference jitters centroids with Go's global math/rand (accelerationstructures.go:267);
// that stream is unseeded and shared, so any fixed stream is an equally valid draw.  This one
// is splitmix64 from a fixed seed (documented in DESIGN.md so the CPU oracle can use the same).
struct Jitter {
    uint64_t s = 0x6A09E667F3BCC909ull;
    double next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
};

constexpr double kJitterFactor = 0.0000000001;  // bvhCentroidJitterFactor, raytracer.go:12

// computeSquareBounds: shape.go:91-93 (sphere), :183-207 (triangle)
void shape_bounds(const rt_shape& s, double lo[3], double hi[3]) {
    if (s.kind == RT_SHAPE_SPHERE) {
        const double r = s.b.x;
        lo[0] = s.a.x - r; lo[1] = s.a.y - r; lo[2] = s.a.z - r;
        hi[0] = s.a.x + r; hi[1] = s.a.y + r; hi[2] = s.a.z + r;
        return;
    }
    const double px[3] = {s.a.x, s.b.x, s.c.x}, py[3] = {s.a.y, s.b.y, s.c.y}, pz[3] = {s.a.z, s.b.z, s.c.z};
    lo[0] = std::fmin(std::fmin(px[0], px[1]), px[2]); hi[0] = std::fmax(std::fmax(px[0], px[1]), px[2]);
    lo[1] = std::fmin(std::fmin(py[0], py[1]), py[2]); hi[1] = std::fmax(std::fmax(py[0], py[1]), py[2]);
    lo[2] = std::fmin(std::fmin(pz[0], pz[1]), pz[2]); hi[2] = std::fmax(std::fmax(pz[0], pz[1]), pz[2]);
}

// centroid: shape.go:95-97 (sphere), :209-211 (triangle: Scale(1/3.0, A + (B + C)))
void shape_centroid(const rt_shape& s, double c[3]) {
    if (s.kind == RT_SHAPE_SPHERE) { c[0] = s.a.x; c[1] = s.a.y; c[2] = s.a.z; return; }
    const double k = 1 / 3.0;
    c[0] = k * (s.a.x + (s.b.x + s.c.x));
    c[1] = k * (s.a.y + (s.b.y + s.c.y));
    c[2] = k * (s.a.z + (s.b.z + s.c.z));
}

// getBvhQuadrantIndex: accelerationstructures.go:266-279 (bit0 = X, bit1 = Y, bit2 = Z)
int octant_of(const rt_shape& s, const HostNode& cell, Jitter& rng) {
    double c[3];
    shape_centroid(s, c);
    const double jx = rng.next(), jy = rng.next(), jz = rng.next();
    c[0] += kJitterFactor * jx; c[1] += kJitterFactor * jy; c[2] += kJitterFactor * jz;
    int idx = 0;
    if (c[0] > cell.lo[0] + (cell.hi[0] - cell.lo[0]) / 2) idx += 1;
    if (c[1] > cell.lo[1] + (cell.hi[1] - cell.lo[1]) / 2) idx += 2;
    if (c[2] > cell.lo[2] + (cell.hi[2] - cell.lo[2]) / 2) idx += 4;
    return idx;
}

// splitBvhQuadrant: accelerationstructures.go:282-353
void split_cell(HostNodes& nodes, int32_t at) {
    const int32_t base = (int32_t)nodes.size();
    double lo[3], hi[3], half[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = nodes[at].lo[a]; hi[a] = nodes[at].hi[a];
        half[a] = (hi[a] - lo[a]) / 2;
    }
    for (int o = 0; o < 8; ++o) {
        HostNode c;
        for (int a = 0; a < 3; ++a) {
            const bool upper = (o >> a) & 1;
            c.lo[a] = upper ? lo[a] + half[a] : lo[a];
            c.hi[a] = upper ? hi[a] : lo[a] + half[a];
        }
        nodes.push_back(c);
    }
    nodes[at].child_base = base;
}

float round_down_f32(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
float round_up_f32(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

// Skip nodes that have exactly one non-empty child (same box as that child after the refit).
int32_t skip_chain(const HostNodes& nodes, int32_t n) {
    while (nodes[n].child_base >= 0) {
        int32_t only = -1, cnt = 0;
        for (int o = 0; o < 8; ++o) {
            const int32_t c = nodes[n].child_base + o;
            if (!nodes[c].empty) { only = c; ++cnt; }
        }
        if (cnt != 1) break;
        n = only;
    }
    return n;
}

}  // namespace

constexpr int kMaxOctreeLevels = 1300;
HostNodes build_octree(const rt_shape* shapes, int64_t n) {
    HostNodes nodes;
    nodes.reserve((size_t)(n * 4 + 16));
    HostNode root;
    // accelerationstructures.go:29-42: union of all shape bounds, widened by the max jitter
    for (int a = 0; a < 3; ++a) { root.lo[a] = DBL_MAX; root.hi[a] = -9223372036854775808.0; }
    for (int64_t i = 0; i < n; ++i) {
        double lo[3], hi[3];
        shape_bounds(shapes[i], lo, hi);
        for (int a = 0; a < 3; ++a) { root.lo[a] = std::fmin(root.lo[a], lo[a]); root.hi[a] = std::fmax(root.hi[a], hi[a]); }
    }
    for (int a = 0; a < 3; ++a) { root.lo[a] -= kJitterFactor * 1; root.hi[a] += kJitterFactor * 1; }
    nodes.push_back(root);

    // accelerationstructures.go:57-60 + addToBVH :226-255.  The recursion of the reference is
    // unrolled: after a split all 8 children are empty, so the evicted shape lands in its child
    // in one step (its 3 jitter draws come first, as in the reference), then the new shape
    // keeps descending.
    Jitter rng;
    for (int64_t i = 0; i < n; ++i) {
        int32_t cur = 0;
        for (int level = 0;; ++level) {
            // Two shapes whose centroids the 1e-10 jitter cannot tell apart (equal centroids at |c| >~ 1e6, where the
            // jitter is below one ulp) land in the same octant at every level: the cell's half-width reaches 0 and the
            // reference recurses until its stack overflows (accelerationstructures.go:226-255).  A cell can be halved
            // about 1200 times between 1e30 and the smallest double; beyond that the scene is refused.
            if (level > kMaxOctreeLevels)
                throw std::invalid_argument("shapes " + std::to_string(nodes[cur].shape >= 0 ? nodes[cur].shape : i) + " and " + std::to_string(i) +
                                            " have centroids that the insertion octree cannot separate (RTGO_TREE=octree; the default tree has no such limit)");
            if (nodes[cur].child_base < 0) {
                if (nodes[cur].shape < 0) { nodes[cur].shape = (int32_t)i; break; }
                const int32_t evicted = nodes[cur].shape;
                nodes[cur].shape = -1;
                split_cell(nodes, cur);
                const int32_t dst = nodes[cur].child_base + octant_of(shapes[evicted], nodes[cur], rng);
                nodes[dst].shape = evicted;
            }
            cur = nodes[cur].child_base + octant_of(shapes[i], nodes[cur], rng);
        }
    }

    // recomputeNodeBounds (:189-211).  Children are always appended after their parent, so a
    // reverse sweep is a post-order walk.  `empty` stands for the reference's
    // (+MaxFloat64, float64(MinInt64)) sentinel box that destroyUselessNodes (:213-224) prunes.
    for (int64_t k = (int64_t)nodes.size() - 1; k >= 0; --k) {
        HostNode& nd = nodes[k];
        if (nd.child_base < 0) {
            nd.empty = nd.shape < 0;
            nd.n_shapes = nd.empty ? 0 : 1;
            if (!nd.empty) shape_bounds(shapes[nd.shape], nd.lo, nd.hi);
            continue;
        }
        nd.empty = true;
        nd.n_shapes = 0;
        for (int a = 0; a < 3; ++a) { nd.lo[a] = DBL_MAX; nd.hi[a] = -9223372036854775808.0; }
        for (int o = 0; o < 8; ++o) {
            const HostNode& c = nodes[nd.child_base + o];
            if (c.empty) continue;
            nd.empty = false;
            nd.n_shapes += c.n_shapes;
            for (int a = 0; a < 3; ++a) { nd.lo[a] = std::fmin(nd.lo[a], c.lo[a]); nd.hi[a] = std::fmax(nd.hi[a], c.hi[a]); }
        }
    }
    return nodes;
}

// ---- binned-SAH BVH emitted in the same 8-slot HostNode form (the default tree; spec in octree.hpp) ----
namespace {
struct SahItem { double lo[3], hi[3]; int32_t shape; };  // centroid = (lo + hi) * 0.5, recomputed where needed
struct Bvh2 { double lo[3], hi[3]; int32_t left, right, b, e; };
double area3(const double lo[3], const double hi[3]) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0 * ((dx * dy + dy * dz) + dz * dx);
}
constexpr int kMaxSahDepth = 48;
constexpr int32_t kChunkMin = 131072;    // ranges at least this long are swept by several threads
struct SahBuilder {
    // min / max of finite numbers as one instruction each (std::fmin's NaN rule costs three).  The items' coordinates
    // are finite (checked at scene creation) and no -0 is among them (build_sah), so which argument an equal pair
    // returns cannot be seen and merging the chunks' minima in any order gives the same bits.
    static double mn(double a, double b) { return b < a ? b : a; }
    static double mx(double a, double b) { return b > a ? b : a; }
    // The items live in two arrays of the same length.  A range is split by ONE sweep that reads it in the array it is
    // in and writes its left and right halves, each in the old order, to the same range of the other array (a long
    // range streams through memory once per level instead of three times: the top levels are bound by that), and
    // gathers the halves' bounds on the way, which the children would otherwise sweep for.  A leaf's items end up in
    // buf[0], where emit() reads them.
    std::unique_ptr<SahItem[]> buf[2];
    std::unique_ptr<Bvh2[]> b2;        // 2n - 1 nodes at most (every leaf holds a shape); numbered as they are made
    std::atomic<int32_t> n_nodes{0};
    int leaf_size;
    // depth: SAH may split 1 : n-1 again and again on adversarial input; beyond kMaxSahDepth levels the
    // range is halved instead, so the recursion stays O(kMaxSahDepth + log n) deep.
    // threads: > 1 only on the thread that builds the top of the tree (chunked sweeps, see build_all); the other
    // ranges are tasks.  The two halves of a range touch disjoint items and only the node numbers depend on the
    // interleaving; emit() follows left/right links, so the tree that comes out is the same for any number of threads.
    static constexpr int kBins = 16;
    struct Bin { double lo[3], hi[3]; int n; };
    struct Box {  // of a range: its items' bounds and their centroids' bounds
        double lo[3], hi[3], clo[3], chi[3];
        void clear() { for (int a = 0; a < 3; ++a) { lo[a] = clo[a] = DBL_MAX; hi[a] = chi[a] = -DBL_MAX; } }
        void add(const SahItem& it) {
            for (int a = 0; a < 3; ++a) {
                const double c = (it.lo[a] + it.hi[a]) * 0.5;
                lo[a] = mn(lo[a], it.lo[a]); hi[a] = mx(hi[a], it.hi[a]);
                clo[a] = mn(clo[a], c); chi[a] = mx(chi[a], c);
            }
        }
        void add(const Box& o) {  // min and max merge exactly: chunking a sweep changes no value
            for (int a = 0; a < 3; ++a) {
                lo[a] = mn(lo[a], o.lo[a]); hi[a] = mx(hi[a], o.hi[a]);
                clo[a] = mn(clo[a], o.clo[a]); chi[a] = mx(chi[a], o.chi[a]);
            }
        }
    };
    struct Pass {  // what one chunk of a range contributes to the sweeps over it
        Bin bins[3][kBins];
        Box left, right;
        int32_t n_left;
    };
    // f(chunk, first, last) over nt equal chunks of [b, e), chunk 0 on the calling thread
    template <class F>
    static void chunks(int32_t b, int32_t e, int nt, F&& f) {
        if (nt <= 1) { f(0, b, e); return; }
        const int64_t len = e - b;
        ThreadGroup group;
        for (int c = 1; c < nt; ++c)
            group.run([&f, b, len, nt, c] { f(c, b + (int32_t)(len * c / nt), b + (int32_t)(len * (c + 1) / nt)); });
        f(0, b, b + (int32_t)(len / nt));
        group.join();
    }
    Box range_box(int32_t b, int32_t e, int threads) const {  // of the whole input; every other range gets its box from its parent's sweep
        const int nt = (e - b >= kChunkMin) ? threads : 1;
        std::vector<Box> part((size_t)nt);
        const SahItem* items = buf[0].get();
        chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {
            Box bx; bx.clear();
            for (int32_t i = cb; i < ce; ++i) bx.add(items[i]);
            part[(size_t)c] = bx;
        });
        Box all; all.clear();
        for (const Box& p : part) all.add(p);
        return all;
    }
    // the subtree of the range [b, e) of buf[in]; bx = the range's box
    int32_t build(int32_t b, int32_t e, int depth, int threads, int in, const Box& bx) {
        const int nt = (e - b >= kChunkMin) ? threads : 1;
        Bvh2 nd;
        nd.b = b; nd.e = e; nd.left = nd.right = -1;
        for (int a = 0; a < 3; ++a) { nd.lo[a] = bx.lo[a]; nd.hi[a] = bx.hi[a]; }
        const double* clo = bx.clo; const double* chi = bx.chi;
        const int32_t me = n_nodes.fetch_add(1, std::memory_order_relaxed);
        b2[me] = nd;
        if (e - b <= leaf_size) {
            if (in != 0) std::copy(buf[1].get() + b, buf[1].get() + e, buf[0].get() + b);
            return me;
        }
        const SahItem* items = buf[in].get();

        Pass one;
        std::vector<Pass> many;
        Pass* ps = &one;
        if (nt > 1) { many.resize((size_t)nt); ps = many.data(); }
        bool use[3]; double scale[3];
        bool any = false;
        for (int a = 0; a < 3; ++a) {
            const double ext = chi[a] - clo[a];
            use[a] = depth < kMaxSahDepth && ext > 0;
            scale[a] = use[a] ? kBins / ext : 0;
            any = any || use[a];
        }
        int best_axis = -1, best_split = 0;
        double best_cost = DBL_MAX;
        if (any) {
            pool_chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {  // the three axes' bins in one sweep
                Pass& p = ps[c];
                for (int a = 0; a < 3; ++a)
                    for (auto& bn : p.bins[a]) { bn.n = 0; for (int k = 0; k < 3; ++k) { bn.lo[k] = DBL_MAX; bn.hi[k] = -DBL_MAX; } }
                for (int32_t i = cb; i < ce; ++i)
                    for (int a = 0; a < 3; ++a) {
                        if (!use[a]) continue;
                        int k = (int)(((items[i].lo[a] + items[i].hi[a]) * 0.5 - clo[a]) * scale[a]);  // coordinates are finite (checked at scene creation)
                        if (k >= kBins) k = kBins - 1;
                        if (k < 0) k = 0;
                        Bin& bn = p.bins[a][k];
                        bn.n++;
                        for (int q = 0; q < 3; ++q) { bn.lo[q] = mn(bn.lo[q], items[i].lo[q]); bn.hi[q] = mx(bn.hi[q], items[i].hi[q]); }
                    }
            });
            for (int a = 0; a < 3; ++a) {
                if (!use[a]) continue;
                Bin bins[kBins];  // the chunks' bins merged (theirs are kept: a chunk's own counts place its items below)
                for (int k = 0; k < kBins; ++k) bins[k] = ps[0].bins[a][k];
                for (int c = 1; c < nt; ++c)
                    for (int k = 0; k < kBins; ++k) {
                        const Bin& o = ps[c].bins[a][k];
                        bins[k].n += o.n;
                        for (int q = 0; q < 3; ++q) { bins[k].lo[q] = mn(bins[k].lo[q], o.lo[q]); bins[k].hi[q] = mx(bins[k].hi[q], o.hi[q]); }
                    }
                double ra[kBins]; int rn[kBins];
                double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
                // an empty bin changes neither side's box nor count: the boundary behind it costs what the one before it
                // does and cannot win (strictly smaller wins), so most of a short range's 16 bins are passed over
                int cnt = 0;
                double area = 0;
                for (int k = kBins - 1; k >= 1; --k) {
                    if (bins[k].n) {
                        for (int q = 0; q < 3; ++q) { lo[q] = mn(lo[q], bins[k].lo[q]); hi[q] = mx(hi[q], bins[k].hi[q]); }
                        cnt += bins[k].n;
                        area = area3(lo, hi);
                    }
                    ra[k] = area; rn[k] = cnt;
                }
                for (int q = 0; q < 3; ++q) { lo[q] = DBL_MAX; hi[q] = -DBL_MAX; }
                cnt = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    if (bins[k].n == 0) continue;
                    for (int q = 0; q < 3; ++q) { lo[q] = mn(lo[q], bins[k].lo[q]); hi[q] = mx(hi[q], bins[k].hi[q]); }
                    cnt += bins[k].n;
                    if (rn[k + 1] == 0) continue;
                    const double cost = area3(lo, hi) * cnt + ra[k + 1] * rn[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
                }
            }
        }
        int32_t mid;
        int out = in;  // the array the halves are in
        Box left, right;
        if (best_axis >= 0) {
            // one sweep: every chunk writes its left and right items, in their order, to their final places in the
            // other array (a chunk's share of the left half is what its own bins counted) and gathers the halves' boxes
            const int a = best_axis;
            const double sc = scale[a], c0 = clo[a];
            int32_t total_left = 0;
            for (int c = 0; c < nt; ++c) {
                int32_t n = 0;
                for (int k = 0; k <= best_split; ++k) n += ps[c].bins[a][k].n;
                ps[c].n_left = n;
                total_left += n;
            }
            SahItem* dst = buf[in ^ 1].get();
            pool_chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {
                int32_t nl = b, nr = b + total_left + (cb - b);
                for (int k = 0; k < c; ++k) { nl += ps[k].n_left; nr -= ps[k].n_left; }
                Box l, r; l.clear(); r.clear();
                for (int32_t i = cb; i < ce; ++i) {
                    const SahItem& it = items[i];
                    int k = (int)(((it.lo[a] + it.hi[a]) * 0.5 - c0) * sc);
                    if (k >= kBins) k = kBins - 1;
                    if (k <= best_split) { l.add(it); dst[nl++] = it; } else { r.add(it); dst[nr++] = it; }
                }
                ps[c].left = l; ps[c].right = r;
            });
            left.clear(); right.clear();
            for (int c = 0; c < nt; ++c) { left.add(ps[c].left); right.add(ps[c].right); }
            mid = b + total_left;
            out = in ^ 1;
        }
        if (best_axis < 0 || mid == b || mid == e) {  // all centroids coincide (or too deep): halve the range where it is
            mid = b + (e - b) / 2;
            const SahItem* src = buf[out].get();
            left.clear(); right.clear();
            for (int32_t i = b; i < mid; ++i) left.add(src[i]);
            for (int32_t i = mid; i < e; ++i) right.add(src[i]);
        }
        // The halves.  On the thread that builds the top of the tree (threads > 1: ranges long enough for chunked sweeps)
        // a shorter half goes to the task list; on a worker (threads == 1) the right half does when both are worth a task.
        const bool queue_left = pool_on && threads > 1 && mid - b < kChunkMin;
        const bool queue_right = pool_on && (threads > 1 ? e - mid < kChunkMin : (mid - b >= grain && e - mid >= grain));
        if (queue_right) push({mid, e, depth + 1, out, right, &b2[me].right});
        if (queue_left) push({b, mid, depth + 1, out, left, &b2[me].left});
        if (!queue_left) { const int32_t l = build(b, mid, depth + 1, threads, out, left); b2[me].left = l; }
        if (!queue_right) { const int32_t r = build(mid, e, depth + 1, threads, out, right); b2[me].right = r; }
        return me;
    }
    // ---- tasks: subtrees that any thread may build (a split into two halves says little about how long each takes
    // to build, so the threads are not shared out over the halves: they take the longest range waiting) ----
    struct Task { int32_t b, e; int depth, in; Box bx; int32_t* link; };  // link: where the subtree's node number goes
    std::mutex qm;
    std::condition_variable qcv;
    std::vector<Task> tasks;
    int64_t outstanding = 0;  // tasks pushed and not finished (a task pushes its own before it finishes)
    bool pool_on = false, failed = false;
    std::exception_ptr failure;
    int32_t grain = INT32_MAX;  // ranges shorter than this are not worth a task
    void push(const Task& t) {
        {
            std::lock_guard<std::mutex> g(qm);
            tasks.push_back(t);
            ++outstanding;
        }
        qcv.notify_one();
    }
    // A chunk of a sweep over a long range: taken before any subtree, by whichever thread is free, and by the thread
    // that asked for the sweep itself as long as chunks of it are waiting (so a sweep never waits for a busy thread
    // to start a chunk, only for the chunks that are running to end).
    struct ChunkJob { void (*run)(void*, int, int32_t, int32_t); void* ctx; int c; int32_t cb, ce; std::atomic<int>* left; };
    std::vector<ChunkJob> chunk_jobs;
    void run_chunk(const ChunkJob& j) {
        try {
            j.run(j.ctx, j.c, j.cb, j.ce);
        } catch (...) {
            std::lock_guard<std::mutex> g(qm);
            if (!failed) { failed = true; failure = std::current_exception(); }
        }
        j.left->fetch_sub(1, std::memory_order_acq_rel);
    }
    template <class F>
    void pool_chunks(int32_t b, int32_t e, int nt, F&& f) {
        if (nt <= 1) { f(0, b, e); return; }
        if (!pool_on) { chunks(b, e, nt, f); return; }
        using Fn = typename std::remove_reference<F>::type;
        std::atomic<int> left{nt - 1};
        const int64_t len = e - b;
        {
            std::lock_guard<std::mutex> g(qm);
            for (int c = 1; c < nt; ++c)
                chunk_jobs.push_back({[](void* ctx, int c_, int32_t cb, int32_t ce) { (*static_cast<Fn*>(ctx))(c_, cb, ce); }, (void*)&f, c,
                                      b + (int32_t)(len * c / nt), b + (int32_t)(len * (c + 1) / nt), &left});
        }
        qcv.notify_all();
        f(0, b, b + (int32_t)(len / nt));
        for (;;) {
            ChunkJob j;
            {
                std::lock_guard<std::mutex> g(qm);
                if (chunk_jobs.empty()) break;
                j = chunk_jobs.back();
                chunk_jobs.pop_back();
            }
            run_chunk(j);
        }
        while (left.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    }
    void work() {
        for (;;) {
            Task t;
            ChunkJob j;
            bool is_chunk = false;
            {
                std::unique_lock<std::mutex> lk(qm);
                qcv.wait(lk, [&] { return failed || !chunk_jobs.empty() || !tasks.empty() || (done_top && outstanding == 0); });
                if (!chunk_jobs.empty()) {
                    j = chunk_jobs.back();
                    chunk_jobs.pop_back();
                    is_chunk = true;
                } else {
                    if (failed || tasks.empty()) return;
                    size_t pick = 0;
                    for (size_t i = 1; i < tasks.size(); ++i)
                        if (tasks[i].e - tasks[i].b > tasks[pick].e - tasks[pick].b) pick = i;
                    t = tasks[pick];
                    tasks[pick] = tasks.back();
                    tasks.pop_back();
                }
            }
            if (is_chunk) { run_chunk(j); continue; }
            try {
                *t.link = build(t.b, t.e, t.depth, 1, t.in, t.bx);
            } catch (...) {
                std::lock_guard<std::mutex> g(qm);
                if (!failed) { failed = true; failure = std::current_exception(); }
            }
            bool wake;
            {
                std::lock_guard<std::mutex> g(qm);
                --outstanding;
                wake = failed || (done_top && outstanding == 0);
            }
            if (wake) qcv.notify_all();
        }
    }
    bool done_top = false;  // the top of the tree is built: no task comes from there any more
    int32_t build_all(int32_t n, int threads) {
        const Box all = range_box(0, n, threads);
        if (threads <= 1) return build(0, n, 0, 1, 0, all);
        pool_on = true;
        grain = std::max<int32_t>(1024, n / (threads * 8));
        const auto t_begin = std::chrono::steady_clock::now();
        int32_t root = -1;
        {
            ThreadGroup group;
            for (int c = 1; c < threads; ++c) group.run([this] { work(); });
            try {
                root = build(0, n, 0, threads, 0, all);
            } catch (...) {
                std::lock_guard<std::mutex> g(qm);
                if (!failed) { failed = true; failure = std::current_exception(); }
            }
            {
                std::lock_guard<std::mutex> g(qm);
                done_top = true;
            }
            qcv.notify_all();
            if (std::getenv("RTGO_TIMING"))
                std::fprintf(stderr, "[rtgo]   build_sah: top of the tree done %.2f ms after the split began\n",
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
            work();
            group.join();
        }
        if (failed) std::rethrow_exception(failure);
        return root;
    }
    // ---- widening: which Bvh2 nodes become the up-to-8 children of a wide node, and in which slots ----
    struct Wide { int32_t slot[8]; };
    std::unique_ptr<Wide[]> wide;  // by Bvh2 node; filled for the inner nodes that end up as wide nodes
    void choose_slots(int32_t n) {
        const Bvh2& nd = b2[n];
        int32_t kids[8]; int nk = 2;
        kids[0] = nd.left; kids[1] = nd.right;
        while (nk < 8) {
            int pick = -1; double pa = -1;
            for (int i = 0; i < nk; ++i) {
                if (b2[kids[i]].left < 0) continue;
                const double ar = area3(b2[kids[i]].lo, b2[kids[i]].hi);
                if (ar > pa) { pa = ar; pick = i; }
            }
            if (pick < 0) break;
            const int32_t k = kids[pick];
            kids[pick] = b2[k].left;
            kids[nk++] = b2[k].right;
        }
        double m[3];
        for (int a = 0; a < 3; ++a) m[a] = (nd.lo[a] + nd.hi[a]) * 0.5;
        int32_t* slot = wide[n].slot;
        for (int sl = 0; sl < 8; ++sl) slot[sl] = -1;
        bool placed[8] = {};
        double score[8][8];  // child i in slot sl
        for (int i = 0; i < nk; ++i) {
            const Bvh2& c = b2[kids[i]];
            for (int sl = 0; sl < 8; ++sl) {
                double sc = 0;
                for (int a = 0; a < 3; ++a) sc += ((sl >> a) & 1 ? 1.0 : -1.0) * ((c.lo[a] + c.hi[a]) * 0.5 - m[a]);
                score[i][sl] = sc;
            }
        }
        for (int round = 0; round < nk; ++round) {
            int bi = -1, bs = -1; double bsc = 0;
            for (int i = 0; i < nk; ++i) {
                if (placed[i]) continue;
                for (int sl = 0; sl < 8; ++sl) {
                    if (slot[sl] >= 0) continue;
                    if (bi < 0 || score[i][sl] > bsc) { bi = i; bs = sl; bsc = score[i][sl]; }
                }
            }
            slot[bs] = kids[bi]; placed[bi] = true;
        }
    }
    // slots of every wide node below n, and span[n] = the HostNodes emit() writes below n (8 per wide node and per
    // leaf); subtrees of at most `grain` shapes are left to `later` (the threads) and `top` lists the others
    std::unique_ptr<int32_t[]> span;
    int32_t widen(int32_t n, int32_t grain, std::vector<int32_t>* later, std::vector<int32_t>* top) {
        if (b2[n].left < 0) return span[n] = 8;
        if (later && b2[n].e - b2[n].b <= grain) { later->push_back(n); return 0; }
        if (top) top->push_back(n);
        choose_slots(n);
        int32_t sp = 8;
        for (int sl = 0; sl < 8; ++sl)
            if (wide[n].slot[sl] >= 0) sp += widen(wide[n].slot[sl], grain, later, top);
        return span[n] = sp;
    }
    void widen_all(int threads) {
        const int32_t total = n_nodes.load();
        wide.reset(new Wide[(size_t)total]);
        span.reset(new int32_t[(size_t)total]);
        const int32_t n_items = b2[0].e - b2[0].b;
        if (threads <= 1 || n_items < kChunkMin) { widen(0, 0, nullptr, nullptr); return; }
        std::vector<int32_t> later, top;
        widen(0, n_items / (threads * 8) + 1, &later, &top);
        run_jobs(threads, later.size(), [&](size_t i) { widen(later[i], 0, nullptr, nullptr); });
        for (size_t i = top.size(); i-- > 0;) {  // children come after their parents in `top`
            const int32_t n = top[i];
            int32_t sp = 8;
            for (int sl = 0; sl < 8; ++sl)
                if (wide[n].slot[sl] >= 0) sp += span[wide[n].slot[sl]];
            span[n] = sp;
        }
    }
    template <class F>
    static void run_jobs(int threads, size_t n_jobs, F&& f) {
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i = next.fetch_add(1); i < n_jobs; i = next.fetch_add(1)) f(i);
        };
        ThreadGroup group;
        for (int c = 1; c < threads; ++c) group.run(work);
        work();
        group.join();
    }
    // wide node for Bvh2 node `n` written into HostNode `at`, its 8 children from `base` on and their subtrees behind
    // them in slot order (so a subtree's place follows from the spans before it, and subtrees can be written by
    // different threads: those of at most `grain` shapes go to `jobs`)
    struct EmitJob { int32_t n, at, base; };
    void emit(HostNode* out, int32_t n, int32_t at, int32_t base, int32_t grain, std::vector<EmitJob>* jobs) const {
        const Bvh2& nd = b2[n];
        HostNode me;
        for (int a = 0; a < 3; ++a) { me.lo[a] = nd.lo[a]; me.hi[a] = nd.hi[a]; }
        me.empty = false;
        me.n_shapes = nd.e - nd.b;
        me.shape = -1;
        me.child_base = base;
        out[at] = me;
        if (nd.left < 0) {  // leaf: one single-shape child per shape
            for (int32_t k = 0; k < 8; ++k) {
                HostNode c;
                if (k < nd.e - nd.b) {
                    const SahItem& it = buf[0][(size_t)(nd.b + k)];
                    for (int a = 0; a < 3; ++a) { c.lo[a] = it.lo[a]; c.hi[a] = it.hi[a]; }
                    c.shape = it.shape; c.empty = false; c.n_shapes = 1;
                } else {
                    for (int a = 0; a < 3; ++a) c.lo[a] = c.hi[a] = 0;
                }
                out[base + k] = c;
            }
            return;
        }
        int32_t next = base + 8;
        for (int sl = 0; sl < 8; ++sl) {
            const int32_t c = wide[n].slot[sl];
            if (c < 0) {
                HostNode none;
                for (int a = 0; a < 3; ++a) none.lo[a] = none.hi[a] = 0;
                out[base + sl] = none;
                continue;
            }
            if (jobs && b2[c].e - b2[c].b <= grain) jobs->push_back({c, base + sl, next});
            else emit(out, c, base + sl, next, grain, jobs);
            next += span[c];
        }
    }
    void emit_all(HostNodes& out, int threads) const {
        out.resize(1 + (size_t)span[0]);
        const int32_t n_items = b2[0].e - b2[0].b;
        if (threads <= 1 || n_items < kChunkMin) { emit(out.data(), 0, 0, 1, 0, nullptr); return; }
        std::vector<EmitJob> jobs;
        emit(out.data(), 0, 0, 1, n_items / (threads * 8) + 1, &jobs);
        run_jobs(threads, jobs.size(), [&](size_t i) { emit(out.data(), jobs[i].n, jobs[i].at, jobs[i].base, 0, nullptr); });
    }
};
}  // namespace

HostNodes build_sah(const rt_shape* shapes, int64_t n, int leaf_size) {
    HostNodes out;
    if (n == 0) { out.push_back(HostNode{}); return out; }
    SahBuilder sb;
    sb.leaf_size = leaf_size < 1 ? 1 : (leaf_size > 8 ? 8 : leaf_size);
    // threads building at one time: RTGO_BUILD_THREADS, by default the hardware's but at most 16 (one GPU's
    // share of a node's cores)
    const int threads = builder_threads();
    const bool timing = std::getenv("RTGO_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rtgo]   build_sah: %-10s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    sb.buf[0].reset(new SahItem[(size_t)n]);  // not zeroed: the threads that fill the arrays touch their pages first
    sb.buf[1].reset(new SahItem[(size_t)n]);
    SahBuilder::chunks(0, (int32_t)n, n >= kChunkMin ? threads : 1, [&](int, int32_t cb, int32_t ce) {
        for (int32_t i = cb; i < ce; ++i) {
            SahItem& it = sb.buf[0][(size_t)i];
            shape_bounds(shapes[i], it.lo, it.hi);
            for (int a = 0; a < 3; ++a) { it.lo[a] += 0.0; it.hi[a] += 0.0; }  // -0 -> +0 (see SahBuilder::mn)
            it.shape = i;
        }
    });
    lap("items");
    sb.b2.reset(new Bvh2[2 * (size_t)n]);
    sb.build_all((int32_t)n, threads);
    lap("split");
    sb.widen_all(threads);
    lap("widen");
    sb.emit_all(out, threads);
    lap("emit");
    return out;
}

// depth-first, octant order
static void collect_shapes(const HostNodes& nodes, int32_t n, std::vector<int32_t>* out) {
    if (nodes[n].child_base < 0) {
        if (nodes[n].shape >= 0) out->push_back(nodes[n].shape);
        return;
    }
    for (int o = 0; o < 8; ++o)
        if (!nodes[nodes[n].child_base + o].empty) collect_shapes(nodes, nodes[n].child_base + o, out);
}

// The up-to-8 children of the block rooted at effective inner node `x`, by slot.
//   1. slot o = the (chain-skipped) child in octant o;
//   2. while some inner child e (more than leaf_size shapes) can be replaced by its own children without
//      exceeding 8 slots, replace the one with the largest box area (first in slot order on ties): its
//      slot is freed and its children go to free slots by greedy matching on
//      score(child, slot) = sum over axes of (+/-)(child box centre - block box centre), + where the
//      slot's octant bit is set; largest score first, ties to the lower child octant, then lower slot.
// A block visit costs the wavefront the same whether 3 or 8 slots are filled, so fuller blocks mean
// fewer visits per ray; slot ~ octant keeps the k ^ sign visiting order roughly front to back.
// The order only decides which of two EQUAL-t hits is found first; every box still encloses its shapes.
static int n_children(const HostNodes& nodes, int32_t e) {
    int n = 0;
    for (int o = 0; o < 8; ++o) n += nodes[nodes[e].child_base + o].empty ? 0 : 1;
    return n;
}
static double box_area(const HostNode& n) {
    const double dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
    return 2.0 * ((dx * dy + dy * dz) + dz * dx);
}
// Which subtrees become ONE leaf: at most leaf_size shapes, and either a small box or a tight one.  A leaf's shapes
// are all tested by every ray that meets the leaf's box; for a handful of small neighbouring triangles that is cheaper
// than another block visit, but NOT for shapes that only a builder's bookkeeping put together: the 6 largest triangles
// of a box room ended up in one leaf whose box was the whole room, and every shadow ray of config 3 paid 6 of its 15
// shape tests there (scripts/sim_groupwalk.cpp sim_big_shapes).  So a subtree whose box has at least 1/64 of the area
// of the scene's box becomes a leaf only if that box is hardly larger (x 1.5 in area) than the largest box of a single
// shape in it — two triangles of one wall — and is otherwise kept as a block of smaller leaves, whose flat boxes
// nearly every ray misses.  Mirrored by the oracle (rtgo_oracle.cpp mark_leaves).
constexpr int kTightLeafMax = 6;
struct LeafRule {
    int leaf_size;
    double root_area;
    bool collapse;  // step 2 of block_slots (off: RTGO_NO_COLLAPSE, an A/B switch for measurements; the oracle mirror has the same one)
};
static double max_shape_area(const HostNodes& nodes, int32_t n) {
    if (nodes[n].child_base < 0) return nodes[n].shape >= 0 ? box_area(nodes[n]) : 0.0;
    double m = 0;
    for (int o = 0; o < 8; ++o)
        if (!nodes[nodes[n].child_base + o].empty) m = std::max(m, max_shape_area(nodes, nodes[n].child_base + o));
    return m;
}
static bool is_leaf(const HostNodes& nodes, int32_t e, const LeafRule& rule) {
    if (nodes[e].n_shapes > rule.leaf_size) return false;
    if (nodes[e].n_shapes <= 1 || nodes[e].child_base < 0) return true;
    const double a = box_area(nodes[e]);
    if (a * 64.0 < rule.root_area) return true;
    // (a tight box makes a leaf of a few large shapes, two triangles of a wall; the many shapes of a wide leaf — leaf
    // sizes above 6 pay because a leaf's slab, device_math.hpp, rejects nearly every ray that meets its box and not its
    // surface — must be a small patch: the 11 shapes of the reference's example scene, spheres over a floor whose box is
    // the scene's, would otherwise be ONE leaf that every ray and every pair's beam meets)
    return nodes[e].n_shapes <= kTightLeafMax && a <= 1.5 * max_shape_area(nodes, e);
}

static void block_slots(const HostNodes& nodes, int32_t x, const LeafRule& rule, int32_t slot[8]) {
    int n = 0;
    for (int o = 0; o < 8; ++o) {
        const int32_t c = nodes[x].child_base + o;
        slot[o] = nodes[c].empty ? -1 : skip_chain(nodes, c);
        n += slot[o] >= 0;
    }
    if (!rule.collapse) return;
    double m[3];
    for (int a = 0; a < 3; ++a) m[a] = (nodes[x].lo[a] + nodes[x].hi[a]) * 0.5;
    for (;;) {
        int pick = -1, pick_k = 0;
        double pick_area = 0;
        for (int s = 0; s < 8; ++s) {
            const int32_t e = slot[s];
            if (e < 0 || is_leaf(nodes, e, rule)) continue;
            const int k = n_children(nodes, e);
            if (n - 1 + k > 8) continue;
            const double a = box_area(nodes[e]);
            if (pick < 0 || a > pick_area) { pick = s; pick_area = a; pick_k = k; }
        }
        if (pick < 0) break;
        const int32_t e = slot[pick];
        slot[pick] = -1;
        int32_t kids[8]; double ctr[8][3]; bool placed[8];
        int nk = 0;
        for (int o = 0; o < 8; ++o) {
            const int32_t c = nodes[e].child_base + o;
            if (nodes[c].empty) continue;
            kids[nk] = skip_chain(nodes, c);
            for (int a = 0; a < 3; ++a) ctr[nk][a] = (nodes[kids[nk]].lo[a] + nodes[kids[nk]].hi[a]) * 0.5;
            placed[nk] = false;
            ++nk;
        }
        for (int round = 0; round < nk; ++round) {
            int bi = -1, bs = -1;
            double bscore = 0;
            for (int i = 0; i < nk; ++i) {
                if (placed[i]) continue;
                for (int sl = 0; sl < 8; ++sl) {
                    if (slot[sl] >= 0) continue;
                    const double sx = (sl & 1) ? 1.0 : -1.0, sy = (sl & 2) ? 1.0 : -1.0, sz = (sl & 4) ? 1.0 : -1.0;
                    const double sc = (sx * (ctr[i][0] - m[0]) + sy * (ctr[i][1] - m[1])) + sz * (ctr[i][2] - m[2]);
                    if (bi < 0 || sc > bscore) { bi = i; bs = sl; bscore = sc; }
                }
            }
            slot[bs] = kids[bi];
            placed[bi] = true;
        }
        n += pick_k - 1;
    }
}

// The block whose child records are rec[0, count) (slot order, absent slots omitted), as a QBlock written to w.
// (The records' boxes are padded already: flatten_octree.)
static void write_qblock(const ChildRecord* rec, int count, uint32_t present, uint32_t leaf_mask, uint32_t child_base, uint32_t* w) {
    for (int i = 0; i < kQBlockWords; ++i) w[i] = 0;
    float origin[3], top[3];
    for (int a = 0; a < 3; ++a) {
        origin[a] = rec[0].bmin[a]; top[a] = rec[0].bmax[a];
        for (int k = 1; k < count; ++k) { origin[a] = std::min(origin[a], rec[k].bmin[a]); top[a] = std::max(top[a], rec[k].bmax[a]); }
    }
    uint32_t exps = 0;
    double step[3];
    for (int a = 0; a < 3; ++a) {
        const double ext = (double)top[a] - (double)origin[a];
        int e = 1;  // biased exponent of the grid step: the smallest with ceil(ext / step) <= 255
        if (ext > 0) {
            int x;
            (void)std::frexp(ext / 255.0, &x);  // ext / 255 = m * 2^x, 0.5 <= m < 1: step 2^x is enough
            e = std::max(1, std::min(254, x + 127));
            while (e < 254 && std::ceil(ext / std::ldexp(1.0, e - 127)) > 255.0) ++e;
        }
        step[a] = std::ldexp(1.0, e - 127);
        exps |= (uint32_t)e << (8 * a);
    }
    // The grid's origin may lie up to a step below the block's box.  It is placed so that a THIN child on the block's lowest
    // face (else one on its highest) has its inner face — the one a ray that starts on it leaves through — on a grid plane:
    // a flat leaf (a floor, a wall: as thick as its pad) is a whole step thick either way, but now towards the outside,
    // and the shadow rays that start on it are out of its box before tMin instead of entering it (config 1, whose rays
    // mostly start on the floor: 9.6 -> 8.8 ms per frame; config 3: 209.0 -> 207.8).
    for (int a = 0; a < 3; ++a) {
        const double s = step[a];
        float cand = origin[a];
        int best = -1;
        for (int k = 0; k < count; ++k)
            if (rec[k].bmin[a] == origin[a] && (best < 0 || rec[k].bmax[a] < rec[best].bmax[a])) best = k;
        if (best >= 0 && (double)rec[best].bmax[a] - (double)origin[a] < s) {
            cand = round_up_f32((double)rec[best].bmax[a] - s);  // its upper face = origin + 1 step (or a hair below)
        } else {
            best = -1;
            for (int k = 0; k < count; ++k)
                if (rec[k].bmax[a] == top[a] && (best < 0 || rec[k].bmin[a] > rec[best].bmin[a])) best = k;
            if (best >= 0 && (double)top[a] - (double)rec[best].bmin[a] < s) {
                const double kk = std::ceil(((double)rec[best].bmin[a] - (double)origin[a]) / s);
                cand = round_down_f32((double)rec[best].bmin[a] - kk * s);  // its lower face = origin + kk steps (or a hair above)
            }
        }
        if (cand < origin[a] && std::isfinite(cand) && (double)origin[a] - (double)cand <= s && std::ceil(((double)top[a] - (double)cand) / s) <= 255.0)
            origin[a] = cand;
    }
    for (int a = 0; a < 3; ++a) std::memcpy(&w[a], &origin[a], 4);
    w[3] = exps | (present << 24);
    w[4] = child_base | (leaf_mask << 24);
    uint8_t* meta = reinterpret_cast<uint8_t*>(&w[6]);
    uint8_t* qlo = reinterpret_cast<uint8_t*>(&w[8]);
    uint8_t* qhi = reinterpret_cast<uint8_t*>(&w[14]);
    int k = 0, rank = 0;
    bool have_leaf = false;
    for (int o = 0; o < 8; ++o) {
        if (!((present >> o) & 1u)) continue;
        const ChildRecord& r = rec[k++];
        if ((leaf_mask >> o) & 1u) {
            if (!have_leaf) { w[5] = r.w0 >> 8; have_leaf = true; }
            meta[o] = (uint8_t)(r.w0 & 0xffu);
        } else {
            meta[o] = (uint8_t)(rank++ << 5);
        }
        for (int a = 0; a < 3; ++a) {
            const double lo = std::floor(((double)r.bmin[a] - (double)origin[a]) / step[a]);
            const double hi = std::ceil(((double)r.bmax[a] - (double)origin[a]) / step[a]);
            qlo[8 * a + o] = (uint8_t)std::max(0.0, std::min(255.0, lo));
            qhi[8 * a + o] = (uint8_t)std::max(0.0, std::min(255.0, hi));
        }
    }
}

int builder_threads() {
    int threads = (int)std::thread::hardware_concurrency();
    if (threads > 16) threads = 16;
    if (const char* ev = std::getenv("RTGO_BUILD_THREADS")) threads = std::atoi(ev);
    return threads < 1 ? 1 : (threads > 32 ? 32 : threads);
}

// Blocks are numbered breadth-first, and so are their records and the leaves' shapes.  One level at a time: the
// level's blocks are dealt to the threads in contiguous runs; every thread works its run's blocks out into arrays of
// its own (slots, leaves, records with run-local indices, the next level's blocks); the runs' lengths are summed up
// in order; then every thread moves its run to its place in the tree.  The result does not depend on the runs.
bool flatten_octree(const HostNodes& nodes, int leaf_size, FlatTree* out, const char** err, double origin_reach) {
    out->records.clear();
    out->qblocks.clear();
    out->order.clear();
    out->block_first.assign(1, 0u);
    out->block_box.clear();
    out->root_ref = 0; out->n_inner = 0; out->depth = 0; out->stack_bound = 1; out->qpad = 0.0;
    if (nodes.empty() || nodes[0].empty) {  // no shapes: present mask 0 => every ray misses
        out->qblocks.assign(kQBlockWords, 0u);
        out->block_box.assign(8, 0.0f);
        return true;
    }
    if (leaf_size < 1) leaf_size = 1;
    if (leaf_size > kMaxLeafSize) leaf_size = kMaxLeafSize;

    // Every box of the flat tree is the node's box grown by `qpad` = 2^-19 of the scene's largest coordinate M (over the
    // shapes' box and the camera's lens, where the primary rays start: origin_reach), then rounded outward to float32.  The device's slab test t = fma(plane, 1 / d, -(o / d)) rounds o / d before the subtraction and
    // so places a plane up to 2^-24 (|o| + 2 |plane - o|) off — 5 M 2^-24 at most for origins and boxes within M of the
    // world's origin (a secondary ray starts on a shape; the quantised form of the shadow rays' walk, fma(q, step / d,
    // fma(origin, 1 / d, -(o / d))), adds |origin - o|: 7 M 2^-24) — while a shape's own test, which subtracts a vertex or
    // a centre from the ray's origin first, errs by a few 2^-24 of the distance between the two.  Far from the world's
    // origin a ray could therefore miss a leaf's box and still pass the test of a shape inside it, and the answer
    // depended on the path: scripts/stress_slabs.py, seed 10050 (flat sheets 5,000 units out) — a shadow ray through the
    // shared face of two leaves missed both boxes from the root and hit the triangle when its walk resumed below them;
    // and the same rays through the SAH tree with 20- and 4-shape leaves and through the octree: 1 answer in 1,000
    // differed at 5,000 units.  With every plane moved outward by more than both errors each axis' computed interval
    // holds the true one: a walk reaches every shape whose own test the ray can pass, wherever it starts and whatever
    // the tree — the answer is the OR (the minimum) over the shapes' tests.  The oracle's float32 mode grows its boxes
    // the same way (fill_f32_boxes).
    // The price is not the size of the pad but that it is not zero: a flat, axis-aligned leaf (a floor, a wall) is now one
    // grid step of its quantised block thick, and the shadow rays that START on it entered it (config 3: + 2 % shape
    // loops, + 0.4 % frame time, whatever the pad) until write_qblock placed the grids so that the step goes outwards.
    const int32_t root = skip_chain(nodes, 0);
    double qpad = std::isfinite(origin_reach) ? std::fabs(origin_reach) : 0.0;
    for (int a = 0; a < 3; ++a) qpad = std::max(qpad, std::max(std::fabs(nodes[root].lo[a]), std::fabs(nodes[root].hi[a])));
    qpad = std::isfinite(qpad) ? qpad * 1.9073486328125e-06 : 0.0;
    out->qpad = qpad;
    auto set_box = [&](int32_t e, ChildRecord* r) {
        for (int a = 0; a < 3; ++a) { r->bmin[a] = round_down_f32(nodes[e].lo[a] - qpad); r->bmax[a] = round_up_f32(nodes[e].hi[a] + qpad); }
    };
    const LeafRule rule{leaf_size, box_area(nodes[root]), std::getenv("RTGO_NO_COLLAPSE") == nullptr};
    if (is_leaf(nodes, root, rule)) {
        // everything fits one leaf: synthetic root block with that leaf in octant 0
        ChildRecord r;
        set_box(root, &r);
        collect_shapes(nodes, root, &out->order);
        r.w0 = (0u << 8) | (uint32_t)out->order.size();
        r.w1 = kLeafFlag | 1u;
        out->records.push_back(r);
        out->root_ref = (0u << 8) | 1u;
        out->n_inner = 1; out->depth = 1; out->stack_bound = 8;
        out->qblocks.resize(kQBlockWords);
        write_qblock(out->records.data(), 1, 1u, 1u, 0u, out->qblocks.data());
        out->block_first.assign(1, 0u);
        out->block_box.assign(8, 0.0f);
        for (int a = 0; a < 3; ++a) { out->block_box[(size_t)a] = r.bmin[a]; out->block_box[(size_t)(4 + a)] = r.bmax[a]; }
        return true;
    }

    struct Pending { int32_t node; int64_t patch; };  // patch = record whose w0 names this block
    struct BlockTmp { int32_t first; uint8_t count, present, leaf_mask, inner; int32_t inner_before; };
    struct Run {  // of a level's blocks
        size_t b, e;
        std::vector<ChildRecord> recs;
        std::vector<int32_t> order;
        std::vector<Pending> next;
        std::vector<BlockTmp> blocks;
        int64_t rec_base, order_base, pushed_base;
    };
    struct BlockInfo { uint32_t child_base; int32_t inner; };
    std::vector<BlockInfo> info;  // by block
    const int threads = builder_threads();
    std::vector<Pending> level{{root, -1}};
    int depth = 0;
    while (!level.empty()) {
        ++depth;
        const size_t n_blocks_before = info.size();
        const int nt = (int)std::min<size_t>((size_t)threads, level.size() / 256 + 1);
        std::vector<Run> runs((size_t)nt);
        for (int c = 0; c < nt; ++c) { runs[(size_t)c].b = level.size() * (size_t)c / (size_t)nt; runs[(size_t)c].e = level.size() * (size_t)(c + 1) / (size_t)nt; }
        auto on_runs = [&](auto&& f) {
            if (nt == 1) { f(runs[0]); return; }
            ThreadGroup group;
            for (int c = 1; c < nt; ++c) group.run([&f, &runs, c] { f(runs[(size_t)c]); });
            f(runs[0]);
            group.join();
        };
        on_runs([&](Run& run) {
            int32_t inner_before = 0;
            for (size_t i = run.b; i < run.e; ++i) {
                int32_t slot[8];
                block_slots(nodes, level[i].node, rule, slot);
                BlockTmp bt{(int32_t)run.recs.size(), 0, 0, 0, 0, inner_before};
                for (int o = 0; o < 8; ++o) {
                    const int32_t e = slot[o];
                    if (e < 0) continue;
                    bt.present |= (uint8_t)(1u << o);
                    ChildRecord r;
                    set_box(e, &r);
                    if (is_leaf(nodes, e, rule)) {
                        const uint32_t first = (uint32_t)run.order.size();
                        collect_shapes(nodes, e, &run.order);
                        r.w0 = (first << 8) | (uint32_t)(run.order.size() - first);
                        r.w1 = kLeafFlag;
                        bt.leaf_mask |= (uint8_t)(1u << o);
                    } else {
                        r.w0 = 0; r.w1 = 0;
                        run.next.push_back({e, (int64_t)run.recs.size()});
                        ++bt.inner;
                    }
                    run.recs.push_back(r);
                }
                bt.count = (uint8_t)(run.recs.size() - (size_t)bt.first);
                inner_before += bt.inner;
                run.blocks.push_back(bt);
            }
        });
        int64_t n_rec = (int64_t)out->records.size(), n_order = (int64_t)out->order.size();
        int64_t n_pushed = (int64_t)(n_blocks_before + level.size());  // the i-th block found is block i (root = 0)
        size_t n_next = 0;
        for (Run& run : runs) {
            run.rec_base = n_rec; run.order_base = n_order; run.pushed_base = n_pushed;
            n_rec += (int64_t)run.recs.size(); n_order += (int64_t)run.order.size(); n_pushed += (int64_t)run.next.size();
            n_next += run.next.size();
        }
        if (n_rec > kMaxChildRecords || n_order >= kMaxChildRecords) {
            *err = "scene too large: the flattened octree exceeds 2^24 child records";
            return false;
        }
        out->records.resize((size_t)n_rec);
        out->order.resize((size_t)n_order);
        info.resize(n_blocks_before + level.size());
        out->qblocks.resize(info.size() * kQBlockWords);
        out->block_first.resize(info.size());
        out->block_box.resize(info.size() * 8);
        std::vector<Pending> next(n_next);
        on_runs([&](Run& run) {
            std::copy(run.order.begin(), run.order.end(), out->order.begin() + run.order_base);
            ChildRecord* recs = out->records.data() + run.rec_base;
            for (size_t k = 0; k < run.recs.size(); ++k) {
                recs[k] = run.recs[k];
                if (recs[k].w1 & kLeafFlag) recs[k].w0 += (uint32_t)run.order_base << 8;
            }
            size_t at = (size_t)(run.pushed_base - (int64_t)(n_blocks_before + level.size()));
            for (const Pending& p : run.next) next[at++] = {p.node, run.rec_base + p.patch};
            for (size_t j = 0; j < run.blocks.size(); ++j) {
                const BlockTmp& bt = run.blocks[j];
                const size_t block = n_blocks_before + run.b + j;
                ChildRecord* rec = recs + bt.first;
                for (int k = 0; k < bt.count; ++k) rec[k].w1 |= bt.leaf_mask;
                const uint32_t child_base = bt.inner ? (uint32_t)(run.pushed_base + bt.inner_before) : 0u;  // its first inner child's block
                write_qblock(rec, bt.count, bt.present, bt.leaf_mask, child_base, out->qblocks.data() + block * kQBlockWords);
                const uint32_t first = (uint32_t)(run.rec_base + bt.first);
                const uint32_t ref = (first << 8) | bt.present;
                const int64_t patch = level[run.b + j].patch;  // a record of the level above: final, and named by this block alone
                if (patch < 0) out->root_ref = ref; else out->records[(size_t)patch].w0 = ref;
                out->block_first[block] = first;
                float* bb = out->block_box.data() + block * 8;
                bb[3] = bb[7] = 0.0f;
                for (int a = 0; a < 3; ++a) {
                    bb[a] = rec[0].bmin[a]; bb[4 + a] = rec[0].bmax[a];
                    for (int k = 1; k < bt.count; ++k) { bb[a] = std::min(bb[a], rec[k].bmin[a]); bb[4 + a] = std::max(bb[4 + a], rec[k].bmax[a]); }
                }
                info[block] = {child_base, bt.inner};
            }
        });
        level.swap(next);
    }
    out->n_inner = (int64_t)info.size();
    out->depth = depth;
    // Worst-case traversal stack: entering a block pushes at most (inner children - 1) entries and descends
    // into one inner child, so the deepest stack below block b is need(b) = (inner(b) - 1) + max over its
    // inner children of need(child).  Blocks are in breadth-first order: children come after their parent, and a
    // block's inner children are consecutive blocks.
    {
        std::vector<int32_t> need(info.size(), 0);
        for (int64_t i = (int64_t)info.size() - 1; i >= 0; --i) {
            const BlockInfo& bi = info[(size_t)i];
            int deepest = 0;
            for (int k = 0; k < bi.inner; ++k) deepest = std::max(deepest, need[(size_t)bi.child_base + (size_t)k]);
            need[(size_t)i] = (bi.inner > 0 ? bi.inner - 1 : 0) + deepest;
        }
        out->stack_bound = (need.empty() ? 0 : need[0]) + 1;
    }
    return true;
}

}  // namespace rtgo
