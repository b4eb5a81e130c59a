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
void split_cell(std::vector<HostNode>& nodes, int32_t at) {
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
int32_t skip_chain(const std::vector<HostNode>& nodes, int32_t n) {
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
std::vector<HostNode> build_octree(const rt_shape* shapes, int64_t n) {
    std::vector<HostNode> nodes;
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
struct SahItem { double lo[3], hi[3], c[3]; int32_t shape; };
struct Bvh2 { double lo[3], hi[3]; int32_t left, right, b, e; };
double area3(const double lo[3], const double hi[3]) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0 * ((dx * dy + dy * dz) + dz * dx);
}
constexpr int kMaxSahDepth = 48;
constexpr int32_t kParallelMin = 16384;  // ranges at least this long may be split off to a thread of their own
constexpr int32_t kChunkMin = 131072;    // ranges at least this long are swept by several threads
struct SahBuilder {
    std::vector<SahItem> items;
    std::unique_ptr<SahItem[]> tmp;    // same length as items: a range's right half while it is being split
    std::unique_ptr<Bvh2[]> b2;        // 2n - 1 nodes at most (every leaf holds a shape); numbered as they are made
    std::atomic<int32_t> n_nodes{0};
    int leaf_size;
    // depth: SAH may split 1 : n-1 again and again on adversarial input; beyond kMaxSahDepth levels the
    // range is halved instead, so the recursion stays O(kMaxSahDepth + log n) deep.
    // threads: how many threads this subtree may use; a split shares them out by the halves' lengths.  The
    // two halves of a range touch disjoint items and only the node numbers depend on the interleaving;
    // emit() follows left/right links, so the tree that comes out is the same for any number of threads.
    static constexpr int kBins = 16;
    struct Bin { double lo[3], hi[3]; int n; };
    struct Pass {  // what one chunk of a range contributes to the passes over it
        double lo[3], hi[3], clo[3], chi[3];
        Bin bins[3][kBins];
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
    int32_t build(int32_t b, int32_t e, int depth, int threads) {
        // a long range is swept by all the threads this subtree may use; min, max and counts merge exactly,
        // so the chunking changes no value
        const int nt = (e - b >= kChunkMin) ? threads : 1;
        Pass one;
        std::vector<Pass> many;
        Pass* ps = &one;
        if (nt > 1) { many.resize((size_t)nt); ps = many.data(); }

        chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {
            Pass& p = ps[c];
            for (int a = 0; a < 3; ++a) { p.lo[a] = p.clo[a] = DBL_MAX; p.hi[a] = p.chi[a] = -DBL_MAX; }
            for (int32_t i = cb; i < ce; ++i)
                for (int a = 0; a < 3; ++a) {
                    p.lo[a] = std::fmin(p.lo[a], items[i].lo[a]); p.hi[a] = std::fmax(p.hi[a], items[i].hi[a]);
                    p.clo[a] = std::fmin(p.clo[a], items[i].c[a]); p.chi[a] = std::fmax(p.chi[a], items[i].c[a]);
                }
        });
        Bvh2 nd;
        nd.b = b; nd.e = e; nd.left = nd.right = -1;
        double clo[3], chi[3];
        for (int a = 0; a < 3; ++a) { nd.lo[a] = clo[a] = DBL_MAX; nd.hi[a] = chi[a] = -DBL_MAX; }
        for (int c = 0; c < nt; ++c)
            for (int a = 0; a < 3; ++a) {
                nd.lo[a] = std::fmin(nd.lo[a], ps[c].lo[a]); nd.hi[a] = std::fmax(nd.hi[a], ps[c].hi[a]);
                clo[a] = std::fmin(clo[a], ps[c].clo[a]); chi[a] = std::fmax(chi[a], ps[c].chi[a]);
            }
        const int32_t me = n_nodes.fetch_add(1, std::memory_order_relaxed);
        b2[me] = nd;
        if (e - b <= leaf_size) return me;

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
            chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {  // the three axes' bins in one sweep
                Pass& p = ps[c];
                for (int a = 0; a < 3; ++a)
                    for (auto& bn : p.bins[a]) { bn.n = 0; for (int k = 0; k < 3; ++k) { bn.lo[k] = DBL_MAX; bn.hi[k] = -DBL_MAX; } }
                for (int32_t i = cb; i < ce; ++i)
                    for (int a = 0; a < 3; ++a) {
                        if (!use[a]) continue;
                        int k = (int)((items[i].c[a] - clo[a]) * scale[a]);  // coordinates are finite (checked at scene creation)
                        if (k >= kBins) k = kBins - 1;
                        if (k < 0) k = 0;
                        Bin& bn = p.bins[a][k];
                        bn.n++;
                        for (int q = 0; q < 3; ++q) { bn.lo[q] = std::fmin(bn.lo[q], items[i].lo[q]); bn.hi[q] = std::fmax(bn.hi[q], items[i].hi[q]); }
                    }
            });
            for (int a = 0; a < 3; ++a) {
                if (!use[a]) continue;
                Bin* bins = ps[0].bins[a];
                for (int c = 1; c < nt; ++c)
                    for (int k = 0; k < kBins; ++k) {
                        const Bin& o = ps[c].bins[a][k];
                        bins[k].n += o.n;
                        for (int q = 0; q < 3; ++q) { bins[k].lo[q] = std::fmin(bins[k].lo[q], o.lo[q]); bins[k].hi[q] = std::fmax(bins[k].hi[q], o.hi[q]); }
                    }
                double ra[kBins]; int rn[kBins];
                double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
                int cnt = 0;
                for (int k = kBins - 1; k >= 1; --k) {
                    for (int q = 0; q < 3; ++q) { lo[q] = std::fmin(lo[q], bins[k].lo[q]); hi[q] = std::fmax(hi[q], bins[k].hi[q]); }
                    cnt += bins[k].n;
                    ra[k] = cnt ? area3(lo, hi) : 0; rn[k] = cnt;
                }
                for (int q = 0; q < 3; ++q) { lo[q] = DBL_MAX; hi[q] = -DBL_MAX; }
                cnt = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    for (int q = 0; q < 3; ++q) { lo[q] = std::fmin(lo[q], bins[k].lo[q]); hi[q] = std::fmax(hi[q], bins[k].hi[q]); }
                    cnt += bins[k].n;
                    if (cnt == 0 || rn[k + 1] == 0) continue;
                    const double cost = area3(lo, hi) * cnt + ra[k + 1] * rn[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
                }
            }
        }
        int32_t mid;
        if (best_axis < 0) {
            mid = b + (e - b) / 2;  // all centroids coincide
        } else {
            const int a = best_axis;
            const double sc = scale[a];
            auto goes_left = [&](const SahItem& it) {
                int k = (int)((it.c[a] - clo[a]) * sc);
                if (k >= kBins) k = kBins - 1;
                return k <= best_split;
            };
            if (nt == 1) {
                // stable split in place: left items close up, right items wait in tmp[b..] (the same range, so
                // threads working on disjoint ranges do not meet) and are copied back behind them
                int32_t nl = b, nr = b;
                for (int32_t i = b; i < e; ++i)
                    if (goes_left(items[i])) items[nl++] = items[i]; else tmp[nr++] = items[i];
                std::copy(tmp.get() + b, tmp.get() + nr, items.begin() + nl);
                mid = nl;
            } else {
                // the same split by chunks: count, then every chunk writes its left and right items to their
                // final places in tmp[b, e), then the range is copied back
                chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {
                    int32_t n = 0;
                    for (int32_t i = cb; i < ce; ++i) n += goes_left(items[i]) ? 1 : 0;
                    ps[c].n_left = n;
                });
                int32_t total_left = 0;
                for (int c = 0; c < nt; ++c) total_left += ps[c].n_left;
                chunks(b, e, nt, [&](int c, int32_t cb, int32_t ce) {
                    int32_t nl = b, nr = b + total_left + (cb - b);
                    for (int k = 0; k < c; ++k) { nl += ps[k].n_left; nr -= ps[k].n_left; }
                    for (int32_t i = cb; i < ce; ++i)
                        if (goes_left(items[i])) tmp[nl++] = items[i]; else tmp[nr++] = items[i];
                });
                chunks(b, e, nt, [&](int, int32_t cb, int32_t ce) { std::copy(tmp.get() + cb, tmp.get() + ce, items.begin() + cb); });
                mid = b + total_left;
            }
            if (mid == b || mid == e) mid = b + (e - b) / 2;
        }
        int32_t l, r;
        if (threads > 1 && mid - b >= kParallelMin && e - mid >= kParallelMin) {
            int tl = (int)(((int64_t)threads * (mid - b) + (e - b) / 2) / (e - b));
            tl = tl < 1 ? 1 : (tl > threads - 1 ? threads - 1 : tl);
            ThreadGroup left;
            left.run([&] { l = build(b, mid, depth + 1, tl); });
            r = build(mid, e, depth + 1, threads - tl);
            left.join();
        } else {  // a short half is not worth a thread: the long one keeps them all
            l = build(b, mid, depth + 1, mid - b >= e - mid ? threads : 1);
            r = build(mid, e, depth + 1, mid - b >= e - mid ? 1 : threads);
        }
        b2[me].left = l; b2[me].right = r;
        return me;
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
    // slots of every wide node below n; subtrees of at most `grain` shapes are left to `later` (the threads)
    void widen(int32_t n, int32_t grain, std::vector<int32_t>* later) {
        if (b2[n].left < 0) return;
        if (later && b2[n].e - b2[n].b <= grain) { later->push_back(n); return; }
        choose_slots(n);
        for (int sl = 0; sl < 8; ++sl)
            if (wide[n].slot[sl] >= 0) widen(wide[n].slot[sl], grain, later);
    }
    void widen_all(int threads) {
        const int32_t total = n_nodes.load();
        wide.reset(new Wide[(size_t)total]);
        const int32_t n_items = b2[0].e - b2[0].b;
        if (threads <= 1 || n_items < kChunkMin) { widen(0, 0, nullptr); return; }
        std::vector<int32_t> later;
        widen(0, n_items / (threads * 8) + 1, &later);
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t i = next.fetch_add(1); i < later.size(); i = next.fetch_add(1)) widen(later[i], 0, nullptr);
        };
        ThreadGroup group;
        for (int c = 1; c < threads; ++c) group.run(work);
        work();
        group.join();
    }
    // wide node for Bvh2 node `n` written into HostNode `at` (sequential: the order fixes the node numbers)
    void emit(std::vector<HostNode>& out, int32_t n, int32_t at) {
        const Bvh2& nd = b2[n];
        for (int a = 0; a < 3; ++a) { out[at].lo[a] = nd.lo[a]; out[at].hi[a] = nd.hi[a]; }
        out[at].empty = false;
        out[at].n_shapes = nd.e - nd.b;
        out[at].shape = -1;
        const int32_t base = (int32_t)out.size();
        out[at].child_base = base;
        out.resize(out.size() + 8);
        if (nd.left < 0) {  // leaf: one single-shape child per shape
            for (int32_t i = nd.b; i < nd.e; ++i) {
                HostNode& c = out[base + (i - nd.b)];
                for (int a = 0; a < 3; ++a) { c.lo[a] = items[i].lo[a]; c.hi[a] = items[i].hi[a]; }
                c.shape = items[i].shape; c.empty = false; c.n_shapes = 1;
            }
            return;
        }
        for (int sl = 0; sl < 8; ++sl)
            if (wide[n].slot[sl] >= 0) emit(out, wide[n].slot[sl], base + sl);
    }
};
}  // namespace

std::vector<HostNode> build_sah(const rt_shape* shapes, int64_t n, int leaf_size) {
    std::vector<HostNode> out;
    if (n == 0) { out.push_back(HostNode{}); return out; }
    SahBuilder sb;
    sb.leaf_size = leaf_size < 1 ? 1 : (leaf_size > 8 ? 8 : leaf_size);
    // threads building at one time: RTGO_BUILD_THREADS, by default the hardware's but at most 16 (one GPU's
    // share of a node's cores)
    int threads = (int)std::thread::hardware_concurrency();
    if (threads > 16) threads = 16;
    if (const char* ev = std::getenv("RTGO_BUILD_THREADS")) threads = std::atoi(ev);
    threads = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    sb.items.resize((size_t)n);
    SahBuilder::chunks(0, (int32_t)n, n >= kChunkMin ? threads : 1, [&](int, int32_t cb, int32_t ce) {
        for (int32_t i = cb; i < ce; ++i) {
            SahItem& it = sb.items[(size_t)i];
            shape_bounds(shapes[i], it.lo, it.hi);
            for (int a = 0; a < 3; ++a) it.c[a] = (it.lo[a] + it.hi[a]) * 0.5;
            it.shape = i;
        }
    });
    sb.b2.reset(new Bvh2[2 * (size_t)n]);
    sb.tmp.reset(new SahItem[(size_t)n]);
    sb.build(0, (int32_t)n, 0, threads);
    out.reserve((size_t)n * 3);
    out.push_back(HostNode{});
    sb.widen_all(threads);
    sb.emit(out, 0, 0);
    return out;
}

// depth-first, octant order
static void collect_shapes(const std::vector<HostNode>& nodes, int32_t n, std::vector<int32_t>* out) {
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
static int n_children(const std::vector<HostNode>& nodes, int32_t e) {
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
struct LeafRule {
    int leaf_size;
    double root_area;
};
static double max_shape_area(const std::vector<HostNode>& nodes, int32_t n) {
    if (nodes[n].child_base < 0) return nodes[n].shape >= 0 ? box_area(nodes[n]) : 0.0;
    double m = 0;
    for (int o = 0; o < 8; ++o)
        if (!nodes[nodes[n].child_base + o].empty) m = std::max(m, max_shape_area(nodes, nodes[n].child_base + o));
    return m;
}
static bool is_leaf(const std::vector<HostNode>& nodes, int32_t e, const LeafRule& rule) {
    if (nodes[e].n_shapes > rule.leaf_size) return false;
    if (nodes[e].n_shapes <= 1 || nodes[e].child_base < 0) return true;
    const double a = box_area(nodes[e]);
    if (a * 64.0 < rule.root_area) return true;
    return a <= 1.5 * max_shape_area(nodes, e);
}

static void block_slots(const std::vector<HostNode>& nodes, int32_t x, const LeafRule& rule, int32_t slot[8]) {
    int n = 0;
    for (int o = 0; o < 8; ++o) {
        const int32_t c = nodes[x].child_base + o;
        slot[o] = nodes[c].empty ? -1 : skip_chain(nodes, c);
        n += slot[o] >= 0;
    }
    if (std::getenv("RTGO_NO_COLLAPSE")) return;  // A/B switch for measurements; the oracle mirror has the same one
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

// The block whose child records are records[first, first + count) (slot order, absent slots omitted), as a QBlock.
static void append_qblock(FlatTree* out, int64_t first, int count, uint32_t present, uint32_t leaf_mask, uint32_t child_base) {
    uint32_t w[kQBlockWords] = {};
    const ChildRecord* rec = out->records.data() + first;
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
    out->qblocks.insert(out->qblocks.end(), w, w + kQBlockWords);
}

bool flatten_octree(const std::vector<HostNode>& nodes, int leaf_size, FlatTree* out, const char** err) {
    out->records.clear();
    out->qblocks.clear();
    out->order.clear();
    out->block_first.assign(1, 0u);
    out->block_box.clear();
    out->root_ref = 0; out->n_inner = 0; out->depth = 0; out->stack_bound = 1;
    if (nodes.empty() || nodes[0].empty) {  // no shapes: present mask 0 => every ray misses
        out->qblocks.assign(kQBlockWords, 0u);
        out->block_box.assign(8, 0.0f);
        return true;
    }
    if (leaf_size < 1) leaf_size = 1;
    if (leaf_size > kMaxLeafSize) leaf_size = kMaxLeafSize;

    auto make_leaf = [&](int32_t e, ChildRecord* r) {
        const uint32_t first = (uint32_t)out->order.size();
        collect_shapes(nodes, e, &out->order);
        r->w0 = (first << 8) | (uint32_t)(out->order.size() - first);
        r->w1 = kLeafFlag;
    };
    auto set_box = [&](int32_t e, ChildRecord* r) {
        for (int a = 0; a < 3; ++a) { r->bmin[a] = round_down_f32(nodes[e].lo[a]); r->bmax[a] = round_up_f32(nodes[e].hi[a]); }
    };

    struct Pending { int32_t node; int64_t patch; int32_t level; };  // patch = record whose w0 names this block
    std::deque<Pending> queue;
    struct Block { int64_t first; int count; };
    std::vector<Block> blocks;
    std::unordered_map<int64_t, int64_t> block_of_first;
    const int32_t root = skip_chain(nodes, 0);
    const LeafRule rule{leaf_size, box_area(nodes[root])};
    if (is_leaf(nodes, root, rule)) {
        // everything fits one leaf: synthetic root block with that leaf in octant 0
        ChildRecord r;
        set_box(root, &r);
        make_leaf(root, &r);
        r.w1 |= 1u;
        out->records.push_back(r);
        out->root_ref = (0u << 8) | 1u;
        out->n_inner = 1; out->depth = 1; out->stack_bound = 8;
        append_qblock(out, 0, 1, 1u, 1u, 0u);
        out->block_first.assign(1, 0u);
        out->block_box.assign(8, 0.0f);
        for (int a = 0; a < 3; ++a) { out->block_box[(size_t)a] = r.bmin[a]; out->block_box[(size_t)(4 + a)] = r.bmax[a]; }
        return true;
    }
    queue.push_back({root, -1, 1});
    uint32_t n_pushed = 1;  // the i-th node pushed becomes block i (first in, first out)
    while (!queue.empty()) {
        const Pending p = queue.front();
        queue.pop_front();
        const HostNode& nd = nodes[p.node];
        const int64_t first = (int64_t)out->records.size();
        int32_t slot[8];
        block_slots(nodes, p.node, rule, slot);
        uint32_t present = 0, leaf_mask = 0, child_base = 0;
        for (int o = 0; o < 8; ++o) {
            const int32_t e = slot[o];
            if (e < 0) continue;
            present |= 1u << o;
            ChildRecord r;
            set_box(e, &r);
            if (is_leaf(nodes, e, rule)) {
                make_leaf(e, &r);
                leaf_mask |= 1u << o;
            } else {
                r.w0 = 0; r.w1 = 0;
                if ((present & ~leaf_mask) == (1u << o)) child_base = n_pushed;  // first inner child of this block
                ++n_pushed;
                queue.push_back({e, (int64_t)out->records.size(), p.level + 1});
            }
            out->records.push_back(r);
        }
        append_qblock(out, first, (int)(out->records.size() - (size_t)first), present, leaf_mask, child_base);
        (void)nd;
        for (size_t k = (size_t)first; k < out->records.size(); ++k) out->records[k].w1 |= leaf_mask;
        if ((int64_t)out->records.size() > kMaxChildRecords || (int64_t)out->order.size() >= kMaxChildRecords) {
            *err = "scene too large: the flattened octree exceeds 2^24 child records";
            return false;
        }
        const uint32_t ref = ((uint32_t)first << 8) | present;
        block_of_first[first] = (int64_t)blocks.size();
        blocks.push_back({first, (int)(out->records.size() - (size_t)first)});
        if (p.patch < 0) out->root_ref = ref; else out->records[(size_t)p.patch].w0 = ref;
        out->n_inner++;
        if (p.level > out->depth) out->depth = p.level;
    }
    out->block_first.assign(blocks.size(), 0u);
    out->block_box.assign(blocks.size() * 8, 0.0f);
    for (size_t i = 0; i < blocks.size(); ++i) {
        out->block_first[i] = (uint32_t)blocks[i].first;
        float* bb = out->block_box.data() + i * 8;
        const ChildRecord* rec = out->records.data() + blocks[i].first;
        for (int a = 0; a < 3; ++a) {
            bb[a] = rec[0].bmin[a]; bb[4 + a] = rec[0].bmax[a];
            for (int k = 1; k < blocks[i].count; ++k) { bb[a] = std::min(bb[a], rec[k].bmin[a]); bb[4 + a] = std::max(bb[4 + a], rec[k].bmax[a]); }
        }
    }
    // Worst-case traversal stack: entering a block pushes at most (inner children - 1) entries and descends
    // into one inner child, so the deepest stack below block b is need(b) = (inner(b) - 1) + max over its
    // inner children of need(child).  Blocks are in breadth-first order: children come after their parent.
    {
        std::vector<int32_t> need(blocks.size(), 0);
        for (int64_t i = (int64_t)blocks.size() - 1; i >= 0; --i) {
            const int64_t first = blocks[(size_t)i].first;
            const int n = blocks[(size_t)i].count;
            int inner = 0, deepest = 0;
            for (int k = 0; k < n; ++k) {
                const ChildRecord& r = out->records[(size_t)(first + k)];
                if (r.w1 & kLeafFlag) continue;
                ++inner;
                const auto it = block_of_first.find((int64_t)(r.w0 >> 8));
                if (it != block_of_first.end()) deepest = std::max(deepest, need[(size_t)it->second]);
            }
            need[(size_t)i] = (inner > 0 ? inner - 1 : 0) + deepest;
        }
        out->stack_bound = (need.empty() ? 0 : need[0]) + 1;
    }
    return true;
}

}  // namespace rtgo
