// Host-side acceleration structure of the MI355X backend.
//
// build_octree() restates NewBoundingVolumeHierarchy (reference
// raytracer/accelerationstructures.go:27-65): sequential insertion octree with one shape per
// leaf (addToBVH :226-255, getBvhQuadrantIndex :266-279, splitBvhQuadrant :282-353), then the
// bottom-up tight refit and the pruning of empty children (recomputeNodeBounds :189-211,
// destroyUselessNodes :213-224).
//
// flatten_octree() turns that tree into the arrays the HIP traversal reads from HBM:
//   - every node with >= 2 non-empty children becomes one contiguous block of 32-byte child
//     records {fp32 AABB rounded outward, w0, w1}, children in octant order, empty ones
//     omitted; blocks are laid out breadth-first so the top of the tree is contiguous;
//   - a chain of nodes with a single non-empty child is dropped: after the refit such a node
//     has exactly its child's box, so removing it cannot change any hit;
//   - a subtree that holds at most `leaf_size` shapes becomes ONE leaf record listing those shapes
//     (depth-first, octant order); shapes are re-ordered so that a leaf's shapes are contiguous
//     (`order[q]` = position in Scene.Shapes of re-ordered shape q).  leaf_size = 1 keeps the
//     reference's one-shape-per-leaf tree;
//   - inner child : w0 = (first record of its block << 8) | 8-bit octant-present mask, w1 = 0 (+ mask below)
//     leaf child  : w0 = (first re-ordered shape << 8) | shape count, w1 = kLeafFlag;
//     every record's w1 also holds, in its low 8 bits, the leaf mask of the block it is in (octant order).
//   - the same blocks once more as 80-byte QUANTISED blocks (QBlock) for the shadow-ray walk, whose answer is an
//     OR over shapes and so needs boxes that enclose, not boxes that are exact: one block = the 8 slots' boxes as
//     8-bit offsets on a per-block power-of-two grid, 5 x 16 bytes fetched per visit instead of up to 16 x 16
//     (k_shadow is bound by the vector-memory path, profiles/r02).
#pragma once
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rtgo.h"

namespace rtgo {

constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr int64_t kMaxChildRecords = (int64_t)1 << 24;  // w0 keeps 24 bits for the block / shape index
constexpr int kMaxLeafSize = 31;  // 8 leaves of a block: their shape counts and the running sums fit bytes of a QBlock

// Block b of the flattened tree for the shadow-ray walk (device_math.hpp visit_qblock), 20 words:
//   [0..2]  origin: the minimum corner of the block's box (fp32, exact minimum of the children's rounded boxes)
//   [3]     biased fp32 exponents of the grid step per axis (bytes 0..2: x, y, z) | slot-present mask << 24
//   [4]     index of the block of the first inner child (inner children's blocks are consecutive, slot order)
//           | leaf mask << 24
//   [5]     first shape of the first leaf child (the leaf children's shapes are consecutive, slot order)
//   [6..7]  one byte per slot: leaf: shape count (1..31); inner child: (rank among the inner children) << 5; absent: 0
//   [8..13] lower planes: 8 bytes (slots 0..7) per axis x, y, z: floor((bmin - origin) / step)
//   [14..19] upper planes likewise: ceil((bmax - origin) / step) <= 255
// so that origin + q * step encloses the child's fp32 box in exact arithmetic.
constexpr int kQBlockWords = 20;
constexpr int64_t kMaxBlocks = (int64_t)1 << 24;

// A few host threads that are always joined, and whose first exception (std::bad_alloc, std::system_error)
// is thrown again by join() on the thread that started them, instead of ending the process.
class ThreadGroup {
public:
    template <class F>
    void run(F f) {
        threads_.emplace_back([this, f]() mutable {
            try {
                f();
            } catch (...) {
                std::lock_guard<std::mutex> g(m_);
                if (!err_) err_ = std::current_exception();
            }
        });
    }
    void join() {
        for (auto& t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
        if (err_) std::rethrow_exception(std::exchange(err_, nullptr));
    }
    ~ThreadGroup() {
        for (auto& t : threads_)
            if (t.joinable()) t.join();
    }

private:
    std::vector<std::thread> threads_;
    std::exception_ptr err_;
    std::mutex m_;
};

struct HostNode {
    double lo[3], hi[3];  // cell bounds while inserting, tight bounds after refit()
    int32_t shape = -1;   // leaf payload, -1 = empty
    int32_t child_base = -1;  // index of the first of 8 consecutive children, -1 = leaf
    bool empty = true;    // after refit(): subtree holds no shape
    int32_t n_shapes = 0; // after refit(): shapes in the subtree
};

// std::vector<HostNode> whose resize(n) leaves the new elements unwritten (HostNode is trivially copyable and
// destructible; whoever resizes writes every new element in full): the SAH builder's threads fill the nodes of a
// 10^6-shape tree themselves, and with that touch the pages of their own parts first
template <class T>
struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    DefaultInit() = default;
    template <class U> DefaultInit(const DefaultInit<U>&) {}
    template <class U> void construct(U*) {}
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using HostNodes = std::vector<HostNode, DefaultInit<HostNode>>;

struct ChildRecord {  // 32 bytes, read by the kernels as two float4
    float bmin[3];
    float bmax[3];
    uint32_t w0, w1;
};

struct FlatTree {
    std::vector<ChildRecord> records;
    std::vector<uint32_t> qblocks;  // kQBlockWords per block, block i of the breadth-first order (root = 0)
    std::vector<int32_t> order;  // re-ordered shape q -> index in Scene.Shapes
    std::vector<float> block_box;  // per block: {min.xyz, 0, max.xyz, 0} of its children's fp32 boxes (what a ray must meet to hit anything below the block)
    std::vector<uint32_t> block_first;  // per block: its first child record
    uint32_t root_ref = 0;  // (first << 8) | present of the root block
    int64_t n_inner = 0;    // blocks
    int32_t depth = 0;      // levels of blocks below the root (root block = 1)
    int32_t stack_bound = 1;  // exact worst case of the traversal stack for this tree
    double qpad = 0.0;        // what every box of the flat tree is grown by before it is rounded outward to float32: 2^-19 of the scene's largest coordinate
};

// shapes: desc->shapes[0..n). Returns nodes[0] = root.
HostNodes build_octree(const rt_shape* shapes, int64_t n);
// The default tree ("faster builder"): a binned-SAH BVH in the same 8-slot node form, so
// that flatten_octree() and the traversal are shared with the reference's octree (RTGO_TREE=octree).
//   split(range): boxes and centroid bounds (centroid = centre of the shape's box); at most leaf_size
//     (<= 8) shapes -> leaf.  Otherwise, per axis with a positive centroid extent, 16 equal bins;
//     for every bin boundary with shapes on both sides cost = area(left) * n_left + area(right) *
//     n_right; strictly smallest cost wins (first axis / boundary on ties); the range is split stably;
//     with no valid boundary, or more than 48 levels down, in the middle.
//   widen(node): children = {left, right}; while fewer than 8 and some child is inner, the inner child
//     with the largest area (first on ties) is replaced by its left child, its right child appended;
//     children take slots by the greedy octant matching of block_slots.  A leaf's shapes become
//     single-shape children in slots 0.. in range order.
//   Threads (RTGO_BUILD_THREADS, default min(hardware, 16)): one thread builds the top of the tree, whose ranges
//     (>= 131072 shapes) are swept in chunks by whichever threads are free; every shorter range is a task that the
//     threads take longest first, and a task hands its right half on as a new task while both halves are worth one.
//     A range is split by one sweep from one of two arrays into the other, which also gathers its halves' boxes.
//     min, max and counts merge exactly and a range's split depends on the range alone, so the tree does not
//     depend on the number of threads or on who builds what (tests/cpp/builder_check.cpp).
// Any tree whose boxes enclose their shapes gives the same hits; only the order in which equal-t hits
// are found depends on it, which is why the oracle's fp32 mode restates this builder too.
HostNodes build_sah(const rt_shape* shapes, int64_t n, int leaf_size);
// Threads the builders and the flattener use: RTGO_BUILD_THREADS, by default the hardware's but at most 16.
int builder_threads();
// Returns false (and sets *err) if the tree exceeds kMaxChildRecords.
// origin_reach: the largest coordinate a ray can start at outside the shapes' box (the camera's lens); the boxes' pad grows with it
bool flatten_octree(const HostNodes& nodes, int leaf_size, FlatTree* out, const char** err, double origin_reach = 0.0);

}  // namespace rtgo
