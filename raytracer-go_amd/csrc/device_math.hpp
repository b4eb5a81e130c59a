// Device-side arithmetic of the MI355X backend: r3.Vec helpers, Philox4x32-10, ray/shape
// intersection, the ordered stack traversal of the flattened octree, textures and materials.
//
// All arithmetic is float32 with contraction OFF (the translation unit is built with
// -ffp-contract=off and the pragma below): every expression is evaluated in the operation
// order of the reference's float64 code, so that a float32 CPU restatement is bit-identical
// (primitive IDs, t, camera rays).  The one deliberate FMA is the slab test, written with an
// explicit fmaf on both sides.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rtgo {

#define RT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 scale3(float f, f3 a) { return mk3(f * a.x, f * a.y, f * a.z); }
RT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 cross3(f3 p, f3 q) { return mk3(p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x); }
RT_DEV float norm3(f3 p) { return sqrtf(p.x * p.x + p.y * p.y + p.z * p.z); }
// r3.Unit: Scale(1/Norm(p), p); Unit(0) = NaN vector
RT_DEV f3 unit3(f3 p) {
    if (p.x == 0.0f && p.y == 0.0f && p.z == 0.0f) { const float n = __builtin_nanf(""); return mk3(n, n, n); }
    return scale3(1.0f / norm3(p), p);
}
RT_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// ---- Philox4x32-10; counter = (pixel, sample, bounce*256+purpose, sub*256+try), key = seed ----
struct u4 { uint32_t x, y, z, w; };
RT_DEV u4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}
enum { P_AA = 0, P_LENS = 1, P_FUZZ = 2, P_SCHLICK = 3, P_SHADOW = 4 };
constexpr int kMaxTries = 64;
struct RngKey { uint32_t k0, k1, pixel, sample; };
RT_DEV u4 draw(const RngKey& k, uint32_t bounce, uint32_t purpose, uint32_t sub, uint32_t tr) {
    return philox4x32(k.pixel, k.sample, bounce * 256u + purpose, sub * 256u + tr, k.k0, k.k1);
}
RT_DEV float u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// randomInUnitSphere, material.go:134-143 (rejection loop capped at kMaxTries).  ONE Philox block per
// call: try 0 takes its first three words; further tries take three words each from a xorshift128
// stream seeded with the block's four words.  (A Philox block per try cost the 64 lanes of a wavefront
// ~5.7 blocks — the slowest lane's try count — and was 16 % of the shadow kernel's instructions.)
RT_DEV uint32_t xorshift128(u4* s) {
    const uint32_t t = s->x ^ (s->x << 11);
    s->x = s->y; s->y = s->z; s->z = s->w;
    s->w = s->w ^ (s->w >> 19) ^ t ^ (t >> 8);
    return s->w;
}
RT_DEV f3 random_in_unit_sphere(const RngKey& k, uint32_t bounce, uint32_t purpose, uint32_t sub) {
    u4 s = draw(k, bounce, purpose, sub, 0u);
    uint32_t a = s.x, b = s.y, c = s.z;
    for (int tr = 0; tr < kMaxTries; ++tr) {
        const f3 p = sub3(scale3(2.0f, mk3(u01(a), u01(b), u01(c))), mk3(1.0f, 1.0f, 1.0f));
        if (p.x * p.x + p.y * p.y + p.z * p.z < 1.0f) return p;
        a = xorshift128(&s); b = xorshift128(&s); c = xorshift128(&s);
    }
    return mk3(0.0f, 0.0f, 0.0f);
}

// ---- scene tables in HBM ------------------------------------------------------------------
struct DMaterial { int kind, texture; float color[3], spec[3], param; };
struct DTexture { int kind, w, h; float c1[3], c2[3], cw, ch; const uint8_t* rgba; };
struct DLight { int kind; float color[3], pos[3], look_dir[3], intensity, spec_intensity, cos_angle, decay; };
struct DCamera { float u[3], v[3], origin[3], ll[3], hor[3], ver[3], lens; };

constexpr uint32_t kLeaf = 0x80000000u;   // ChildRecord.w1
constexpr uint32_t kPrimTriangle = 1u;    // prims[3i+2].y bits: bit0 triangle, bit1 single sided, material << 2
constexpr uint32_t kPrimSingleSided = 2u;

constexpr int kLeafBoundQuads = 4;
struct DScene {
    const float4* recs;    // 2 float4 per child record: {min.xyz, max.x}, {max.y, max.z, w0, w1}
    const uint4* qblocks;  // 5 uint4 per block: the same tree with quantised boxes (octree.hpp QBlock), shadow rays only
    const float4* prims;   // 3 float4 per shape, in LEAF order: tri {A, B.x}{B.yz, C.xy}{C.z, bits, -, -}; sphere {C, R}{-}{-, bits}
    const float4* tprims;  // the same for the intersection tests: edges e1 = B - A, e2 = C - A in place of B, C
    const int* prim_ids;   // leaf-order shape q -> position in Scene.Shapes
    const uint32_t* block_first;     // per block (qblocks): its first child record (recs)
    const float4* block_box;         // 2 float4 per block: {min.xyz, -}, {max.xyz, -} of its children's boxes
    const float4* leaf_bound;        // kLeafBoundQuads float4 per SHAPE index, filled at every leaf's FIRST shape (k_leaf_bounds): the leaf's
                                     // exact box and its two slabs {min.xyz, max.x} {max.yz, n.xy} {n.z, dmin, dmax, n2.x} {n2.yz, d2min, d2max}; nullptr: none
    const uint8_t* vox;              // per positional light and cell of a grid over the scene's box: 1 = every ray from any point of the
                                     // cell to any jittered position of the light is unoccluded (k_voxel_beams); nullptr: no grid
    float vox_min[3], vox_cs, vox_inv_cs;
    float beam_r, beam_n1, beam_3n1;  // radius of the ball round a positional light that holds every jittered (and rounded) light position (pair beams, voxel beams), and 1.00001 / 3.00003 times that
    int vox_n[3];
    const DMaterial* materials;
    const DTexture* textures;
    const DLight* lights;
    const DLight* pos_lights;  // the positional (point and spot) lights of `lights`, in the same order
    int n_lights, n_pos_lights;
    uint32_t root_ref;
};

// ---- intersection: shape.go:47-78 (sphere), :117-163 (triangle); returns t or -1 ------------
RT_DEV float sphere_hit(const float4 p0, f3 o, f3 d, float tmin, float tmax) {
    const f3 oc = sub3(o, mk3(p0.x, p0.y, p0.z));
    const float a = dot3(d, d);
    const float b = dot3(oc, d);
    const float c = dot3(oc, oc) - p0.w * p0.w;
    const float disc = b * b - a * c;
    if (disc > 0.0f) {
        const float first = (-b - sqrtf(b * b - a * c)) / a;
        if (first > tmin && first <= tmax) return first;
        // shape.go:64: the "second root" repeats the first, so it can never add a hit
    }
    return -1.0f;
}
// e1 = B - A, e2 = C - A come precomputed (DScene::tprims): the same fp32 values shape.go:119-120 would give
RT_DEV float triangle_hit(f3 A, f3 e1, f3 e2, bool single_sided, f3 o, f3 d, float tmin, float tmax) {
    const f3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    if (single_sided && det < 0.0f) return -1.0f;
    const float inv = 1.0f / det;
    const f3 tvec = sub3(o, A);
    const float u = dot3(tvec, pvec) * inv;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 qvec = cross3(tvec, e1);
    const float v = dot3(d, qvec) * inv;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float t = dot3(e2, qvec) * inv;
    if (t < tmin || t > tmax) return -1.0f;
    return t;
}
// Shadow rays only need "is there a hit with tmin <= t <= tmax": the same Moller-Trumbore quantities,
// but fused (explicit fmaf) and without the division — u, v, t are compared against |det| instead of
// being scaled by 1/det.  The outcome can differ from
// triangle_hit's only for a ray within a rounding error of an edge or of tmin/tmax; nearest-hit rays
// (whose t, shape and barycentrics are visible in the image) keep the reference's expression order.
RT_DEV float dotf(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RT_DEV f3 crossf(f3 a, f3 b) {
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
RT_DEV bool triangle_occludes(f3 A, f3 e1, f3 e2, bool single_sided, f3 o, f3 d, float tmin, float tmax) {
    const f3 pvec = crossf(d, e2);
    const float det = dotf(e1, pvec);
    const f3 tvec = sub3(o, A);
    const f3 qvec = crossf(tvec, e1);
    const uint32_t sb = __float_as_uint(det) & 0x80000000u;  // move det's sign onto u, v, t
    const float ad = __uint_as_float(__float_as_uint(det) ^ sb);
    const float us = __uint_as_float(__float_as_uint(dotf(tvec, pvec)) ^ sb);
    const float vs = __uint_as_float(__float_as_uint(dotf(d, qvec)) ^ sb);
    const float ts = __uint_as_float(__float_as_uint(dotf(e2, qvec)) ^ sb);
    // t = ts / ad is never formed: t > 0 is ts > 0, tmin <= t <= tmax is compared against ad
    return ad > 0.0f && !(single_sided && sb) && us >= 0.0f && vs >= 0.0f && us + vs <= ad &&
           ts >= tmin * ad && ts <= tmax * ad && ts > 0.0f;
}
// all three quads of a shape are fetched together (a sphere ignores p1): one memory latency per shape
struct PrimQuads { float4 p0, p1, p2; };
RT_DEV PrimQuads load_tprim(const DScene& S, uint32_t prim) {
    PrimQuads q;
    q.p0 = S.tprims[3 * (size_t)prim];
    q.p1 = S.tprims[3 * (size_t)prim + 1];
    q.p2 = S.tprims[3 * (size_t)prim + 2];
    // keeps the compiler from sinking the p1 / p2 loads below the triangle-or-sphere branch, which would
    // make them a second, dependent memory round trip per shape
    asm volatile("" : "+v"(q.p0.x), "+v"(q.p1.x), "+v"(q.p2.x), "+v"(q.p2.y));
    return q;
}
// nearest-hit rays: t of the hit inside (tmin, tmax] / [tmin, tmax], or -1
RT_DEV float prim_hit(const DScene& S, uint32_t prim, f3 o, f3 d, float tmin, float tmax) {
    const PrimQuads q = load_tprim(S, prim);
    const uint32_t bits = __float_as_uint(q.p2.y);
    if (bits & kPrimTriangle)
        return triangle_hit(mk3(q.p0.x, q.p0.y, q.p0.z), mk3(q.p0.w, q.p1.x, q.p1.y), mk3(q.p1.z, q.p1.w, q.p2.x),
                            (bits & kPrimSingleSided) != 0, o, d, tmin, tmax);
    return sphere_hit(q.p0, o, d, tmin, tmax);
}
// shadow rays: is there a hit with t > 0 inside the interval (the traversal's acceptance rule, light.go:141-160)
RT_DEV bool prim_occludes(const DScene& S, uint32_t prim, f3 o, f3 d, float tmin, float tmax) {
    const PrimQuads q = load_tprim(S, prim);
    const uint32_t bits = __float_as_uint(q.p2.y);
    if (bits & kPrimTriangle)
        return triangle_occludes(mk3(q.p0.x, q.p0.y, q.p0.z), mk3(q.p0.w, q.p1.x, q.p1.y), mk3(q.p1.z, q.p1.w, q.p2.x),
                                 (bits & kPrimSingleSided) != 0, o, d, tmin, tmax);
    const float t = sphere_hit(q.p0, o, d, tmin, tmax);
    return t > 0.0f && t <= tmax;
}

// ---- optional loop instrumentation (make prof -> librtgo_prof.so, scripts/prof_loops.py) ----
#ifdef RTGO_PROF
__device__ unsigned long long g_prof[8];
__device__ unsigned long long g_prof3[65 + 64];  // shadow-ray walk: [w] iterations with w walking lanes, [65 + i] wave rounds of i iterations (63: or more)
__device__ unsigned long long g_prof5[4];  // pairs of consecutive shadow-ray rounds of a wavefront: [0] pairs, [1] sum of (iterations a + iterations b), [2] sum of max over lanes of (visits a + visits b), [3] sum of max(iters a, iters b, ceil(visits of both / 64))
__device__ unsigned long long g_prof4[64];  // walking lanes summed per iteration index (63: and later)
__device__ unsigned long long g_prof2[4];  // shadow-ray walk: iterations, lanes only holding leaves, idle lanes, shape loops  // [2i] = times a wavefront passed point i, [2i+1] = lanes active there
#define RT_PROF_POINT(i)                                                          \
    do {                                                                          \
        const unsigned long long m_ = __ballot(true);                             \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) {                     \
            atomicAdd(&g_prof[2 * (i)], 1ull);                                    \
            atomicAdd(&g_prof[2 * (i) + 1], (unsigned long long)__popcll(m_));    \
        }                                                                         \
    } while (0)
#else
#define RT_PROF_POINT(i)
#endif

// ---- ordered stack traversal --------------------------------------------------------------
// Replaces bvh.trace (accelerationstructures.go:81-124): same nearest hit, found with a stack
// instead of a heap.  Per block: (1) every present child's box is slab-tested against the best t
// known at block entry, in ray order k = 0..7, slot = k ^ sign mask; (2) hit leaf children are queued
// in that order; (3) hit inner children are pushed far-to-near and the nearest is entered.
// The queued leaves' shapes are intersected later, when enough lanes of the wavefront hold leaves
// (kFlushLanes, or as many lanes as are still walking): a shape loop entered for the few lanes that hit
// a leaf in the current block ran at 16 % lane utilisation and was half the kernel's instructions.
// Holding a leaf back cannot change the result: a lane's shapes are still tested in the same order,
// and a box that a smaller `best` would have culled holds no hit below that `best`.
#ifndef RTGO_LDS_STACK
#define RTGO_LDS_STACK 11
#endif
constexpr int kLdsStackDepth = RTGO_LDS_STACK;
#ifndef RTGO_PEND_CAP
#define RTGO_PEND_CAP 8
#endif
#ifndef RTGO_FLUSH_LANES
#define RTGO_FLUSH_LANES 16
#endif
#ifndef RTGO_STEAL
#define RTGO_STEAL 16  // occluded(): idle lanes take stack entries of walking lanes when this many are idle (0: never)
#endif
#ifndef RTGO_VISIT_BATCH
#define RTGO_VISIT_BATCH 4
#endif
#ifndef RTGO_BEAM_CAP
#define RTGO_BEAM_CAP 32  // beam_is_clear(): block visits after which a beam's walk gives up (0: never).  With leaf slabs a walk no longer stops at the first leaf BOX it meets, and the longest of a wavefront's 64 walks sets its pace: frame 231.5 ms without a cap, 237.3 / 232.6 / 228.8 / 228.9 with 12 / 16 / 24 / 32 (k_pair_beam alone 28.9 -> 17 ms; 0.5 % fewer rays decided by beams at 32)
#endif
#ifndef RTGO_BEAM_SLAB2
#define RTGO_BEAM_SLAB2 1
#endif
#ifndef RTGO_LEAF_SLAB
#define RTGO_LEAF_SLAB 1  // occluded(), beam_is_clear(): leaves have a slab besides their box (0: A/B switch)
#endif
constexpr int kVisitBatch = RTGO_VISIT_BATCH;  // child records fetched together in a block visit (8, 4 or 2)
constexpr int kPendCap = RTGO_PEND_CAP;        // leaves a lane can hold back (one block visit adds at most 8)
constexpr int kFlushLanes = RTGO_FLUSH_LANES;  // occluded(): run the shared shape loop when this many lanes hold leaves
constexpr int kFlushLanesNearest = 64;         // traverse(): every lane loops over its own shapes, so wait for all
constexpr int kLdsPerLane = kLdsStackDepth + kPendCap + 1;  // +1: the branch-free append stores one past the last
static_assert(kPendCap >= 8, "one block visit can queue 8 leaves: a smaller queue is overrun (and then names shapes that do not exist)");
static_assert(kLdsStackDepth >= 1 && kVisitBatch >= 1 && 8 % kVisitBatch == 0, "bad traversal knobs");
struct TraceStack {
    uint32_t* lds;        // &s_stack[threadIdx.x], entries kBlock apart: [0, kLdsStackDepth) stack, then the leaf queue
    uint32_t* overflow;   // &overflow[global thread], entries `stride` apart (worst-case depth)
    uint32_t lds_stride, stride;
    uint32_t* wave;       // 128 words shared by the wavefront (occluded() only)
    RT_DEV void put(int sp, uint32_t v) const {
        if (sp < kLdsStackDepth) lds[sp * lds_stride] = v; else overflow[(size_t)(sp - kLdsStackDepth) * stride] = v;
    }
    RT_DEV uint32_t get(int sp) const {
        return sp < kLdsStackDepth ? lds[sp * lds_stride] : overflow[(size_t)(sp - kLdsStackDepth) * stride];
    }
};
struct TraceRay {
    f3 o, d, inv, oi;
    uint32_t sign;
};
RT_DEV TraceRay make_trace_ray(f3 o, f3 d) {
    TraceRay r;
    r.o = o; r.d = d;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.oi = mk3(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
    r.sign = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    return r;
}

// (1): slab tests of the 8 slots of block `cur` in ray order; bit k of the masks = ray-order position k
RT_DEV void visit_block(const DScene& S, uint32_t cur, const TraceRay& r, float t_enter, float entry_best, uint32_t ref[8],
                        uint32_t* leaf_mask, uint32_t* inner_mask) {
    RT_PROF_POINT(0);  // block visits
    const uint32_t first = cur >> 8, present = cur & 0xffu;
    // All child records are fetched before any is tested, with no branch around the loads: one memory
    // latency per block instead of one per child.  A slot's record is first + (present slots below it);
    // for an absent slot that is the record of the next present slot, or the record after the block's
    // last (the array is padded by one), fetched and then masked out.
    const uint32_t leaves = __float_as_uint(S.recs[2 * (size_t)first + 1].w);
    uint32_t hit = 0;
#pragma unroll
    for (int h = 0; h < 8; h += kVisitBatch) {  // batches of kVisitBatch records: fewer registers in flight
        float4 ra[kVisitBatch], rb[kVisitBatch];
#pragma unroll
        for (int k = 0; k < kVisitBatch; ++k) {
            const uint32_t oct = (uint32_t)(h + k) ^ r.sign;
            const size_t idx = (size_t)first + (uint32_t)__popc(present & ((1u << oct) - 1u));
            ra[k] = S.recs[2 * idx];
            rb[k] = S.recs[2 * idx + 1];
        }
#pragma unroll
        for (int k = 0; k < kVisitBatch; ++k) {
            const float4 a = ra[k], b = rb[k];
            float tn = t_enter, tf = entry_best;
            {
                const float lo = fmaf(a.x, r.inv.x, -r.oi.x), hi = fmaf(a.w, r.inv.x, -r.oi.x);
                tn = fmaxf(tn, fminf(lo, hi)); tf = fminf(tf, fmaxf(lo, hi));
            }
            {
                const float lo = fmaf(a.y, r.inv.y, -r.oi.y), hi = fmaf(b.x, r.inv.y, -r.oi.y);
                tn = fmaxf(tn, fminf(lo, hi)); tf = fminf(tf, fmaxf(lo, hi));
            }
            {
                const float lo = fmaf(a.z, r.inv.z, -r.oi.z), hi = fmaf(b.y, r.inv.z, -r.oi.z);
                tn = fmaxf(tn, fminf(lo, hi)); tf = fminf(tf, fmaxf(lo, hi));
            }
            hit |= tn <= tf ? (1u << (h + k)) : 0u;
            ref[h + k] = __float_as_uint(b.z);
        }
        asm volatile("" ::: "memory");
    }
    // present and leaf masks are in octant order; ray-order position k holds octant k ^ sign, and that
    // permutation of the bits is three conditional swaps (halves, pairs, neighbours) of both masks at once
    uint32_t m = present | ((leaves & 0xffu) << 8);
    const uint32_t s4 = r.sign & 4u, s2 = r.sign & 2u, s1 = r.sign & 1u;
    m = ((m >> s4) & 0x0f0fu) | ((m << s4) & 0xf0f0u);
    m = ((m >> s2) & 0x3333u) | ((m << s2) & 0xccccu);
    m = ((m >> s1) & 0x5555u) | ((m << s1) & 0xaaaau);
    hit &= m;  // (bits 8..15 of hit are zero)
    *leaf_mask = hit & (m >> 8);
    *inner_mask = hit & ~(m >> 8);
}
// (2): branch-free append in ray order — every slot stores at the queue's end, which only moves for hit leaves
RT_DEV void queue_leaves(uint32_t* pend, uint32_t stride, const uint32_t ref[8], uint32_t leaf_mask, uint32_t* npend) {
    if (leaf_mask) {
        uint32_t n = *npend;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pend[n * stride] = ref[k];
            n += (leaf_mask >> k) & 1u;
        }
        *npend = n;
    }
}
// (3): far-to-near pushes of every hit inner child; the next block is the top of the stack, i.e. the nearest
// of them, or an older entry when none was hit (so the stack holds one entry more than
// FlatTree::stack_bound for a moment).  Returns false when there is no next block.
RT_DEV bool next_block(const TraceStack& st, const uint32_t ref[8], uint32_t inner_mask, uint32_t* cur, int* sp_io) {
    int sp = *sp_io;
    if (inner_mask) {
        if (sp + 8 <= kLdsStackDepth) {
            // common case, no branch per entry: every slot stores at the current top and the top only
            // moves for the slots that were hit; the nearest is pushed last and popped right away
#pragma unroll
            for (int k = 7; k >= 0; --k) {
                st.lds[sp * st.lds_stride] = ref[k];
                sp += (int)((inner_mask >> k) & 1u);
            }
        } else {
#pragma unroll
            for (int k = 7; k >= 0; --k)
                if ((inner_mask >> k) & 1u) st.put(sp++, ref[k]);
        }
    }
    if (sp == 0) return false;
    *cur = st.get(--sp);
    *sp_io = sp;
    return true;
}

// Nearest hit (camera and scattered rays).  Accepts t > 0 && t < best (strict: first found wins ties,
// accelerationstructures.go:101).
RT_DEV bool traverse(const DScene& S, const TraceStack& st, f3 o, f3 d, float tmin, float* t_out, int* prim_out) {
    const TraceRay r = make_trace_ray(o, d);
    float best = 3.402823466e+38f;
    int best_prim = -1;
    uint32_t cur = S.root_ref;
    int sp = 0;
    bool walking = true;   // has a block to visit
    uint32_t npend = 0;    // queued leaves
    uint32_t* const pend = st.lds + kLdsStackDepth * st.lds_stride;
    // the shape loop: one flat loop over the shapes of all queued leaves, in queue order
    auto flush = [&]() {
        uint32_t qhead = 0, pnext = 0, pleft = 0;
        for (;;) {
            if (pleft == 0) {
                if (qhead == npend) break;
                const uint32_t leaf = pend[qhead * st.lds_stride];
                ++qhead;
                pnext = leaf >> 8;  // the leaf's shapes are contiguous
                pleft = leaf & 0xffu;
            }
            RT_PROF_POINT(1);  // shape tests
            const uint32_t prim = pnext++;
            --pleft;
            const float t = prim_hit(S, prim, o, d, tmin, best);
            if (t > 0.0f && t < best) { best = t; best_prim = (int)prim; }
        }
        npend = 0;
    };
    for (;;) {
        uint32_t ref[8];
        uint32_t leaf_mask = 0, inner_mask = 0;
        if (walking) visit_block(S, cur, r, 0.0f, best, ref, &leaf_mask, &inner_mask);
        // room for this visit's leaves in every lane's queue?
        if (__ballot(npend + (uint32_t)__popc(leaf_mask) > (uint32_t)kPendCap) != 0ull) flush();
        queue_leaves(pend, st.lds_stride, ref, leaf_mask, &npend);
        if (walking) walking = next_block(st, ref, inner_mask, &cur, &sp);
        const int n_walking = __popcll(__ballot(walking));
        const int n_holding = __popcll(__ballot(npend > 0u));
        if (n_holding > 0 && n_holding >= (n_walking < kFlushLanesNearest ? n_walking : kFlushLanesNearest)) flush();
        if (!walking && npend == 0u) break;
    }
    *t_out = best;
    *prim_out = best_prim;
    return best_prim >= 0;
}

// ---- leaf slabs -------------------------------------------------------------------------------
// A leaf's axis-aligned box is loose around a tilted patch of surface: on config 3 a shadow ray that ends up occluded
// met 4.5 leaf boxes and tested 19.5 shapes, an unoccluded one 1.0 and 4.9 (scripts/sim_leaf_slab.py).  Every leaf
// therefore has a second bound, two parallel planes around its shapes, {x : dmin <= n . (x - box.min) <= dmax} with n
// the leaf's mean triangle normal (DScene::leaf_bound, made on the device by k_leaf_bounds): a ray can only hit a
// shape of the leaf where it is inside the leaf's box AND between the planes; a second pair of planes, normal to the
// in-plane direction in which the leaf is narrowest, bounds the leaf inside the first (a third fewer shape tests again).
// With the first slab the same rays test 7.2 and
// 0.1 shapes, and a pair's beam (below) that meets a leaf's box but not its slab there goes on: 64 % instead of 53 %
// of the shadow rays belong to pairs that a beam proves clear.  The tests only ever say "cannot hit" with margins
// (2^-16 of the distances involved, against float32 rounding of 2^-23 in the shape tests), so the answers are the
// ones the shape tests alone would give; a sphere in a leaf adds its extent along n; a leaf of spheres only has no slab (n = 0, dmin = -inf, dmax = +inf).
constexpr float kSlabMargin = 1.52587890625e-05f;  // 2^-16
// true: the ray cannot hit a shape of the leaf.  Box interval in local coordinates ((plane - o) * inv: a rounding
// error relative to the distance, not to |o|), never used to cull by itself; the slab is evaluated at its two ends.
RT_DEV bool ray_misses_leaf_slab(const float4 b0, const float4 b1, const float4 b2, const float4 b3, f3 o, f3 d, f3 inv, float tmin, float tmax) {
    // With d_a = 0 (1 / d_a an infinity) a plane's distance is an infinity of the right sign — or a NaN beside the other
    // plane's infinity when the origin lies exactly ON the plane, and fminf / fmaxf would then keep the infinity: a ray
    // that runs along a face of the leaf's box, where a triangle's edge can lie, would count as outside
    // (scripts/stress_slabs.py found such rays 10^5 units from the origin, where coordinates are multiples of 2^-7).
    // So the filed box is 8 ulps of its coordinates larger than the leaf's (k_leaf_bounds): on its own faces no shape lies.
    const float x0 = (b0.x - o.x) * inv.x, x1 = (b0.w - o.x) * inv.x;
    const float y0 = (b0.y - o.y) * inv.y, y1 = (b1.x - o.y) * inv.y;
    const float z0 = (b0.z - o.z) * inv.z, z1 = (b1.y - o.z) * inv.z;
    const float tn0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
    const float tf0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    const float tn = fmaf(-4.0f * kSlabMargin, fabsf(tn0), tn0), tf = fmaf(4.0f * kSlabMargin, fabsf(tf0), tf0);  // widened, whatever their signs
    const f3 n = mk3(b1.z, b1.w, b2.x);
    const f3 rel = mk3(o.x - b0.x, o.y - b0.y, o.z - b0.z);
    const float nob = dotf(n, rel), nd = dotf(n, d);
    const float a0 = fmaf(tn, nd, nob), a1 = fmaf(tf, nd, nob);
    const float m = kSlabMargin * ((fabsf(rel.x) + fabsf(rel.y)) + fabsf(rel.z));  // |n| <= 1: the size of the terms that cancel
    // (comparisons with a NaN are false: such a leaf is kept)
    if (fmaxf(a0, a1) + m < b2.y || fminf(a0, a1) - m > b2.z) return true;
    // The second slab, along the in-plane direction n2 in which the leaf is narrowest, over the piece of [tn, tf] on
    // which the ray is between the first slab's (widened) planes: t = (plane - n . rel) / (n . d).
    const float ind = 1.0f / nd;  // (n . d = 0: infinities, or a NaN that fminf / fmaxf drop; a piece that comes out empty decides nothing, below)
    const float ta = ((b2.y - m) - nob) * ind, tb = ((b2.z + m) - nob) * ind;
    const float sn0 = fmaxf(tn, fminf(ta, tb)), sf0 = fminf(tf, fmaxf(ta, tb));
    if (!(sn0 <= sf0)) return false;  // no piece left, though the first slab is met: rounding at an edge; the second slab then decides nothing
    const float sn = fmaf(-4.0f * kSlabMargin, fabsf(sn0), sn0), sf = fmaf(4.0f * kSlabMargin, fabsf(sf0), sf0);
    const f3 n2 = mk3(b2.w, b3.x, b3.y);
    const float nob2 = dotf(n2, rel), nd2 = dotf(n2, d);
    const float c0 = fmaf(sn, nd2, nob2), c1 = fmaf(sf, nd2, nob2);
    return fmaxf(c0, c1) + m < b3.z || fminf(c0, c1) - m > b3.w;
}

// ---- one conservative walk for all the shadow rays of a (hit, light) pair ------------------------------------
// Every jittered light position of a pair lies in the cube of half-width R around the light (light.go:94: the light
// moved by 0.25 * randomInUnitSphere), so every shadow ray of the pair lies in the beam
//     { p + s (D + j) : s0 <= s <= s1, |j_a| <= R },   D = light - p.
// At parameter s the beam spans [p_a + s (D_a - R), p_a + s (D_a + R)] on axis a; it meets a box's slab there iff
// s (D_a - R) <= hi_a - p_a and s (D_a + R) >= lo_a - p_a, two half-lines in s.  With both coefficients positive the
// second bounds s from below through the LOWER plane and the first from above through the upper plane — a ray's slab
// test with different divisors for entry and exit; both negative: the planes swap roles; of different sign (the beam
// straddles the axis): both bound s from below, one is kept, and nothing bounds it from above.  If the bounds of the
// three axes and [s0, s1] leave no s, no ray of the pair meets the box.  A walk over the EXACT boxes that meets no leaf
// proves every ray of the pair unoccluded (shapes lie inside their leaves' exact boxes; the quantised boxes of the
// per-ray walk are only larger); it stops at the first leaf box met, and the pair's rays are then traced one by one.
// R, s0 and s1 carry margins far above float32 rounding.  (CPU simulation on config 3, scripts/sim_groupwalk.cpp
// sim_ball_beam: 55 % of the pairs are proven clear by 4.7 block visits each; the others give up after 8.9.)
struct Beam {
    f3 inv_n, c_n, inv_f, c_f;  // entry bound = near plane * inv_n + c_n, exit bound = far plane * inv_f + c_f
    uint32_t sign;              // bit a: the near plane on axis a is the UPPER one
    float s0, s1;
};
// A bound is fma(plane, 1 / c, -(P / c)) with P / c rounded BEFORE the subtraction: it errs by up to an ulp of P / c,
// which far from the origin is more than a leaf is wide in s (scripts/stress_slabs.py: at 10^5 units a beam passed by a
// box that it met).  Every entry bound is therefore lowered and every exit bound raised by 4 ulps of the largest P / c
// in use, once per beam: nothing at ordinary distances (10^-6 of the way to the light), a fatter beam far out.
RT_DEV void beam_slack(float cn[3], float cf[3]) {
    const float inf = __builtin_inff();
    float m = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (fabsf(cn[a]) < inf) m = fmaxf(m, fabsf(cn[a]));
        if (fabsf(cf[a]) < inf) m = fmaxf(m, fabsf(cf[a]));
    }
    const float slack = m * 9.5367431640625e-07f;  // 2^-20
#pragma unroll
    for (int a = 0; a < 3; ++a) { cn[a] -= slack; cf[a] += slack; }  // (an infinite bound stays what it is; a NaN never bounded anything)
}
RT_DEV Beam make_beam(f3 p, f3 light, float R, float tmin) {
    Beam b;
    const float D[3] = {light.x - p.x, light.y - p.y, light.z - p.z}, P[3] = {p.x, p.y, p.z};
    float in[3], cn[3], jf[3], cf[3];
    b.sign = 0u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float c1 = D[a] - R, c2 = D[a] + R;
        const float inf = __builtin_inff();
        if (c1 > 0.0f) {            // entry through the lower plane (divisor c2), exit through the upper (c1)
            in[a] = 1.0f / c2; jf[a] = 1.0f / c1; cf[a] = -P[a] * jf[a];
        } else if (c2 < 0.0f) {     // entry through the upper plane (divisor c1), exit through the lower (c2)
            in[a] = 1.0f / c1; jf[a] = 1.0f / c2; cf[a] = -P[a] * jf[a];
            b.sign |= 1u << a;
        } else {                    // straddles: one of the two lower bounds, no upper bound
            if (D[a] < 0.0f) { in[a] = 1.0f / c1; b.sign |= 1u << a; } else in[a] = 1.0f / c2;
            jf[a] = 0.0f; cf[a] = inf;
        }
        cn[a] = -P[a] * in[a];
    }
    beam_slack(cn, cf);
    b.inv_n = mk3(in[0], in[1], in[2]); b.c_n = mk3(cn[0], cn[1], cn[2]);
    b.inv_f = mk3(jf[0], jf[1], jf[2]); b.c_f = mk3(cf[0], cf[1], cf[2]);
    const float len = sqrtf(D[0] * D[0] + D[1] * D[1] + D[2] * D[2]);
    b.s0 = 0.999f * tmin / (len + 1.7320508f * R);
    b.s1 = 1.0001f;
    return b;
}
// The same for every ray from ANY point of the box [vlo, vhi] to any jittered light position (the voxel beams of
// k_voxel_beams): at parameter s the beam spans [vlo_a + s (L_a - R - vlo_a), vhi_a + s (L_a + R - vhi_a)] on axis a, so
// it meets a box's slab iff s c1 <= hi_a - vlo_a and s c2 >= lo_a - vhi_a with c1 = L_a - R - vlo_a, c2 = L_a + R - vhi_a:
// make_beam's two half-lines with a point of their own each (vlo = vhi = p gives make_beam).  A box wider than 2R can
// have c1 > 0 > c2: two upper bounds, of which one is kept, and no lower bound.
RT_DEV Beam make_box_beam(f3 vlo3, f3 vhi3, f3 light, float R, float tmin) {
    Beam b;
    const float lo[3] = {vlo3.x, vlo3.y, vlo3.z}, hi[3] = {vhi3.x, vhi3.y, vhi3.z}, Lp[3] = {light.x, light.y, light.z};
    float in[3], cn[3], jf[3], cf[3], far2 = 0.0f;
    b.sign = 0u;
    const float inf = __builtin_inff();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float c1 = (Lp[a] - R) - lo[a], c2 = (Lp[a] + R) - hi[a];
        if (c1 > 0.0f && c2 > 0.0f) {
            in[a] = 1.0f / c2; cn[a] = -hi[a] * in[a]; jf[a] = 1.0f / c1; cf[a] = -lo[a] * jf[a];
        } else if (c1 < 0.0f && c2 < 0.0f) {
            in[a] = 1.0f / c1; cn[a] = -lo[a] * in[a]; jf[a] = 1.0f / c2; cf[a] = -hi[a] * jf[a];
            b.sign |= 1u << a;
        } else if (c1 <= 0.0f && c2 >= 0.0f) {
            if (Lp[a] - 0.5f * (lo[a] + hi[a]) < 0.0f) { in[a] = 1.0f / c1; cn[a] = -lo[a] * in[a]; b.sign |= 1u << a; } else { in[a] = 1.0f / c2; cn[a] = -hi[a] * in[a]; }
            jf[a] = 0.0f; cf[a] = inf;
        } else {
            in[a] = 0.0f; cn[a] = -inf; jf[a] = 1.0f / c1; cf[a] = -lo[a] * jf[a];
        }
        const float reach = fmaxf(fabsf(Lp[a] - lo[a]), fabsf(Lp[a] - hi[a])) + R;  // the farthest a ray of the beam goes on this axis
        far2 += reach * reach;
    }
    beam_slack(cn, cf);
    b.inv_n = mk3(in[0], in[1], in[2]); b.c_n = mk3(cn[0], cn[1], cn[2]);
    b.inv_f = mk3(jf[0], jf[1], jf[2]); b.c_f = mk3(cf[0], cf[1], cf[2]);
    b.s0 = 0.999f * tmin / sqrtf(far2);
    b.s1 = 1.0001f;
    return b;
}
// ---- the shadow-ray walk over quantised blocks ------------------------------------------------
// k_shadow is bound by the vector-memory path (TA busy 77-90 %, about one L1 access per clock and CU,
// profiles/r02): a visit of the exact blocks issues 17 loads of 16 bytes per lane.  A shadow ray only asks
// "is anything hit", an OR over shapes, so its walk may use any boxes that ENCLOSE the children: here the 8
// slots' boxes as bytes on a power-of-two grid anchored at the block's minimum corner (octree.hpp QBlock),
// 5 loads per visit.  A box that is a little too large costs a visit, never a hit.
struct QRay {
    f3 inv, oi;      // 1 / d and o / d as in TraceRay
    uint32_t sign;   // bit a: the SIGN BIT of d on axis a (so -0 counts as negative, like 1 / -0 = -inf)
};
RT_DEV QRay make_qray(f3 o, f3 d) {
    QRay r;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.oi = mk3(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
    r.sign = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    return r;
}
RT_DEV float qbyte(uint32_t w, int b) { return (float)((w >> (8 * b)) & 0xffu); }  // v_cvt_f32_ubyteN
constexpr int kQBlockQuads = 5;
// Slab tests of the 8 slots of block `blk` in SLOT order.  hit: bit per slot whose box the ray meets inside
// [tmin, tmax]; h1 = {child base | leaf mask << 24, first leaf shape, meta bytes of slots 0..3, 4..7}.
RT_DEV uint32_t visit_qblock(const DScene& S, uint32_t blk, const QRay& r, float tmin, float tmax, uint4* h1_out, uint32_t* present_out) {
    RT_PROF_POINT(0);  // block visits
    const uint4* q = S.qblocks + kQBlockQuads * (size_t)blk;
    const uint4 h0 = q[0], h1 = q[1], qa = q[2], qb = q[3], qc = q[4];
    // t = ((origin + q * step) - o) / d = q * (step / d) + (origin / d - o / d)
    const float sx = __uint_as_float((h0.w & 0xffu) << 23) * r.inv.x;
    const float sy = __uint_as_float((h0.w & 0xff00u) << 15) * r.inv.y;
    const float sz = __uint_as_float((h0.w & 0xff0000u) << 7) * r.inv.z;
    const float bx = fmaf(__uint_as_float(h0.x), r.inv.x, -r.oi.x);
    const float by = fmaf(__uint_as_float(h0.y), r.inv.y, -r.oi.y);
    const float bz = fmaf(__uint_as_float(h0.z), r.inv.z, -r.oi.z);
    // the plane the ray enters through is the lower one for a positive direction, the upper one for a negative
    const bool nx = (r.sign & 1u) != 0u, ny = (r.sign & 2u) != 0u, nz = (r.sign & 4u) != 0u;
    uint32_t near_[3][2], far_[3][2];
    near_[0][0] = nx ? qb.z : qa.x; near_[0][1] = nx ? qb.w : qa.y; far_[0][0] = nx ? qa.x : qb.z; far_[0][1] = nx ? qa.y : qb.w;
    near_[1][0] = ny ? qc.x : qa.z; near_[1][1] = ny ? qc.y : qa.w; far_[1][0] = ny ? qa.z : qc.x; far_[1][1] = ny ? qa.w : qc.y;
    near_[2][0] = nz ? qc.z : qb.x; near_[2][1] = nz ? qc.w : qb.y; far_[2][0] = nz ? qb.x : qc.z; far_[2][1] = nz ? qb.y : qc.w;
    uint32_t hit = 0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {  // slot 7 first: every slot shifts the mask left and adds its bit
        // a NaN (0 * inf for a direction component of 0) drops out of fmaxf / fminf: that axis then culls nothing
        const int w = k >> 2, b = k & 3;
        const float tn = fmaxf(fmaxf(fmaf(qbyte(near_[0][w], b), sx, bx), fmaf(qbyte(near_[1][w], b), sy, by)),
                               fmaxf(fmaf(qbyte(near_[2][w], b), sz, bz), tmin));
        const float tf = fminf(fminf(fmaf(qbyte(far_[0][w], b), sx, bx), fmaf(qbyte(far_[1][w], b), sy, by)),
                               fminf(fmaf(qbyte(far_[2][w], b), sz, bz), tmax));
#ifdef RTGO_HITMASK_CARRY
        // hit = 2 * hit + (tn <= tf): the compare's lane mask goes in as the carry of an add (k_shadow 354 -> 353 ms
        // against compare + select + or)
        asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hit) : "v"(tn), "v"(tf) : "vcc");
#else
        // the sign of tf - tn shifted in, so a set bit is a MISS (tn and tf are never NaN after fmaxf / fminf against
        // tmin and tmax, and tf - tn is -0 only for tf = -0 < tmin): a subtraction and a v_alignbit_b32 per slot
        // (k_shadow 351.8 -> 347.2 ms against the compare-and-carry form)
        hit = __builtin_amdgcn_alignbit(hit, __float_as_uint(tf - tn), 31u);
#endif
    }
    *h1_out = h1;
    *present_out = h0.w >> 24;
#ifndef RTGO_HITMASK_CARRY
    hit = ~hit;
#endif
    return hit & (h0.w >> 24);
}
// slot-order mask -> ray-order mask: position k holds slot k ^ sign (three conditional swaps)
RT_DEV uint32_t ray_order8(uint32_t m, uint32_t sign) {
    const uint32_t s4 = sign & 4u, s2 = sign & 2u, s1 = sign & 1u;
    m = ((m >> s4) & 0x0fu) | ((m << s4) & 0xf0u);
    m = ((m >> s2) & 0x33u) | ((m << s2) & 0xccu);
    m = ((m >> s1) & 0x55u) | ((m << s1) & 0xaau);
    return m;
}
// Hit leaf children -> the lane's leaf queue as (first shape << 8 | count), in slot order (the order of a shadow
// ray's shape tests is free).  The leaves' shapes are consecutive: first = base + counts of the leaves before it;
// the low 5 bits of a meta byte are the shape count of a leaf and 0 for an inner or absent slot, so a byte-wise
// running sum over the two meta words gives every slot's offset.  One iteration per hit leaf (0.4 per visit).
RT_DEV void queue_qleaves(uint32_t* pend, uint32_t stride, const uint4& h1, uint32_t leaf_hits, uint32_t* npend) {
    if (leaf_hits == 0u) return;
    const uint32_t c_lo = h1.z & 0x1f1f1f1fu, c_hi = h1.w & 0x1f1f1f1fu;
    uint32_t i_lo = c_lo + (c_lo << 8); i_lo += i_lo << 16;  // inclusive byte-wise sums (8 x 31 < 256: no carry)
    uint32_t i_hi = c_hi + (c_hi << 8); i_hi += i_hi << 16;
    const uint32_t p_lo = i_lo << 8;                                       // exclusive sums of slots 0..3
    const uint32_t p_hi = (i_hi << 8) + (i_lo >> 24) * 0x01010101u;        // slots 4..7: + everything in slots 0..3
    uint32_t n = *npend;
    uint32_t m = leaf_hits;
    while (m) {
        const int k = __ffs((int)m) - 1;
        m &= m - 1u;
        const uint32_t sh = (uint32_t)(k & 3) * 8u;
        const uint32_t first = h1.y + (((k < 4 ? p_lo : p_hi) >> sh) & 0xffu);
        pend[n * stride] = (first << 8) | (((k < 4 ? c_lo : c_hi) >> sh) & 0x1fu);
        ++n;
    }
    *npend = n;
}
// Hit inner children: the first is entered, the others pushed last to first.  A child's block is the
// block's first child block + its rank among the inner children (slot order).
#ifdef RTGO_PUSH_BRANCHFREE
RT_DEV bool next_qblock(const TraceStack& st, const uint4& h1, uint32_t, uint32_t inner_hits, uint32_t sign, uint32_t* cur, int* sp_io) {
    int sp = *sp_io;
    if (inner_hits) {
        const uint32_t m = ray_order8(inner_hits, sign);
        const uint32_t base = h1.x & 0xffffffu;
        // meta bytes brought to ray order: position i <- slot i ^ sign
        const uint32_t sel = 0x03020100u ^ (sign * 0x01010101u);
        const uint32_t mlo = __builtin_amdgcn_perm(h1.w, h1.z, sel), mhi = __builtin_amdgcn_perm(h1.w, h1.z, sel ^ 0x04040404u);
        if (sp + 8 <= kLdsStackDepth) {
#pragma unroll
            for (int k = 7; k >= 0; --k) {
                const uint32_t rank = ((k < 4 ? mlo : mhi) >> (8 * (k & 3) + 5)) & 7u;
                st.lds[sp * st.lds_stride] = base + rank;
                sp += (int)((m >> k) & 1u);
            }
        } else {
#pragma unroll
            for (int k = 7; k >= 0; --k)
                if ((m >> k) & 1u) st.put(sp++, base + (((k < 4 ? mlo : mhi) >> (8 * (k & 3) + 5)) & 7u));
        }
    }
    if (sp == 0) return false;
    *cur = st.get(--sp);
    *sp_io = sp;
    return true;
}
#else
// One loop iteration per hit inner child beyond the nearest (0.9 hit inner children per visit on average): the
// branch-free form above stores and counts all 8 slots on every visit (66 VALU + 9 LDS operations).
RT_DEV bool next_qblock(const TraceStack& st, const uint4& h1, uint32_t present, uint32_t inner_hits, uint32_t sign, uint32_t* cur, int* sp_io) {
    int sp = *sp_io;
    if (inner_hits) {
        // in SLOT order, whatever the ray's direction: the answer is an OR over the subtrees, and bringing the mask into
        // ray order (16 instructions per visit, + 1 per child for its slot) bought these walks nothing — config 3,
        // k_shadow alone: 256.1 ms in octant order, 256.5 with the whole order reversed for rays that go down two or
        // three axes, 252.0 like this
#ifdef RTGO_RAY_ORDER
        uint32_t m = ray_order8(inner_hits, sign);
#else
        uint32_t m = inner_hits;
        sign = 0u;
#endif
        const uint32_t base = h1.x & 0xffffffu;
        const uint32_t inner = present & ~(h1.x >> 24);  // the slots that hold inner children
        const int kn = __ffs((int)m) - 1;             // entered now: the lowest position
        m &= m - 1u;
        while (m) {
            const int k = 31 - __clz((int)m);          // highest first
            m ^= 1u << k;
            const uint32_t slot = (uint32_t)k ^ sign;
            st.put(sp++, base + (uint32_t)__popc(inner & ((1u << slot) - 1u)));
        }
        const uint32_t slot = (uint32_t)kn ^ sign;
        *cur = base + (uint32_t)__popc(inner & ((1u << slot) - 1u));
        *sp_io = sp;
        return true;
    }
    if (sp == 0) return false;
    *cur = st.get(--sp);
    *sp_io = sp;
    return true;
}
#endif

// visit_qblock() for a beam: the slots whose quantised box the beam meets, in SLOT order (the quantised boxes enclose
// the exact ones, so a miss here is a miss there)
RT_DEV uint32_t beam_visit_qblock(const DScene& S, uint32_t blk, const Beam& bm, uint4* h1_out, uint32_t* present_out) {
    const uint4* q = S.qblocks + kQBlockQuads * (size_t)blk;
    const uint4 h0 = q[0], h1 = q[1], qa = q[2], qb = q[3], qc = q[4];
    const float ex = __uint_as_float((h0.w & 0xffu) << 23), ey = __uint_as_float((h0.w & 0xff00u) << 15), ez = __uint_as_float((h0.w & 0xff0000u) << 7);
    // bound = (origin + q * step) * inv + c, entry and exit with their own inv and c (Beam)
    const float snx = ex * bm.inv_n.x, sny = ey * bm.inv_n.y, snz = ez * bm.inv_n.z;
    const float sfx = ex * bm.inv_f.x, sfy = ey * bm.inv_f.y, sfz = ez * bm.inv_f.z;
    const float bnx = fmaf(__uint_as_float(h0.x), bm.inv_n.x, bm.c_n.x), bny = fmaf(__uint_as_float(h0.y), bm.inv_n.y, bm.c_n.y), bnz = fmaf(__uint_as_float(h0.z), bm.inv_n.z, bm.c_n.z);
    const float bfx = fmaf(__uint_as_float(h0.x), bm.inv_f.x, bm.c_f.x), bfy = fmaf(__uint_as_float(h0.y), bm.inv_f.y, bm.c_f.y), bfz = fmaf(__uint_as_float(h0.z), bm.inv_f.z, bm.c_f.z);
    const bool nx = (bm.sign & 1u) != 0u, ny = (bm.sign & 2u) != 0u, nz = (bm.sign & 4u) != 0u;
    uint32_t near_[3][2], far_[3][2];
    near_[0][0] = nx ? qb.z : qa.x; near_[0][1] = nx ? qb.w : qa.y; far_[0][0] = nx ? qa.x : qb.z; far_[0][1] = nx ? qa.y : qb.w;
    near_[1][0] = ny ? qc.x : qa.z; near_[1][1] = ny ? qc.y : qa.w; far_[1][0] = ny ? qa.z : qc.x; far_[1][1] = ny ? qa.w : qc.y;
    near_[2][0] = nz ? qc.z : qb.x; near_[2][1] = nz ? qc.w : qb.y; far_[2][0] = nz ? qb.x : qc.z; far_[2][1] = nz ? qb.y : qc.w;
    uint32_t miss = 0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        // a NaN (0 * inf, inf - inf) drops out of fmaxf / fminf: that bound then culls nothing
        const int w = k >> 2, b = k & 3;
        const float lo = fmaxf(fmaxf(fmaf(qbyte(near_[0][w], b), snx, bnx), fmaf(qbyte(near_[1][w], b), sny, bny)),
                               fmaxf(fmaf(qbyte(near_[2][w], b), snz, bnz), bm.s0));
        const float hi = fminf(fminf(fmaf(qbyte(far_[0][w], b), sfx, bfx), fmaf(qbyte(far_[1][w], b), sfy, bfy)),
                               fminf(fmaf(qbyte(far_[2][w], b), sfz, bfz), bm.s1));
        miss = (miss << 1) | (lo <= hi ? 0u : 1u);
    }
    *h1_out = h1;
    *present_out = h0.w >> 24;
    return ~miss & (h0.w >> 24);
}
// the beam against the EXACT box of child record `rec`
// (the walk without slabs, RTGO_SLAB=0; *first_shape = the leaf's first shape, [*lo_out, *hi_out] = the beam's parameters inside the box)
RT_DEV bool beam_meets_record(const DScene& S, uint32_t rec, const Beam& bm, uint32_t* first_shape, float* lo_out, float* hi_out) {
    const float4 a = S.recs[2 * (size_t)rec], b = S.recs[2 * (size_t)rec + 1];  // {min.xyz, max.x}, {max.y, max.z, w0, w1}
    const bool nx = (bm.sign & 1u) != 0u, ny = (bm.sign & 2u) != 0u, nz = (bm.sign & 4u) != 0u;
    const float ex = fmaf(nx ? a.w : a.x, bm.inv_n.x, bm.c_n.x), xx = fmaf(nx ? a.x : a.w, bm.inv_f.x, bm.c_f.x);
    const float ey = fmaf(ny ? b.x : a.y, bm.inv_n.y, bm.c_n.y), xy = fmaf(ny ? a.y : b.x, bm.inv_f.y, bm.c_f.y);
    const float ez = fmaf(nz ? b.y : a.z, bm.inv_n.z, bm.c_n.z), xz = fmaf(nz ? a.z : b.y, bm.inv_f.z, bm.c_f.z);
    const float lo = fmaxf(fmaxf(ex, ey), fmaxf(ez, bm.s0));
    const float hi = fminf(fminf(xx, xy), fminf(xz, bm.s1));
    *first_shape = __float_as_uint(b.z) >> 8;
    *lo_out = lo; *hi_out = hi;
    return lo <= hi;
}
// ---- voxel beams ----------------------------------------------------------------------------------
// A grid of cubic cells over the scene's box; per positional light and cell one beam from the WHOLE cell to the light's
// ball (make_box_beam), walked once per scene (k_voxel_beams): if it meets no leaf, every (hit, light) pair whose point
// lies in the cell is clear, and k_phong_pairs settles it without a beam walk of its own.  The cell's box as the beam saw
// it (padded: the point's cell index is rounded) is recomputed at the look-up, and the point must lie inside.
RT_DEV void vox_cell_box(const DScene& S, int ix, int iy, int iz, f3* lo, f3* hi) {
    const int i[3] = {ix, iy, iz};
    float l[3], h[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x0 = fmaf((float)i[a], S.vox_cs, S.vox_min[a]), x1 = fmaf((float)(i[a] + 1), S.vox_cs, S.vox_min[a]);
        const float pad = S.vox_cs * 9.765625e-4f + fmaxf(fabsf(x0), fabsf(x1)) * 9.5367431640625e-07f;  // 2^-10 of a cell + 2^-20 of the coordinate
        l[a] = x0 - pad; h[a] = x1 + pad;
    }
    *lo = mk3(l[0], l[1], l[2]); *hi = mk3(h[0], h[1], h[2]);
}
RT_DEV bool vox_clear(const DScene& S, f3 p, int lp) {
    if (!S.vox) return false;
    const float f[3] = {(p.x - S.vox_min[0]) * S.vox_inv_cs, (p.y - S.vox_min[1]) * S.vox_inv_cs, (p.z - S.vox_min[2]) * S.vox_inv_cs};
    int i[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(f[a] > -1.0f && f[a] < 16777216.0f)) return false;  // also NaN
        const int k = (int)f[a];
        i[a] = k < 0 ? 0 : (k >= S.vox_n[a] ? S.vox_n[a] - 1 : k);
    }
    f3 lo, hi;
    vox_cell_box(S, i[0], i[1], i[2], &lo, &hi);
    if (!(p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z)) return false;
    return S.vox[(((size_t)lp * S.vox_n[2] + i[2]) * S.vox_n[1] + i[1]) * S.vox_n[0] + i[0]] != 0;
}
// true: no leaf box of the tree meets the beam.  Every lane of the wavefront calls it (live = this lane has a beam).
// The walk goes over the quantised blocks (5 loads per visit); a leaf whose quantised box the beam meets is checked
// against its exact box (the quantised box of a wall's flat triangle is thick enough to be met by a beam that leaves
// the wall).  A walk that meets a leaf stops there and leaves its state for the pair's rays: every block it has been
// through held no leaf that the beam meets — so none that a ray of the pair could hit — and every inner child the beam
// met is either visited already or still on its stack.  The rays therefore need not start at the root: they resume at
// the block where the beam stopped, with the beam's stack (config 3, CPU simulation: 8.1 entries; 10.7 instead of 13.2
// block visits per ray, the visits of inherited blocks that a ray then misses included; 7.7 with those filtered out).
//   resume[0] = the block to enter << 8 | entries of the stack (kResumeRoot: start at the root instead),
//   resume[1 ..] = the stack, bottom first; all as indices of quantised blocks (below 2^24).  48 bytes per pair.
constexpr int kResumeWords = 12;
constexpr int kResumeStack = kLdsStackDepth < kResumeWords - 1 ? kLdsStackDepth : kResumeWords - 1;  // entries handed over at most
constexpr uint32_t kResumeRoot = 0xffffffffu;
constexpr uint32_t kResumeClear = 0xfffffffeu;  // the beam met no leaf: every ray of the pair is unoccluded, only the spot cone is left to test per ray
// The pair's point p and D = light - p wait in the lane's leaf-queue words of LDS (the beam walk queues no leaves), for
// the few leaves whose box the beam meets: kept in registers through the walk they would be spilled.
// (h: half the size of the box the beam starts from, k_voxel_beams; zero for a pair's beam)
RT_DEV void beam_store_origin(const TraceStack& st, f3 p, f3 D, f3 h = mk3(0.0f, 0.0f, 0.0f)) {
    uint32_t* const w = st.lds + kLdsStackDepth * st.lds_stride;
    w[0] = __float_as_uint(p.x); w[st.lds_stride] = __float_as_uint(p.y); w[2 * st.lds_stride] = __float_as_uint(p.z);
    w[3 * st.lds_stride] = __float_as_uint(D.x); w[4 * st.lds_stride] = __float_as_uint(D.y); w[5 * st.lds_stride] = __float_as_uint(D.z);
    w[6 * st.lds_stride] = __float_as_uint(h.x); w[7 * st.lds_stride] = __float_as_uint(h.y); w[8 * st.lds_stride] = __float_as_uint(h.z);
}
static_assert(kPendCap >= 8, "beam_store_origin() uses nine of the leaf-queue words (kPendCap + 1)");
// The beam against a leaf's exact box and its slab as filed in DScene::leaf_bound (one fetch).  [lo, hi] = the beam's
// parameters inside the box; no ray of the beam { p + s (D + j), |j_a| <= R } is between the slab's planes there if the
// range of n . (x - box.min) over the beam's points misses [dmin, dmax]: at parameter s that range is
// n . (p - box.min) + s (n . D -/+ R |n|_2), linear in s, so the two ends of [lo, hi] bound it.
RT_DEV bool beam_meets_leaf(const DScene& S, const TraceStack& st, uint32_t first_shape, const Beam& bm) {
    const float4* const lb = S.leaf_bound + kLeafBoundQuads * (size_t)first_shape;
    const float4 a = lb[0], b = lb[1], c = lb[2], q3 = lb[3];
    const uint32_t* const w = st.lds + kLdsStackDepth * st.lds_stride;
    const f3 p = mk3(__uint_as_float(w[0]), __uint_as_float(w[st.lds_stride]), __uint_as_float(w[2 * st.lds_stride]));
    const f3 D = mk3(__uint_as_float(w[3 * st.lds_stride]), __uint_as_float(w[4 * st.lds_stride]), __uint_as_float(w[5 * st.lds_stride]));
    const f3 h = mk3(__uint_as_float(w[6 * st.lds_stride]), __uint_as_float(w[7 * st.lds_stride]), __uint_as_float(w[8 * st.lds_stride]));
    const bool nx = (bm.sign & 1u) != 0u, ny = (bm.sign & 2u) != 0u, nz = (bm.sign & 4u) != 0u;
    const float ex = fmaf(nx ? a.w : a.x, bm.inv_n.x, bm.c_n.x), xx = fmaf(nx ? a.x : a.w, bm.inv_f.x, bm.c_f.x);
    const float ey = fmaf(ny ? b.x : a.y, bm.inv_n.y, bm.c_n.y), xy = fmaf(ny ? a.y : b.x, bm.inv_f.y, bm.c_f.y);
    const float ez = fmaf(nz ? b.y : a.z, bm.inv_n.z, bm.c_n.z), xz = fmaf(nz ? a.z : b.y, bm.inv_f.z, bm.c_f.z);
    const float lo = fmaxf(fmaxf(ex, ey), fmaxf(ez, bm.s0));  // (widened by the beam's slack already: beam_slack)
    const float hi = fminf(fminf(xx, xy), fminf(xz, bm.s1));
    if (!(lo <= hi)) return false;
    const f3 n = mk3(b.z, b.w, c.x);
    const f3 rel = mk3(p.x - a.x, p.y - a.y, p.z - a.z);
    const float nob = dotf(n, rel), nD = dotf(n, D);
    // the jittered light positions lie in the BALL of radius R = S.beam_r (light.go:94: 0.25, + the rounding of light + jitter), so
    // |n . j| <= R |n|_2, and |n|_2 <= 1 + 2^-22: n1 = S.beam_n1 = 1.00001 R (a scalar of the scene, so it costs the walk no register)
    const float n1 = S.beam_n1;
    const float s_a = lo - 64.0f * kSlabMargin * fabsf(lo), s_b = hi + 64.0f * kSlabMargin * fabsf(hi);
    // (a beam that starts from a box: its points at s have n . (x - box.min) within (1 - s) (n . rel -/+ e) + s (n . rel + n . D -/+ n1),
    // e = the box's half extent along n)
    const float e1 = (fabsf(n.x) * h.x + fabsf(n.y) * h.y) + fabsf(n.z) * h.z;
    const float vlo = fminf(fmaf(s_a, (nD - n1) + e1, nob - e1), fmaf(s_b, (nD - n1) + e1, nob - e1));
    const float vhi = fmaxf(fmaf(s_a, (nD + n1) - e1, nob + e1), fmaf(s_b, (nD + n1) - e1, nob + e1));
    const float m = kSlabMargin * ((((fabsf(rel.x) + fabsf(rel.y)) + fabsf(rel.z)) + ((h.x + h.y) + h.z)) + s_b * (((fabsf(D.x) + fabsf(D.y)) + fabsf(D.z)) + S.beam_3n1));
    if (vhi + m < c.y || vlo - m > c.z) return false;
#if !RTGO_BEAM_SLAB2
    (void)q3;
    return true;
#endif
    // the second slab (the in-plane direction n2 in which the leaf is narrowest), over the same [lo, hi]
    const f3 n2 = mk3(c.w, q3.x, q3.y);
    const float nob2 = dotf(n2, rel), nD2 = dotf(n2, D);
    const float e2 = (fabsf(n2.x) * h.x + fabsf(n2.y) * h.y) + fabsf(n2.z) * h.z;
    const float wlo = fminf(fmaf(s_a, (nD2 - n1) + e2, nob2 - e2), fmaf(s_b, (nD2 - n1) + e2, nob2 - e2));
    const float whi = fmaxf(fmaf(s_a, (nD2 + n1) - e2, nob2 + e2), fmaf(s_b, (nD2 + n1) - e2, nob2 + e2));
    return !(whi + m < q3.z || wlo - m > q3.w);
}
RT_DEV bool beam_is_clear(const DScene& S, const TraceStack& st, bool live, const Beam& bm, uint32_t* cur_out, int* sp_out) {
    uint32_t cur = 0u;  // block index; the root is block 0
    int sp = 0;
    bool walking = live, clear = true;
#if RTGO_BEAM_CAP > 0
    int budget = RTGO_BEAM_CAP;  // block visits after which a walk gives up: the pair's rays then resume at the block it was about to enter
#endif
    while (__ballot(walking) != 0ull) {
#if RTGO_BEAM_CAP > 0
        if (walking && budget-- == 0) { clear = false; walking = false; }
#endif
        if (walking) {
            uint4 h1;
            uint32_t present;
            const uint32_t hit = beam_visit_qblock(S, cur, bm, &h1, &present);
            uint32_t leaf_hits = hit & (h1.x >> 24);
            bool met = false;
            if (leaf_hits) {
                if (S.leaf_bound) {
                    // a leaf's first shape as in queue_qleaves(): the block's first leaf shape + the counts of the leaves before it
                    const uint32_t c_lo = h1.z & 0x1f1f1f1fu, c_hi = h1.w & 0x1f1f1f1fu;
                    uint32_t i_lo = c_lo + (c_lo << 8); i_lo += i_lo << 16;
                    uint32_t i_hi = c_hi + (c_hi << 8); i_hi += i_hi << 16;
                    const uint32_t p_lo = i_lo << 8, p_hi = (i_hi << 8) + (i_lo >> 24) * 0x01010101u;
                    while (leaf_hits && !met) {
                        const int k = __ffs((int)leaf_hits) - 1;
                        leaf_hits &= leaf_hits - 1u;
                        met = beam_meets_leaf(S, st, h1.y + (((k < 4 ? p_lo : p_hi) >> ((uint32_t)(k & 3) * 8u)) & 0xffu), bm);
                    }
                } else {
                    const uint32_t first = S.block_first[cur];
                    while (leaf_hits && !met) {
                        const int k = __ffs((int)leaf_hits) - 1;
                        leaf_hits &= leaf_hits - 1u;
                        uint32_t first_shape;
                        float lo, hi;
                        met = beam_meets_record(S, first + (uint32_t)__popc(present & ((1u << k) - 1u)), bm, &first_shape, &lo, &hi);
                    }
                }
            }
            if (met) { clear = false; walking = false; }
            else walking = next_qblock(st, h1, present, hit & ~(h1.x >> 24), bm.sign, &cur, &sp);
        }
    }
    *cur_out = cur;  // where a walk that met a leaf stopped; its stack is st.get(0 .. sp - 1)
    *sp_out = sp;
    return clear;
}
// The state of a stopped beam walk, as the pair's rays will read it
RT_DEV void write_resume(const TraceStack& st, uint32_t cur, int sp, uint32_t* resume) {
    if (sp <= kResumeStack) {
        resume[0] = (cur << 8) | (uint32_t)sp;
        for (int k = 0; k < sp; ++k) resume[1 + k] = st.get(k);
    } else {
        resume[0] = kResumeRoot;
    }
}

// Inclusive scans over the 64 lanes of a wavefront through the DPP path (shifts within rows of 16 lanes,
// then the last lane of a row broadcast to the rows after it); every lane of the wavefront must be active.
template <int CTRL, int ROWS>
RT_DEV uint32_t dpp_or_zero(uint32_t x) {  // lanes that get no source, or whose row is masked out, read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xf, false);
}
RT_DEV uint32_t wave_scan_add(uint32_t x) {
    x += dpp_or_zero<0x111, 0xf>(x);  // row_shr:1
    x += dpp_or_zero<0x112, 0xf>(x);  // row_shr:2
    x += dpp_or_zero<0x114, 0xf>(x);  // row_shr:4
    x += dpp_or_zero<0x118, 0xf>(x);  // row_shr:8
    x += dpp_or_zero<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
    x += dpp_or_zero<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
    return x;
}
RT_DEV uint32_t wave_scan_max(uint32_t x) {
    uint32_t y;
    y = dpp_or_zero<0x111, 0xf>(x); x = x > y ? x : y;
    y = dpp_or_zero<0x112, 0xf>(x); x = x > y ? x : y;
    y = dpp_or_zero<0x114, 0xf>(x); x = x > y ? x : y;
    y = dpp_or_zero<0x118, 0xf>(x); x = x > y ? x : y;
    y = dpp_or_zero<0x142, 0xa>(x); x = x > y ? x : y;
    y = dpp_or_zero<0x143, 0xc>(x); x = x > y ? x : y;
    return x;
}

// Shadow rays: "is anything hit with tmin <= t <= tmax" (doesReachLight, light.go:141-160); the first
// accepted hit ends the ray.  Same walk and leaf queue as traverse().
// (A first version dealt the queue out LEAF by leaf, at a time when the shape loop still ran after most
// block visits: utilisation 0.20 -> 0.43 but frame 1.31 s -> 1.65 s, branch exp/balanced-shape-loop.  The
// form below deals out single shapes and runs 2.3 times per 10 block visits.)
// SAME_TMIN: tmin is the same for every lane of the wavefront (k_shadow's constant), so a lane that tests
// another lane's shape need not fetch it with the rest of that lane's ray.
template <bool SAME_TMIN>
RT_DEV bool occluded(const DScene& S, const TraceStack& st, bool live, f3 o, f3 d, float tmin, float tmax, const uint32_t* resume = nullptr
#ifdef RTGO_PROF
                     , uint32_t* prof_visits = nullptr, uint32_t* prof_iters_out = nullptr
#endif
                     ) {
    // Called by ALL 64 lanes of a wavefront (live = this lane has a ray).  The lanes walk their own rays, but
    // the shape loop is shared: the queued shapes of all lanes are dealt out evenly, one shape per lane and
    // round, whoever's ray it belongs to.  (Lane utilisation of the shape tests 0.91, against 0.38 when every
    // lane looped over its own shapes.)  The answer is an OR over the shapes: any order will do.
    QRay r = make_qray(o, d);
    uint32_t cur = 0u;  // block index; the root is block 0
    int sp = 0;
    bool walking = live;
    if (live && resume) {  // the state a beam walk left for this ray's pair (beam_is_clear): at most kResumeStack entries,
        // all fetched at once and stored to the LDS part of the stack without a branch per entry
        const uint4* q = reinterpret_cast<const uint4*>(resume);
        const uint4 w0 = q[0], w1 = q[1], w2 = q[2];
        const uint32_t e[kResumeWords] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        if (e[0] != kResumeRoot) {
            cur = e[0] >> 8;
            // Of the blocks on the beam's stack this ray keeps those whose box it meets (config 3, CPU simulation: it
            // misses 2.9 of the 8.1, each a block visit that would find nothing).  No branch per entry: every entry is
            // stored at the top, which only moves for the ones that are kept; entries past e[0] test block 0 and are dropped.
            const uint32_t n = e[0] & 0xffu;
            // fma(box, 1 / d, -(o / d)) is off by up to an ulp of o / d (rounded before the subtraction): far from the origin
            // that is more than a block is wide.  The walk itself errs the same way with or without a beam; this filter
            // is an extra test, so it allows for the error (scripts/stress_slabs.py: one ray at 10^5 units lost its occluder).
            const float slack = 9.5367431640625e-07f * fmaxf(fmaxf(fabsf(r.oi.x), fabsf(r.oi.y)), fabsf(r.oi.z));  // 2^-20 max |o / d|; inf or NaN keeps every block
#pragma unroll
            for (int k = 0; k < kResumeStack; ++k) {
                const bool in_use = (uint32_t)k < n;
                if (__ballot(in_use) == 0ull) break;  // wave-uniform
                const uint32_t b = in_use ? e[1 + k] : 0u;
                const float4 lo4 = S.block_box[2 * (size_t)b], hi4 = S.block_box[2 * (size_t)b + 1];
                const float x0 = fmaf(lo4.x, r.inv.x, -r.oi.x), x1 = fmaf(hi4.x, r.inv.x, -r.oi.x);
                const float y0 = fmaf(lo4.y, r.inv.y, -r.oi.y), y1 = fmaf(hi4.y, r.inv.y, -r.oi.y);
                const float z0 = fmaf(lo4.z, r.inv.z, -r.oi.z), z1 = fmaf(hi4.z, r.inv.z, -r.oi.z);
                // NaNs (0 * inf) drop out of fmaxf / fminf: that axis then culls nothing
                const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
                const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
                st.lds[sp * st.lds_stride] = b;
                sp += (in_use && !(tn > tf + slack)) ? 1 : 0;
            }
        }
    }
    bool occ = false;
    uint32_t npend = 0;
    uint32_t* const pend = st.lds + kLdsStackDepth * st.lds_stride;
    const int lane = (int)__lane_id();
#if RTGO_STEAL > 0
    // Work sharing inside the wavefront (SAME_TMIN callers: k_shadow).  The answer is an OR over the subtrees a ray
    // meets, so a lane with nothing left to do may take an entry of another lane's stack and walk it with that lane's
    // ray.  work_of = the lane whose ray this lane is walking for (itself at first); hit_rays = the rays (by their own
    // lane) known to be occluded, wave-uniform: every lane that walks for such a ray stops at the next shape loop.
    // Config 3: 13.6 -> 10.6 walk iterations per round of 64 rays, lane utilisation of the block visits 0.59 -> 0.81,
    // 6 % more block visits (subtrees walked for rays that another lane finds occluded) — and k_shadow 271 -> 262 ms:
    // a visit with few lanes is cheap on the memory side, so the iterations saved were the cheap ones.  Giving the
    // OLDEST entry (the farthest, largest subtree) instead of the newest: 277 ms, most of that work is never needed.
    constexpr bool kSteal = SAME_TMIN;
    int work_of = lane;
    unsigned long long hit_rays = 0ull;
#endif
    uint32_t* const own = st.wave;         // per item of a round: (round stamp << 6) | owner lane where an owner's shapes begin
    uint32_t* const flag = st.wave + 64;   // per owner lane: one of its queued shapes was hit
    uint32_t stamp = 0;                    // rounds so far: entries of earlier rounds lose against this round's
    own[lane] = 0u;

    auto flush = [&]() {
#ifdef RTGO_PROF
        if (lane == 0) atomicAdd(&g_prof2[3], 1ull);
#endif
#if RTGO_LEAF_SLAB
        if (S.leaf_bound) {
            // First the queued LEAVES are dealt out, one per lane and round, whoever's ray they belong to: a leaf whose
            // slab its ray misses inside the leaf's box (ray_misses_leaf_slab) loses its shapes (count 0 in the
            // owner's queue) before they are dealt out below.
            const uint32_t lsum = wave_scan_add(npend), lexcl = lsum - npend;
            const uint32_t ltotal = (uint32_t)__builtin_amdgcn_readlane((int)lsum, 63);
            for (uint32_t base = 0; base < ltotal; base += 64u) {
                ++stamp;
                if (npend != 0u && lsum > base && lexcl < base + 64u) own[lexcl > base ? lexcl - base : 0u] = (stamp << 6) | (uint32_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t item = base + (uint32_t)lane;
                const int owner = (int)(wave_scan_max(own[lane]) & 63u);
                uint32_t* const oq = pend + (owner - lane);
                const uint32_t first_excl = (uint32_t)__shfl((int)lexcl, owner);
                const f3 oo = mk3(__shfl(o.x, owner), __shfl(o.y, owner), __shfl(o.z, owner));
                const f3 dd = mk3(__shfl(d.x, owner), __shfl(d.y, owner), __shfl(d.z, owner));
                const f3 ii = mk3(__shfl(r.inv.x, owner), __shfl(r.inv.y, owner), __shfl(r.inv.z, owner));
                const float tmx = __shfl(tmax, owner);
                const float tmn = SAME_TMIN ? tmin : __shfl(tmin, owner);
                if (item < ltotal) {
                    RT_PROF_POINT(3);  // leaf slab tests
                    uint32_t* const slot = oq + (item - first_excl) * st.lds_stride;
                    const uint32_t leaf = *slot;
                    const float4* const lb = S.leaf_bound + kLeafBoundQuads * (size_t)(leaf >> 8);
                    const float4 b0 = lb[0], b1 = lb[1], b2 = lb[2], b3 = lb[3];
                    if (ray_misses_leaf_slab(b0, b1, b2, b3, oo, dd, ii, tmn, tmx)) *slot = leaf & ~0xffu;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#endif
        uint32_t mine = 0;  // all entries are read at once (those past npend are stale and masked): one LDS latency
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)kPendCap; ++q) {
            const uint32_t e = pend[q * st.lds_stride];
            mine += q < npend ? (e & 0xffu) : 0u;
        }
        const uint32_t sum = wave_scan_add(mine);  // this lane's shapes are items [sum - mine, sum) of the wavefront
        const uint32_t excl = sum - mine;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)sum, 63);  // wave-uniform: the round loop is scalar
        flag[lane] = 0u;
        for (uint32_t base = 0; base < total; base += 64u) {
            // every lane whose items reach into this round marks where they begin; the owner of an item is the
            // last mark at or before it (item 0 always has one: the lane whose items span `base`)
            ++stamp;
            if (mine != 0u && sum > base && excl < base + 64u) own[excl > base ? excl - base : 0u] = (stamp << 6) | (uint32_t)lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t item = base + (uint32_t)lane;
            const bool has = item < total;
            const int owner = (int)(wave_scan_max(own[lane]) & 63u);
            // everything that depends on the owner alone is fetched together, for lanes without an item too (their
            // owner is a valid lane): one LDS latency instead of four in a row
            const uint32_t* const oq = pend + (owner - lane);          // the owner's queue
            const uint32_t first_excl = (uint32_t)__shfl((int)excl, owner);
            uint32_t leaf = oq[0];
            const uint32_t decided = flag[owner];
            const f3 oo = mk3(__shfl(o.x, owner), __shfl(o.y, owner), __shfl(o.z, owner));
            const f3 dd = mk3(__shfl(d.x, owner), __shfl(d.y, owner), __shfl(d.z, owner));
            const float tmx = __shfl(tmax, owner);
            const float tmn = SAME_TMIN ? tmin : __shfl(tmin, owner);
            if (has && decided == 0u) {
                uint32_t g = item - first_excl;  // which of the owner's shapes
                for (uint32_t q = 1; g >= (leaf & 0xffu); ++q) { g -= leaf & 0xffu; leaf = oq[q * st.lds_stride]; }
                RT_PROF_POINT(1);  // shape tests
                if (prim_occludes(S, (leaf >> 8) + g, oo, dd, tmn, tmx)) flag[owner] = 1u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (flag[lane] != 0u) { occ = true; walking = false; }
        npend = 0;
#if RTGO_STEAL > 0
        if (kSteal) {
            unsigned long long found = __ballot(flag[lane] != 0u);  // lanes whose queued shapes held a hit: for whose ray?
            while (found) {
                const int l = __ffsll((long long)found) - 1;
                found &= found - 1ull;
                hit_rays |= 1ull << (uint32_t)__builtin_amdgcn_readlane(work_of, l);
            }
            if ((hit_rays >> (uint32_t)work_of) & 1ull) { occ = true; walking = false; }
        }
#endif
    };

#ifdef RTGO_PROF
    int prof_iters = 0;
    uint32_t prof_my_visits = 0;
#endif
    for (;;) {
        uint4 h1 = make_uint4(0u, 0u, 0u, 0u);
        uint32_t leaf_mask = 0, inner_mask = 0, present = 0;
#ifdef RTGO_PROF
        {   // per iteration of the walk: lanes that only hold leaves, lanes with nothing to do
            const unsigned long long hold = __ballot(!walking && npend > 0u), idle = __ballot(!walking && npend == 0u);
            const int n_walk = __popcll(__ballot(walking));
            if (lane == 0) {
                atomicAdd(&g_prof3[n_walk], 1ull);
                atomicAdd(&g_prof4[prof_iters < 63 ? prof_iters : 63], (unsigned long long)n_walk);
                atomicAdd(&g_prof2[0], 1ull);
                atomicAdd(&g_prof2[1], (unsigned long long)__popcll(hold));
                atomicAdd(&g_prof2[2], (unsigned long long)__popcll(idle));
            }
            ++prof_iters;
        }
#endif
        if (walking) {  // a box that ends before tmin holds no accepted hit
#ifdef RTGO_PROF
            ++prof_my_visits;
#endif
            const uint32_t hit = visit_qblock(S, cur, r, tmin, tmax, &h1, &present);
            leaf_mask = hit & (h1.x >> 24);
            inner_mask = hit & ~(h1.x >> 24);
        }
        if (__ballot(npend + (uint32_t)__popc(leaf_mask) > (uint32_t)kPendCap) != 0ull) {
            flush();
            if (occ) leaf_mask = 0u;  // this lane's ray is decided: nothing more to queue
        }
        queue_qleaves(pend, st.lds_stride, h1, leaf_mask, &npend);
        if (walking) walking = next_qblock(st, h1, present, inner_mask, r.sign, &cur, &sp);
        const int n_walking = __popcll(__ballot(walking));
        const int n_holding = __popcll(__ballot(npend > 0u));
        if (n_holding > 0 && n_holding >= (n_walking < kFlushLanes ? n_walking : kFlushLanes)) flush();
#if RTGO_STEAL > 0
        if (kSteal) {
            // (only entries in the LDS part of a stack are given away: st.get() of either part is a flat load)
            const bool idle = !walking && npend == 0u, rich = walking && sp > 0 && sp <= kLdsStackDepth;
            const unsigned long long idle_m = __ballot(idle), rich_m = __ballot(rich);
            if (rich_m != 0ull && __popcll(idle_m) >= RTGO_STEAL) {
                const int n_take = __popcll(idle_m) < __popcll(rich_m) ? __popcll(idle_m) : __popcll(rich_m);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int rich_rank = __popcll(rich_m & below), idle_rank = __popcll(idle_m & below);
                // the k-th lane with a stack gives its top entry — the nearest subtree it has not entered — to the k-th idle
                // lane, through the words that the shape loop uses for its hit flags (free between two shape loops)
                if (rich && rich_rank < n_take) { --sp; flag[rich_rank] = (st.lds[sp * st.lds_stride] << 6) | (uint32_t)lane; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool take = idle && idle_rank < n_take;
                const uint32_t got = take ? flag[idle_rank] : (uint32_t)lane;
                // the giver's ray comes over by cross-lane reads, its derived values (1 / d, o / d, signs) too: those
                // reads go through the LDS pipe, which has room, where recomputing them would cost VALU issue slots; a
                // lane that takes nothing reads from itself, so no lane needs a select
                const int from = (int)(got & 63u);
                o = mk3(__shfl(o.x, from), __shfl(o.y, from), __shfl(o.z, from));
                d = mk3(__shfl(d.x, from), __shfl(d.y, from), __shfl(d.z, from));
                r.inv = mk3(__shfl(r.inv.x, from), __shfl(r.inv.y, from), __shfl(r.inv.z, from));
                r.oi = mk3(__shfl(r.oi.x, from), __shfl(r.oi.y, from), __shfl(r.oi.z, from));
                r.sign = (uint32_t)__shfl((int)r.sign, from);
                tmax = __shfl(tmax, from);
                work_of = __shfl(work_of, from);
                if (take) { cur = got >> 6; sp = 0; walking = true; occ = false; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
#endif
        if (__ballot(walking || npend > 0u) == 0ull) break;  // wave-uniform exit: idle lanes stay as workers
    }
#ifdef RTGO_PROF
    if (lane == 0) atomicAdd(&g_prof3[65 + (prof_iters < 63 ? prof_iters : 63)], 1ull);
    if (prof_visits) *prof_visits = prof_my_visits;
    if (prof_iters_out) *prof_iters_out = (uint32_t)prof_iters;
#endif
#if RTGO_STEAL > 0
    if (kSteal) return live && ((hit_rays >> (uint32_t)lane) & 1ull) != 0ull;
#endif
    return occ;
}

// ---- textures (texture.go:27-53) and UV maps (shape.go:100-105, 213-227) --------------------
RT_DEV long long floor_to_int(float x) {
    if (!(x == x)) return 0;
    const float f = floorf(x);
    if (f >= 9.0e18f) return 0x7fffffffffffffffLL;
    if (f <= -9.0e18f) return (long long)0x8000000000000000ULL;
    return (long long)f;
}
RT_DEV f3 texture_color(const DTexture& t, float u, float v) {
    if (t.kind == RT_TEX_CHECKERS) {
        const long long u2 = floor_to_int(u * t.cw), v2 = floor_to_int(v * t.ch);
        return (((u2 + v2) & 1LL) == 0) ? ld3(t.c1) : ld3(t.c2);
    }
    const long long u2 = floor_to_int(u * (float)t.w), v2 = floor_to_int(v * (float)t.h);
    uint32_t r = 0, g = 0, b = 0, a = 0;
    if (u2 >= 0 && u2 < t.w && v2 >= 0 && v2 < t.h) {  // no clamp/wrap: outside = zero colour
        const uchar4 px = reinterpret_cast<const uchar4*>(t.rgba)[(size_t)v2 * t.w + (size_t)u2];
        r = px.x * 257u; g = px.y * 257u; b = px.z * 257u; a = px.w * 257u;
    }
    const float r256 = (float)r / 255.99f, g256 = (float)g / 255.99f, b256 = (float)b / 255.99f;
    const float aa = (float)a / (255.99f * 255.99f);
    const float a1 = aa < 1.0f ? aa : 1.0f;
    return mk3((1.0f - a1) * 0.0f + (a1 * r256) / 255.99f,
               (1.0f - a1) * 0.0f + (a1 * g256) / 255.99f,
               (1.0f - a1) * 0.0f + (a1 * b256) / 255.99f);
}
struct PrimData { float4 p0, p1, p2; };
RT_DEV PrimData load_prim(const DScene& S, int prim) {
    PrimData P;
    P.p0 = S.prims[3 * (size_t)prim]; P.p1 = S.prims[3 * (size_t)prim + 1]; P.p2 = S.prims[3 * (size_t)prim + 2];
    return P;
}
RT_DEV void texture_map(const PrimData& P, f3 point, float* u, float* v) {
    const uint32_t bits = __float_as_uint(P.p2.y);
    const float kPi = 3.14159265358979323846f;
    if (!(bits & kPrimTriangle)) {  // shape.go:100-105
        const f3 q = sub3(point, mk3(P.p0.x, P.p0.y, P.p0.z));
        const float theta = atan2f(-1.0f * q.z, q.x);
        const float phi = acosf(-1.0f * q.y / P.p0.w);
        *u = (theta + kPi) / (2.0f * kPi);
        *v = phi / kPi;
    } else {  // shape.go:213-227: u = 1 - v(=0, the unset named result) - w
        const f3 A = mk3(P.p0.x, P.p0.y, P.p0.z), B = mk3(P.p0.w, P.p1.x, P.p1.y), C = mk3(P.p1.z, P.p1.w, P.p2.x);
        const f3 v0 = sub3(B, A), v1 = sub3(C, A), v2 = sub3(point, A);
        const float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
        const float denom = d00 * d11 - d01 * d01;
        const float w = (d00 * d21 - d01 * d20) / denom;
        *u = 1.0f - 0.0f - w;
        *v = (d11 * d20 - d01 * d21) / denom;
    }
}
// hitRecord.normal: shape.go:59 (sphere: (p - C)/R), :122 (triangle: Unit(e1 x e2), never flipped)
RT_DEV f3 prim_normal(const PrimData& P, f3 o, f3 d, float t) {
    const uint32_t bits = __float_as_uint(P.p2.y);
    if (!(bits & kPrimTriangle))
        return scale3(1.0f / P.p0.w, sub3(add3(o, scale3(t, d)), mk3(P.p0.x, P.p0.y, P.p0.z)));
    const f3 A = mk3(P.p0.x, P.p0.y, P.p0.z), B = mk3(P.p0.w, P.p1.x, P.p1.y), C = mk3(P.p1.z, P.p1.w, P.p2.x);
    return unit3(cross3(sub3(B, A), sub3(C, A)));
}

// ---- material helpers (material.go:145-172) -------------------------------------------------
RT_DEV f3 reflected(f3 v, f3 n) { return unit3(sub3(v, scale3(2.0f * dot3(v, n), n))); }
RT_DEV f3 refracted(f3 v, f3 n, float eta) {
    const f3 uv = unit3(v);
    const float dt = dot3(scale3(-1.0f, uv), n);
    const float cos_theta = (1.0f < dt) ? 1.0f : dt;  // math.Min(dt, 1.0)
    const f3 perp = scale3(eta, add3(uv, scale3(cos_theta, n)));
    // material.go:153 precedence quirk kept: |1 - x^2 + y^2 + z^2|
    const f3 par = scale3(-sqrtf(fabsf(1.0f - perp.x * perp.x + perp.y * perp.y + perp.z * perp.z)), n);
    return unit3(add3(perp, par));
}
RT_DEV float schlick(float cosine, float ri) {
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f);
}
RT_DEV float saturate(float i) { return i > 1.0f ? 1.0f : (i < 0.0f ? 0.0f : i); }
RT_DEV float min1(float x) { return x < 1.0f ? x : 1.0f; }  // std::min(1, x) / math.Min(1.0, x)

}  // namespace rtgo
