// librtgo_hip.so — wavefront path tracer for MI355X (gfx950) behind include/rtgo.h.
//
// Stages per batch of (pixel, sample) paths, all on one HIP stream:
//   raygen     camera.go:37-55 + raytracer.go:113-116   1 lane / path
//   extend     accelerationstructures.go:81-124          1 lane / live ray, ordered stack traversal
//   shade      raytracer.go:143-161 + material.go:34-70  1 lane / hit; Metal/Dielectric continue,
//                                                         PhongBlinn hits go to the phong queue
//   pairs      material.go:83-119 (visibility-free part) 1 lane / PhongBlinn hit: which (hit, light) pairs need rays
//   shadow     material.go:76-82 + light.go:93-160       1 lane / (queued pair, repetition); any-hit
//   phong      material.go:83-131                         1 lane / PhongBlinn hit
//   accumulate raytracer.go:117-119                       1 lane / pixel (fixed-order sample sum)
//   resolve    raytracer.go:93-96,125                     1 lane / pixel -> RGBA8
// Queues are compacted with a 64-lane ballot + popcount prefix and one atomic per 256-thread block.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: librccl is opened with dlopen when a multi-device frame needs it

#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtgo.h"
#include "device_math.hpp"
#include "octree.hpp"

#pragma clang fp contract(off)

using namespace rtgo;

namespace {

constexpr int kBlock = 256;
#ifndef RTGO_MAX_GRID
#define RTGO_MAX_GRID 16384
#endif
constexpr int kMaxGridBlocks = RTGO_MAX_GRID;  // grid-stride kernels; k_shadow ms per frame at config 3: 1024 -> 708, 2048 -> 688, 4096 -> 666, 8192 -> 651, 16384 -> 634, 65536 -> 636
constexpr int64_t kDefaultMaxPaths = (int64_t)1 << 23;
constexpr int kDefaultLeafSizeSah = 20;  // shapes per leaf (the SAH builder splits down to 8, the flattener makes one leaf of a subtree of at most this many).
// Frame ms at config 3 with leaf slabs, which make a leaf that is met but not hit nearly free: 4 -> 245.9, 6 -> 228.1, 8 -> 223.6, 12 -> 218.7, 16 -> 216.0, 20 -> 214.3, 24 -> 214.7, 31 -> 220.4;
// before them 6 was best (k_shadow ms: 2 -> 833, 3 -> 816, 4 -> 823, 6 -> 791, 8 -> 809 in round 1; 4 -> 257, 6 -> 250.6, 8 -> 252.7, 12 -> 254.7 at the start of round 3)
constexpr int kDefaultLeafSize = 4;  // measured on config 3: 1 -> 2.84 s, 2 -> 2.58 s, 4 -> 2.42 s, 8 -> 2.48 s, 16 -> 2.76 s per frame

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// ------------------------------------------------------------------------------------------
// Batch geometry: which (pixel, sample) a path slot is
// ------------------------------------------------------------------------------------------
struct Batch {
    int W, H;
    int tile, tiles_x;         // square tiles of `tile` pixels (multiple of 8), row-major tile grid
    int rank, world, by_tiles; // by_tiles: this rank owns tiles k with k % world == rank
    uint32_t tj0;              // first local tile of the batch (a batch starts on a tile; a frame has fewer than 2^32 tiles)
    int npix, ns, s0;          // local pixels, samples per pixel in this batch, first sample index
    int max_depth, reps;
    uint32_t k0, k1;           // Philox key
};

// pixel of the batch (counted from its first tile) -> (column x, output row); false when the tile cell lies outside
// the image.  All in 32 bits: a batch has at most 2^30 pixels and a frame fewer than 2^32 tiles (64-bit divisions were
// most of k_raygen's instructions).
RT_DEV bool pixel_of(const Batch& B, uint32_t lpl, int* x, int* row) {
    const uint32_t tt = (uint32_t)(B.tile * B.tile);
    const uint32_t tjl = lpl / tt;
    const uint32_t w = lpl - tjl * tt;
    const uint32_t tj = B.tj0 + tjl;
    const uint32_t k = B.by_tiles ? tj * (uint32_t)B.world + (uint32_t)B.rank : tj;
    const uint32_t ty = k / (uint32_t)B.tiles_x, tx = k - ty * (uint32_t)B.tiles_x;
    const uint32_t sub_per_row = (uint32_t)B.tile >> 3;
    const uint32_t b = w >> 6, i = w & 63u;  // 8x8 sub-blocks keep a wavefront's pixels together
    const uint32_t by = b / sub_per_row, bx = b - by * sub_per_row;
    *x = (int)(tx * (uint32_t)B.tile + bx * 8u + (i & 7u));
    *row = (int)(ty * (uint32_t)B.tile + by * 8u + (i >> 3));
    return *x < B.W && *row < B.H;
}

constexpr int kSubCounters = 64;
constexpr int kCounterSums = 3;  // frame_counters[0 .. 3): the sums; then kSubCounters words per sum that the wavefronts add to
struct Work {
    float4* ray_o;   // {origin.xyz, hit t}
    float4* ray_d;   // {direction.xyz, bits of hit shape index (-1 = miss)}
    float4* thr;     // path throughput (product of attenuations)
    float4* rad;     // terminal radiance of the path (written once)
    uint2* path_key; // {pixel index, sample index} of the path: the Philox counter words every later draw needs
    uint32_t* queue[2];
    uint32_t* phong_queue;
    uint32_t* vis;   // visible-repetition count per (phong hit, positional light)
    uint32_t* pairs; // (hit, light) pairs whose shading term is not exactly zero: hit << light_shift | light, light-major
    uint32_t light_shift;  // bits of a pair that hold the positional light's number (2^light_shift >= their count): no division per pair
    uint32_t* pair_seg;    // the same before compaction: light l's hits at [l * n_hits, l * n_hits + light_count[l])
    uint32_t* light_count; // [0, L) pairs per positional light, [L, 2L) their exclusive prefix sums, [2L] pairs k_pair_beam passes on
    uint32_t* counters;  // per batch: [0..D] ray-queue sizes per bounce, [D+1..2D] phong-queue sizes,
                         // [2D+4..3D+3] pair-queue sizes per bounce
    unsigned long long* frame_counters;  // per call: [0] shadow rays traced, [1] shadow rays decided by a pair's beam walk, [2] beam walks: the
                                         // sums (k_sum_counters) of [3, 3 + kSubCounters) and of the two runs of kSubCounters words after those, to
                                         // which the wavefronts add (one address takes an atomic every 11 ns, 65,536 per launch)
    uint32_t* overflow;  // traversal stack spill
    uint32_t overflow_stride;
    uint32_t* beam_state;  // per (phong hit, positional light), like vis: kResumeWords words the pair's rays resume from (nullptr: none kept)
};

// Queue compaction: ballot + popcount prefix inside a wavefront, the wavefronts' counts combined through
// LDS, ONE atomic per 256-thread block (a queue counter is a single address: with one atomic per
// wavefront the 131k same-address atomics of a batch took longer than the kernels' real work).
// Every thread of the block must call it.
RT_DEV uint32_t block_push(uint32_t* counter, bool pred) {
    __shared__ uint32_t s_cnt[kBlock / 64 + 1];
    const unsigned long long mask = __ballot(pred);
    const int lane = (int)__lane_id(), wave = (int)(threadIdx.x >> 6);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kBlock / 64; ++w) total += s_cnt[w];
        s_cnt[kBlock / 64] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t at = s_cnt[kBlock / 64];
    for (int w = 0; w < wave; ++w) at += s_cnt[w];
    at += (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // s_cnt is reused by the next call
    return at;
}

// block_push for up to N queues at once (k_phong_pairs: one queue per positional light): the same three barriers
// serve all of them, and thread q of the block does queue q's atomic.
template <int N>
RT_DEV void block_push_multi(uint32_t* counters, int n, const bool (&pred)[N], uint32_t (&at)[N]) {
    __shared__ uint32_t s_cnt[N][kBlock / 64 + 1];
    const int lane = (int)__lane_id(), wave = (int)(threadIdx.x >> 6);
    unsigned long long mask[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        mask[q] = __ballot(q < n && pred[q]);
        if (lane == 0) s_cnt[q][wave] = (uint32_t)__popcll(mask[q]);
    }
    __syncthreads();
    if ((int)threadIdx.x < n) {
        uint32_t total = 0;
        for (int w = 0; w < kBlock / 64; ++w) total += s_cnt[threadIdx.x][w];
        s_cnt[threadIdx.x][kBlock / 64] = total ? atomicAdd(counters + threadIdx.x, total) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        uint32_t a = s_cnt[q][kBlock / 64];
        for (int w = 0; w < wave; ++w) a += s_cnt[q][w];
        at[q] = a + (uint32_t)__popcll(mask[q] & ((1ull << lane) - 1ull));
    }
    __syncthreads();  // s_cnt is reused by the next call
}

// block_push for K elements per thread (element k of thread t is item (k * kBlock + t) of the block): ONE atomic
// per block and queue for all K * kBlock of them, and the queue keeps the items' order.  A queue counter is a single
// address, and atomics on one address go through at about 11 ns each whatever else the GPU does: with one per
// 256 items they took most of the time of k_raygen, k_shade and k_phong_pairs.
// Q queues at once (bit q of pred[k]: element k goes to queue q); at[k][q] is only meaningful where that bit is set.
template <int Q, int K>
RT_DEV void block_push_qk(uint32_t* const* counters, int n_queues, const uint32_t (&pred)[K], uint32_t (&at)[K][Q]) {
    __shared__ uint32_t s_cnt[Q][K * (kBlock / 64) + 1];
    const int lane = (int)__lane_id(), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(((pred[k] >> q) & 1u) != 0u);
            if (lane == 0) s_cnt[q][k * (kBlock / 64) + wave] = (uint32_t)__popcll(m);
        }
    __syncthreads();
    if ((int)threadIdx.x < n_queues && (int)threadIdx.x < Q) {  // thread q: exclusive prefix sums of queue q's counts, then its atomic
        uint32_t run = 0;
        for (int j = 0; j < K * (kBlock / 64); ++j) { const uint32_t c = s_cnt[threadIdx.x][j]; s_cnt[threadIdx.x][j] = run; run += c; }
        s_cnt[threadIdx.x][K * (kBlock / 64)] = run ? atomicAdd(counters[threadIdx.x], run) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(((pred[k] >> q) & 1u) != 0u);
            at[k][q] = s_cnt[q][K * (kBlock / 64)] + s_cnt[q][k * (kBlock / 64) + wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
    __syncthreads();  // s_cnt is reused by the next call
}

// raytracer.go:114-116 + camera.go:37-55
RT_DEV void camera_ray(const DCamera& cam, int W, int H, int i, int j, const RngKey& key, f3* o, f3* d) {
    const u4 aa = draw(key, 0, P_AA, 0, 0);
    const float u = ((float)i + u01(aa.x)) / (float)W;
    const float v = ((float)j + u01(aa.y)) / (float)H;
    f3 disk = mk3(0.0f, 0.0f, 0.0f);
    for (int tr = 0; tr < kMaxTries; ++tr) {  // randomInUnitDisk runs even when the aperture is 0
        const u4 w = draw(key, 0, P_LENS, 0, (uint32_t)tr);
        const f3 p = sub3(scale3(2.0f, mk3(u01(w.x), u01(w.y), 0.0f)), mk3(1.0f, 1.0f, 0.0f));
        if (dot3(p, p) < 1.0f) { disk = p; break; }
    }
    const f3 rd = scale3(cam.lens, disk);
    const f3 offset = add3(scale3(rd.x, ld3(cam.u)), scale3(rd.y, ld3(cam.v)));
    *o = add3(ld3(cam.origin), offset);
    *d = unit3(sub3(sub3(add3(add3(ld3(cam.ll), scale3(u, ld3(cam.hor))), scale3(v, ld3(cam.ver))), ld3(cam.origin)), offset));
}

constexpr int kPerThread = 4;  // elements per thread in the kernels that fill a queue (block_push_qk)
__global__ __launch_bounds__(kBlock) void k_raygen(Batch B, DCamera cam, Work Wk) {
    const uint32_t total = (uint32_t)B.npix * (uint32_t)B.ns;  // (a batch holds at most 2^30 paths: plan_render)
    uint32_t valid[kPerThread];
    uint32_t pixel[kPerThread];  // row * W + x
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {  // which of the block's slots are paths: only the pixel arithmetic
        const uint32_t slot = (blockIdx.x * kPerThread + k) * kBlock + threadIdx.x;
        valid[k] = 0u; pixel[k] = 0u;
        if (slot < total) {
            int x = 0, row = 0;
            if (pixel_of(B, slot / (uint32_t)B.ns, &x, &row)) { valid[k] = 1u; pixel[k] = (uint32_t)(row * B.W + x); }
            Wk.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    uint32_t at[kPerThread][1];
    uint32_t* const counters[1] = {&Wk.counters[0]};
    block_push_qk<1, kPerThread>(counters, 1, valid, at);
#pragma unroll 1
    for (int k = 0; k < kPerThread; ++k) {
        if (!valid[k]) continue;
        const uint32_t slot = (blockIdx.x * kPerThread + k) * kBlock + threadIdx.x;
        const uint32_t lpl = slot / (uint32_t)B.ns;
        const int smp = B.s0 + (int)(slot - lpl * (uint32_t)B.ns);
        const int row = (int)(pixel[k] / (uint32_t)B.W), x = (int)(pixel[k] - (uint32_t)row * (uint32_t)B.W);
        const RngKey key{B.k0, B.k1, pixel[k], (uint32_t)smp};
        Wk.path_key[slot] = make_uint2(key.pixel, key.sample);
        f3 o, d;
        camera_ray(cam, B.W, B.H, x, B.H - 1 - row, key, &o, &d);
        Wk.ray_o[slot] = make_float4(o.x, o.y, o.z, 0.0f);
        Wk.ray_d[slot] = make_float4(d.x, d.y, d.z, __int_as_float(-1));
        Wk.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        Wk.queue[0][at[k][0]] = slot;
    }
}

// Workgroups go to the 8 XCDs round-robin and every XCD has an L2 of its own: block b works on virtual
// block (b % 8) * (blocks / 8) + b / 8, so that an XCD's blocks cover one contiguous eighth of each round
// of a grid-stride loop (neighbouring rays, the same part of the tree) instead of every eighth block of it.
__device__ __forceinline__ uint32_t xcd_block() {
    return (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
}

// The queue lengths live in Wk.counters: the host launches on an upper bound (the previous bounce's length) and
// every kernel reads the real one, so no launch waits for a read-back.
#ifndef RTGO_EXTEND_WAVES
#define RTGO_EXTEND_WAVES 6  // frame ms at config 3: 4 waves (104 VGPRs, the compiler's own choice) 305.2-306.1, 5 (88) 303.4-303.7, 6 (80, 36 B scratch) 302.3-303.4, 7 (72) 304.3
#endif
__global__ __launch_bounds__(kBlock, RTGO_EXTEND_WAVES) void k_extend(DScene S, Work Wk, const uint32_t* queue, const uint32_t* n_ptr) {
    __shared__ uint32_t s_stack[kLdsPerLane * kBlock];
    const uint32_t n = *n_ptr;
    // (plain block order here: with xcd_block() the XCDs get image regions of unequal cost, 12.1 -> 15.9 ms per frame)
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t gstride = gridDim.x * kBlock;
    const TraceStack st{s_stack + threadIdx.x, Wk.overflow + gtid, (uint32_t)kBlock, Wk.overflow_stride, nullptr};
    for (uint32_t i = gtid; i < n; i += gstride) {
        const uint32_t slot = queue[i];
        const float4 ro = Wk.ray_o[slot], rdv = Wk.ray_d[slot];
        float t; int prim;
        traverse(S, st, mk3(ro.x, ro.y, ro.z), mk3(rdv.x, rdv.y, rdv.z), 0.0f, &t, &prim);
        Wk.ray_o[slot].w = t;
        Wk.ray_d[slot].w = __int_as_float(prim);
    }
}

// Metal.scatter (material.go:43-50) and Dielectric.scatter (material.go:52-70): the continuation ray and its attenuation
struct Scattered { bool cont; f3 o, d, att; };
RT_DEV Scattered scatter_specular(const DMaterial& m, f3 d, f3 p, f3 nrm, const RngKey& key, uint32_t bounce) {
    Scattered r;
    r.cont = false; r.o = p; r.d = d; r.att = mk3(1.0f, 1.0f, 1.0f);
    if (m.kind == RT_MAT_METAL) {
        const float fuzz = m.param < 1.0f ? m.param : 1.0f;
        const f3 refl = reflected(d, nrm);
        const f3 rnd = random_in_unit_sphere(key, bounce, P_FUZZ, 0);  // drawn even when it is not used
        r.cont = dot3(refl, nrm) > 0.0f;                  // back face: terminal (0,0,0)
        r.d = add3(refl, scale3(fuzz, rnd));              // not re-normalised
        r.o = p;                                          // origin = p, no epsilon
        r.att = ld3(m.color);
    } else {
        const float ratio = dot3(d, nrm) > 0.0f ? m.param : 1.0f / m.param;
        const float dt = dot3(scale3(-1.0f, d), nrm);
        const float cos_theta = (1.0f < dt) ? 1.0f : dt;
        const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        const bool cannot = ratio * sin_theta > 1.0f;
        f3 dir;
        if (cannot || schlick(cos_theta, ratio) > u01(draw(key, bounce, P_SCHLICK, 0, 0).x))
            dir = reflected(d, nrm);
        else
            dir = refracted(d, nrm, ratio);
        r.cont = true;
        r.d = dir;
        r.o = add3(p, scale3(0.00001f, dir));
    }
    return r;
}

// color() (raytracer.go:135-162) one bounce at a time + Standard/Metal/Dielectric scatter
__global__ __launch_bounds__(kBlock) void k_shade(DScene S, Batch B, Work Wk, const uint32_t* queue, const uint32_t* n_ptr,
                                                  uint32_t* next_queue, int bounce) {
    // kPerThread queue entries per thread, then ONE push per block and queue (block_push_qk); what an entry adds to
    // which queue waits in LDS meanwhile: its path slot (below 2^30) | goes on << 30 | PhongBlinn hit << 31
    __shared__ uint32_t s_item[kPerThread][kBlock];
    const uint32_t n = *n_ptr;
    const uint32_t first = blockIdx.x * (kBlock * kPerThread);
    if (first >= n) return;  // whole block beyond the queue (the grid covers an upper bound)
#pragma unroll 1
    for (int k = 0; k < kPerThread; ++k) {
    const uint32_t i = first + (uint32_t)k * kBlock + threadIdx.x;
    bool cont = false, phong = false;
    uint32_t slot = 0;
    if (i < n) {
        slot = queue[i];
        const float4 ro = Wk.ray_o[slot], rdv = Wk.ray_d[slot];
        const int prim = __float_as_int(rdv.w);
        if (prim >= 0) {  // miss: background (0,0,0), raytracer.go:161
            const f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rdv.x, rdv.y, rdv.z);
            const float t = ro.w;
            const PrimData P = load_prim(S, prim);
            const DMaterial m = S.materials[__float_as_uint(P.p2.y) >> 2];
            const f3 p = add3(o, scale3(t, d));
            const float4 th = Wk.thr[slot];
            if (m.kind == RT_MAT_STANDARD) {  // material.go:34-41: unlit terminal colour
                f3 col = ld3(m.color);
                if (m.texture >= 0) {
                    float u, v;
                    texture_map(P, p, &u, &v);
                    col = texture_color(S.textures[m.texture], u, v);
                }
                Wk.rad[slot] = make_float4(th.x * col.x, th.y * col.y, th.z * col.z, 0.0f);
            } else if (m.kind == RT_MAT_PHONGBLINN) {
                phong = true;
            } else {
                const uint2 pk = Wk.path_key[slot];
                const RngKey key{B.k0, B.k1, pk.x, pk.y};
                const Scattered sc = scatter_specular(m, d, p, prim_normal(P, o, d, t), key, (uint32_t)bounce);
                if (sc.cont) {
                    Wk.ray_o[slot] = make_float4(sc.o.x, sc.o.y, sc.o.z, 0.0f);
                    Wk.ray_d[slot] = make_float4(sc.d.x, sc.d.y, sc.d.z, __int_as_float(-1));
                    Wk.thr[slot] = make_float4(th.x * sc.att.x, th.y * sc.att.y, th.z * sc.att.z, 0.0f);
                    cont = true;
                }
            }
        }
    }
    // a ray scattered at the last allowed depth can only return black (raytracer.go:145)
    cont = cont && (bounce + 1 < B.max_depth);
    s_item[k][threadIdx.x] = slot | (cont ? 1u << 30 : 0u) | (phong ? 1u << 31 : 0u);
    }
    uint32_t pred[kPerThread], at[kPerThread][2];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) pred[k] = s_item[k][threadIdx.x] >> 30;  // (a thread reads what it wrote itself)
    uint32_t* const counters[2] = {&Wk.counters[bounce + 1], &Wk.counters[B.max_depth + 1 + bounce]};
    block_push_qk<2, kPerThread>(counters, 2, pred, at);
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const uint32_t slot = s_item[k][threadIdx.x] & 0x3fffffffu;
        if (pred[k] & 1u) next_queue[at[k][0]] = slot;
        if (pred[k] & 2u) Wk.phong_queue[at[k][1]] = slot;
    }
}

// What one visible repetition of light L adds at a PhongBlinn hit (material.go:83-119).  The light position
// used here is the un-jittered one, so the value is the same for every repetition of a (hit, light).
RT_DEV f3 phong_term(const DLight& L, const DMaterial& m, const f3& mc, const f3& p, const f3& nrm, const f3& d) {
    const f3 l2p = sub3(ld3(L.pos), p);  // material.go:83-84
    const f3 ldir = unit3(l2p);
    const float d2 = l2p.x * l2p.x + l2p.y * l2p.y + l2p.z * l2p.z;
    float decay = L.decay * d2;
    if (decay <= 1.0f) decay = 1.0f;
    const float inten = saturate(dot3(nrm, ldir));
    const f3 diff = scale3(inten * L.intensity / decay, mk3(mc.x * L.color[0], mc.y * L.color[1], mc.z * L.color[2]));
    const f3 h = unit3(add3(ldir, d));  // material.go:107: ray direction, not view direction
    const float si = powf(saturate(dot3(nrm, h)), m.param);
    const f3 spec = scale3(si * L.spec_intensity / decay, mk3(m.spec[0] * L.color[0], m.spec[1] * L.color[1], m.spec[2] * L.color[2]));
    return mk3(min1(diff.x + spec.x), min1(diff.y + spec.y), min1(diff.z + spec.z));
}

struct PhongHit { f3 p, nrm, d, mc; DMaterial m; };
RT_DEV PhongHit load_phong_hit(const DScene& S, const Work& Wk, uint32_t slot) {
    const float4 ro = Wk.ray_o[slot], rdv = Wk.ray_d[slot];
    const f3 o = mk3(ro.x, ro.y, ro.z);
    PhongHit h;
    h.d = mk3(rdv.x, rdv.y, rdv.z);
    const float t = ro.w;
    const PrimData P = load_prim(S, __float_as_int(rdv.w));
    h.m = S.materials[__float_as_uint(P.p2.y) >> 2];
    h.p = add3(o, scale3(t, h.d));
    h.nrm = prim_normal(P, o, h.d, t);
    h.mc = ld3(h.m.color);
    if (h.m.texture >= 0) {
        float u, v;
        texture_map(P, h.p, &u, &v);
        h.mc = texture_color(S.textures[h.m.texture], u, v);
    }
    return h;
}

// True only if the spot-cone test of k_shadow (light.go:126-130) fails for EVERY jittered light position:
// the jitter moves the light by less than 0.25, which turns the direction light -> p by at most
// asin(0.25 / |p - L|); a point whose angle to the cone axis exceeds Angle by more than that (plus a
// margin far above fp32 rounding) is outside the cone for all repetitions, so the pair has no ray to trace.
RT_DEV bool surely_outside_cone(const DLight& L, const f3& p) {
    if (L.kind != RT_LIGHT_SPOT) return false;
    const f3 v = sub3(p, ld3(L.pos));
    const float dist = norm3(v);
    const float sin_d = 0.25f / dist * 1.001f + 1e-4f;  // NaN / Inf for dist = 0 fail the next test
    if (!(sin_d < 0.99f)) return false;
    const float cos_d = sqrtf(1.0f - sin_d * sin_d);
    const float cos_a = L.cos_angle;
    const float sin_a = sqrtf(fmaxf(0.0f, 1.0f - cos_a * cos_a));  // Angle in [0, 180] degrees
    if (!(sin_a * cos_d + cos_a * sin_d > 1e-3f)) return false;     // Angle + delta must stay below 180 degrees
    const float cos_limit = cos_a * cos_d - sin_a * sin_d;          // cos(Angle + delta)
    const float cos_p = dot3(ld3(L.look_dir), scale3(1.0f / dist, v));
    return cos_p < cos_limit - 1e-4f;
}

// True only if the spot-cone test passes for EVERY jittered light position (the same bound on the turn of the direction
// light -> p as above): such a pair traces all of its repetitions.  Only cones up to 90 degrees off the axis are judged.
RT_DEV bool surely_inside_cone(const DLight& L, const f3& p) {
    if (L.kind != RT_LIGHT_SPOT) return true;
    const f3 v = sub3(p, ld3(L.pos));
    const float dist = norm3(v);
    const float sin_d = 0.25f / dist * 1.001f + 1e-4f;
    if (!(sin_d < 0.99f)) return false;
    const float cos_d = sqrtf(1.0f - sin_d * sin_d);
    const float cos_p = dot3(ld3(L.look_dir), scale3(1.0f / dist, v));
    if (!(cos_p > 0.0f)) return false;  // also NaN
    const float sin_p = sqrtf(fmaxf(0.0f, 1.0f - cos_p * cos_p));
    return cos_p * cos_d - sin_p * sin_d > L.cos_angle + 1e-4f;  // cos(angle + delta) > cos(Angle)
}

// One lane per PhongBlinn hit.  A (hit, light) pair whose shading term is exactly (0,0,0) — light behind
// the surface and no specular lobe towards it — adds count/reps * 0 whatever its soft-shadow rays would
// find, so those rays are never generated; nor are those of a pair that lies outside a spot light's cone
// for every jittered light position.  Only the other pairs are queued for k_shadow, per light
// (segment l of pair_seg), so that the queue k_pair_compact builds from the segments is light-major.
// Every pair's visible-repetition count starts at 0 here.
__global__ __launch_bounds__(kBlock) void k_phong_pairs(DScene S, Work Wk, const uint32_t* n_hits_ptr, uint32_t reps) {
    // kPerThread hits per thread and up to kAtOnce lights per round, then ONE push per block and light
    // (block_push_qk; uniform trip counts: every thread of the block takes part)
    constexpr int kAtOnce = 4;
    __shared__ uint32_t s_keep[kPerThread][kBlock];  // bit q: the hit's pair with light lp0 + q is kept
    const uint32_t n_hits = *n_hits_ptr;
    const uint32_t first = blockIdx.x * (kBlock * kPerThread);
    if (first >= n_hits) return;
    unsigned long long decided = 0;  // shadow rays settled by voxel beams
    for (int lp0 = 0; lp0 < S.n_pos_lights; lp0 += kAtOnce) {
        const int n = S.n_pos_lights - lp0 < kAtOnce ? S.n_pos_lights - lp0 : kAtOnce;
#pragma unroll 1
        for (int k = 0; k < kPerThread; ++k) {
            const uint32_t hit_i = first + (uint32_t)k * kBlock + threadIdx.x;
            uint32_t keep = 0u;
            if (hit_i < n_hits) {
                const PhongHit h = load_phong_hit(S, Wk, Wk.phong_queue[hit_i]);
                for (int q = 0; q < n; ++q) {
                    const DLight& L = S.pos_lights[lp0 + q];
                    const f3 comb = phong_term(L, h.m, h.mc, h.p, h.nrm, h.d);
                    uint32_t v = 0u;
                    if (!(comb.x == 0.0f && comb.y == 0.0f && comb.z == 0.0f) && !surely_outside_cone(L, h.p)) {
                        // a pair in a cell whose voxel beam towards this light is clear, and inside the cone for every jittered
                        // light position, has all its repetitions visible: no beam walk, no rays
                        if (vox_clear(S, h.p, lp0 + q) && surely_inside_cone(L, h.p)) { v = reps; decided += reps; }
                        else keep |= 1u << q;
                    }
                    Wk.vis[(size_t)hit_i * S.n_pos_lights + (lp0 + q)] = v;
                }
            }
            s_keep[k][threadIdx.x] = keep;
        }
        uint32_t pred[kPerThread], at[kPerThread][kAtOnce];
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) pred[k] = s_keep[k][threadIdx.x];  // (a thread reads what it wrote itself)
        uint32_t* counters[kAtOnce];
#pragma unroll
        for (int q = 0; q < kAtOnce; ++q) counters[q] = Wk.light_count + lp0 + (q < n ? q : 0);
        block_push_qk<kAtOnce, kPerThread>(counters, n, pred, at);
#pragma unroll
        for (int k = 0; k < kPerThread; ++k)
#pragma unroll
            for (int q = 0; q < kAtOnce; ++q)
                if ((pred[k] >> q) & 1u) Wk.pair_seg[(size_t)(lp0 + q) * n_hits + at[k][q]] = first + (uint32_t)k * kBlock + threadIdx.x;
    }
    if (S.vox) {  // (wave-uniform: every lane of the block comes here)
        for (int off = 32; off > 0; off >>= 1) decided += __shfl_down(decided, off);
        if (__lane_id() == 0 && decided) atomicAdd(Wk.frame_counters + kCounterSums + kSubCounters + (blockIdx.x & (kSubCounters - 1)), decided);
    }
}
// exclusive prefix sums of the per-light pair counts; the total is the pair-queue length k_shadow reads
__global__ void k_pair_offsets(Work Wk, int n_pos_lights, uint32_t* pair_counter) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t run = 0;
    for (int l = 0; l < n_pos_lights; ++l) {
        Wk.light_count[n_pos_lights + l] = run;
        run += Wk.light_count[l];
    }
    *pair_counter = run;
    Wk.light_count[2 * n_pos_lights] = 0u;  // length of the queue k_pair_beam builds
}
__global__ __launch_bounds__(kBlock) void k_pair_compact(Work Wk, const uint32_t* n_hits_ptr, int n_pos_lights) {
    // grid: x over the hits, y over the positional lights (a light per grid row, no division per element)
    const uint32_t n_hits = *n_hits_ptr;
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    for (uint32_t l = blockIdx.y; l < (uint32_t)n_pos_lights; l += gridDim.y)
        if (j < Wk.light_count[l]) Wk.pairs[Wk.light_count[n_pos_lights + l] + j] = (Wk.pair_seg[(size_t)l * n_hits + j] << Wk.light_shift) | l;
}

// One lane per queued (hit, light) pair: ONE conservative walk for all of the pair's shadow rays (device_math.hpp
// Beam).  A pair whose beam meets no leaf box, and that lies inside the spot cone for every jittered light position, has
// all `reps` repetitions visible: its count is written here and none of its rays is generated.  The other pairs go on
// to k_shadow through a second queue (the space of pair_seg, which k_pair_compact has consumed).
#ifndef RTGO_BEAM_WAVES
#define RTGO_BEAM_WAVES 7  // frame ms at config 3: 5 waves (86 VGPRs, the compiler's own choice) 308-310, 6 (78) 307.5, 7 (72) 303.5-304.2; the block's 20.5 KB of LDS allow no 8th
#endif
constexpr float kBeamR = 0.25f * 1.0001f + 1e-5f;  // radius of the ball that holds every jittered light position (light.go:94), with margin
// ... and the jittered position is ROUNDED: light + jitter in float32 is up to half an ulp of the light's coordinates off per
// axis (0.00024 at 5,000 units from the origin: more than the margin above, and many times the thickness of a flat leaf's
// slab).  DScene::beam_r = kBeamR + 2^-22 of the largest sum of a positional light's coordinates, over four times that
// (one radius per scene: a scalar, where a radius per light would be one more register across the beam's walk).
__global__ __launch_bounds__(kBlock, RTGO_BEAM_WAVES) void k_pair_beam(DScene S, Batch B, Work Wk, const uint32_t* n_hits_ptr, const uint32_t* pair_counter) {
    __shared__ uint32_t s_stack[kLdsPerLane * kBlock];
    (void)n_hits_ptr;
    const uint32_t n_pairs = *pair_counter;
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t gstride = gridDim.x * kBlock;
    const TraceStack st{s_stack + threadIdx.x, Wk.overflow + gtid, (uint32_t)kBlock, Wk.overflow_stride, nullptr};
    uint32_t* const out_counter = Wk.light_count + 2 * S.n_pos_lights;
    unsigned long long decided = 0, walks = 0;  // (walks: wave-uniform)
    const uint32_t rounds = (n_pairs + gstride - 1) / gstride;  // the same for every thread: block_push_qk is a block-wide operation
    // kPerThread rounds, then ONE push per block for all of them (a pair's place in the queue is not needed before:
    // its resume state is filed under (hit, light), not under that place)
    for (uint32_t it0 = 0; it0 < rounds; it0 += kPerThread) {
      uint32_t keep_bits = 0;
#pragma unroll 1
      for (uint32_t k = 0; k < (uint32_t)kPerThread && it0 + k < rounds; ++k) {
        const uint32_t i = (it0 + k) * gstride + gtid;
        const bool live = i < n_pairs;
        uint32_t hl = 0, lp = 0, hit_i = 0;
        bool judge = false, inside = false;
        Beam bm{};
        if (live) {
            hl = Wk.pairs[i];
            lp = hl & ((1u << Wk.light_shift) - 1u);
            hit_i = hl >> Wk.light_shift;
            const uint32_t slot = Wk.phong_queue[hit_i];
            const float4 ro = Wk.ray_o[slot], rdv = Wk.ray_d[slot];
            const f3 p = add3(mk3(ro.x, ro.y, ro.z), scale3(ro.w, mk3(rdv.x, rdv.y, rdv.z)));
            const DLight& L = S.pos_lights[lp];
            // A pair that is not inside the spot cone for EVERY jittered light position (one in five of the queued pairs
            // at config 3: a band of asin(0.25 / distance) either side of the cone's edge) has a beam all the same: a
            // clear beam leaves its rays only the cone test, which they take in k_shadow without a walk (kResumeClear).
            inside = surely_inside_cone(L, p);
            judge = inside || Wk.beam_state != nullptr;
            if (judge) {
                bm = make_beam(p, ld3(L.pos), S.beam_r, 0.01f);
                beam_store_origin(st, p, sub3(ld3(L.pos), p));
            }
        }
        uint32_t stop_block = 0;
        int stop_sp = 0;
        walks += (unsigned long long)__popcll(__ballot(judge));
        const bool beam_clear = beam_is_clear(S, st, judge, bm, &stop_block, &stop_sp) && judge;
        const bool clear = beam_clear && inside;
        if (clear) {
            Wk.vis[(size_t)hit_i * S.n_pos_lights + lp] = (uint32_t)B.reps;
            decided += (unsigned long long)B.reps;
        }
        if (live && !clear) {
            keep_bits |= 1u << k;
            if (Wk.beam_state) {  // the walk's stack is still in place (LDS + spill region) until the next round
                uint32_t* dst = Wk.beam_state + ((size_t)hit_i * S.n_pos_lights + lp) * kResumeWords;
                if (beam_clear) dst[0] = kResumeClear; else if (judge) write_resume(st, stop_block, stop_sp, dst); else dst[0] = kResumeRoot;
            }
        }
      }
      uint32_t pred[kPerThread], at[kPerThread][1];
#pragma unroll
      for (int k = 0; k < kPerThread; ++k) pred[k] = (keep_bits >> k) & 1u;
      uint32_t* const counters[1] = {out_counter};
      block_push_qk<1, kPerThread>(counters, 1, pred, at);
#pragma unroll
      for (int k = 0; k < kPerThread; ++k)
          if (pred[k]) Wk.pair_seg[at[k][0]] = Wk.pairs[(it0 + (uint32_t)k) * gstride + gtid];
    }
    for (int off = 32; off > 0; off >>= 1) decided += __shfl_down(decided, off);
    if (__lane_id() == 0 && decided) atomicAdd(Wk.frame_counters + kCounterSums + kSubCounters + (blockIdx.x & (kSubCounters - 1)), decided);
    if (__lane_id() == 0 && walks) atomicAdd(Wk.frame_counters + kCounterSums + 2 * kSubCounters + (blockIdx.x & (kSubCounters - 1)), walks);
}
// the call's three totals from the words the wavefronts added to (one wavefront)
__global__ void k_sum_counters(unsigned long long* fc) {
    unsigned long long a = fc[kCounterSums + threadIdx.x], b = fc[kCounterSums + kSubCounters + threadIdx.x], c = fc[kCounterSums + 2 * kSubCounters + threadIdx.x];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); }
    if (threadIdx.x == 0) { fc[0] = a; fc[1] = b; fc[2] = c; }
}

// Scene set-up: one thread per child record; a leaf's record files the leaf's exact box and its slab under the leaf's
// first shape (DScene::leaf_bound, device_math.hpp "leaf slabs").  n = the mean of the leaf's triangle normals (each
// turned to the side of the first), narrowed to float32; dmin / dmax = the range of n . (vertex - box.min) over the
// leaf's vertices in float64 with that float32 n, widened by 2^-16 of the box's size and rounded outward; the same for
// n2, the one of 16 directions normal to n along which the leaf is narrowest (the second slab).  A sphere
// of the leaf adds its extent along n (centre -/+ radius) and no normal; a leaf of spheres only, or whose normals cancel,
// gets n = 0 and [-inf, +inf]: it is never culled.
__global__ __launch_bounds__(kBlock) void k_leaf_bounds(const float4* recs, const float4* prims, uint32_t n_recs, float4* leaf_bound) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_recs) return;
    const float4 ra = recs[2 * (size_t)i], rb = recs[2 * (size_t)i + 1];
    const uint32_t w0 = __float_as_uint(rb.z), w1 = __float_as_uint(rb.w);
    if (!(w1 & kLeafFlag)) return;
    const uint32_t first = w0 >> 8, count = w0 & 0xffu;
    // The box filed with the slabs must hold the shapes AS THE DEVICE TESTS THEM, in float32.  The record's box holds
    // the float64 shapes (rounded outward), hence every float32 vertex, but a sphere whose centre was rounded to float32
    // can stick out of it by half an ulp of the centre (scripts/stress_slabs.py: spheres of radius 0.004 at 10^5 units
    // from the origin): the box is grown to the float32 spheres' extents.
    double blo[3] = {(double)ra.x, (double)ra.y, (double)ra.z}, bhi[3] = {(double)ra.w, (double)rb.x, (double)rb.y};
    for (uint32_t k = 0; k < count; ++k) {
        const float4 p0 = prims[3 * (size_t)(first + k)], p2 = prims[3 * (size_t)(first + k) + 2];
        if (__float_as_uint(p2.y) & kPrimTriangle) continue;
        const double c[3] = {(double)p0.x, (double)p0.y, (double)p0.z}, r = fabs((double)p0.w) * 1.000001;
        for (int q = 0; q < 3; ++q) { blo[q] = c[q] - r < blo[q] ? c[q] - r : blo[q]; bhi[q] = c[q] + r > bhi[q] ? c[q] + r : bhi[q]; }
    }
    // ... and then by 8 ulps of its coordinates, so that a ray origin or a vertex exactly on a face of the leaf's box is strictly
    // inside the filed one (ray_misses_leaf_slab: a direction component of 0 turns the distance to a plane into 0 or an infinity)
    for (int q = 0; q < 3; ++q) { blo[q] -= fabs(blo[q]) * 9.5367431640625e-07 + 1e-30; bhi[q] += fabs(bhi[q]) * 9.5367431640625e-07 + 1e-30; }
    const float4 a = make_float4(__double2float_rd(blo[0]), __double2float_rd(blo[1]), __double2float_rd(blo[2]), __double2float_ru(bhi[0]));
    const float4 b = make_float4(__double2float_ru(bhi[1]), __double2float_ru(bhi[2]), rb.z, rb.w);
    double ns[3] = {0.0, 0.0, 0.0};
    for (uint32_t k = 0; k < count; ++k) {
        const float4 p0 = prims[3 * (size_t)(first + k)], p1 = prims[3 * (size_t)(first + k) + 1], p2 = prims[3 * (size_t)(first + k) + 2];
        if (!(__float_as_uint(p2.y) & kPrimTriangle)) continue;  // a sphere: no normal of its own, its extent is added below
        const double e1[3] = {(double)p0.w - p0.x, (double)p1.x - p0.y, (double)p1.y - p0.z}, e2[3] = {(double)p1.z - p0.x, (double)p1.w - p0.y, (double)p2.x - p0.z};
        double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        if (n[0] * ns[0] + n[1] * ns[1] + n[2] * ns[2] < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        ns[0] += n[0]; ns[1] += n[1]; ns[2] += n[2];
    }
    const double len = sqrt(ns[0] * ns[0] + ns[1] * ns[1] + ns[2] * ns[2]);
    float nf[3] = {0.0f, 0.0f, 0.0f};
    float dmin = -__builtin_inff(), dmax = __builtin_inff();
    if (len > 0.0 && len < 1e300) {
        for (int k = 0; k < 3; ++k) nf[k] = (float)(ns[k] / len);
        double lo = 1e300, hi = -1e300;
        for (uint32_t k = 0; k < count; ++k) {
            const float4 p0 = prims[3 * (size_t)(first + k)], p1 = prims[3 * (size_t)(first + k) + 1], p2 = prims[3 * (size_t)(first + k) + 2];
            if (!(__float_as_uint(p2.y) & kPrimTriangle)) {  // sphere {centre, radius}: n . (centre - box.min) -/+ radius |n|, |n| <= 1 + 2^-22
                const double t = ((double)nf[0] * ((double)p0.x - (double)a.x) + (double)nf[1] * ((double)p0.y - (double)a.y)) + (double)nf[2] * ((double)p0.z - (double)a.z);
                const double r = fabs((double)p0.w) * 1.000001;
                lo = t - r < lo ? t - r : lo; hi = t + r > hi ? t + r : hi;
                continue;
            }
            const double v[3][3] = {{p0.x, p0.y, p0.z}, {p0.w, p1.x, p1.y}, {p1.z, p1.w, p2.x}};
            for (int q = 0; q < 3; ++q) {
                const double t = ((double)nf[0] * (v[q][0] - (double)a.x) + (double)nf[1] * (v[q][1] - (double)a.y)) + (double)nf[2] * (v[q][2] - (double)a.z);
                lo = t < lo ? t : lo; hi = t > hi ? t : hi;
            }
        }
        const double pad = 1.52587890625e-05 * ((((double)a.w - a.x) + ((double)b.x - a.y)) + ((double)b.y - a.z)) + 1e-30;
        dmin = __double2float_rd(lo - pad);
        dmax = __double2float_ru(hi + pad);
    }
    // the second slab: of 16 directions in the plane normal to n, the one along which the leaf is narrowest
    float n2f[3] = {0.0f, 0.0f, 0.0f};
    float d2min = -__builtin_inff(), d2max = __builtin_inff();
    if (dmax < __builtin_inff()) {
        const double nn[3] = {(double)nf[0], (double)nf[1], (double)nf[2]};
        const int ax = fabs(nn[0]) < fabs(nn[1]) ? (fabs(nn[0]) < fabs(nn[2]) ? 0 : 2) : (fabs(nn[1]) < fabs(nn[2]) ? 1 : 2);
        double e[3] = {0.0, 0.0, 0.0};
        e[ax] = 1.0;
        double u[3] = {nn[1] * e[2] - nn[2] * e[1], nn[2] * e[0] - nn[0] * e[2], nn[0] * e[1] - nn[1] * e[0]};
        const double ul = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int k = 0; k < 3; ++k) u[k] /= ul;  // |n x e| >= sqrt(2/3) |n|: the smallest component of n was crossed out
        const double v[3] = {nn[1] * u[2] - nn[2] * u[1], nn[2] * u[0] - nn[0] * u[2], nn[0] * u[1] - nn[1] * u[0]};
        double best = 1e300;
        for (int q = 0; q < 16; ++q) {
            const double c = cos(q * (M_PI / 16.0)), sn = sin(q * (M_PI / 16.0));
            const float df[3] = {(float)(c * u[0] + sn * v[0]), (float)(c * u[1] + sn * v[1]), (float)(c * u[2] + sn * v[2])};
            double lo = 1e300, hi = -1e300;
            for (uint32_t k = 0; k < count; ++k) {
                const float4 p0 = prims[3 * (size_t)(first + k)], p1 = prims[3 * (size_t)(first + k) + 1], p2 = prims[3 * (size_t)(first + k) + 2];
                if (!(__float_as_uint(p2.y) & kPrimTriangle)) {
                    const double t = ((double)df[0] * ((double)p0.x - (double)a.x) + (double)df[1] * ((double)p0.y - (double)a.y)) + (double)df[2] * ((double)p0.z - (double)a.z);
                    const double r = fabs((double)p0.w) * 1.000001;
                    lo = t - r < lo ? t - r : lo; hi = t + r > hi ? t + r : hi;
                    continue;
                }
                const double vx[3][3] = {{p0.x, p0.y, p0.z}, {p0.w, p1.x, p1.y}, {p1.z, p1.w, p2.x}};
                for (int w = 0; w < 3; ++w) {
                    const double t = ((double)df[0] * (vx[w][0] - (double)a.x) + (double)df[1] * (vx[w][1] - (double)a.y)) + (double)df[2] * (vx[w][2] - (double)a.z);
                    lo = t < lo ? t : lo; hi = t > hi ? t : hi;
                }
            }
            if (hi - lo < best) {
                best = hi - lo;
                const double pad = 1.52587890625e-05 * ((((double)a.w - a.x) + ((double)b.x - a.y)) + ((double)b.y - a.z)) + 1e-30;
                n2f[0] = df[0]; n2f[1] = df[1]; n2f[2] = df[2];
                d2min = __double2float_rd(lo - pad); d2max = __double2float_ru(hi + pad);
            }
        }
    }
    float4* out = leaf_bound + kLeafBoundQuads * (size_t)first;
    out[0] = a;
    out[1] = make_float4(b.x, b.y, nf[0], nf[1]);
    out[2] = make_float4(nf[2], dmin, dmax, n2f[0]);
    out[3] = make_float4(n2f[1], n2f[2], d2min, d2max);
}

// Scene set-up: one lane per (positional light, cell) of the voxel grid (device_math.hpp "voxel beams"); grid-stride,
// every lane of a wavefront runs the same number of rounds (beam_is_clear is a wavefront-wide call).
__global__ __launch_bounds__(kBlock, RTGO_BEAM_WAVES) void k_voxel_beams(DScene S, uint32_t* overflow, uint32_t overflow_stride, uint8_t* vox, unsigned long long n_cells) {
    __shared__ uint32_t s_stack[kLdsPerLane * kBlock];
    const unsigned long long gtid = (unsigned long long)blockIdx.x * kBlock + threadIdx.x, gstride = (unsigned long long)gridDim.x * kBlock;
    const TraceStack st{s_stack + threadIdx.x, overflow + gtid, (uint32_t)kBlock, overflow_stride, nullptr};
    const unsigned long long total = n_cells * (unsigned long long)S.n_pos_lights;
    for (unsigned long long base = (unsigned long long)blockIdx.x * kBlock; base < total; base += gstride) {
        const unsigned long long i = base + threadIdx.x;
        const bool live = i < total;
        Beam bm{};
        if (live) {
            const int lp = (int)(i / n_cells);
            unsigned long long c = i - (unsigned long long)lp * n_cells;
            const int ix = (int)(c % (unsigned long long)S.vox_n[0]); c /= (unsigned long long)S.vox_n[0];
            const int iy = (int)(c % (unsigned long long)S.vox_n[1]), iz = (int)(c / (unsigned long long)S.vox_n[1]);
            f3 lo, hi;
            vox_cell_box(S, ix, iy, iz, &lo, &hi);
            const f3 L = ld3(S.pos_lights[lp].pos);
            const f3 ctr = mk3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
            // (half extents a little over hi - ctr: the centre is rounded)
            const f3 h = mk3((hi.x - lo.x) * 0.50001f, (hi.y - lo.y) * 0.50001f, (hi.z - lo.z) * 0.50001f);
            bm = make_box_beam(lo, hi, L, S.beam_r, 0.01f);
            beam_store_origin(st, ctr, sub3(L, ctr), h);
        }
        uint32_t stop_block = 0;
        int stop_sp = 0;
        const bool clear = beam_is_clear(S, st, live, bm, &stop_block, &stop_sp);
        if (live) vox[i] = clear ? 1 : 0;
    }
}

// One lane per (PhongBlinn hit, positional light, repetition lane).  group = lanes per (hit, light):
// a power of two <= 64; each lane runs repetitions r, r+group, ...  The visible repetitions of a
// (hit, light) are counted with a ballot popcount: integer, so the result has no summation order.
#ifndef RTGO_SHADOW_WAVES
#define RTGO_SHADOW_WAVES 7  // k_shadow ms per frame at config 3 (quantised walk): 5 waves (LDS stack 20) 416, 6 (15) 395, 7 (11) 392, 8 (8) 404; with pair beams (hard rays only): 6 (15) 299.4, 6 (13) 300.6, 7 (11) 293.1, 7 (10) 293.5
#endif
template <bool MANY_REPS>  // more than 64 repetitions: a lane runs several of a pair's rays, one after the other
__global__ __launch_bounds__(kBlock, RTGO_SHADOW_WAVES) void k_shadow(DScene S, Batch B, Work Wk, const uint32_t* n_hits_ptr, int bounce,
                                                   int group, int group_shift, const uint32_t* pairs, const uint32_t* pair_counter, const uint32_t* beam_state) {
    __shared__ uint32_t s_stack[kLdsPerLane * kBlock];
    (void)n_hits_ptr;
    __shared__ uint32_t s_wave[(kBlock / 64) * 128];
    const uint32_t gtid = xcd_block() * kBlock + threadIdx.x;
    const uint32_t gstride = gridDim.x * kBlock;
    const TraceStack st{s_stack + threadIdx.x, Wk.overflow + gtid, (uint32_t)kBlock, Wk.overflow_stride, s_wave + (threadIdx.x >> 6) * 128};
    // Little besides the ray itself may stay live across the walk (7 waves per SIMD leave 72 registers, and what does not
    // fit goes to scratch memory once per round: 44 bytes per ray, 89 GB of writes per frame at config 3, before this
    // loop kept its state in 32-bit and wave-uniform values and looked the pair up again after the walk).
    const uint32_t n_pairs = *pair_counter;  // written by k_pair_offsets / k_pair_beam, same stream
    const uint32_t pairs_per_round = gstride >> group_shift;  // the grid's threads are a multiple of 256, a group is at most 64
    const uint32_t rounds = (n_pairs + pairs_per_round - 1) / pairs_per_round;
    const uint32_t my_pair = gtid >> group_shift;
    const uint32_t r = gtid & (uint32_t)(group - 1);  // the same in every round: the stride is a multiple of the group
    const unsigned long long low_mask = group == 64 ? ~0ull : ((1ull << group) - 1ull);
    const int lane = (int)__lane_id();
    unsigned long long launched = 0, by_beam = 0;  // wave-uniform: rays walked, rays of pairs whose beam was clear (cone test only)
    const int iters = MANY_REPS ? (B.reps + group - 1) >> group_shift : 1;  // > 1 only when reps > 64
#ifdef RTGO_PROF
    uint32_t prof_prev_visits = 0, prof_prev_iters = 0;
#endif
    for (uint32_t it = 0; it < rounds; ++it) {
        // interleaved, not a contiguous range per block: ranges differ too much in cost (1024 blocks: 1760 vs 708 ms)
        const uint32_t pi = it * pairs_per_round + my_pair;  // (plan_render keeps n_pairs + a round of pairs below 2^32)
        const bool active = pi < n_pairs;
        uint32_t count = 0;
        for (int itn = 0; itn < iters; ++itn) {
            // every lane of the wavefront runs the same number of rounds so the ballot is convergent
            const uint32_t rep = r + (uint32_t)itn * (uint32_t)group;
            bool traced = false, beam_clear = false;
            f3 p = mk3(0.0f, 0.0f, 0.0f), dir = mk3(1.0f, 1.0f, 1.0f);
            float dist = 0.0f;
            const uint32_t* resume = nullptr;  // what the pair's beam walk left (k_pair_beam), filed under (hit, light)
            if (active && rep < (uint32_t)B.reps) {
                // everything about the ray is derived here, so nothing but the pair's index stays live
                // across the traversal below.  The pair queue is light-major: the lane groups of a wavefront are
                // consecutive hits (neighbouring pixels / samples) towards the SAME light, so all 64 shadow rays
                // are nearly parallel
                const uint32_t hl = pairs[pi];
                const uint32_t lp = hl & ((1u << Wk.light_shift) - 1u);
                const uint32_t hit_i = hl >> Wk.light_shift;
                const uint32_t slot = Wk.phong_queue[hit_i];
                const float4 ro = Wk.ray_o[slot], rdv = Wk.ray_d[slot];
                p = add3(mk3(ro.x, ro.y, ro.z), scale3(ro.w, mk3(rdv.x, rdv.y, rdv.z)));
                const uint2 pk = Wk.path_key[slot];
                const RngKey key{B.k0, B.k1, pk.x, pk.y};
                const DLight& L = S.pos_lights[lp];
                if (beam_state) resume = beam_state + ((size_t)hit_i * S.n_pos_lights + lp) * kResumeWords;
                const f3 jit = scale3(0.25f, random_in_unit_sphere(key, (uint32_t)bounce, P_SHADOW, lp * (uint32_t)B.reps + rep));
                const f3 shifted = add3(ld3(L.pos), jit);  // light.go:94 / :123
                const f3 ld = sub3(shifted, p);  // doesReachLight, light.go:141-160
                dist = norm3(ld);
                dir = scale3(1.0f / dist, ld);
                // light.go:126-130 as cos(angle) >= cos(Angle).  Unit(p - L') is -dir bit for bit (x - y = -(y - x), so the
                // same norm and the same quotient; only a zero component's sign differs, which a dot product does not see;
                // p = L' gives NaN either way), and a dot product with -dir is the negated dot product with dir: one
                // square root and one division per ray instead of two of each
                traced = L.kind != RT_LIGHT_SPOT || -dot3(ld3(L.look_dir), dir) >= L.cos_angle;
                // the pair's beam met no leaf (k_pair_beam, kResumeClear): this ray is unoccluded without a walk
                if (resume && resume[0] == kResumeClear) { beam_clear = traced; traced = false; }
            }
            launched += (unsigned long long)__popcll(__ballot(traced));
            by_beam += (unsigned long long)__popcll(__ballot(beam_clear));
            RT_PROF_POINT(2);  // shadow-ray rounds
            // all 64 lanes call: lanes without a ray of their own test shapes for the others
#ifdef RTGO_PROF
            // what fusing two consecutive rounds of a wavefront (a lane going on to its second ray as soon as its first is done) could give
            uint32_t pv = 0, pit = 0;
            const bool occ = occluded<true>(S, st, traced, p, dir, 0.01f, dist, resume, &pv, &pit);
            if (it & 1u) {
                uint32_t both = pv + prof_prev_visits, sum = both, mx = both;
                for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); const uint32_t o2 = __shfl_xor(mx, off); mx = mx > o2 ? mx : o2; }
                if (lane == 0) {
                    const uint32_t lower = (sum + 63u) / 64u, m2 = pit > prof_prev_iters ? pit : prof_prev_iters;
                    atomicAdd(&g_prof5[0], 1ull);
                    atomicAdd(&g_prof5[1], (unsigned long long)(pit + prof_prev_iters));
                    atomicAdd(&g_prof5[2], (unsigned long long)mx);
                    atomicAdd(&g_prof5[3], (unsigned long long)(lower > m2 ? lower : m2));
                }
            }
            prof_prev_visits = pv; prof_prev_iters = pit;
#else
            const bool occ = occluded<true>(S, st, traced, p, dir, 0.01f, dist, resume);
#endif
#ifdef RTGO_DEBUG_CHECK
            {   // every ray again from the root, without slabs: what do the bounds get wrong?
                DScene S2 = S;
                S2.leaf_bound = nullptr;
                const bool want_ray = traced || beam_clear;
                const bool occ2 = occluded<true>(S2, st, want_ray, p, dir, 0.01f, dist, nullptr);
                const bool vis1 = (traced && !occ) || beam_clear, vis2 = want_ray && !occ2;
                if (vis1 != vis2)
                    printf("WRONG pair %u rep %u: visible %d, in truth %d; beam_clear %d resume0 %08x p %.9g %.9g %.9g dir %.9g %.9g %.9g dist %.9g\n",
                           pi, rep, (int)vis1, (int)vis2, (int)beam_clear, resume ? resume[0] : 0u, p.x, p.y, p.z, dir.x, dir.y, dir.z, dist);
            }
#endif
            // the visible repetitions of a (hit, light) counted over its lane group: an integer, no summation order
            const unsigned long long m = __ballot((traced && !occ) || beam_clear);
            count += (uint32_t)__popcll((m >> (lane & ~(group - 1))) & low_mask);
        }
        if (r == 0) {
            // the pair is looked up again (from the lane's constant and the round, not from `pi`, which the compiler
            // would otherwise keep in scratch memory across the walk, address and all)
            uint32_t mine = my_pair;
            asm volatile("" : "+v"(mine));
            const uint32_t pj = it * pairs_per_round + mine;
            if (pj < n_pairs) {
                const uint32_t hl = pairs[pj];
                Wk.vis[(size_t)(hl >> Wk.light_shift) * S.n_pos_lights + (hl & ((1u << Wk.light_shift) - 1u))] = count;
            }
        }
    }
    // shadow rays actually traced: one 64-bit atomic per wavefront
    if (lane == 0 && launched) atomicAdd(Wk.frame_counters + kCounterSums + (blockIdx.x & (kSubCounters - 1)), launched);
    if (lane == 0 && by_beam) atomicAdd(Wk.frame_counters + kCounterSums + kSubCounters + (blockIdx.x & (kSubCounters - 1)), by_beam);
}


// PhongBlinn.scatter (material.go:73-132) given the visible-repetition counts
__global__ __launch_bounds__(kBlock) void k_phong(DScene S, Batch B, Work Wk, const uint32_t* n_hits_ptr) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= *n_hits_ptr) return;
    const uint32_t slot = Wk.phong_queue[i];
    const PhongHit h = load_phong_hit(S, Wk, slot);
    f3 c = mk3(0.0f, 0.0f, 0.0f);
    int lp = 0;
    for (int q = 0; q < S.n_lights; ++q) {
        const DLight L = S.lights[q];
        if (L.kind != RT_LIGHT_AMBIENT) {
            const uint32_t cnt = Wk.vis[(size_t)i * S.n_pos_lights + lp];
            ++lp;
            if (cnt == 0) continue;
            const f3 comb = phong_term(L, h.m, h.mc, h.p, h.nrm, h.d);
            const float w = (float)cnt / (float)B.reps;  // cnt visible repetitions, each worth comb / reps
            c = add3(c, scale3(w, comb));
        } else {
            c = add3(c, scale3(L.intensity, ld3(L.color)));  // material.go:125
        }
    }
    c = mk3(min1(c.x), min1(c.y), min1(c.z));  // material.go:128-130
    const float4 th = Wk.thr[slot];
    Wk.rad[slot] = make_float4(th.x * c.x, th.y * c.y, th.z * c.z, 0.0f);
}

// raytracer.go:117: pixel += color(...), samples summed in index order (deterministic)
__global__ __launch_bounds__(kBlock) void k_accumulate(Batch B, Work Wk, float* accum) {
    const long long lpl = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (lpl >= B.npix) return;
    int x, row;
    if (!pixel_of(B, (uint32_t)lpl, &x, &row)) return;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    const float4* r = Wk.rad + lpl * B.ns;
    for (int k = 0; k < B.ns; ++k) { const float4 v = r[k]; sx += v.x; sy += v.y; sz += v.z; }
    float* a = accum + ((size_t)row * B.W + x) * 3;
    a[0] += sx; a[1] += sy; a[2] += sz;
}

// raytracer.go:119 (/spp) + :93-96 (uint8(min(255, c*255.99)), alpha 255)
__global__ __launch_bounds__(kBlock) void k_resolve(const float* accum, int n_pixels, float inv_spp, uchar4* out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pixels) return;
    uchar4 px;
    const float r = (inv_spp * accum[3 * i + 0]) * 255.99f, g = (inv_spp * accum[3 * i + 1]) * 255.99f, b = (inv_spp * accum[3 * i + 2]) * 255.99f;
    px.x = (unsigned char)(r < 255.0f ? r : 255.0f);
    px.y = (unsigned char)(g < 255.0f ? g : 255.0f);
    px.z = (unsigned char)(b < 255.0f ? b : 255.0f);
    px.w = 255;
    out[i] = px;
}

// accum[i] += other[i]: the shares of one frame summed on one device (only when a device is listed more than once,
// i.e. a rehearsal on fewer GPUs than shares; distinct devices are reduced by RCCL)
__global__ __launch_bounds__(kBlock) void k_add(float* accum, const float* other, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) accum[i] += other[i];
}

__global__ __launch_bounds__(kBlock) void k_trace_probe(DScene S, Work Wk, const float* rays, uint32_t n, int any_hit,
                                                        int* prim_out, float* t_out) {
    __shared__ uint32_t s_stack[kLdsPerLane * kBlock];
    __shared__ uint32_t s_wave[(kBlock / 64) * 128];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t gstride = gridDim.x * kBlock;
    const TraceStack st{s_stack + threadIdx.x, Wk.overflow + gtid, (uint32_t)kBlock, Wk.overflow_stride, s_wave + (threadIdx.x >> 6) * 128};
    // wave-uniform trip count: occluded() needs all 64 lanes of a wavefront
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gstride) {
        const uint32_t i = base + threadIdx.x;
        const bool live = i < n;
        const float* q = rays + (size_t)(live ? i : 0u) * 8;
        if (any_hit == 2) {
            // the instantiation k_shadow uses — one tMin for the whole call (ray 0's), idle lanes taking over
            // subtrees of walking lanes — on the caller's own rays
            const bool occ = occluded<true>(S, st, live, mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), rays[6], q[7]);
            if (live) { prim_out[i] = occ ? 1 : 0; t_out[i] = 0.0f; }
        } else if (any_hit) {
            const bool occ = occluded<false>(S, st, live, mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), q[6], q[7]);
            if (live) { prim_out[i] = occ ? 1 : 0; t_out[i] = 0.0f; }
        } else if (live) {
            float t; int prim;
            traverse(S, st, mk3(q[0], q[1], q[2]), mk3(q[3], q[4], q[5]), q[6], &t, &prim);
            prim_out[i] = prim >= 0 ? S.prim_ids[prim] : -1;  // report the position in Scene.Shapes
            t_out[i] = t;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_camera_probe(DCamera cam, int W, int H, uint32_t k0, uint32_t k1,
                                                         const int* pixels, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int x = pixels[3 * i], j = pixels[3 * i + 1], s = pixels[3 * i + 2];
    const RngKey key{k0, k1, (uint32_t)((H - 1 - j) * W + x), (uint32_t)s};
    f3 o, d;
    camera_ray(cam, W, H, x, j, key, &o, &d);
    float* r = out + (size_t)i * 6;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}

// material.scatter (material.go:34-132) of one ray per lane, for the parity probes: nearest hit, then what the
// shading stages compute for it.  out: 20 floats {should_trace, attenuation, scattered origin, scattered direction,
// colour, t, p, normal}; a miss leaves zeros and t = -1.  PhongBlinn hits trace their soft-shadow rays here, all
// 64 lanes of the wavefront together (occluded() shares its shape loop over the wavefront).
__global__ __launch_bounds__(kBlock) void k_scatter_probe(DScene S, Work Wk, const float* rays, const uint32_t* keys, uint32_t n,
                                                          uint32_t k0, uint32_t k1, int reps, float* out) {
    __shared__ uint32_t s_stack[kLdsPerLane * kBlock];
    __shared__ uint32_t s_wave[(kBlock / 64) * 128];
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t gstride = gridDim.x * kBlock;
    const TraceStack st{s_stack + threadIdx.x, Wk.overflow + gtid, (uint32_t)kBlock, Wk.overflow_stride, s_wave + (threadIdx.x >> 6) * 128};
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gstride) {  // wave-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool live = i < n;
        const float* q = rays + (size_t)(live ? i : 0u) * 6;
        const uint32_t* kq = keys + (size_t)(live ? i : 0u) * 3;
        const f3 o = mk3(q[0], q[1], q[2]), d = mk3(q[3], q[4], q[5]);
        const RngKey key{k0, k1, kq[0], kq[1]};
        const uint32_t bounce = kq[2];
        float t = -1.0f; int prim = -1;
        if (live) traverse(S, st, o, d, 0.0f, &t, &prim);
        float r[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) r[k] = 0.0f;
        r[13] = -1.0f;
        bool phong = false;
        PhongHit h{};
        if (live && prim >= 0) {
            const PrimData P = load_prim(S, prim);
            const DMaterial m = S.materials[__float_as_uint(P.p2.y) >> 2];
            const f3 p = add3(o, scale3(t, d));
            const f3 nrm = prim_normal(P, o, d, t);
            r[13] = t; r[14] = p.x; r[15] = p.y; r[16] = p.z; r[17] = nrm.x; r[18] = nrm.y; r[19] = nrm.z;
            if (m.kind == RT_MAT_STANDARD) {
                f3 col = ld3(m.color);
                if (m.texture >= 0) {
                    float u, v;
                    texture_map(P, p, &u, &v);
                    col = texture_color(S.textures[m.texture], u, v);
                }
                r[4] = p.x; r[5] = p.y; r[6] = p.z;
                r[10] = col.x; r[11] = col.y; r[12] = col.z;
            } else if (m.kind == RT_MAT_PHONGBLINN) {
                phong = true;
                h.p = p; h.nrm = nrm; h.d = d; h.m = m; h.mc = ld3(m.color);
                if (m.texture >= 0) {
                    float u, v;
                    texture_map(P, p, &u, &v);
                    h.mc = texture_color(S.textures[m.texture], u, v);
                }
            } else {
                const Scattered sc = scatter_specular(m, d, p, nrm, key, bounce);
                r[0] = sc.cont ? 1.0f : 0.0f;
                r[1] = sc.att.x; r[2] = sc.att.y; r[3] = sc.att.z;
                r[4] = sc.o.x; r[5] = sc.o.y; r[6] = sc.o.z;
                r[7] = sc.d.x; r[8] = sc.d.y; r[9] = sc.d.z;
            }
        }
        // PhongBlinn.scatter (material.go:73-132) with the rules of k_phong_pairs / k_shadow / k_phong
        f3 c = mk3(0.0f, 0.0f, 0.0f);
        uint32_t lp = 0;
        for (int ql = 0; ql < S.n_lights; ++ql) {
            const DLight L = S.lights[ql];
            if (L.kind == RT_LIGHT_AMBIENT) {
                c = add3(c, scale3(L.intensity, ld3(L.color)));
                continue;
            }
            f3 comb = mk3(0.0f, 0.0f, 0.0f);
            bool keep = false;
            if (phong) {
                comb = phong_term(L, h.m, h.mc, h.p, h.nrm, h.d);
                keep = !(comb.x == 0.0f && comb.y == 0.0f && comb.z == 0.0f);
            }
            uint32_t visible = 0;
            for (int rep = 0; rep < reps; ++rep) {
                bool traced = false;
                f3 dir = mk3(1.0f, 1.0f, 1.0f);
                float dist = 0.0f;
                if (keep) {
                    const f3 jit = scale3(0.25f, random_in_unit_sphere(key, bounce, P_SHADOW, lp * (uint32_t)reps + (uint32_t)rep));
                    const f3 shifted = add3(ld3(L.pos), jit);
                    bool in_cone = true;
                    if (L.kind == RT_LIGHT_SPOT) in_cone = dot3(ld3(L.look_dir), unit3(sub3(h.p, shifted))) >= L.cos_angle;
                    if (in_cone) {
                        traced = true;
                        const f3 ld = sub3(shifted, h.p);
                        dist = norm3(ld);
                        dir = scale3(1.0f / dist, ld);
                    }
                }
                const bool occ = occluded<true>(S, st, traced, h.p, dir, 0.01f, dist);
                if (traced && !occ) ++visible;
            }
            if (visible > 0) c = add3(c, scale3((float)visible / (float)reps, comb));
            ++lp;
        }
        if (phong) { r[10] = min1(c.x); r[11] = min1(c.y); r[12] = min1(c.z); }
        if (live) {
            float* w = out + (size_t)i * 20;
#pragma unroll
            for (int k = 0; k < 20; ++k) w[k] = r[k];
        }
    }
}

}  // namespace

// ============================================================================================
// Host runtime
// ============================================================================================
// A device buffer that is kept between calls and only ever grows (frame and probe buffers of the host-pointer
// entry points: no hipMalloc / hipFree per call).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        const hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// Batch workspace of a render call: ONE device allocation, carved into the arrays of `Work`.
struct Arena {
    Work wk{};
    unsigned char* base = nullptr;   // start of the allocation (owned: hipMalloc'd by the scene)
    size_t bytes = 0;
    bool owned = true;
    int64_t paths = 0; int depth = 0, lpos = 0;  // what wk was carved for
    unsigned long long overflow_entries = 0;     // (spill + 1) * stride
    int grid_cap = kMaxGridBlocks;               // most blocks a grid-stride launch may use with this arena
};

constexpr int kSnapshots = 3;  // pinned copies of the counter block: two alternate per bounce, one for the end of a call

// What one stream of batches needs for itself.  A render call deals its batches out over kLanes lanes, each with a
// host thread, a stream and a workspace of its own: k_shadow is bound by VALU issue and everything else in a batch
// by HBM, so one lane's ray generation, nearest-hit walk, shading and accumulation run under the other's shadow
// rays (config 3: 393 -> 373 ms per frame with two lanes; with the k_shadow launches of the two lanes made to take
// turns through events, so that two of them never share the GPU: 381 ms; with the pair beams in place 2 / 3 / 4 / 6 / 8
// lanes: 334 / 330 / 325 / 331 / 328 ms over the frame's 16 batches).  A launch's duration now includes the time slices the
// other lane's kernels got, so the per-kernel times of a two-lane frame add up to more than the frame
// (RTGO_LANES=1 measures a kernel alone).  A pixel's batches always go to the same lane, in the order of the
// single-lane loop, so the frame is bit-identical for any number of lanes.
struct Lane {
    Arena arena;     // grown on demand
    uint32_t* h_counters = nullptr;  // pinned, kSnapshots * h_counter_words
    size_t h_counter_words = 0;
    hipEvent_t ev_snap[2] = {}, ev_frame[2] = {};
    bool ev_ok = false;
    DevBuf beam_state;  // per (path, light) of a batch: what a pair's rays resume from (48 bytes); only calls that walk pair beams make it
    std::vector<hipEvent_t> ev_pool;  // per-stage timing of a call that asks for stats: 4 per (batch, bounce), reused
    hipStream_t stream = nullptr;     // lanes beyond the first: their own stream (lane 0 runs on the caller's)
    hipEvent_t ev_join = nullptr;
};
constexpr int kLanes = 4;

struct rt_scene {
    int device = 0;
    DScene d{};
    void* d_recs = nullptr; void* d_qblocks = nullptr; void* d_prims = nullptr; void* d_tprims = nullptr; void* d_prim_ids = nullptr; void* d_block_first = nullptr; void* d_block_box = nullptr; void* d_leaf_bound = nullptr; void* d_vox = nullptr; void* d_mats = nullptr; void* d_texs = nullptr; void* d_lights = nullptr;
    std::vector<void*> d_tex_pixels;
    rt_scene_info info{};
    double ms_build = 0;
    // camera inputs (NewCamera needs the aspect ratio, so the camera is finished per render)
    rt_vec3 from, at, up; double aperture = 0, fov = 0;
    Lane lanes[kLanes];
    int spill = 0;   // traversal-stack entries beyond the LDS part (from the tree's exact worst case)
    hipEvent_t ev_fork = nullptr, ev_wall[2] = {};
    DevBuf accum, rgba, probe_in, probe_out, probe_out2;
};

// ---- workspace of rt_render_accum_device_static, the entry point that allocates nothing ------------------------
// A file-scope device array: the loader places one copy on every GPU the library runs on, at an address of its
// own there; it is held until the process ends.  In host code the NAME below is only a stand-in for the symbol:
// the memory is reached through the pointer hipGetSymbolAddress() gives for the current device (rt_init), never
// through the name.
constexpr int64_t kStaticPaths = (int64_t)1 << 18;  // (pixel, sample) paths of one batch
constexpr int kStaticLights = 4;                    // positional lights
constexpr int kStaticDepth = 64;                    // RayTracingMaxDepth
constexpr int kStaticSpill = 44;                    // traversal-stack entries per lane beyond the LDS part
constexpr int kStaticGrid = 768;                    // blocks of a grid-stride launch (the spill region is per thread of it)
constexpr size_t kScratchBytes = (size_t)72 << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[kMaxDevices] = {};  // per device: where `scratch` lives there

namespace {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

// Lays the arrays of a batch workspace out over `base` (nullptr: only measure).  Returns the bytes needed.
size_t carve_work(Work* w, unsigned char* base, int64_t paths, int depth, int lpos, unsigned long long overflow_entries) {
    size_t off = 0;
    auto take = [&](size_t n) { unsigned char* p = base ? base + off : nullptr; off += align_up(n); return p; };
    const size_t P = (size_t)paths, L = (size_t)(lpos > 0 ? lpos : 1);
    w->ray_o = (float4*)take(P * sizeof(float4));
    w->ray_d = (float4*)take(P * sizeof(float4));
    w->thr = (float4*)take(P * sizeof(float4));
    w->rad = (float4*)take(P * sizeof(float4));
    w->path_key = (uint2*)take(P * sizeof(uint2));
    w->queue[0] = (uint32_t*)take(P * sizeof(uint32_t));
    w->queue[1] = (uint32_t*)take(P * sizeof(uint32_t));
    w->phong_queue = (uint32_t*)take(P * sizeof(uint32_t));
    w->vis = (uint32_t*)take(P * L * sizeof(uint32_t));
    w->pairs = (uint32_t*)take(P * L * sizeof(uint32_t));
    w->pair_seg = (uint32_t*)take(P * L * sizeof(uint32_t));
    w->light_count = (uint32_t*)take((2 * L + 2) * sizeof(uint32_t));
    w->counters = (uint32_t*)take((size_t)(3 * depth + 8) * sizeof(uint32_t));
    w->frame_counters = (unsigned long long*)take((kCounterSums + kCounterSums * kSubCounters) * sizeof(unsigned long long));
    w->overflow = (uint32_t*)take((size_t)overflow_entries * sizeof(uint32_t));
    w->beam_state = nullptr;  // a buffer of its own (Lane::beam_state), made by the calls that walk pair beams
    return off;
}

// Blocks of the widest grid-stride launch over `lanes` work items, and the spill region that goes with it:
// (spill + 1) entries for every thread of that grid.
inline unsigned long long grid_blocks(unsigned long long lanes, int cap) {
    unsigned long long blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks < 256) blocks = 256;
    return blocks > (unsigned long long)cap ? (unsigned long long)cap : blocks;
}

int ensure_work(rt_scene* s, Lane& ln, int64_t paths, int depth, int lpos, unsigned long long lanes) {
    Arena& a = ln.arena;
    const unsigned long long stride = grid_blocks(lanes, a.grid_cap) * kBlock;
    const unsigned long long entries = s->spill > 0 ? (unsigned long long)(s->spill + 1) * stride : 0ull;
    if (a.base && paths <= a.paths && depth <= a.depth && lpos <= a.lpos && entries <= a.overflow_entries) {
        a.wk.overflow_stride = (uint32_t)(a.overflow_entries / (unsigned long long)(s->spill + 1));
        return RT_OK;
    }
    // grow, never shrink: a later, smaller call finds everything in place
    const int64_t np = std::max(paths, a.paths);
    const int nd = std::max(depth, a.depth), nl = std::max(lpos, a.lpos);
    const unsigned long long ne = std::max(entries, a.overflow_entries);
    Work w{};
    const size_t need = carve_work(&w, nullptr, np, nd, nl, ne);
    if (a.base) { (void)hipFree(a.base); a.base = nullptr; a.bytes = 0; a.paths = 0; a.depth = 0; a.lpos = 0; a.overflow_entries = 0; }
    HIP_TRY(hipMalloc((void**)&a.base, need));
    a.bytes = need;
    (void)carve_work(&a.wk, a.base, np, nd, nl, ne);
    a.paths = np; a.depth = nd; a.lpos = nl; a.overflow_entries = ne;
    a.wk.overflow_stride = s->spill > 0 ? (uint32_t)(ne / (unsigned long long)(s->spill + 1)) : 0u;
    const size_t words = (size_t)(3 * nd + 8);
    if (words > ln.h_counter_words) {
        if (ln.h_counters) { (void)hipHostFree(ln.h_counters); ln.h_counters = nullptr; ln.h_counter_words = 0; }
        HIP_TRY(hipHostMalloc((void**)&ln.h_counters, kSnapshots * words * sizeof(uint32_t)));
        ln.h_counter_words = words;
    }
    return RT_OK;
}

// Makes the spill region large enough for a probe launch over `lanes` rays (the render workspace is kept).
int ensure_overflow(rt_scene* s, unsigned long long lanes) {
    const Arena& a = s->lanes[0].arena;
    return ensure_work(s, s->lanes[0], std::max<int64_t>(1, a.paths), std::max(1, a.depth), std::max(1, a.lpos), lanes);
}

// NewCamera, camera.go:18-35, with the arguments GenerateImage passes (raytracer.go:59-68)
DCamera make_camera(const rt_scene* s, int W, int H) {
    const double fx = s->from.x - s->at.x, fy = s->from.y - s->at.y, fz = s->from.z - s->at.z;
    const double focus = std::sqrt(fx * fx + fy * fy + fz * fz);
    const double aspect = (double)W / (double)H;
    const double theta = s->fov * M_PI / 180.0;
    const double hh = std::tan(theta / 2), hw = aspect * hh;
    auto unit = [](double v[3]) {
        if (v[0] == 0 && v[1] == 0 && v[2] == 0) { v[0] = v[1] = v[2] = NAN; return; }
        const double k = 1 / std::hypot(v[0], std::hypot(v[1], v[2]));
        v[0] *= k; v[1] *= k; v[2] *= k;
    };
    double w[3] = {fx, fy, fz};
    unit(w);
    double u[3] = {s->up.y * w[2] - s->up.z * w[1], s->up.z * w[0] - s->up.x * w[2], s->up.x * w[1] - s->up.y * w[0]};
    unit(u);
    const double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
    const double from[3] = {s->from.x, s->from.y, s->from.z};
    DCamera c;
    for (int a = 0; a < 3; ++a) {
        c.u[a] = (float)u[a]; c.v[a] = (float)v[a]; c.origin[a] = (float)from[a];
        c.ll[a] = (float)(((from[a] - (hw * focus) * u[a]) - (hh * focus) * v[a]) - focus * w[a]);
        c.hor[a] = (float)((2 * hw * focus) * u[a]);
        c.ver[a] = (float)((2 * hh * focus) * v[a]);
    }
    c.lens = (float)(s->aperture / 2);
    return c;
}

int validate_spec(const rt_image_spec* spec) {
    if (!spec) return fail(RT_ERR_INVALID, "spec is NULL");
    if (spec->bvh_traversal_algorithm != RT_BVH_DIJKSTRA && spec->bvh_traversal_algorithm != RT_BVH_DEPTH_FIRST_SEARCH)
        return fail(RT_ERR_ALGORITHM, "No trace algorithm found for " + std::to_string(spec->bvh_traversal_algorithm));
    if (spec->width <= 0 || spec->height <= 0) return fail(RT_ERR_INVALID, "width and height must be positive");
    if ((int64_t)spec->width * spec->height > ((int64_t)1 << 31) - 1) return fail(RT_ERR_LIMIT, "image has more than 2^31-1 pixels");
    if (spec->anti_aliasing_factor < 0 || spec->ray_tracing_max_depth < 0 || spec->soft_shadow_monte_carlo_repetitions < 0)
        return fail(RT_ERR_INVALID, "negative AntiAliasingFactor / RayTracingMaxDepth / SoftShadowMonteCarloRepetitions");
    if (spec->ray_tracing_max_depth > (1 << 20)) return fail(RT_ERR_LIMIT, "RayTracingMaxDepth > 2^20");
    if (spec->soft_shadow_monte_carlo_repetitions > (1 << 16)) return fail(RT_ERR_LIMIT, "SoftShadowMonteCarloRepetitions > 65536");
    return RT_OK;
}

inline int grid_for(unsigned long long n, int cap = kMaxGridBlocks) {
    unsigned long long b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return (int)(b < (unsigned long long)cap ? b : (unsigned long long)cap);
}

}  // namespace

// f(first, last) over [0, n) in chunks on a few host threads (long tables only)
template <class F>
static void host_chunks(int64_t n, F&& f) {
    int nt = (int)std::thread::hardware_concurrency();
    nt = nt > 8 ? 8 : (nt < 1 ? 1 : nt);
    if (n < 65536) nt = 1;
    ThreadGroup group;
    for (int c = 1; c < nt; ++c) group.run([&f, n, nt, c] { f(n * c / nt, n * (c + 1) / nt); });
    f(0, n / nt);
    group.join();
}

// The public entry points: no C++ exception crosses the C boundary (the reference panics; here the caller
// gets an error code and rt_last_error()).
template <class F>
static int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_LIMIT, "out of host memory");
    } catch (const std::invalid_argument& e) {
        return fail(RT_ERR_INVALID, e.what());
    } catch (const std::exception& e) {
        return fail(RT_ERR_DEVICE, std::string("host runtime error: ") + e.what());
    } catch (...) {
        return fail(RT_ERR_DEVICE, "unknown host runtime error");
    }
}

// f(rank) on one host thread per share; the first error (code + text) comes back on the calling thread
template <class F>
static int on_every_rank(int n, F&& f) {
    std::vector<int> rc((size_t)n, RT_OK);
    std::vector<std::string> msg((size_t)n);
    {
        ThreadGroup group;
        for (int r = 1; r < n; ++r)
            group.run([&, r] { rc[(size_t)r] = guarded([&] { return f(r); }); if (rc[(size_t)r] != RT_OK) msg[(size_t)r] = g_err; });
        rc[0] = guarded([&] { return f(0); });
        if (rc[0] != RT_OK) msg[0] = g_err;
        group.join();
    }
    for (int r = 0; r < n; ++r)
        if (rc[(size_t)r] != RT_OK) return fail(rc[(size_t)r], "share " + std::to_string(r) + ": " + msg[(size_t)r]);
    return RT_OK;
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

// The HIP runtime sets up its copy path on the first hipMemcpy of the process (about 150 ms on an MI355X box,
// RTGO_TIMING=1; a first hipMalloc alone does not trigger it).  The first rt_init of a device starts a thread that
// pays that with a small copy and memset while the caller goes on (flattening its scene, then the tree build in
// rt_scene_create, which waits for the thread before its first upload): one-shot frame 0.83 s -> 0.71 s when the thread
// started with the tree build, and less again when the caller calls rt_init before it prepares the scene.
static std::mutex g_warm_mutex;
static std::shared_future<void> g_warm[kMaxDevices];
static void start_warm_up(int device) {
    if (device < 0 || device >= kMaxDevices) return;
    std::lock_guard<std::mutex> g(g_warm_mutex);
    if (g_warm[device].valid()) return;
    g_warm[device] = std::async(std::launch::async, [device] {
        void* p = nullptr;
        std::vector<char> host((size_t)4 << 20, 1);
        if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, host.size()) != hipSuccess) return;
        (void)hipMemcpy(p, host.data(), host.size(), hipMemcpyHostToDevice);
        (void)hipMemset(p, 0, host.size());
        (void)hipFree(p);
    }).share();
}
static void wait_warm_up(int device) {
    std::shared_future<void> f;
    {
        std::lock_guard<std::mutex> g(g_warm_mutex);
        if (device >= 0 && device < kMaxDevices) f = g_warm[device];
    }
    if (f.valid()) f.wait();
}

int rt_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RT_ERR_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(RT_ERR_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_DEVICE, std::string("librtgo_hip is built for gfx950 only, found ") + prop.gcnArchName);
    if (device < kMaxDevices) {  // once per device, with that device current: this GPU's copy of the static workspace
        std::lock_guard<std::mutex> g(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            unsigned char* ptr = nullptr;
            HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
            g_scratch_ptr[device] = ptr;
        }
    }
    start_warm_up(device);
    return RT_OK;
}

// ---- scene creation, in the steps rt_scene_create and rt_multi_create put together ----
// RTGO_TIMING=1: the phases of scene creation on stderr (host tree build, flattening, tables, uploads)
struct Laps {
    bool on = std::getenv("RTGO_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void operator()(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rtgo] scene create: %-10s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

static int validate_desc(const rt_scene_desc* desc) {
    if (!desc) return fail(RT_ERR_INVALID, "desc is NULL");
    if (desc->n_shapes < 0 || desc->n_materials < 0 || desc->n_textures < 0 || desc->n_lights < 0)
        return fail(RT_ERR_INVALID, "negative table size");
    if (desc->n_shapes >= kMaxChildRecords) return fail(RT_ERR_LIMIT, "scene has " + std::to_string(desc->n_shapes) + " shapes; the limit is 2^24 - 1 (a leaf keeps 24 bits for its first shape)");
    if (desc->n_materials >= ((int64_t)1 << 30)) return fail(RT_ERR_LIMIT, "more than 2^30 materials");
    if ((desc->n_shapes > 0 && !desc->shapes) || (desc->n_materials > 0 && !desc->materials) ||
        (desc->n_textures > 0 && !desc->textures) || (desc->n_lights > 0 && !desc->lights))
        return fail(RT_ERR_INVALID, "table pointer is NULL but its size is not 0");
    // fp32 on the device: coordinates must be finite and well inside the float range
    auto usable = [](const rt_vec3& v) {
        return std::fabs(v.x) < 1e30 && std::fabs(v.y) < 1e30 && std::fabs(v.z) < 1e30;  // false for NaN / Inf too
    };
    for (int64_t i = 0; i < desc->n_shapes; ++i) {
        const rt_shape& s = desc->shapes[i];
        if (s.kind != RT_SHAPE_SPHERE && s.kind != RT_SHAPE_TRIANGLE) return fail(RT_ERR_INVALID, "unknown shape kind");
        if (s.material < 0 || s.material >= desc->n_materials) return fail(RT_ERR_INVALID, "shape material index out of range");
        // the reference's BVH build does not terminate on a NaN centroid (accelerationstructures.go:226-279)
        if (!usable(s.a) || !(std::fabs(s.b.x) < 1e30) || (s.kind == RT_SHAPE_TRIANGLE && (!usable(s.b) || !usable(s.c))))
            return fail(RT_ERR_INVALID, "shape " + std::to_string(i) + " has a non-finite (or > 1e30) coordinate");
    }
    for (int64_t i = 0; i < desc->n_materials; ++i) {
        const rt_material& m = desc->materials[i];
        if (m.kind < RT_MAT_STANDARD || m.kind > RT_MAT_PHONGBLINN) return fail(RT_ERR_INVALID, "unknown material kind");
        if (m.texture >= desc->n_textures) return fail(RT_ERR_INVALID, "material texture index out of range");
    }
    for (int64_t i = 0; i < desc->n_textures; ++i) {
        const rt_texture& t = desc->textures[i];
        if (t.kind != RT_TEX_CHECKERS && t.kind != RT_TEX_IMAGE) return fail(RT_ERR_INVALID, "unknown texture kind");
        if (t.kind == RT_TEX_IMAGE && (!t.rgba || t.width <= 0 || t.height <= 0)) return fail(RT_ERR_INVALID, "image texture without pixels");
    }
    for (int64_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].kind < RT_LIGHT_AMBIENT || desc->lights[i].kind > RT_LIGHT_SPOT) return fail(RT_ERR_INVALID, "unknown light kind");
    return RT_OK;
}

// What every device's copy of a scene is uploaded from: built once on the host, whatever the number of devices.
struct HostScene {
    FlatTree flat;
    size_t prim_words = 0;
    std::unique_ptr<float[]> prims, tprims;
    std::vector<DMaterial> mats;
    std::vector<DLight> lights;
    int n_pos = 0;
};


static int build_host_scene(const rt_scene_desc* desc, HostScene* h, Laps& lap) {
    {
        const char* err = nullptr;
        // shapes per leaf: 1 = the reference's tree; larger leaves trade box tests for shape tests
        const char* ls = std::getenv("RTGO_LEAF_SIZE");
        const char* tree = std::getenv("RTGO_TREE");
        const bool octree = tree && std::strcmp(tree, "octree") == 0;
        const int leaf_size = ls ? std::atoi(ls) : (octree ? kDefaultLeafSize : kDefaultLeafSizeSah);
        // default: binned-SAH tree (octree.hpp); RTGO_TREE=octree builds the reference's insertion octree
        const HostNodes nodes = octree
                                                ? build_octree(desc->shapes, desc->n_shapes)
                                                : build_sah(desc->shapes, desc->n_shapes, leaf_size);
        lap("tree");
        // (the boxes' pad covers the slab test's error for rays that start anywhere among the shapes or on the camera's lens)
        const double lens_reach = std::max(std::max(std::fabs(desc->camera_look_from.x), std::fabs(desc->camera_look_from.y)), std::fabs(desc->camera_look_from.z)) +
                                  std::fabs(desc->camera_aperture) * 0.5;
        if (!flatten_octree(nodes, leaf_size, &h->flat, &err, lens_reach)) return fail(RT_ERR_LIMIT, err);
        lap("flatten");
    }
    lap("free nodes");
    // shapes -> 3 float4 each, narrowed to fp32 (round to nearest), stored in leaf order
    // tprims: the same records for the intersection tests, with the triangle's edges e1 = B - A, e2 = C - A (fp32
    // differences of the fp32 vertices, exactly what the tests would compute) in place of B and C
    // (not zeroed first: the threads write all 12 words of every record, and the pages are theirs to touch)
    const size_t prim_words = h->prim_words = (size_t)desc->n_shapes * 12 + 12;
    h->prims.reset(new float[prim_words]);
    h->tprims.reset(new float[prim_words]);
    float* const prims = h->prims.get();
    float* const tprims = h->tprims.get();
    for (size_t k = prim_words - 12; k < prim_words; ++k) prims[k] = tprims[k] = 0.0f;  // the padding record
    host_chunks(desc->n_shapes, [&](int64_t q0, int64_t q1) {
        for (int64_t q = q0; q < q1; ++q) {
            const rt_shape& s = desc->shapes[h->flat.order[(size_t)q]];
            float* p = prims + (size_t)q * 12;
            float* t = tprims + (size_t)q * 12;
            for (int k = 0; k < 12; ++k) p[k] = t[k] = 0.0f;
            uint32_t bits = (uint32_t)s.material << 2;
            if (s.kind == RT_SHAPE_TRIANGLE) {
                bits |= kPrimTriangle | (s.single_sided ? kPrimSingleSided : 0u);
                p[0] = (float)s.a.x; p[1] = (float)s.a.y; p[2] = (float)s.a.z;
                p[3] = (float)s.b.x; p[4] = (float)s.b.y; p[5] = (float)s.b.z;
                p[6] = (float)s.c.x; p[7] = (float)s.c.y; p[8] = (float)s.c.z;
                t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
                t[3] = p[3] - p[0]; t[4] = p[4] - p[1]; t[5] = p[5] - p[2];
                t[6] = p[6] - p[0]; t[7] = p[7] - p[1]; t[8] = p[8] - p[2];
            } else {
                p[0] = (float)s.a.x; p[1] = (float)s.a.y; p[2] = (float)s.a.z; p[3] = (float)s.b.x;
                t[0] = p[0]; t[1] = p[1]; t[2] = p[2]; t[3] = p[3];
            }
            std::memcpy(&p[9], &bits, 4);
            std::memcpy(&t[9], &bits, 4);
        }
    });
    std::vector<DMaterial>& mats = h->mats;
    mats.assign((size_t)desc->n_materials + 1, DMaterial{});
    for (int64_t i = 0; i < desc->n_materials; ++i) {
        const rt_material& m = desc->materials[i];
        DMaterial& d = mats[(size_t)i];
        d.kind = m.kind; d.texture = m.texture < 0 ? -1 : m.texture;
        d.color[0] = (float)m.color.x; d.color[1] = (float)m.color.y; d.color[2] = (float)m.color.z;
        d.spec[0] = (float)m.specular_color.x; d.spec[1] = (float)m.specular_color.y; d.spec[2] = (float)m.specular_color.z;
        d.param = (float)m.param;
    }


    std::vector<DLight>& lights = h->lights;
    lights.assign((size_t)desc->n_lights + 1, DLight{});
    int n_pos = 0;
    for (int64_t i = 0; i < desc->n_lights; ++i) {
        const rt_light& l = desc->lights[i];
        DLight& d = lights[(size_t)i];
        d.kind = l.kind;
        d.color[0] = (float)l.color.x; d.color[1] = (float)l.color.y; d.color[2] = (float)l.color.z;
        d.pos[0] = (float)l.position.x; d.pos[1] = (float)l.position.y; d.pos[2] = (float)l.position.z;
        double ld[3] = {l.look_at.x - l.position.x, l.look_at.y - l.position.y, l.look_at.z - l.position.z};
        if (l.kind == RT_LIGHT_SPOT) {  // light.go:127: Unit(LookAt - Position)
            if (ld[0] == 0 && ld[1] == 0 && ld[2] == 0) ld[0] = ld[1] = ld[2] = NAN;
            else { const double k = 1 / std::hypot(ld[0], std::hypot(ld[1], ld[2])); ld[0] *= k; ld[1] *= k; ld[2] *= k; }
        } else { ld[0] = ld[1] = ld[2] = 0; }
        d.look_dir[0] = (float)ld[0]; d.look_dir[1] = (float)ld[1]; d.look_dir[2] = (float)ld[2];
        d.intensity = (float)l.intensity; d.spec_intensity = (float)l.specular_intensity;
        d.cos_angle = (float)std::cos(l.angle_deg * M_PI / 180.0);
        d.decay = (float)l.decay;
        if (l.kind != RT_LIGHT_AMBIENT) ++n_pos;
    }
    h->n_pos = n_pos;
    for (int64_t i = 0; i < desc->n_lights; ++i)  // the positional lights once more, in order: pair -> light without a search
        if (lights[(size_t)i].kind != RT_LIGHT_AMBIENT) lights.push_back(lights[(size_t)i]);
    lap("tables");
    return RT_OK;
}

using LaneSetup = std::future<std::pair<hipError_t, const char*>>;
// A device's scene before its tables: rt_init (which starts the runtime's warm-up there) and a thread that makes the
// lanes' events, streams and pinned counters while the host builds the tree.  The caller owns *out from here on and
// lets `lane_setup` go (a std::async future waits in its destructor) BEFORE it destroys the scene.
static int begin_scene(int device, rt_scene** out, LaneSetup* lane_setup_out) {
    int rc = rt_init(device);
    if (rc != RT_OK) return rc;
    rt_scene* s = new rt_scene();
    s->device = device;
    *out = s;
    *lane_setup_out = std::async(std::launch::async, [s, device]() -> std::pair<hipError_t, const char*> {
#define HIP_TRY_L(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return {e_, #expr}; } while (0)
        HIP_TRY_L(hipSetDevice(device));
        for (int l = 0; l < kLanes; ++l) {
            Lane& ln = s->lanes[l];
            ln.ev_ok = true;  // from here on rt_scene_destroy() releases whatever of the lane exists
            ln.h_counter_words = (size_t)(3 * kStaticDepth + 8);
            HIP_TRY_L(hipHostMalloc((void**)&ln.h_counters, kSnapshots * ln.h_counter_words * sizeof(uint32_t)));
            for (auto& e : ln.ev_snap) HIP_TRY_L(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& e : ln.ev_frame) HIP_TRY_L(hipEventCreate(&e));
            HIP_TRY_L(hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
            if (l > 0) HIP_TRY_L(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        }
        HIP_TRY_L(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        for (auto& e : s->ev_wall) HIP_TRY_L(hipEventCreate(&e));
        return {hipSuccess, ""};
#undef HIP_TRY_L
    });
    return RT_OK;
}

// The tables of `h` uploaded to the scene's device.
static int finish_scene(const rt_scene_desc* desc, const HostScene& h, rt_scene* s, LaneSetup* lane_setup, Laps& lap) {
    const int device = s->device;
    HIP_TRY(hipSetDevice(device));
    wait_warm_up(device);  // started by rt_init
    lap("hip warm");
    s->from = desc->camera_look_from; s->at = desc->camera_look_at; s->up = desc->camera_up;
    s->aperture = desc->camera_aperture; s->fov = desc->camera_fov_deg;
    auto bail = [&](int code) { return code; };  // the guard releases the scene
#define HIP_TRY_S(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)

    std::vector<DTexture> texs((size_t)desc->n_textures + 1);
    int64_t tex_bytes = 0;
    for (int64_t i = 0; i < desc->n_textures; ++i) {
        const rt_texture& t = desc->textures[i];
        DTexture& d = texs[(size_t)i];
        d.kind = t.kind; d.w = t.width; d.h = t.height;
        d.c1[0] = (float)t.color1.x; d.c1[1] = (float)t.color1.y; d.c1[2] = (float)t.color1.z;
        d.c2[0] = (float)t.color2.x; d.c2[1] = (float)t.color2.y; d.c2[2] = (float)t.color2.z;
        d.cw = (float)t.checkers_width; d.ch = (float)t.checkers_height;
        d.rgba = nullptr;
        if (t.kind == RT_TEX_IMAGE) {
            const size_t n = (size_t)t.width * t.height * 4;
            void* px = nullptr;
            HIP_TRY_S(hipMalloc(&px, n));
            s->d_tex_pixels.push_back(px);
            HIP_TRY_S(hipMemcpy(px, t.rgba, n, hipMemcpyHostToDevice));
            d.rgba = (const uint8_t*)px;
            tex_bytes += (int64_t)n;
        }
    }
    lap("textures");
    const size_t rec_bytes = (h.flat.records.size() + 1) * sizeof(ChildRecord);
    if (h.flat.qblocks.empty()) return bail(fail(RT_ERR_DEVICE, "flattener produced no quantised blocks"));
    HIP_TRY_S(hipMalloc(&s->d_recs, rec_bytes));
    HIP_TRY_S(hipMemset(s->d_recs, 0, rec_bytes));  // the padding record is fetched (and masked out) for trailing absent slots
    if (!h.flat.records.empty()) HIP_TRY_S(hipMemcpy(s->d_recs, h.flat.records.data(), h.flat.records.size() * sizeof(ChildRecord), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_qblocks, h.flat.qblocks.size() * sizeof(uint32_t)));
    HIP_TRY_S(hipMemcpy(s->d_qblocks, h.flat.qblocks.data(), h.flat.qblocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_prims, h.prim_words * sizeof(float)));
    HIP_TRY_S(hipMemcpy(s->d_prims, h.prims.get(), h.prim_words * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_tprims, h.prim_words * sizeof(float)));
    HIP_TRY_S(hipMemcpy(s->d_tprims, h.tprims.get(), h.prim_words * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_prim_ids, (h.flat.order.size() + 1) * sizeof(int32_t)));
    if (!h.flat.order.empty()) HIP_TRY_S(hipMemcpy(s->d_prim_ids, h.flat.order.data(), h.flat.order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_block_first, h.flat.block_first.size() * sizeof(uint32_t)));
    HIP_TRY_S(hipMemcpy(s->d_block_first, h.flat.block_first.data(), h.flat.block_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_block_box, h.flat.block_box.size() * sizeof(float)));
    HIP_TRY_S(hipMemcpy(s->d_block_box, h.flat.block_box.data(), h.flat.block_box.size() * sizeof(float), hipMemcpyHostToDevice));
    // the leaves' slabs, filed under each leaf's first shape: made on the device from the records and shapes just uploaded
    // (RTGO_SLAB=0: none, an A/B switch).  Unfiled entries are never read; the kernel runs on the null stream, before any render.
    const char* slab_env = std::getenv("RTGO_SLAB");
    const bool slabs = RTGO_LEAF_SLAB && !(slab_env && std::strcmp(slab_env, "0") == 0) && !h.flat.records.empty();
    const size_t leaf_bound_bytes = slabs ? ((size_t)desc->n_shapes + 1) * kLeafBoundQuads * sizeof(float4) : 0;
    if (slabs) {
        HIP_TRY_S(hipMalloc(&s->d_leaf_bound, leaf_bound_bytes));
        const uint32_t n_recs = (uint32_t)h.flat.records.size();
        hipLaunchKernelGGL(k_leaf_bounds, dim3((n_recs + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, (const float4*)s->d_recs, (const float4*)s->d_prims, n_recs,
                           (float4*)s->d_leaf_bound);
        HIP_TRY_S(hipGetLastError());
        HIP_TRY_S(hipDeviceSynchronize());
    }
    HIP_TRY_S(hipMalloc(&s->d_mats, h.mats.size() * sizeof(DMaterial)));
    HIP_TRY_S(hipMemcpy(s->d_mats, h.mats.data(), h.mats.size() * sizeof(DMaterial), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_texs, texs.size() * sizeof(DTexture)));
    HIP_TRY_S(hipMemcpy(s->d_texs, texs.data(), texs.size() * sizeof(DTexture), hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_lights, h.lights.size() * sizeof(DLight)));
    HIP_TRY_S(hipMemcpy(s->d_lights, h.lights.data(), h.lights.size() * sizeof(DLight), hipMemcpyHostToDevice));

    lap("uploads");
    s->d.recs = (const float4*)s->d_recs; s->d.qblocks = (const uint4*)s->d_qblocks; s->d.prims = (const float4*)s->d_prims; s->d.tprims = (const float4*)s->d_tprims; s->d.prim_ids = (const int*)s->d_prim_ids; s->d.block_first = (const uint32_t*)s->d_block_first; s->d.block_box = (const float4*)s->d_block_box; s->d.leaf_bound = (const float4*)s->d_leaf_bound;
    s->d.materials = (const DMaterial*)s->d_mats; s->d.textures = (const DTexture*)s->d_texs;
    s->d.lights = (const DLight*)s->d_lights;
    s->d.pos_lights = s->d.lights + desc->n_lights + 1;
    s->d.n_lights = (int)desc->n_lights; s->d.n_pos_lights = h.n_pos; s->d.root_ref = h.flat.root_ref;
    s->d.beam_r = kBeamR;
    for (int l = 0; l < h.n_pos; ++l) {
        const DLight& pl = h.lights[(size_t)desc->n_lights + 1 + (size_t)l];
        const float sum = (std::fabs(pl.pos[0]) + std::fabs(pl.pos[1])) + std::fabs(pl.pos[2]);
        if (std::isfinite(sum)) s->d.beam_r = std::max(s->d.beam_r, kBeamR + 2.384185791015625e-07f * sum);
    }
    s->d.beam_n1 = s->d.beam_r * 1.00001f;
    s->d.beam_3n1 = 3.0f * s->d.beam_n1;
    // The voxel beams (device_math.hpp): a grid of at most 128 cubic cells a side over the box of the tree's top block,
    // one beam per positional light and cell, walked here once.  Only for scenes that get pair beams by default (256
    // shapes or more), with slabs, and few enough lights for 32 MiB of cells (RTGO_VOXEL=0: none).
    size_t vox_bytes = 0;
    {
        const char* vox_env = std::getenv("RTGO_VOXEL");
        const bool want = slabs && desc->n_shapes >= 256 && h.n_pos > 0 && !(vox_env && vox_env[0] == '0') && h.flat.block_box.size() >= 8;
        if (want) {
            const float* bb = h.flat.block_box.data();  // block 0: {min.xyz, -, max.xyz, -}
            float ext = 0.0f;
            for (int a = 0; a < 3; ++a) ext = std::max(ext, bb[4 + a] - bb[a]);
            int res = 128;
            if (const char* ev = std::getenv("RTGO_VOXEL_RES")) res = std::max(4, std::min(256, std::atoi(ev)));
            const float cs = ext / (float)res;
            unsigned long long n_cells = 1;
            if (cs > 0.0f && std::isfinite(cs)) {
                for (int a = 0; a < 3; ++a) {
                    s->d.vox_min[a] = bb[a];
                    s->d.vox_n[a] = std::max(1, std::min(res, (int)std::ceil((bb[4 + a] - bb[a]) / cs)));
                    n_cells *= (unsigned long long)s->d.vox_n[a];
                }
                s->d.vox_cs = cs; s->d.vox_inv_cs = 1.0f / cs;
                const unsigned long long total = n_cells * (unsigned long long)h.n_pos;
                if (total <= (32ull << 20)) {
                    vox_bytes = (size_t)total;
                    HIP_TRY_S(hipMalloc(&s->d_vox, vox_bytes));
                    const int grid = (int)std::min<unsigned long long>((total + kBlock - 1) / kBlock, 2048ull);
                    const int spill_v = h.flat.stack_bound + 1 > kLdsStackDepth ? h.flat.stack_bound + 1 - kLdsStackDepth : 0;
                    const size_t stride = (size_t)grid * kBlock;
                    void* tmp = nullptr;  // the walks' stack spill region, for this launch only
                    HIP_TRY_S(hipMalloc(&tmp, std::max<size_t>(1, (size_t)(spill_v + 1) * stride) * sizeof(uint32_t)));
                    hipLaunchKernelGGL(k_voxel_beams, dim3(grid), dim3(kBlock), 0, 0, s->d, (uint32_t*)tmp, (uint32_t)stride, (uint8_t*)s->d_vox, n_cells);
                    const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
                    (void)hipFree(tmp);
                    HIP_TRY_S(e1);
                    HIP_TRY_S(e2);
                    s->d.vox = (const uint8_t*)s->d_vox;
                }
            }
        }
    }
    lap("voxel beams");
    // traversal stack spill: worst case 7 pending siblings per level, beyond the LDS part
    const int spill = h.flat.stack_bound + 1 > kLdsStackDepth ? h.flat.stack_bound + 1 - kLdsStackDepth : 0;  // +1: the nearest child is pushed, then popped
    s->spill = spill;  // the region itself is sized by ensure_overflow() for the grids a call really launches

    {  // the lanes' events, streams and pinned counters: made by a thread of their own since the tree build began
        const std::pair<hipError_t, const char*> made = lane_setup->get();
        if (made.first != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string(made.second) + ": " + hipGetErrorString(made.first)));
    }

    s->info.n_shapes = desc->n_shapes; s->info.n_inner_nodes = h.flat.n_inner;
    s->info.n_child_records = (int64_t)h.flat.records.size();
    s->info.tree_depth = h.flat.depth; s->info.stack_bound = h.flat.stack_bound;
    s->info.device_bytes = (int64_t)(rec_bytes + h.flat.qblocks.size() * 4 + h.prim_words * 4 + h.prim_words * 4 + leaf_bound_bytes + vox_bytes + h.mats.size() * sizeof(DMaterial) + texs.size() * sizeof(DTexture) +
                                     h.lights.size() * sizeof(DLight)) + tex_bytes;  // + the spill region, see ensure_overflow()
    lap("events");
    return RT_OK;
#undef HIP_TRY_S
}

static int rt_scene_create_impl(const rt_scene_desc* desc, int device, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "desc/out is NULL");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc != RT_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    Laps lap;
    struct Guard {  // an exception (std::bad_alloc in a table, std::system_error from a thread) must not leak the device buffers
        rt_scene* s = nullptr;
        ~Guard() { if (s) rt_scene_destroy(s); }
    } guard;
    LaneSetup lane_setup;  // declared after the guard: its thread has ended before the guard releases the scene
    rc = begin_scene(device, &guard.s, &lane_setup);
    if (rc != RT_OK) return rc;
    lap("rt_init");
    HostScene host;
    rc = build_host_scene(desc, &host, lap);
    if (rc != RT_OK) return rc;
    rc = finish_scene(desc, host, guard.s, &lane_setup, lap);
    if (rc != RT_OK) return rc;
    guard.s->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = guard.s;
    guard.s = nullptr;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();  // a call without stats may have left its tail on a stream
    for (Lane& ln : s->lanes) {
        if (ln.arena.owned && ln.arena.base) (void)hipFree(ln.arena.base);
        ln.beam_state.release();
        if (ln.h_counters) (void)hipHostFree(ln.h_counters);
        if (ln.ev_ok) {
            for (auto& e : ln.ev_snap) if (e) (void)hipEventDestroy(e);
            for (auto& e : ln.ev_frame) if (e) (void)hipEventDestroy(e);
            if (ln.ev_join) (void)hipEventDestroy(ln.ev_join);
        }
        for (auto& e : ln.ev_pool) (void)hipEventDestroy(e);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    for (auto& e : s->ev_wall) if (e) (void)hipEventDestroy(e);
    void* ptrs[] = {s->d_recs, s->d_qblocks, s->d_prims, s->d_tprims, s->d_prim_ids, s->d_block_first, s->d_block_box, s->d_leaf_bound, s->d_vox, s->d_mats, s->d_texs, s->d_lights};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (void* p : s->d_tex_pixels) (void)hipFree(p);
    for (DevBuf* b : {&s->accum, &s->rgba, &s->probe_in, &s->probe_out, &s->probe_out2}) b->release();
    delete s;
}

int rt_scene_get_info(const rt_scene* s, rt_scene_info* out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "scene/out is NULL");
    *out = s->info;
    return RT_OK;
}

// What a render call needs besides the scene: its share of the frame, as batches of (pixel, sample) paths.
static int light_shift_of(int n_pos_lights) {  // bits for a positional light's number
    int b = 0;
    while ((1 << b) < n_pos_lights) ++b;
    return b;
}
struct RenderPlan {
    rt_partition part;
    int tile, tiles_x, sA, sB, ns_max, group, group_shift;
    long long n_local_pixels, pix_per_batch;
    int64_t batch_paths;
    bool by_tiles, empty;
};

// Which part of a frame a rank renders: tiles k with k % world == rank of the row-major tile grid, or the sample
// range [rank * spp / world, (rank + 1) * spp / world) of every pixel.
static void share_of(int W, int H, int spp, const rt_partition& part, int tile, rt_share* out) {
    const long long tiles_x = (W + tile - 1) / tile, tiles_y = (H + tile - 1) / tile;
    const long long n_tiles = tiles_x * tiles_y;
    out->n_tiles = part.mode == RT_PART_TILES ? (n_tiles - part.rank + part.world_size - 1) / part.world_size : n_tiles;
    out->sample_begin = 0; out->sample_end = spp;
    if (part.mode == RT_PART_SAMPLES) {
        out->sample_begin = (int)((long long)spp * part.rank / part.world_size);
        out->sample_end = (int)((long long)spp * (part.rank + 1) / part.world_size);
    }
    out->tile_size = tile;
}

static int plan_render(const rt_scene* s, const rt_image_spec* spec, const rt_partition* part_in, int64_t paths_cap, RenderPlan* P) {
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    rt_partition part{};
    part.seed = 0x5EED; part.mode = RT_PART_ALL; part.rank = 0; part.world_size = 1;
    if (part_in) part = *part_in;
    if (part.mode < RT_PART_ALL || part.mode > RT_PART_SAMPLES) return fail(RT_ERR_INVALID, "unknown partition mode");
    if (part.mode == RT_PART_ALL) { part.rank = 0; part.world_size = 1; }
    if (part.world_size < 1 || part.rank < 0 || part.rank >= part.world_size) return fail(RT_ERR_INVALID, "bad rank/world_size");
    const int tile = part.tile_size > 0 ? part.tile_size : 64;
    if (tile % 8 != 0 || tile > 1024) return fail(RT_ERR_INVALID, "tile_size must be a multiple of 8, at most 1024");
    const int W = spec->width, H = spec->height, D = spec->ray_tracing_max_depth;
    const int spp = spec->anti_aliasing_factor, reps = spec->soft_shadow_monte_carlo_repetitions;
    P->part = part; P->tile = tile;
    P->tiles_x = (W + tile - 1) / tile;
    P->by_tiles = part.mode == RT_PART_TILES;
    rt_share sh{};
    share_of(W, H, spp, part, tile, &sh);
    P->sA = sh.sample_begin; P->sB = sh.sample_end;
    P->n_local_pixels = sh.n_tiles * tile * tile;
    P->empty = !(P->sB > P->sA && D > 0 && P->n_local_pixels > 0);
    P->ns_max = 1; P->pix_per_batch = (long long)tile * tile; P->batch_paths = 0;
    // lanes per (hit, light) in the shadow kernel: smallest power of two >= min(reps, 64)
    P->group = 1; P->group_shift = 0;
    while (P->group < reps && P->group < 64) { P->group <<= 1; ++P->group_shift; }
    if (P->empty) return RT_OK;
    int64_t max_paths = part.max_paths_per_batch > 0 ? part.max_paths_per_batch : kDefaultMaxPaths;
    if (const char* ev = std::getenv("RTGO_MAX_PATHS")) {  // developer knob: log2 of the default batch size
        const int lg = std::atoi(ev);
        if (part.max_paths_per_batch <= 0 && lg >= 12 && lg <= 30) max_paths = (int64_t)1 << lg;
    }
    if (paths_cap > 0 && max_paths > paths_cap) max_paths = paths_cap;
    if (max_paths < (int64_t)tile * tile) {
        if (paths_cap > 0) return fail(RT_ERR_LIMIT, "tile_size^2 paths do not fit the workspace of this entry point; use a smaller tile_size");
        max_paths = (int64_t)tile * tile;
    }
    if (max_paths > ((int64_t)1 << 30)) max_paths = (int64_t)1 << 30;
    // a (hit, light) pair travels as one u32: light * hits + hit
    // (k_shadow indexes pairs with 32 bits and may look one grid-stride round past the end: 2^22 threads at most)
    const int64_t pair_cap = (((int64_t)1 << 32) - 1 - (int64_t)kMaxGridBlocks * kBlock) / std::max(1, s->d.n_pos_lights);
    if (pair_cap < (int64_t)tile * tile) return fail(RT_ERR_LIMIT, "too many positional lights for one tile of paths; use a smaller tile_size");
    if (max_paths > pair_cap) max_paths = pair_cap;
    if (max_paths > ((int64_t)1 << (32 - light_shift_of(s->d.n_pos_lights)))) max_paths = (int64_t)1 << (32 - light_shift_of(s->d.n_pos_lights));  // hit << light_shift | light
    P->ns_max = (int)std::min<long long>(P->sB - P->sA, std::min<long long>(64, max_paths / ((long long)tile * tile)));
    P->pix_per_batch = std::max<long long>(1, max_paths / P->ns_max / ((long long)tile * tile)) * tile * tile;
    // a share of fewer batches than lanes is cut into as many batches as there are lanes, as long as a batch keeps
    // 2^20 paths (which pixels share a batch changes no value: a pixel's samples are summed in index order)
    if (paths_cap <= 0 && part.max_paths_per_batch <= 0) {
        const long long tt = (long long)tile * tile;
        const long long n_tiles = P->n_local_pixels / tt;
        const long long n_batches = (P->n_local_pixels + P->pix_per_batch - 1) / P->pix_per_batch;
        int target = kLanes;  // batches wanted at least (RTGO_MIN_BATCHES: developer knob)
        if (const char* ev = std::getenv("RTGO_MIN_BATCHES")) target = std::max(1, std::atoi(ev));
        while (target > 1 && (n_tiles < target || ((n_tiles + target - 1) / target) * tt * P->ns_max < ((long long)1 << 20))) --target;
        if (n_batches < target) P->pix_per_batch = ((n_tiles + target - 1) / target) * tt;
    }
    P->batch_paths = (int64_t)std::min<long long>(P->pix_per_batch, P->n_local_pixels) * P->ns_max;
    return RT_OK;
}

// The wavefront loop over an arena that is already in place.  One host wait per bounce, on an event recorded right
// after k_shade: it tells the host how long the next queue is while the device works on this bounce's shadow rays,
// so the stream never drains inside a call.  Every kernel reads its queue length from device memory; the host's
// numbers only size the grids (upper bounds: a queue never outgrows the one it came from).
// Of the pixel batches (in loop order) this call takes those with index % n_lanes == lane_i.
static int run_render(rt_scene* s, Lane& ln, Arena& arena, const rt_image_spec* spec, const RenderPlan& P, float* d_accum, hipStream_t stream,
                      rt_stats* stats, bool may_create_events, int lane_i = 0, int n_lanes = 1,
                      hipEvent_t clock0 = nullptr, std::vector<std::pair<float, float>>* shadow_spans = nullptr) {
    const int W = spec->width, H = spec->height, D = spec->ray_tracing_max_depth, reps = spec->soft_shadow_monte_carlo_repetitions;
    const int L = s->d.n_pos_lights;
    Work& wk = arena.wk;
    rt_stats st{};
    st.ms_build = s->ms_build;
    const size_t n_ctr = (size_t)(3 * D + 8);
    if (n_ctr > ln.h_counter_words) return fail(RT_ERR_LIMIT, "RayTracingMaxDepth exceeds the counter snapshots of this scene");
    const bool timed = stats != nullptr;
    // pair beams: on for scenes of 256 shapes or more (the README scene's 11 shapes: 8.9 ms per frame without, 10.5 with;
    // a walk through two blocks is cheaper than proving it unnecessary) and 4 repetitions or more (a beam's walk costs
    // about a ray's: config 2 at 1080p 16 spp with 1 repetition 9.1 ms without, 10.1 with; with 16 repetitions 20.4 and
    // 15.7).  RTGO_BEAM=0 / 1 forces them off / on.
    // RTGO_SLAB=0 at render time: the shadow kernels of this call ignore the leaves' slabs (A/B switch, like RTGO_BEAM)
    DScene shadow_scene = s->d;
    {
        const char* slab_env = std::getenv("RTGO_SLAB");
        if (slab_env && slab_env[0] == '0') shadow_scene.leaf_bound = nullptr;
    }
    const char* beam_env = std::getenv("RTGO_BEAM");
    const bool beams = beam_env ? beam_env[0] != '0' : (s->info.n_shapes >= 256 && reps >= 4);
    // the voxel beams settle pairs where k_phong_pairs finds them: only with beams on and soft-shadow repetitions to save (RTGO_VOXEL=0: A/B switch)
    DScene pairs_scene = s->d;
    {
        const char* vox_env = std::getenv("RTGO_VOXEL");
        if (!beams || reps <= 0 || (vox_env && vox_env[0] == '0')) pairs_scene.vox = nullptr;
    }
    const char* resume_env = std::getenv("RTGO_RESUME");  // "0": rays start at the root, not where their pair's beam stopped (A/B switch)
    const bool resume_on = !(resume_env && resume_env[0] == '0');
    // what a pair's rays resume from: 48 bytes per (path, light) of a batch, in a buffer of the lane's own that only
    // calls which walk pair beams make (and only entry points that may allocate; without it the rays start at the root)
    wk.beam_state = nullptr;
    wk.light_shift = (uint32_t)light_shift_of(L);
    if (beams && resume_on && may_create_events && L > 0 && reps > 0) {
        const unsigned long long need = (unsigned long long)P.batch_paths * (unsigned long long)L * kResumeWords * sizeof(uint32_t);
        if (need <= (4ull << 30)) {  // (many lights: the rays then start at the root)
            HIP_TRY(ln.beam_state.ensure((size_t)need));
            wk.beam_state = (uint32_t*)ln.beam_state.p;
        }
    }
    const bool staged = timed && may_create_events;  // per-stage events (and their host-side list) only where the entry point may allocate
    size_t ev_used = 0;
    auto stamp = [&]() -> int {  // one timing event on the stream, from the pool (grown only where the entry point may allocate)
        if (ev_used == ln.ev_pool.size()) {
            if (!may_create_events) return -1;
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            ln.ev_pool.push_back(e);
        }
        if (hipEventRecord(ln.ev_pool[ev_used], stream) != hipSuccess) return -1;
        return (int)ev_used++;
    };
    struct Span { int a, b; int kind; };  // kind 0 = k_extend, 1 = k_shadow
    std::vector<Span> spans;
    if (timed) HIP_TRY(hipEventRecord(ln.ev_frame[0], stream));
    HIP_TRY(hipMemsetAsync(wk.frame_counters, 0, (kCounterSums + kCounterSums * kSubCounters) * sizeof(unsigned long long), stream));
    const DCamera cam = make_camera(s, W, H);
    for (int s0 = P.sA; s0 < P.sB; s0 += P.ns_max) {
        const int ns = std::min(P.ns_max, P.sB - s0);
        for (long long lp0 = 0; lp0 < P.n_local_pixels; lp0 += P.pix_per_batch) {
            if ((int)((lp0 / P.pix_per_batch) % n_lanes) != lane_i) continue;
            Batch B{};
            B.W = W; B.H = H; B.tile = P.tile; B.tiles_x = P.tiles_x;
            B.rank = P.part.rank; B.world = P.part.world_size; B.by_tiles = P.by_tiles ? 1 : 0;
            B.tj0 = (uint32_t)(lp0 / ((long long)P.tile * P.tile)); B.npix = (int)std::min<long long>(P.pix_per_batch, P.n_local_pixels - lp0);
            B.ns = ns; B.s0 = s0; B.max_depth = D; B.reps = reps;
            B.k0 = (uint32_t)P.part.seed; B.k1 = (uint32_t)(P.part.seed >> 32);
            const long long slots = (long long)B.npix * ns;
            HIP_TRY(hipMemsetAsync(wk.counters, 0, n_ctr * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_raygen, dim3((unsigned)((slots + kBlock * kPerThread - 1) / (kBlock * kPerThread))), dim3(kBlock), 0, stream, B, cam, wk);
            uint32_t bound = (uint32_t)slots;  // >= the length of the queue of bounce b
            for (int b = 0; b < D && bound > 0; ++b) {
                uint32_t* q = wk.queue[b & 1];
                uint32_t* qn = wk.queue[(b + 1) & 1];
                const uint32_t* n_rays = wk.counters + b;
                const uint32_t* n_phong = wk.counters + D + 1 + b;
                Span se{-1, -1, 0}, ss{-1, -1, 1};
                if (staged) se.a = stamp();
                hipLaunchKernelGGL(k_extend, dim3(grid_for(bound, arena.grid_cap)), dim3(kBlock), 0, stream, s->d, wk, q, n_rays);
                if (staged) se.b = stamp();
                hipLaunchKernelGGL(k_shade, dim3((bound + kBlock * kPerThread - 1) / (kBlock * kPerThread)), dim3(kBlock), 0, stream, s->d, B, wk, q, n_rays, qn, b);
                uint32_t* snap = ln.h_counters + (size_t)(b & 1) * ln.h_counter_words;
                HIP_TRY(hipMemcpyAsync(snap, wk.counters, n_ctr * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipEventRecord(ln.ev_snap[b & 1], stream));
                // the PhongBlinn hits of this bounce (at most `bound`): pairs, shadow rays, shading
                if (L > 0) {
                    const unsigned long long n_pairs = (unsigned long long)bound * L;
                    uint32_t* pair_counter = wk.counters + 2 * D + 4 + b;
                    HIP_TRY(hipMemsetAsync(wk.light_count, 0, (size_t)L * sizeof(uint32_t), stream));
                    hipLaunchKernelGGL(k_phong_pairs, dim3((bound + kBlock * kPerThread - 1) / (kBlock * kPerThread)), dim3(kBlock), 0, stream, pairs_scene, wk, n_phong, (uint32_t)reps);
                    hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(64), 0, stream, wk, L, pair_counter);
                    hipLaunchKernelGGL(k_pair_compact, dim3((bound + kBlock - 1) / kBlock, (unsigned)std::min(L, 65535)), dim3(kBlock), 0, stream, wk, n_phong, L);
                    if (reps > 0) {
                        const uint32_t* queue = wk.pairs;
                        const uint32_t* queue_len = pair_counter;
                        const uint32_t* resume = nullptr;
                        if (beams) {  // pairs whose rays one conservative walk proves unoccluded never reach k_shadow
                            hipLaunchKernelGGL(k_pair_beam, dim3(grid_for(n_pairs, arena.grid_cap)), dim3(kBlock), 0, stream, shadow_scene, B, wk, n_phong, pair_counter);
                            queue = wk.pair_seg;
                            queue_len = wk.light_count + 2 * L;
                            resume = resume_on ? wk.beam_state : nullptr;
                        }
                        if (staged) ss.a = stamp();  // ms_shadow brackets k_shadow alone
                        if (reps > 64)
                            hipLaunchKernelGGL(k_shadow<true>, dim3(grid_for(n_pairs * P.group, arena.grid_cap)), dim3(kBlock), 0, stream, shadow_scene, B, wk, n_phong, b,
                                               P.group, P.group_shift, queue, queue_len, resume);
                        else
                            hipLaunchKernelGGL(k_shadow<false>, dim3(grid_for(n_pairs * P.group, arena.grid_cap)), dim3(kBlock), 0, stream, shadow_scene, B, wk, n_phong, b,
                                               P.group, P.group_shift, queue, queue_len, resume);
                        if (staged) ss.b = stamp();
                    }
                }
                hipLaunchKernelGGL(k_phong, dim3((bound + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, s->d, B, wk, n_phong);
                if (staged) { spans.push_back(se); spans.push_back(ss); }
                // how long are this bounce's queues?  (the device is busy with the kernels above meanwhile)
                HIP_TRY(hipEventSynchronize(ln.ev_snap[b & 1]));
                const uint32_t nq = snap[b], np = snap[D + 1 + b];
                if (b == 0) { st.paths += nq; st.rays_primary += nq; } else st.rays_continuation += nq;
                st.phong_hits += np;
                st.rays_shadow_reference_equivalent += (uint64_t)np * L * reps;
                bound = snap[b + 1];
            }
            hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((B.npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, B, wk, d_accum);
        }
    }
    HIP_TRY(hipGetLastError());
    if (stats) {
        unsigned long long* fc = reinterpret_cast<unsigned long long*>(ln.h_counters + 2 * ln.h_counter_words);
        hipLaunchKernelGGL(k_sum_counters, dim3(1), dim3(kSubCounters), 0, stream, wk.frame_counters);
        HIP_TRY(hipMemcpyAsync(fc, wk.frame_counters, kCounterSums * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(ln.ev_frame[1], stream));
        HIP_TRY(hipEventSynchronize(ln.ev_frame[1]));
        st.rays_shadow = fc[0] + fc[1];
        st.rays_shadow_beam = fc[1];
        st.beam_walks = fc[2];
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ln.ev_frame[0], ln.ev_frame[1]));
        st.ms_render = ms;
        for (const Span& sp : spans) {
            if (sp.a < 0 || sp.b < 0) continue;
            float t = 0;
            if (hipEventElapsedTime(&t, ln.ev_pool[(size_t)sp.a], ln.ev_pool[(size_t)sp.b]) != hipSuccess) continue;
            (sp.kind == 0 ? st.ms_extend : st.ms_shadow) += t;
            if (sp.kind == 1 && clock0 && shadow_spans) {  // when this k_shadow launch ran, on the call's clock
                float t0 = 0;
                if (hipEventElapsedTime(&t0, clock0, ln.ev_pool[(size_t)sp.a]) == hipSuccess) shadow_spans->push_back({t0, t0 + t});
            }
        }
        st.ms_shadow_busy = st.ms_shadow;  // one stream: no two launches overlap
        st.ms_shade = st.ms_render - st.ms_extend - st.ms_shadow;  // everything else between the first and the last launch
        *stats = st;
    }
    return RT_OK;
}

static int rt_render_accum_device_impl(rt_scene* s, const rt_image_spec* spec, const rt_partition* part_in,
                           float* d_accum, void* hip_stream, rt_stats* stats) {
    if (!s || !d_accum) return fail(RT_ERR_INVALID, "scene/accum is NULL");
    RenderPlan P{};
    int rc = plan_render(s, spec, part_in, 0, &P);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(s->device));
    if (P.empty) {
        if (stats) { *stats = rt_stats{}; stats->ms_build = s->ms_build; }
        return RT_OK;
    }
    // lanes: RTGO_LANES (1 to 4, default 4), at most one per pixel batch of the share
    const char* lanes_env = std::getenv("RTGO_LANES");
    int n_lanes = lanes_env ? std::atoi(lanes_env) : kLanes;
    n_lanes = n_lanes < 1 ? 1 : (n_lanes > kLanes ? kLanes : n_lanes);
    const long long n_batches = (P.n_local_pixels + P.pix_per_batch - 1) / P.pix_per_batch;
    if (n_batches < n_lanes) n_lanes = (int)n_batches;
    // a lane's workspace (made on first use, by the lane's own thread: the allocations of a first frame run side by
    // side); the widest launch is k_shadow's: one lane per (pair, repetition lane), at most 64 lanes per pair
    auto workspace = [&](int l) {
        return ensure_work(s, s->lanes[l], P.batch_paths, spec->ray_tracing_max_depth, s->d.n_pos_lights,
                           (unsigned long long)P.batch_paths * (unsigned long long)std::max(1, s->d.n_pos_lights) * 64ull);
    };
    hipStream_t stream = (hipStream_t)hip_stream;
    if (n_lanes == 1) {
        rc = workspace(0);
        if (rc != RT_OK) return rc;
        return run_render(s, s->lanes[0], s->lanes[0].arena, spec, P, d_accum, stream, stats, true);
    }
    // fork: the other lanes start behind what the caller has queued on its stream (the zeroing of d_accum, say)
    if (stats) HIP_TRY(hipEventRecord(s->ev_wall[0], stream));
    HIP_TRY(hipEventRecord(s->ev_fork, stream));
    std::vector<rt_stats> part_stats((size_t)n_lanes);
    std::vector<std::vector<std::pair<float, float>>> spans((size_t)n_lanes);
    const int device = s->device;
    rc = on_every_rank(n_lanes, [&](int l) -> int {
        Lane& ln = s->lanes[l];
        HIP_TRY(hipSetDevice(device));
        const int wrc = workspace(l);
        if (wrc != RT_OK) return wrc;
        hipStream_t st = l == 0 ? stream : ln.stream;
        if (l > 0) HIP_TRY(hipStreamWaitEvent(st, s->ev_fork, 0));
        const int r = run_render(s, ln, ln.arena, spec, P, d_accum, st, stats ? &part_stats[(size_t)l] : nullptr, true, l, n_lanes, s->ev_wall[0], &spans[(size_t)l]);
        if (l > 0 && r == RT_OK) HIP_TRY(hipEventRecord(ln.ev_join, st));
        return r;
    });
    if (rc != RT_OK) {
        (void)hipDeviceSynchronize();  // nothing of a failed call is left running on the lanes' streams
        return rc;
    }
    // join: what the caller queues next on its stream comes after every lane's batches
    for (int l = 1; l < n_lanes; ++l) HIP_TRY(hipStreamWaitEvent(stream, s->lanes[l].ev_join, 0));
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev_wall[1], stream));
        HIP_TRY(hipEventSynchronize(s->ev_wall[1]));
        rt_stats t = part_stats[0];
        for (int l = 1; l < n_lanes; ++l) {
            const rt_stats& x = part_stats[(size_t)l];
            t.paths += x.paths; t.rays_primary += x.rays_primary; t.rays_continuation += x.rays_continuation;
            t.rays_shadow += x.rays_shadow; t.rays_shadow_beam += x.rays_shadow_beam; t.beam_walks += x.beam_walks; t.rays_shadow_reference_equivalent += x.rays_shadow_reference_equivalent;
            t.phong_hits += x.phong_hits;
            t.ms_extend += x.ms_extend; t.ms_shadow += x.ms_shadow; t.ms_shade += x.ms_shade;  // stream time, summed over the lanes
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_wall[0], s->ev_wall[1]));
        t.ms_render = ms;
        // the time during which at least one lane's k_shadow was running: the union of the launches' intervals
        std::vector<std::pair<float, float>> all;
        for (auto& v : spans) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        double busy = 0, end = -1;
        for (const auto& iv : all) {
            if (iv.second <= end) continue;
            busy += iv.second - std::max<double>(iv.first, end);
            end = iv.second;
        }
        t.ms_shadow_busy = busy;
        *stats = t;
    }
    return RT_OK;
}

// rt_render_accum_device for callers that cannot allocate: the batch workspace is this GPU's copy of `scratch`.
static int rt_render_accum_device_static_impl(rt_scene* s, const rt_image_spec* spec, const rt_partition* part_in,
                                              float* d_accum, void* hip_stream, rt_stats* stats) {
    if (!s || !d_accum) return fail(RT_ERR_INVALID, "scene/accum is NULL");
    RenderPlan P{};
    int rc = plan_render(s, spec, part_in, kStaticPaths, &P);
    if (rc != RT_OK) return rc;
    const int D = spec->ray_tracing_max_depth, L = s->d.n_pos_lights;
    if (D > kStaticDepth) return fail(RT_ERR_LIMIT, "rt_render_accum_device_static: RayTracingMaxDepth > 64");
    if (L > kStaticLights) return fail(RT_ERR_LIMIT, "rt_render_accum_device_static: more than 4 positional lights");
    if (s->spill + 1 > kStaticSpill)
        return fail(RT_ERR_LIMIT, "rt_render_accum_device_static: the scene's tree can need " + std::to_string(s->info.stack_bound + 1) +
                                      " traversal-stack entries per ray; the static workspace holds " + std::to_string(kLdsStackDepth + kStaticSpill - 1));
    if (s->device >= kMaxDevices || !g_scratch_ptr[s->device]) return fail(RT_ERR_DEVICE, "no static workspace on this device");
    HIP_TRY(hipSetDevice(s->device));
    if (P.empty) {
        if (stats) { *stats = rt_stats{}; stats->ms_build = s->ms_build; }
        return RT_OK;
    }
    Arena st;
    st.owned = false;
    st.base = g_scratch_ptr[s->device];  // the device address, not the symbol's name
    st.bytes = kScratchBytes;
    st.grid_cap = kStaticGrid;
    const unsigned long long entries = s->spill > 0 ? (unsigned long long)(s->spill + 1) * kStaticGrid * kBlock : 0ull;
    const size_t need = carve_work(&st.wk, st.base, P.batch_paths, D, L, entries);
    if (need > kScratchBytes) return fail(RT_ERR_LIMIT, "rt_render_accum_device_static: the batch does not fit the 72 MiB static workspace");
    st.wk.overflow_stride = (uint32_t)kStaticGrid * kBlock;
    st.paths = P.batch_paths; st.depth = D; st.lpos = L; st.overflow_entries = entries;
    // nothing in the workspace has to survive from an earlier call: run_render zeroes the counters it reads, and
    // every other array is written before it is read
    return run_render(s, s->lanes[0], st, spec, P, d_accum, (hipStream_t)hip_stream, stats, false);
}

static int rt_resolve_device_impl(const float* d_accum, int32_t width, int32_t height, int32_t spp, uint8_t* d_rgba, void* hip_stream) {
    if (!d_accum || !d_rgba || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "bad resolve arguments");
    const int n = width * height;
    // AntiAliasingFactor 0: the reference scales by 1/0 = +Inf and 0 * Inf = NaN -> uint8(NaN); we emit 0
    const float inv = spp > 0 ? 1.0f / (float)spp : 0.0f;
    hipLaunchKernelGGL(k_resolve, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)hip_stream, d_accum, n, inv, (uchar4*)d_rgba);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

static int rt_render_accum_impl(rt_scene* s, const rt_image_spec* spec, const rt_partition* part, float* accum_out, rt_stats* stats) {
    if (!s || !accum_out) return fail(RT_ERR_INVALID, "scene/accum is NULL");
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(s->device));
    const size_t bytes = (size_t)spec->width * spec->height * 3 * sizeof(float);
    HIP_TRY(s->accum.ensure(bytes));  // kept by the scene: no allocation per frame
    HIP_TRY(hipMemsetAsync(s->accum.p, 0, bytes, nullptr));
    rc = rt_render_accum_device(s, spec, part, (float*)s->accum.p, nullptr, stats);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpy(accum_out, s->accum.p, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

static int rt_render_impl(rt_scene* s, const rt_image_spec* spec, const rt_partition* part, uint8_t* rgba_out, rt_stats* stats) {
    if (!s || !rgba_out) return fail(RT_ERR_INVALID, "scene/rgba_out is NULL");
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)spec->width * spec->height;
    HIP_TRY(s->accum.ensure(npx * 3 * sizeof(float)));
    HIP_TRY(s->rgba.ensure(npx * 4));
    HIP_TRY(hipMemsetAsync(s->accum.p, 0, npx * 3 * sizeof(float), nullptr));
    rc = rt_render_accum_device(s, spec, part, (float*)s->accum.p, nullptr, stats);
    if (rc == RT_OK) rc = rt_resolve_device((const float*)s->accum.p, spec->width, spec->height, spec->anti_aliasing_factor, (uint8_t*)s->rgba.p, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpy(rgba_out, s->rgba.p, npx * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

static int rt_generate_image_impl(const rt_scene_desc* desc, const rt_image_spec* spec, const rt_partition* part, uint8_t* rgba_out, rt_stats* stats) {
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    rt_scene* s = nullptr;
    rc = rt_scene_create(desc, 0, &s);
    if (rc != RT_OK) return rc;
    rc = rt_render(s, spec, part, rgba_out, stats);
    rt_scene_destroy(s);
    return rc;
}

static int rt_trace_rays_impl(rt_scene* s, const float* rays, int64_t n, int32_t any_hit, int32_t* prim_out, float* t_out) {
    if (!s || !rays || !prim_out || !t_out || n < 0) return fail(RT_ERR_INVALID, "bad trace arguments");
    if (n == 0) return RT_OK;
    if (n > ((int64_t)1 << 30)) return fail(RT_ERR_LIMIT, "more than 2^30 rays in one call");
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_overflow(s, (unsigned long long)n);
    if (rc != RT_OK) return rc;
    HIP_TRY(s->probe_in.ensure((size_t)n * 8 * sizeof(float)));
    HIP_TRY(s->probe_out.ensure((size_t)n * sizeof(int)));
    HIP_TRY(s->probe_out2.ensure((size_t)n * sizeof(float)));
    HIP_TRY(hipMemcpy(s->probe_in.p, rays, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice));
    DScene probe_scene = s->d;
    {
        const char* slab_env = std::getenv("RTGO_SLAB");  // "0": the any-hit walks of this call ignore the leaves' slabs
        if (slab_env && slab_env[0] == '0') probe_scene.leaf_bound = nullptr;
    }
    hipLaunchKernelGGL(k_trace_probe, dim3(grid_for((unsigned long long)n, s->lanes[0].arena.grid_cap)), dim3(kBlock), 0, nullptr, probe_scene, s->lanes[0].arena.wk,
                       (const float*)s->probe_in.p, (uint32_t)n, any_hit, (int*)s->probe_out.p, (float*)s->probe_out2.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(prim_out, s->probe_out.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t_out, s->probe_out2.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

static int rt_camera_rays_impl(rt_scene* s, const rt_image_spec* spec, uint64_t seed, const int32_t* pixels, int64_t n, float* rays_out) {
    if (!s || !pixels || !rays_out || n < 0) return fail(RT_ERR_INVALID, "bad camera arguments");
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (n > ((int64_t)1 << 30)) return fail(RT_ERR_LIMIT, "more than 2^30 rays in one call");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(s->probe_in.ensure((size_t)n * 3 * sizeof(int)));
    HIP_TRY(s->probe_out.ensure((size_t)n * 6 * sizeof(float)));
    HIP_TRY(hipMemcpy(s->probe_in.p, pixels, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice));
    const DCamera cam = make_camera(s, spec->width, spec->height);
    hipLaunchKernelGGL(k_camera_probe, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, cam, spec->width, spec->height,
                       (uint32_t)seed, (uint32_t)(seed >> 32), (const int*)s->probe_in.p, (uint32_t)n, (float*)s->probe_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rays_out, s->probe_out.p, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

static int rt_scatter_probe_impl(rt_scene* s, const rt_image_spec* spec, uint64_t seed, const float* rays, const uint32_t* keys, int64_t n, float* out) {
    if (!s || !rays || !keys || !out || n < 0) return fail(RT_ERR_INVALID, "bad scatter-probe arguments");
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (n > ((int64_t)1 << 24)) return fail(RT_ERR_LIMIT, "more than 2^24 probes in one call");
    HIP_TRY(hipSetDevice(s->device));
    rc = ensure_overflow(s, (unsigned long long)n);
    if (rc != RT_OK) return rc;
    const size_t ray_bytes = (size_t)n * 6 * sizeof(float), key_bytes = (size_t)n * 3 * sizeof(uint32_t);
    HIP_TRY(s->probe_in.ensure(align_up(ray_bytes) + key_bytes));
    HIP_TRY(s->probe_out.ensure((size_t)n * 20 * sizeof(float)));
    unsigned char* in = (unsigned char*)s->probe_in.p;
    HIP_TRY(hipMemcpy(in, rays, ray_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(in + align_up(ray_bytes), keys, key_bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scatter_probe, dim3(grid_for((unsigned long long)n, s->lanes[0].arena.grid_cap)), dim3(kBlock), 0, nullptr, s->d, s->lanes[0].arena.wk,
                       (const float*)in, (const uint32_t*)(in + align_up(ray_bytes)), (uint32_t)n, (uint32_t)seed, (uint32_t)(seed >> 32),
                       spec->soft_shadow_monte_carlo_repetitions, (float*)s->probe_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, s->probe_out.p, (size_t)n * 20 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---- one frame on several GPUs behind one call (what GenerateImage's goroutine fan-out becomes, raytracer.go:73-76) --
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static int load_rccl(Rccl* r) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r->handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r->handle) break;
    }
    if (!r->handle) return fail(RT_ERR_DEVICE, std::string("librccl is not loadable: ") + dlerror());
    r->CommInitAll = (decltype(r->CommInitAll))dlsym(r->handle, "ncclCommInitAll");
    r->CommDestroy = (decltype(r->CommDestroy))dlsym(r->handle, "ncclCommDestroy");
    r->Reduce = (decltype(r->Reduce))dlsym(r->handle, "ncclReduce");
    r->GroupStart = (decltype(r->GroupStart))dlsym(r->handle, "ncclGroupStart");
    r->GroupEnd = (decltype(r->GroupEnd))dlsym(r->handle, "ncclGroupEnd");
    r->GetErrorString = (decltype(r->GetErrorString))dlsym(r->handle, "ncclGetErrorString");
    if (!r->CommInitAll || !r->CommDestroy || !r->Reduce || !r->GroupStart || !r->GroupEnd || !r->GetErrorString)
        return fail(RT_ERR_DEVICE, "librccl lacks ncclCommInitAll / ncclReduce / ncclGroupStart / ncclGroupEnd");
    return RT_OK;
}

struct rt_multi {
    std::vector<int> devices;
    std::vector<rt_scene*> scenes;      // one replica of the scene per share
    std::vector<hipStream_t> streams;   // one stream per share, on its device
    std::vector<ncclComm_t> comms;      // when the devices are distinct and there are at least two
    Rccl rccl;
    bool distinct = true;
};

static void rt_multi_destroy_impl(rt_multi* m) {
    if (!m) return;
    for (size_t r = 0; r < m->comms.size(); ++r)
        if (m->comms[r]) { (void)hipSetDevice(m->devices[r]); (void)m->rccl.CommDestroy(m->comms[r]); }
    for (size_t r = 0; r < m->streams.size(); ++r)
        if (m->streams[r]) { (void)hipSetDevice(m->devices[r]); (void)hipStreamDestroy(m->streams[r]); }
    for (rt_scene* s : m->scenes) rt_scene_destroy(s);
    if (m->rccl.handle) (void)dlclose(m->rccl.handle);
    delete m;
}

static int rt_multi_create_impl(const rt_scene_desc* desc, const int32_t* devices, int32_t n, rt_multi** out) {
    if (!desc || !devices || !out) return fail(RT_ERR_INVALID, "desc/devices/out is NULL");
    *out = nullptr;
    if (n < 1 || n > 64) return fail(RT_ERR_INVALID, "1 to 64 devices");
    rt_multi* m = new rt_multi();
    struct Guard { rt_multi* m; ~Guard() { if (m) rt_multi_destroy_impl(m); } } guard{m};
    m->devices.assign(devices, devices + n);
    m->scenes.assign((size_t)n, nullptr);
    m->streams.assign((size_t)n, nullptr);
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (devices[a] == devices[b]) m->distinct = false;
    // The replicas: the tree and the tables are built ONCE on the host (while every device's runtime warm-up and lane
    // set-up run on threads of their own) and then uploaded to all devices at the same time.
    int rc = validate_desc(desc);
    if (rc != RT_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    Laps lap;
    std::vector<LaneSetup> lane_setups((size_t)n);  // declared after the guard: their threads end before it destroys the scenes
    for (int r = 0; r < n; ++r) {
        rc = begin_scene(m->devices[(size_t)r], &m->scenes[(size_t)r], &lane_setups[(size_t)r]);
        if (rc != RT_OK) return rc;
    }
    lap("rt_init");
    HostScene host;
    rc = build_host_scene(desc, &host, lap);
    if (rc != RT_OK) return rc;
    rc = on_every_rank(n, [&](int r) {
        Laps mine;
        mine.on = lap.on && r == 0;
        int c = finish_scene(desc, host, m->scenes[(size_t)r], &lane_setups[(size_t)r], mine);
        if (c != RT_OK) return c;
        m->scenes[(size_t)r]->ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        HIP_TRY(hipStreamCreateWithFlags(&m->streams[(size_t)r], hipStreamNonBlocking));
        return RT_OK;
    });
    if (rc != RT_OK) return rc;
    if (n > 1 && m->distinct) {
        rc = load_rccl(&m->rccl);
        if (rc != RT_OK) return rc;
        m->comms.assign((size_t)n, nullptr);
        const ncclResult_t e = m->rccl.CommInitAll(m->comms.data(), n, m->devices.data());
        if (e != ncclSuccess) return fail(RT_ERR_DEVICE, std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(e));
    }
    guard.m = nullptr;
    *out = m;
    return RT_OK;
}

static int rt_multi_render_impl(rt_multi* m, const rt_image_spec* spec, int32_t mode, uint64_t seed, uint8_t* rgba_out, rt_stats* stats) {
    if (!m || !rgba_out) return fail(RT_ERR_INVALID, "multi/rgba_out is NULL");
    int rc = validate_spec(spec);
    if (rc != RT_OK) return rc;
    const int n = (int)m->scenes.size();
    if (mode != RT_PART_TILES && mode != RT_PART_SAMPLES) return fail(RT_ERR_INVALID, "mode must be RT_PART_TILES or RT_PART_SAMPLES");
    const size_t count = (size_t)spec->width * spec->height * 3;
    std::vector<rt_stats> st((size_t)n);
    // RTGO_FAIL_SHARE=r: share r fails before it renders (test hook for the error path below)
    const char* fail_env = std::getenv("RTGO_FAIL_SHARE");
    const int fail_share = fail_env ? std::atoi(fail_env) : -1;
    // Phase 1: every share renders into its own device's accumulation buffer.  on_every_rank() joins all the
    // threads, so every share has either finished its frame (a call with stats blocks until then) or failed
    // before anything below touches a collective.
    rc = on_every_rank(n, [&](int r) {
        rt_scene* s = m->scenes[(size_t)r];
        hipStream_t stream = m->streams[(size_t)r];
        HIP_TRY(hipSetDevice(s->device));
        if (r == fail_share) return fail(RT_ERR_DEVICE, "injected failure (RTGO_FAIL_SHARE)");
        HIP_TRY(s->accum.ensure(count * sizeof(float)));
        HIP_TRY(hipMemsetAsync(s->accum.p, 0, count * sizeof(float), stream));
        rt_partition part{};
        part.seed = seed; part.mode = mode; part.rank = r; part.world_size = n;
        return rt_render_accum_device(s, spec, &part, (float*)s->accum.p, stream, &st[(size_t)r]);
    });
    if (rc != RT_OK) {
        // A share failed: no rank enters the reduce (a rank alone in a collective would wait for ever), and
        // nothing of the frame is left running on any device when the error comes back.
        const std::string msg = g_err;
        for (int r = 0; r < n; ++r)
            if (hipSetDevice(m->scenes[(size_t)r]->device) == hipSuccess) (void)hipDeviceSynchronize();
        return fail(rc, msg);
    }
    // Phase 2, only when every share succeeded: RCCL over xGMI, the shares summed onto the first device, in place
    // there.  One host thread enqueues all ranks' ncclReduce inside one group (the single-process form of the
    // collective: nothing is launched until ncclGroupEnd has every rank's call, so a rank cannot be left alone in it).
    if (!m->comms.empty()) {
        ncclResult_t e = m->rccl.GroupStart();
        for (int r = 0; r < n && e == ncclSuccess; ++r) {
            rt_scene* s = m->scenes[(size_t)r];
            if (hipSetDevice(s->device) != hipSuccess) { e = ncclUnhandledCudaError; break; }  // the rank's stream is this device's
            e = m->rccl.Reduce(s->accum.p, s->accum.p, count, ncclFloat, ncclSum, 0, m->comms[(size_t)r], m->streams[(size_t)r]);
        }
        const ncclResult_t e2 = m->rccl.GroupEnd();
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) return fail(RT_ERR_DEVICE, std::string("ncclReduce: ") + m->rccl.GetErrorString(e));
        for (int r = 0; r < n; ++r) {
            HIP_TRY(hipSetDevice(m->scenes[(size_t)r]->device));
            HIP_TRY(hipStreamSynchronize(m->streams[(size_t)r]));
        }
    }
    rt_scene* root = m->scenes[0];
    HIP_TRY(hipSetDevice(root->device));
    if (m->comms.empty())  // shares that live on the root's own device (a device listed more than once): summed there
        for (int r = 1; r < n; ++r) {
            if (m->scenes[(size_t)r]->device != root->device) return fail(RT_ERR_DEVICE, "shares on several devices need RCCL");
            hipLaunchKernelGGL(k_add, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, m->streams[0], (float*)root->accum.p,
                               (const float*)m->scenes[(size_t)r]->accum.p, count);
        }
    HIP_TRY(root->rgba.ensure(count / 3 * 4));
    rc = rt_resolve_device((const float*)root->accum.p, spec->width, spec->height, spec->anti_aliasing_factor, (uint8_t*)root->rgba.p, m->streams[0]);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgba_out, root->rgba.p, count / 3 * 4, hipMemcpyDeviceToHost, m->streams[0]));
    HIP_TRY(hipStreamSynchronize(m->streams[0]));
    if (stats) {
        rt_stats t{};
        for (const rt_stats& x : st) {
            t.rays_primary += x.rays_primary; t.rays_continuation += x.rays_continuation; t.rays_shadow += x.rays_shadow; t.rays_shadow_beam += x.rays_shadow_beam; t.beam_walks += x.beam_walks;
            t.rays_shadow_reference_equivalent += x.rays_shadow_reference_equivalent; t.phong_hits += x.phong_hits; t.paths += x.paths;
            t.ms_build = std::max(t.ms_build, x.ms_build); t.ms_render = std::max(t.ms_render, x.ms_render); t.ms_shadow_busy += x.ms_shadow_busy;
            t.ms_extend = std::max(t.ms_extend, x.ms_extend); t.ms_shadow = std::max(t.ms_shadow, x.ms_shadow); t.ms_shade = std::max(t.ms_shade, x.ms_shade);
        }
        *stats = t;
    }
    return RT_OK;
}

int rt_multi_get_info(const rt_multi* m, rt_multi_info* out) {
    if (!m || !out) return fail(RT_ERR_INVALID, "multi/out is NULL");
    rt_multi_info i{};
    i.n_shares = (int32_t)m->scenes.size();
    for (size_t a = 0; a < m->devices.size(); ++a) {
        bool seen = false;
        for (size_t b = 0; b < a; ++b) seen = seen || m->devices[b] == m->devices[a];
        if (!seen) ++i.n_distinct_devices;
    }
    for (ncclComm_t c : m->comms) if (c) ++i.rccl_ranks;
    *out = i;
    return RT_OK;
}

// What this library was built from: a hash of its sources (csrc/Makefile) and the compile-time knobs of its kernels.
// bench.py compares it with the build that profiles/*/bounds.json was counted on.
#ifndef RTGO_SRC_HASH
#define RTGO_SRC_HASH "unknown"
#endif
#define RTGO_STR2(x) #x
#define RTGO_STR(x) RTGO_STR2(x)
const char* rt_build_info(void) {
    return "src=" RTGO_SRC_HASH " shadow_waves=" RTGO_STR(RTGO_SHADOW_WAVES) " beam_waves=" RTGO_STR(RTGO_BEAM_WAVES)
           " extend_waves=" RTGO_STR(RTGO_EXTEND_WAVES) " lds_stack=" RTGO_STR(RTGO_LDS_STACK) " pend_cap=" RTGO_STR(RTGO_PEND_CAP)
           " flush_lanes=" RTGO_STR(RTGO_FLUSH_LANES) " steal=" RTGO_STR(RTGO_STEAL) " visit_batch=" RTGO_STR(RTGO_VISIT_BATCH) " max_grid=" RTGO_STR(RTGO_MAX_GRID)
           " leaf_slab=" RTGO_STR(RTGO_LEAF_SLAB) " beam_cap=" RTGO_STR(RTGO_BEAM_CAP);
}

#ifdef RTGO_PROF
int rt_debug_prof(unsigned long long* out8, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(rtgo::g_prof), 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(out8 + 8, HIP_SYMBOL(rtgo::g_prof2), 4 * sizeof(unsigned long long)));  // out8 holds 12 + 129 + 64
    HIP_TRY(hipMemcpyFromSymbol(out8 + 12, HIP_SYMBOL(rtgo::g_prof3), (65 + 64) * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(out8 + 12 + 129, HIP_SYMBOL(rtgo::g_prof4), 64 * sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyFromSymbol(out8 + 12 + 129 + 64, HIP_SYMBOL(rtgo::g_prof5), 4 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long z[8] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rtgo::g_prof), z, sizeof(z)));
    }
    return RT_OK;
}
#endif

// The public entry points (guarded(), above: no C++ exception crosses the C boundary).
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) {
    return guarded([&] { return rt_scene_create_impl(desc, device, out); });
}
int rt_render_accum_device(rt_scene* s, const rt_image_spec* spec, const rt_partition* part_in, float* d_accum, void* hip_stream, rt_stats* stats) {
    return guarded([&] { return rt_render_accum_device_impl(s, spec, part_in, d_accum, hip_stream, stats); });
}
int rt_render_accum_device_static(rt_scene* s, const rt_image_spec* spec, const rt_partition* part_in, float* d_accum, void* hip_stream, rt_stats* stats) {
    return guarded([&] { return rt_render_accum_device_static_impl(s, spec, part_in, d_accum, hip_stream, stats); });
}
int rt_resolve_device(const float* d_accum, int32_t width, int32_t height, int32_t spp, uint8_t* d_rgba, void* hip_stream) {
    return guarded([&] { return rt_resolve_device_impl(d_accum, width, height, spp, d_rgba, hip_stream); });
}
int rt_render_accum(rt_scene* s, const rt_image_spec* spec, const rt_partition* part, float* accum_out, rt_stats* stats) {
    return guarded([&] { return rt_render_accum_impl(s, spec, part, accum_out, stats); });
}
int rt_render(rt_scene* s, const rt_image_spec* spec, const rt_partition* part, uint8_t* rgba_out, rt_stats* stats) {
    return guarded([&] { return rt_render_impl(s, spec, part, rgba_out, stats); });
}
int rt_generate_image(const rt_scene_desc* desc, const rt_image_spec* spec, const rt_partition* part, uint8_t* rgba_out, rt_stats* stats) {
    return guarded([&] { return rt_generate_image_impl(desc, spec, part, rgba_out, stats); });
}
int rt_trace_rays(rt_scene* s, const float* rays, int64_t n, int32_t any_hit, int32_t* prim_out, float* t_out) {
    return guarded([&] { return rt_trace_rays_impl(s, rays, n, any_hit, prim_out, t_out); });
}
int rt_camera_rays(rt_scene* s, const rt_image_spec* spec, uint64_t seed, const int32_t* pixels, int64_t n, float* rays_out) {
    return guarded([&] { return rt_camera_rays_impl(s, spec, seed, pixels, n, rays_out); });
}
int rt_partition_share(const rt_image_spec* spec, const rt_partition* part, rt_share* out) {
    return guarded([&] {
        if (!part || !out) return fail(RT_ERR_INVALID, "part/out is NULL");
        int rc = validate_spec(spec);
        if (rc != RT_OK) return rc;
        rt_partition p = *part;
        if (p.mode < RT_PART_ALL || p.mode > RT_PART_SAMPLES) return fail(RT_ERR_INVALID, "unknown partition mode");
        if (p.mode == RT_PART_ALL) { p.rank = 0; p.world_size = 1; }
        if (p.world_size < 1 || p.rank < 0 || p.rank >= p.world_size) return fail(RT_ERR_INVALID, "bad rank/world_size");
        const int tile = p.tile_size > 0 ? p.tile_size : 64;
        if (tile % 8 != 0 || tile > 1024) return fail(RT_ERR_INVALID, "tile_size must be a multiple of 8, at most 1024");
        share_of(spec->width, spec->height, spec->anti_aliasing_factor, p, tile, out);
        return RT_OK;
    });
}
int rt_multi_create(const rt_scene_desc* desc, const int32_t* devices, int32_t n_devices, rt_multi** out) {
    return guarded([&] { return rt_multi_create_impl(desc, devices, n_devices, out); });
}
void rt_multi_destroy(rt_multi* m) { (void)guarded([&] { rt_multi_destroy_impl(m); return RT_OK; }); }
int rt_multi_render(rt_multi* m, const rt_image_spec* spec, int32_t mode, uint64_t seed, uint8_t* rgba_out, rt_stats* stats) {
    return guarded([&] { return rt_multi_render_impl(m, spec, mode, seed, rgba_out, stats); });
}
int rt_generate_image_multi(const rt_scene_desc* desc, const rt_image_spec* spec, const int32_t* devices, int32_t n_devices, int32_t mode,
                            uint64_t seed, uint8_t* rgba_out, rt_stats* stats) {
    return guarded([&] {
        int rc = validate_spec(spec);
        if (rc != RT_OK) return rc;
        rt_multi* m = nullptr;
        rc = rt_multi_create_impl(desc, devices, n_devices, &m);
        if (rc != RT_OK) return rc;
        rc = rt_multi_render_impl(m, spec, mode, seed, rgba_out, stats);
        rt_multi_destroy_impl(m);
        return rc;
    });
}
int rt_scatter_probe(rt_scene* s, const rt_image_spec* spec, uint64_t seed, const float* rays, const uint32_t* keys, int64_t n, float* out) {
    return guarded([&] { return rt_scatter_probe_impl(s, spec, seed, rays, keys, n, out); });
}

}  // extern "C"
