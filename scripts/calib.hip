// Calibration micro-benchmarks for the k_shadow roofline:
//   valu    register-only v_fma_f32 stream at W waves per SIMD -> cycles per wave64 VALU instruction per SIMD
//   gather  16-byte-per-lane loads of 32-byte records from an L2-resident 8 MiB table (the child-record fetch
//           pattern of visit_block), R records in flight, at W waves per SIMD, with `share` lanes reading the
//           same record (1 = every lane its own, 16 = a lane group walks together)
//           -> cycles per wave-instruction per CU and bytes per cycle per CU through the TA / vector L1
// Build: hipcc --offload-arch=gfx950 -O3 -o calib scripts/calib.hip ; run: ./calib  (prints one JSON line)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

constexpr int kIters = 16384;

template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_valu(float* out, float seed, unsigned long long* cycles) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// the same with packed fp32 (v_pk_fma_f32: two fma per lane and instruction): does it issue at the scalar rate?
typedef float float2v __attribute__((ext_vector_type(2)));
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_valu_pk(float* out, float seed, unsigned long long* cycles) {
    float2v a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = float2v{seed + (float)(threadIdx.x + i), seed - (float)i};
    const float2v m = {1.0000001f, 0.9999999f}, c = {1e-9f, 2e-9f};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = 0;
}

template <int WAVES, int RECS>
__global__ __launch_bounds__(256, WAVES) void k_gather(const float4* __restrict__ table, uint32_t mask, int share, float* out,
                                                      unsigned long long* cycles) {
    const uint32_t lane_key = (blockIdx.x * 256u + threadIdx.x) / (uint32_t)share;
    uint32_t x = lane_key * 2654435761u + 12345u;
    float acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters / 8; ++it) {
        float4 ra[RECS], rb[RECS];
#pragma unroll
        for (int k = 0; k < RECS; ++k) {
            x = x * 1664525u + 1013904223u;
            const uint32_t rec = (x >> 8) & mask;
            ra[k] = table[2 * (size_t)rec];
            rb[k] = table[2 * (size_t)rec + 1];
        }
#pragma unroll
        for (int k = 0; k < RECS; ++k) acc += ra[k].x + rb[k].w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <class F>
static int timed(F&& launch, float* ms) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(ms, a, b));
    CHECK(hipGetLastError());
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk_hz = prop.clockRate * 1e3;
    float *out; unsigned long long* cyc; float4* table;
    const uint32_t n_rec = 1u << 18;  // 8 MiB of 32-byte records
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&table, (size_t)n_rec * 32));
    CHECK(hipMemset(table, 0, (size_t)n_rec * 32));
    std::vector<unsigned long long> h((size_t)cus * 8);
    auto mean_cycles = [&](int blocks) {
        (void)hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[(size_t)i];
        return s / blocks;
    };
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f", prop.gcnArchName, cus, clk_hz / 1e6);
    // ---- VALU: 16 x kIters fma per thread; blocks of 4 waves = one per SIMD
    {
        float ms; double c;
        const double insts = 16.0 * kIters;
        if (timed([&] { hipLaunchKernelGGL(k_valu<1>, dim3(cus), dim3(256), 0, 0, out, 1.0f, cyc); }, &ms)) return 1;
        c = mean_cycles(cus);
        printf(", \"valu_1wave_per_simd\": {\"ms\": %.4f, \"cycles_per_inst_in_kernel\": %.3f, \"cycles_per_inst_by_event\": %.3f}", ms,
               c / insts, ms * 1e-3 * clk_hz / insts);
        if (timed([&] { hipLaunchKernelGGL(k_valu<5>, dim3(cus * 5), dim3(256), 0, 0, out, 1.0f, cyc); }, &ms)) return 1;
        c = mean_cycles(cus * 5);
        printf(", \"valu_5waves_per_simd\": {\"ms\": %.4f, \"cycles_per_inst_per_simd_in_kernel\": %.3f, \"cycles_per_inst_per_simd_by_event\": %.3f}", ms,
               c / (insts * 5), ms * 1e-3 * clk_hz / (insts * 5));
        if (timed([&] { hipLaunchKernelGGL(k_valu<8>, dim3(cus * 8), dim3(256), 0, 0, out, 1.0f, cyc); }, &ms)) return 1;
        c = mean_cycles(cus * 8);
        printf(", \"valu_8waves_per_simd\": {\"ms\": %.4f, \"cycles_per_inst_per_simd_in_kernel\": %.3f, \"cycles_per_inst_per_simd_by_event\": %.3f}", ms,
               c / (insts * 8), ms * 1e-3 * clk_hz / (insts * 8));
    }
    {   // 8 x kIters packed fma per thread = the flops of the scalar loop in half the instructions
        float ms;
        if (timed([&] { hipLaunchKernelGGL(k_valu_pk<5>, dim3(cus * 5), dim3(256), 0, 0, out, 1.0f, cyc); }, &ms)) return 1;
        printf(", \"valu_pk_fma_5waves_per_simd\": {\"ms\": %.4f, \"cycles_per_pk_inst_per_simd_by_event\": %.3f}", ms,
               ms * 1e-3 * clk_hz / (8.0 * kIters * 5));
    }
    // ---- gather: per thread kIters/8 rounds x RECS records x 2 dwordx4 loads; 5 waves per SIMD (k_shadow's occupancy)
    {
        const int shares[] = {1, 4, 16, 64};
        for (int s : shares) {
            float ms;
            if (timed([&] { hipLaunchKernelGGL((k_gather<5, 4>), dim3(cus * 5), dim3(256), 0, 0, table, n_rec - 1, s, out, cyc); }, &ms)) return 1;
            const double wave_insts_per_cu = 20.0 * (kIters / 8) * 4 * 2;  // 20 waves per CU
            const double cyc_per_inst = ms * 1e-3 * clk_hz / wave_insts_per_cu;
            printf(", \"gather_share%d\": {\"ms\": %.4f, \"cycles_per_wave_load_per_cu\": %.2f, \"lane_bytes_per_cycle_per_cu\": %.1f}", s, ms,
                   cyc_per_inst, 1024.0 / cyc_per_inst);
        }
        // the same with a 16 KiB table: every line stays in the CU's 32 KiB vector L1 (tag-lookup rate of L1 hits)
        for (int s : shares) {
            float ms;
            if (timed([&] { hipLaunchKernelGGL((k_gather<5, 4>), dim3(cus * 5), dim3(256), 0, 0, table, 511u, s, out, cyc); }, &ms)) return 1;
            const double wave_insts_per_cu = 20.0 * (kIters / 8) * 4 * 2;
            const double cyc_per_inst = ms * 1e-3 * clk_hz / wave_insts_per_cu;
            printf(", \"gather_l1_share%d\": {\"ms\": %.4f, \"cycles_per_wave_load_per_cu\": %.2f, \"lane_bytes_per_cycle_per_cu\": %.1f}", s, ms,
                   cyc_per_inst, 1024.0 / cyc_per_inst);
        }
        float ms;
        if (timed([&] { hipLaunchKernelGGL((k_gather<5, 8>), dim3(cus * 5), dim3(256), 0, 0, table, n_rec - 1, 16, out, cyc); }, &ms)) return 1;
        const double wave_insts_per_cu = 20.0 * (kIters / 8) * 8 * 2;
        printf(", \"gather_share16_8inflight\": {\"ms\": %.4f, \"cycles_per_wave_load_per_cu\": %.2f}", ms, ms * 1e-3 * clk_hz / wave_insts_per_cu);
    }
    printf("}\n");
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(table);
    return 0;
}
