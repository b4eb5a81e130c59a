// Issue rates of the instructions a cheaper slab test could be built from (design-time micro-benchmark, gfx950):
// v_fma_f32 (baseline), v_fma_mix_f32 (f16 operand, f32 result), v_perm_b32, v_cvt_f32_ubyte0, v_max3_f32.
// Measured on an MI355X (profiles/r02/calib_mix.txt), nominal cycles per wave64 instruction and SIMD at 6 waves per
// SIMD: v_add/sub/mul_f32, v_mov, v_and/or/xor, v_add/sub_u32, v_lshrrev 2.4-2.9; nearly everything else (v_fma_f32
// with three register sources, v_max/min(3), v_cmp, v_cndmask, v_cvt_f32_ubyte, v_perm, v_bfe, v_alignbit, v_pk_fma_f32
// per PAIR) 4.0-4.9; v_rcp_f32 8.2.  What it led to in k_shadow: the slab hit mask through v_sub_f32 + v_alignbit_b32
// instead of v_cmp + v_addc (351.8 -> 347.2 ms); pairing the slab's 48 v_fma_f32 into 24 v_pk_fma_f32 made the
// kernel SLOWER (351.8 -> 373.5 ms), so these figures rank single instructions in a register-only stream and do not
// add up to a kernel's time.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o build/calib_mix scripts/calib_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 8192;
template <int OP>
__global__ __launch_bounds__(256, 6) void k(const uint32_t* in, float* out) {
    float a[16];
    uint32_t w = in[threadIdx.x];
    const h2 h = __builtin_bit_cast(h2, in[threadIdx.x + 256]);
    const float c = 1e-9f;
    const uint32_t hsub = (in[threadIdx.x + 256] & 0x00ff00ffu) | 0x00010001u;  // two float16 subnormals
    const unsigned long long mask = __ballot(w & 1u);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 d[8], d2 = {1.0000001f, 0.9999999f};
    for (int i = 0; i < 8; ++i) d[i] = f2{(float)threadIdx.x, (float)i};
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0000001f, c);
            if (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h), "v"(c));
            if (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(0x04010400u));
            if (OP == 3) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(h));
            if (OP == 5) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(a[i]), "v"(c) : "vcc");
            if (OP == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 11) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 14) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
            if (OP == 16) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
            if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            if (OP == 18) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d2));
            if (OP == 19) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 20) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 21) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(h));
            if (OP == 22) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(h));
            if (OP == 23) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            if (OP == 24) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(mask));
            if (OP == 25) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 26) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 27) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 28) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 29) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
            if (OP == 30) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 31) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 33) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(h));
            if (OP == 34) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 35) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 36) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 37) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 38) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 39) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            if (OP == 40) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == 41) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(a[i]));
            if (OP == 42) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(hsub), "v"(c));  // float16 SUBNORMAL operand
            if (OP == 43) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h), "v"(c));  // upper half
        }
    }
    float s = (float)w;
    for (int i = 0; i < 8; ++i) s += d[i].x + d[i].y;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
static void run(const char* name, const uint32_t* in, float* out, int cus, double clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(cus * 6), dim3(256), 0, 0, in, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus * 6), dim3(256), 0, 0, in, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = (OP == 5 || OP == 23 ? 2.0 : 1.0) * 16.0 * kIters * 6;  // wave instructions per SIMD
    printf("%-18s %8.3f ms  %.3f nominal cycles per wave64 instruction per SIMD\n", name, ms, ms * 1e-3 * clk / per);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
    uint32_t* in; float* out;
    hipMalloc(&in, 4096); hipMemset(in, 0x3c, 4096); hipMalloc(&out, (size_t)cus * 6 * 256 * 4);
    run<0>("v_fma_f32", in, out, cus, clk);
    run<1>("v_fma_mix_f32", in, out, cus, clk);
    run<2>("v_perm_b32", in, out, cus, clk);
    run<3>("v_cvt_f32_ubyte0", in, out, cus, clk);
    run<4>("v_max3_f32", in, out, cus, clk);
    run<5>("v_cmp+v_addc", in, out, cus, clk);
    run<6>("v_max_f32", in, out, cus, clk);
    run<7>("v_add_f32", in, out, cus, clk);
    run<8>("v_mul_f32", in, out, cus, clk);
    run<9>("v_cndmask_b32", in, out, cus, clk);
    run<10>("v_and_b32", in, out, cus, clk);
    run<11>("v_lshlrev_b32", in, out, cus, clk);
    run<12>("v_add_u32", in, out, cus, clk);
    run<13>("v_lshl_or_b32", in, out, cus, clk);
    run<14>("v_sub_f32", in, out, cus, clk);
    run<15>("v_mov_b32", in, out, cus, clk);
    run<16>("v_cmp_le_f32", in, out, cus, clk);
    run<17>("v_bfe_u32", in, out, cus, clk);
    run<18>("v_pk_fma_f32", in, out, cus, clk);
    run<19>("v_ldexp_f32", in, out, cus, clk);
    run<20>("v_min_u32", in, out, cus, clk);
    run<21>("v_fma_f32(asm)", in, out, cus, clk);
    run<22>("v_fmac_f32", in, out, cus, clk);
    run<23>("v_cmp+v_cndmask", in, out, cus, clk);
    run<24>("v_cndmask(sgpr)", in, out, cus, clk);
    run<25>("v_min_f32", in, out, cus, clk);
    run<26>("v_or_b32", in, out, cus, clk);
    run<27>("v_lshrrev_b32", in, out, cus, clk);
    run<28>("v_and_or_b32", in, out, cus, clk);
    run<29>("v_ffbl_b32", in, out, cus, clk);
    run<30>("v_bcnt_u32_b32", in, out, cus, clk);
    run<31>("v_mad_u32_u24", in, out, cus, clk);
    run<32>("v_rcp_f32", in, out, cus, clk);
    run<33>("v_med3_f32", in, out, cus, clk);
    run<34>("v_sub_u32", in, out, cus, clk);
    run<35>("v_xor_b32", in, out, cus, clk);
    run<36>("v_lshl_add_u32", in, out, cus, clk);
    run<37>("v_add3_u32", in, out, cus, clk);
    run<38>("v_mul_lo_u32", in, out, cus, clk);
    run<39>("v_mul_hi_u32", in, out, cus, clk);
    run<40>("v_mov_dpp", in, out, cus, clk);
    run<41>("v_mul_f32_imm", in, out, cus, clk);
    run<42>("v_fma_mix(subnorm)", in, out, cus, clk);
    run<43>("v_fma_mix(hi half)", in, out, cus, clk);
    return 0;
}
