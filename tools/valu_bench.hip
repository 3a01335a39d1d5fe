// Vector-instruction issue rate on one SIMD as a function of the waves that share it: W waves per SIMD (a block of
// 256*W threads per CU) each run a stream of independent instructions of one kind.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_bench tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ void stream_kernel(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((f2){a, a}), "v"((f2){b, b}));
            } else if (KIND == 2) {
                asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                             "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(u0), "v"(u1));
            } else if (KIND == 3) {
                asm volatile("v_max3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 6) {      // mixed-precision fma: half-precision first operand, converted inside the instruction
                asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(u0), "v"(b));
            } else if (KIND == 7) {      // integer and bit-field work
                asm volatile("v_bfe_u32 %0, %8, 8, 8\n v_and_b32 %1, %8, %9\n v_lshlrev_b32 %2, 16, %8\n v_perm_b32 %3, %8, %9, %8\n"
                             "v_bfe_u32 %4, %9, 16, 8\n v_and_b32 %5, %9, %8\n v_lshlrev_b32 %6, 8, %9\n v_perm_b32 %7, %9, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(u0), "v"(u1));
            } else if (KIND == 8) {      // two-operand min/max and mul/add
                asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                             "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 9) {      // selects alone (the condition stays in vcc)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 4) {      // a dependent chain: each instruction needs the one before it
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(x0) : "v"(a), "v"(b));
            } else if (KIND == 5) {      // compare into a scalar pair, then a select that reads it
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n"
                             "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %5, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
static void run(const char* name, float* out, int cus, double ghz)
{
    const int iters = 20000;
    for (int w = 1; w <= 4; w++) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        stream_kernel<KIND><<<cus, 256 * w>>>(out, 100, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e0));
        stream_kernel<KIND><<<cus, 256 * w>>>(out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double per_wave = (double)iters * 64.0;
        printf("%-28s %d wave(s)/SIMD: %.2f cycles per instruction on the SIMD, %.2f per wave\n", name, w, ms * 1e-3 * ghz * 1e9 / (per_wave * w), ms * 1e-3 * ghz * 1e9 / per_wave);
    }
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz (nominal; ratios between rows are what matters)\n", p.name, p.multiProcessorCount, ghz);
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 1024 * p.multiProcessorCount));
    run<0>("v_fma_f32 independent", out, p.multiProcessorCount, ghz);
    run<1>("v_pk_fma_f32 independent", out, p.multiProcessorCount, ghz);
    run<2>("v_cvt_f32_ubyteN", out, p.multiProcessorCount, ghz);
    run<3>("v_max3/min3_f32", out, p.multiProcessorCount, ghz);
    run<4>("v_fma_f32 dependent chain", out, p.multiProcessorCount, ghz);
    run<5>("v_cmp + v_cndmask pairs", out, p.multiProcessorCount, ghz);
    run<6>("v_fma_mix_f32 (f16 operand)", out, p.multiProcessorCount, ghz);
    run<7>("bfe / and / lshl / perm", out, p.multiProcessorCount, ghz);
    run<8>("min / max / mul / add f32", out, p.multiProcessorCount, ghz);
    run<9>("v_cndmask_b32 alone", out, p.multiProcessorCount, ghz);
    CHECK(hipFree(out));
    return 0;
}
