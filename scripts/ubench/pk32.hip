// Micro-benchmark: issue cost of packed fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) against
// the plain instructions, and of the transcendental ones (v_exp_f32, v_rcp_f32), 1..4 waves per SIMD,
// 8 independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk32 pk32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(OP, FMT) \
    asm volatile(OP " %0, " FMT "\n\t" OP " %1, " FMT "\n\t" OP " %2, " FMT "\n\t" OP " %3, " FMT "\n\t" \
                 OP " %4, " FMT "\n\t" OP " %5, " FMT "\n\t" OP " %6, " FMT "\n\t" OP " %7, " FMT \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w))

template <int MODE>
__global__ void k(float* out, int iters) {
    if constexpr (MODE < 3) {
        f2 a[8], x = {1.0f + threadIdx.x * 1e-7f, 1.0f}, w = {1.0f - threadIdx.x * 1e-7f, 0.5f};
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f2{(float)i, 1.0f};
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 0) {
                asm volatile(
                    "v_pk_fma_f32 %0, %8, %9, %0\n\tv_pk_fma_f32 %1, %8, %9, %1\n\tv_pk_fma_f32 %2, %8, %9, %2\n\tv_pk_fma_f32 %3, %8, %9, %3\n\t"
                    "v_pk_fma_f32 %4, %8, %9, %4\n\tv_pk_fma_f32 %5, %8, %9, %5\n\tv_pk_fma_f32 %6, %8, %9, %6\n\tv_pk_fma_f32 %7, %8, %9, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            } else if constexpr (MODE == 1) {
                asm volatile(
                    "v_pk_add_f32 %0, %8, %0\n\tv_pk_add_f32 %1, %8, %1\n\tv_pk_add_f32 %2, %8, %2\n\tv_pk_add_f32 %3, %8, %3\n\t"
                    "v_pk_add_f32 %4, %8, %4\n\tv_pk_add_f32 %5, %8, %5\n\tv_pk_add_f32 %6, %8, %6\n\tv_pk_add_f32 %7, %8, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            } else {
                asm volatile(
                    "v_pk_mul_f32 %0, %8, %0\n\tv_pk_mul_f32 %1, %8, %1\n\tv_pk_mul_f32 %2, %8, %2\n\tv_pk_mul_f32 %3, %8, %3\n\t"
                    "v_pk_mul_f32 %4, %8, %4\n\tv_pk_mul_f32 %5, %8, %5\n\tv_pk_mul_f32 %6, %8, %6\n\tv_pk_mul_f32 %7, %8, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w), "v"(x));
            }
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        float a[8], x = 1.0f + threadIdx.x * 1e-7f, w = 1.0f - threadIdx.x * 1e-7f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (float)i;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 3) {
                asm volatile(
                    "v_fmac_f32_e32 %0, %8, %9\n\tv_fmac_f32_e32 %1, %8, %9\n\tv_fmac_f32_e32 %2, %8, %9\n\tv_fmac_f32_e32 %3, %8, %9\n\t"
                    "v_fmac_f32_e32 %4, %8, %9\n\tv_fmac_f32_e32 %5, %8, %9\n\tv_fmac_f32_e32 %6, %8, %9\n\tv_fmac_f32_e32 %7, %8, %9"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            } else if constexpr (MODE == 5) {
                asm volatile(
                    "v_exp_f32_e32 %0, %0\n\tv_exp_f32_e32 %1, %1\n\tv_exp_f32_e32 %2, %2\n\tv_exp_f32_e32 %3, %3\n\t"
                    "v_exp_f32_e32 %4, %4\n\tv_exp_f32_e32 %5, %5\n\tv_exp_f32_e32 %6, %6\n\tv_exp_f32_e32 %7, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            } else if constexpr (MODE == 6) {
                asm volatile(
                    "v_rcp_f32_e32 %0, %0\n\tv_rcp_f32_e32 %1, %1\n\tv_rcp_f32_e32 %2, %2\n\tv_rcp_f32_e32 %3, %3\n\t"
                    "v_rcp_f32_e32 %4, %4\n\tv_rcp_f32_e32 %5, %5\n\tv_rcp_f32_e32 %6, %6\n\tv_rcp_f32_e32 %7, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            } else if constexpr (MODE == 7) {      // the OTF element of the matrix-core stage: fma + exp
                asm volatile(
                    "v_fma_f32 %0, %8, %9, %0\n\tv_exp_f32_e32 %1, %0\n\tv_fma_f32 %2, %8, %9, %2\n\tv_exp_f32_e32 %3, %2\n\t"
                    "v_fma_f32 %4, %8, %9, %4\n\tv_exp_f32_e32 %5, %4\n\tv_fma_f32 %6, %8, %9, %6\n\tv_exp_f32_e32 %7, %6"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            } else {
                asm volatile(
                    "v_add_f32_e32 %0, %8, %0\n\tv_add_f32_e32 %1, %8, %1\n\tv_add_f32_e32 %2, %8, %2\n\tv_add_f32_e32 %3, %8, %3\n\t"
                    "v_add_f32_e32 %4, %8, %4\n\tv_add_f32_e32 %5, %8, %5\n\tv_add_f32_e32 %6, %8, %6\n\tv_add_f32_e32 %7, %8, %7"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
            }
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int MODE>
void run(const char* name) {
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {         // waves per SIMD: block = 64 * 4 * wps threads, one block per CU
        const int threads = 256 * wps, blocks = 256;
        float* out;
        hipMalloc(&out, sizeof(float) * threads * blocks);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)iters * 8 * wps;
        printf("%-16s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n",
               name, wps, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
        hipFree(out);
    }
}

int main() {
    run<0>("v_pk_fma_f32");
    run<1>("v_pk_add_f32");
    run<2>("v_pk_mul_f32");
    run<3>("v_fmac_f32");
    run<4>("v_add_f32");
    run<5>("v_exp_f32");
    run<6>("v_rcp_f32");
    run<7>("fma + exp pairs");
    return 0;
}
