// Micro-benchmark: issue cost of fp64 multiply-adds with and without the DPP row_newbcast operand,
// 1..4 waves per SIMD, 8 independent accumulators.  Build: hipcc --offload-arch=gfx950 -O3 -o dpp64 dpp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(double* out, long long* cyc, int iters) {
    double a[8];
    double x = 1.0 + threadIdx.x * 1e-9, w = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            asm volatile(
                "v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %9\n\tv_fmac_f64_e32 %2, %8, %9\n\tv_fmac_f64_e32 %3, %8, %9\n\t"
                "v_fmac_f64_e32 %4, %8, %9\n\tv_fmac_f64_e32 %5, %8, %9\n\tv_fmac_f64_e32 %6, %8, %9\n\tv_fmac_f64_e32 %7, %8, %9"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
        } else if constexpr (MODE == 1) {
            asm volatile(
                "v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(w));
        } else if constexpr (MODE == 2) {      // scalar operand
            double sx;
            asm volatile("s_mov_b64 %0, 1.0" : "=s"(sx));
            asm volatile(
                "v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %9\n\tv_fmac_f64_e32 %2, %8, %9\n\tv_fmac_f64_e32 %3, %8, %9\n\t"
                "v_fmac_f64_e32 %4, %8, %9\n\tv_fmac_f64_e32 %5, %8, %9\n\tv_fmac_f64_e32 %6, %8, %9\n\tv_fmac_f64_e32 %7, %8, %9"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "s"(sx), "v"(w));
        } else if constexpr (MODE == 3) {      // v_mov_b64_dpp + plain fmac
            double t;
            asm volatile(
                "v_mov_b64_dpp %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_e32 %0, %8, %10\n\tv_fmac_f64_e32 %1, %8, %10\n\tv_fmac_f64_e32 %2, %8, %10\n\tv_fmac_f64_e32 %3, %8, %10\n\t"
                "v_fmac_f64_e32 %4, %8, %10\n\tv_fmac_f64_e32 %5, %8, %10\n\tv_fmac_f64_e32 %6, %8, %10\n\tv_fmac_f64_e32 %7, %8, %10"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=&v"(t) : "v"(x), "v"(w));
        } else if constexpr (MODE == 4) {      // fp32 dpp fmac for comparison (row_share / quad_perm would be 32-bit)
            float* f = reinterpret_cast<float*>(a);
            float fx = (float)x, fw = (float)w;
            asm volatile(
                "v_fmac_f32_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(fx), "v"(fw));
        } else if constexpr (MODE == 5) {      // plain fp32 fmac
            float* f = reinterpret_cast<float*>(a);
            float fx = (float)x, fw = (float)w;
            asm volatile(
                "v_fmac_f32_e32 %0, %8, %9\n\tv_fmac_f32_e32 %1, %8, %9\n\tv_fmac_f32_e32 %2, %8, %9\n\tv_fmac_f32_e32 %3, %8, %9\n\t"
                "v_fmac_f32_e32 %4, %8, %9\n\tv_fmac_f32_e32 %5, %8, %9\n\tv_fmac_f32_e32 %6, %8, %9\n\tv_fmac_f32_e32 %7, %8, %9"
                : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(fx), "v"(fw));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {         // waves per SIMD: block = 64 * 4 * wps threads, one block per CU
        const int threads = 256 * wps, blocks = 256;
        double* out; long long* cyc;
        hipMalloc(&out, sizeof(double) * threads * blocks);
        hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= h.size();
        // s_memtime counts at 100 MHz; wall time is the reliable figure: instructions per SIMD / time
        const double inst_per_simd = (double)iters * 8 * wps;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz), memtime ticks %.0f\n",
               name, wps, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, mean);
        hipFree(out); hipFree(cyc);
    }
}

int main() {
    run<0>("v_fmac_f64");
    run<1>("v_fmac_f64_dpp newbcast");
    run<2>("v_fmac_f64 sgpr operand");
    run<3>("mov_b64_dpp + 8 fmac_f64");
    run<4>("v_fmac_f32_dpp newbcast");
    run<5>("v_fmac_f32");
    return 0;
}
