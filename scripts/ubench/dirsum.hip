// Micro-benchmark (round 6): the OTF tile of the several-directions kernel (k_otf_mfma1<MULTI>), per
// wave and tile step: 8 elements per lane from LDS planes of 2 KB.
//   mode 0: the kernel's form -- nine directions, one v_pk_fma_f32 + two v_exp_f32 per direction and element pair
//   mode K (4, 6, 8): the moment form -- planes Dbar, mu_2 .. mu_K (+ log2 tel): one exponential per element and a
//           polynomial in c:  2^(c' Dbar + l2tel) (1 + a_2 mu_2 + ... + a_K mu_K)
// 2 waves per SIMD (as the kernel), every wave ITER tile steps over a 52 KB LDS buffer; prints cycles per tile step.
// Build: hipcc --offload-arch=gfx950 -O3 -o dirsum dirsum.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k(float* out, long long* cyc, int iters, float c2) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NPL = MODE == 0 ? 9 : MODE;            // planes besides log2 tel
    constexpr int TILE = (NPL + 1) * 2048;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 52 * 1024 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 1.0f + 1e-3f * (i % 97);
    __syncthreads();
    const int ntile = 52 * 1024 / TILE;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const f2 cc = {c2, c2};
    float a[9];
    for (int k2 = 0; k2 < 9; ++k2) a[k2] = c2 * (0.1f + k2);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned char* tp = smem + ((it + wave) % ntile) * TILE + lane * 16;
        const f4 t0v = *reinterpret_cast<const f4*>(tp + NPL * 2048), t1v = *reinterpret_cast<const f4*>(tp + NPL * 2048 + 1024);
        f2 tt[4] = {f2{t0v[0], t0v[1]}, f2{t0v[2], t0v[3]}, f2{t1v[0], t1v[1]}, f2{t1v[2], t1v[3]}};
        float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < 9; ++d) {
                const f4 d0 = *reinterpret_cast<const f4*>(tp + d * 2048), d1 = *reinterpret_cast<const f4*>(tp + d * 2048 + 1024);
                const f2 dd[4] = {f2{d0[0], d0[1]}, f2{d0[2], d0[3]}, f2{d1[0], d1[1]}, f2{d1[2], d1[3]}};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f2 y;
                    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(cc), "v"(dd[q]), "v"(tt[q]));
                    x[2 * q] += __builtin_amdgcn_exp2f(y[0]);
                    x[2 * q + 1] += __builtin_amdgcn_exp2f(y[1]);
                }
            }
        } else {
            const f4 d0 = *reinterpret_cast<const f4*>(tp), d1 = *reinterpret_cast<const f4*>(tp + 1024);
            const f2 dd[4] = {f2{d0[0], d0[1]}, f2{d0[2], d0[3]}, f2{d1[0], d1[1]}, f2{d1[2], d1[3]}};
            f2 p[4] = {f2{1.f, 1.f}, f2{1.f, 1.f}, f2{1.f, 1.f}, f2{1.f, 1.f}};
#pragma unroll
            for (int m = 1; m < NPL; ++m) {
                const f4 m0 = *reinterpret_cast<const f4*>(tp + m * 2048), m1 = *reinterpret_cast<const f4*>(tp + m * 2048 + 1024);
                const f2 mm[4] = {f2{m0[0], m0[1]}, f2{m0[2], m0[3]}, f2{m1[0], m1[1]}, f2{m1[2], m1[3]}};
                const f2 am = {a[m], a[m]};
#pragma unroll
                for (int q = 0; q < 4; ++q) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[q]) : "v"(am), "v"(mm[q]));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f2 y;
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(cc), "v"(dd[q]), "v"(tt[q]));
                x[2 * q] = __builtin_amdgcn_exp2f(y[0]) * p[q][0];
                x[2 * q + 1] = __builtin_amdgcn_exp2f(y[1]) * p[q][1];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += x[e];
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int e = 0; e < 8; ++e) s += acc[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, float* out, long long* cyc) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 52 * 1024, 0, out, cyc, iters, -0.37f);
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
    long long h[256];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    printf("%-52s %8.1f cycles per tile step and wave (2 waves per SIMD)\n", name, m / 256 / iters);
    return 0;
}

// the reduction pass the moment form needs in stage A: 9 planes in, K planes out, per task
__global__ void __launch_bounds__(256) k_reduce(const float* __restrict__ D, float* __restrict__ M, int plane, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (i >= plane) return;
    float d[9], mean = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) { d[j] = D[((size_t)t * 9 + j) * plane + i]; mean += d[j]; }
    mean *= 1.0f / 9.0f;
    float mu[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float e = d[j] - mean;
        float pw = e;
#pragma unroll
        for (int q = 1; q < 8; ++q) { pw *= e; mu[q] += pw; }
    }
    M[((size_t)t * K) * plane + i] = mean;
    for (int q = 1; q < K; ++q) M[((size_t)t * K + q) * plane + i] = mu[q] * (1.0f / 9.0f);
}

int main() {
    float* out; long long* cyc;
    hipMalloc((void**)&out, 256 * 512 * 4);
    hipMalloc((void**)&cyc, 256 * 8);
    if (run<0>("nine directions: 9 x (pk_fma + 2 exp) per pair", out, cyc)) return 1;
    if (run<4>("moments K = 4: 1 exp + 3 fma per element", out, cyc)) return 1;
    if (run<6>("moments K = 6: 1 exp + 5 fma per element", out, cyc)) return 1;
    if (run<8>("moments K = 8: 1 exp + 7 fma per element", out, cyc)) return 1;
    // reduction pass at configs[3]: 100 tasks x 9 planes of 129 x 256 floats
    const int plane = 129 * 256, T = 100;
    float *D, *M;
    hipMalloc((void**)&D, (size_t)T * 9 * plane * 4);
    hipMalloc((void**)&M, (size_t)T * 8 * plane * 4);
    hipMemset(D, 0, (size_t)T * 9 * plane * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int K : {4, 6, 8}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_reduce, dim3((plane + 255) / 256, T), dim3(256), 0, 0, D, M, plane, K);
        hipEventRecord(a);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_reduce, dim3((plane + 255) / 256, T), dim3(256), 0, 0, D, M, plane, K);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("reduction pass 9 -> %d planes, 100 tasks at 256^2: %.1f us (%.0f MB moved)\n", K, ms * 1e3 / 20, (9.0 + K) * T * plane * 4 / 1e6);
    }
    return 0;
}
