// Shared device helpers and launch plumbing of the hot-path kernels (stage_a.hip, per_lambda.hip,
// stamps.hip).  Everything here has internal linkage.
#pragma once
#include "kernels.h"

#include "fft_lds.h"

namespace mpsfr {
namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kArcminH = 60.0 / 206265.0;   // psfrec.py:279, 440
constexpr double kTi = 1.0e-3;                 // 1/Fsamp, psfrec.py:584
constexpr double kDeltaT = 1.0e-3 + 2.5e-3;    // ti.max() + td, psfrec.py:449, 585

[[maybe_unused]] __device__ __forceinline__ double sinc_pi(double x) {   // np.sinc
    return x == 0.0 ? 1.0 : sinpi(x) / (kPi * x);
}

[[maybe_unused]] __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}


// dynamic LDS bytes of a line-FFT kernel: optional twiddle table + nbuf buffers per slot
template <typename T, int N>
constexpr size_t fft_smem(bool with_table, int nbuf) {
    return (size_t)((with_table ? LineCfg<N>::NPAD : 0) + nbuf * LineCfg<N>::SLOTS * LineCfg<N>::NPAD) *
           sizeof(T) * 2;
}


// kernels that need more than 64 KB of dynamic LDS must opt in
template <typename K>
void allow_smem(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}


}  // namespace
}  // namespace mpsfr

#define DISPATCH_N(N, ...)                                          \
    switch (N) {                                                    \
        case 128: { constexpr int NN = 128; __VA_ARGS__; } break;   \
        case 256: { constexpr int NN = 256; __VA_ARGS__; } break;   \
        case 512: { constexpr int NN = 512; __VA_ARGS__; } break;   \
        case 1024: { constexpr int NN = 1024; __VA_ARGS__; } break; \
        case 1280: { constexpr int NN = 1280; __VA_ARGS__; } break; \
        default: break;                                             \
    }

