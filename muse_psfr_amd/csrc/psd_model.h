// The PSD model of simul_psd_wfm (psfrec.py:36-151) as the kernels of stage A evaluate it: the fitting
// term (psd_fit, psfrec.py:616-626) and max(fit, AO) inside the corrected zone (psfrec.py:148-149).
// Shared by stage_a.hip (full-size transforms) and stage_a2.hip (series + patch form).
#pragma once
#include "device_common.h"

namespace mpsfr {
namespace {

// x^(-11/6) = (x^(-1/6))^11 for x > 0 in the float range.  y = x^(-1/6) from a hardware
// log2/exp2 seed (relative error e0 ~ 1e-6) and Newton steps on y^-6 = x,
//   y <- y (7 - x y^6) / 6,   e' = -3.5 e^2   (1e-6 -> 1e-11 -> 1e-21),
// then five multiplies: ~25 fp64 instructions against ~70 for cbrt(sqrt(x)) / x^2 (this function
// is what K_PSD_ROWFFT spends its VALU time on: every element of every distinct PSD row).
// (NEWTON = 1: relative error ~4e-12, mixed mode -- D is stored as fp32; 2: f64 mode)
template <int NEWTON>
__device__ __forceinline__ double pow_m11_6(double x) {
    double y = (double)__builtin_amdgcn_exp2f(-0.16666667f * __builtin_amdgcn_logf((float)x));
#pragma unroll
    for (int it = 0; it < NEWTON; ++it) {
        const double y2 = y * y, y3 = y2 * y;
        y = y * fma(-(1.0 / 6.0) * x, y3 * y3, 7.0 / 6.0);
    }
    const double y2 = y * y, y4 = y2 * y2;
    return (y4 * y4) * (y2 * y);
}

// fitting term (psd_fit psfrec.py:616-626) at row su, column sv of the half-pixel grid
template <int NEWTON>
__device__ __forceinline__ double psd_fit_value(int su, int sv, const TaskPar& p, double cfit) {
    const double fx = sv + 0.5, fy = su + 0.5;
    const double f2 = (fx * fx + fy * fy) * (1.0 / 256.0);          // L = 16 m, psfrec.py:618
    if (f2 < 2.25) return 0.0;                                       // f >= fc = 1.5, :624
    const double x = f2 + p.inv_l0sq;
    double v = cfit * p.r0m53 * pow_m11_6<NEWTON>(x);
    // basis task (stage_a2.hip): term k = basis - 1 of the expansion in 1/L0^2, x^(-11/6 - k), with
    // the binomial coefficient in r0m53
    if (p.basis > 0) {
        const double xi = 1.0 / x;
        for (int k = 1; k < p.basis; ++k) v *= xi;
    }
    return v;
}

// max(fit, AO) inside the 80 x 80 corrected zone (psfrec.py:148-149), fit elsewhere
template <int NEWTON>
__device__ __forceinline__ double psd_with_ao(double fit, int su, int sv, const TaskPar& p,
                                              const double* __restrict__ tb) {
    if (su >= -NAO / 2 && su < NAO / 2 && sv >= -NAO / 2 && sv < NAO / 2) {
        const int ia = su < 0 ? su + NAO : su, ib = sv < 0 ? sv + NAO : sv;
        const double g2 = (double)(su * su + sv * sv) * (1.0 / 256.0);
        const double vk = 0.0229 * p.r0m53 * pow_m11_6<NEWTON>(g2 + p.inv_l0sq);   // :569-571
        const int o = ia * NAO + ib;
        const double ao = vk * (p.cn2_0 * tb[o] + p.cn2_1 * tb[NAO * NAO + o]) +
                          tb[2 * NAO * NAO + o];
        fit = fmax(fit, ao);                                        // :149
    }
    return fit;
}

}  // namespace
}  // namespace mpsfr
