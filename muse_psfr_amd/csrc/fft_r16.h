// Radix-16 line transforms with a merged last pass, for the per-wavelength kernel (K_OTF_R16).
//
// K_OTF_ROWFFT is bound by the LDS store path (a ds_write_b64 costs 6 cycles per wave whatever the
// banking: SQ_WAIT_INST_LDS is a third of its wave cycles), and a radix-8 plan of a 512-point line
// crosses the LDS three times (8.8.8, the third crossing only feeds the extraction of <= 84 of the
// 512 outputs).  Here a thread holds 16 points and a line is N / 16 threads:
//   pass 0 : radix 16 from registers (x[e] = line[t + e TPR]), no twiddles     -> LDS image 0
//   pass 1 : radix 16 with the thread's 15 register twiddles                    -> LDS image 1
//   pass L : radix RL = N / 256 (1, 2, 4) is NOT executed: the extraction evaluates it for the
//            few outputs it needs,  Z[k + 256 q'] = sum_q W_RL^(q q') W_N^(q k) img1[k + 256 q].
// Two LDS crossings instead of three (four for 1024 = 8.8.4.4), 15 instead of 14 + 14 twiddle
// products per 16 points, and the lines of a wave stay independent (a line is a quarter, a half
// or a whole wavefront: in-order LDS, no s_barrier).
// Forward sign convention as fft_lds.h: X[k] = sum_n x[n] exp(-2 pi i n k / N).
#pragma once
#include "fft_lds.h"

namespace mpsfr {

template <int N>
struct R16 {
    static_assert(N == 256 || N == 512 || N == 1024, "16.16.{1,2,4} plans");
    static constexpr int NIMG = 256;                // image 1: RL blocks of 256 elements
    static constexpr int TPR = N / 16;              // threads per line
    static constexpr int RL = N / 256;              // radix of the merged last pass
    static constexpr int NPAD = N + N / 16;         // image 0 is padded x -> x + x / 16
    static constexpr int THREADS = 256;
    static constexpr int LINES = THREADS / TPR;     // lines per workgroup
};

// image 0 (after pass 0): lane stride of the writes is 16 elements; one pad element per 16 puts
// the 16 lanes of a ds_write_b64 group on 16 distinct bank pairs.  Image 1 is the identity.
__device__ __forceinline__ int r16_img0(int x) { return x + (x >> 4); }

// 16-point DFT, v[n] -> v[k] (natural order).  16 = 4 x 4: n = 4 n1 + n2, k = k1 + 4 k2.
template <typename R>
__device__ __forceinline__ void dft16(cx<R>* v) {
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173;
    constexpr double H = 0.70710678118654752440;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // v[4 k1 + n2] *= W16^(n2 k1)
    auto rot = [](cx<R> a, double c, double s) -> cx<R> {      // a * (c - i s)
        return {(R)c * a.x + (R)s * a.y, (R)c * a.y - (R)s * a.x};
    };
    v[5] = rot(v[5], C1, S1);                                  // W16^1
    v[6] = {(R)H * (v[6].x + v[6].y), (R)H * (v[6].y - v[6].x)};   // W16^2 = (1 - i)/sqrt2
    v[7] = rot(v[7], S1, C1);                                  // W16^3
    v[9] = {(R)H * (v[9].x + v[9].y), (R)H * (v[9].y - v[9].x)};   // W16^2
    v[10] = cmulmi(v[10]);                                     // W16^4 = -i
    v[11] = {(R)H * (v[11].y - v[11].x), -(R)H * (v[11].x + v[11].y)};   // W16^6 = (-1 - i)/sqrt2
    v[13] = rot(v[13], S1, C1);                                // W16^3
    v[14] = {(R)H * (v[14].y - v[14].x), -(R)H * (v[14].x + v[14].y)};   // W16^6
    v[15] = rot(v[15], -C1, -S1);                              // W16^9 = -cos(pi/8) + i sin(pi/8)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    cx<R> o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// The thread's pass-1 twiddles W_N^(q k RL), q = 1..15, k = t % 16.  twg[m] = exp(-2 pi i m / N).
template <typename R, int N>
struct R16Tw {
    cx<R> w[15];
    template <typename RT>
    __device__ __forceinline__ void init(const cx<RT>* __restrict__ twg, int t) {
        const int k = t & 15;
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const cx<RT> v = twg[q * k * R16<N>::RL];
            w[q - 1] = {(R)v.x, (R)v.y};
        }
    }
};

// Passes 0 and 1 of one line: x[e] = line[t + e TPR] in registers, `buf` the line's LDS buffer
// (R16<N>::NPAD elements).  Every thread of the line must call them; the line must not span more
// than one wavefront.  After r16_pass1 image 1 (identity layout, N elements) is visible to the
// line.
template <typename R, int N>
__device__ __forceinline__ void r16_pass0(cx<R>* x, cx<R>* buf, int t) {
    dft16(x);
#pragma unroll
    for (int q = 0; q < 16; ++q) buf[17 * t + q] = x[q];           // r16_img0(16 t + q)
    fft_sync<true>();
}

template <typename R, int N>
__device__ __forceinline__ void r16_pass1(cx<R>* buf, const cx<R>* tw, int t) {
    constexpr int TPR = R16<N>::TPR;
    cx<R> v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = buf[r16_img0(t + q * TPR)];
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], tw[q - 1]);
    dft16(v);
    const int k = t & 15;
#pragma unroll
    for (int q = 0; q < 16; ++q) buf[(t - k) * 16 + k + 16 * q] = v[q];
    fft_sync<true>();
}

}  // namespace mpsfr
