// 64 x 64 frames of the stamp stage: the 64-point line transforms of the FFT convolutions (K_CONV_FFT, stamps.hip)
// and the kernel spectra they multiply by (the body of K_KHAT) -- shared by stamps.hip and stage_a2.hip, where the
// spectra of a chunk's tip-tilt kernels ride as trailing workgroups of K_PATCH_ROWS (round 6: one launch less at
// the head of a call).  Everything here has internal linkage.
#pragma once
#include "device_common.h"

namespace mpsfr {
namespace {

constexpr int CF = 64;            // frame side
constexpr int CFH = CF / 2;       // 32
constexpr int CFP = 36;           // pitch of the half-spectrum frame (bank-conflict free)
constexpr int CFB = CF + CF / 8;  // padded line buffer

// exp(-2 pi i m / 64), m = 0..63: the pass twiddles of the 64-point lines come from this table
// (seven sincospif per thread at kernel start were ~15 % of the kernel's VALU work)
__constant__ float2 kTw64[64] = {
    {1.000000000e+00f, 0.000000000e+00f}, {9.951847267e-01f, -9.801714033e-02f},
    {9.807852804e-01f, -1.950903220e-01f}, {9.569403357e-01f, -2.902846773e-01f},
    {9.238795325e-01f, -3.826834324e-01f}, {8.819212643e-01f, -4.713967368e-01f},
    {8.314696123e-01f, -5.555702330e-01f}, {7.730104534e-01f, -6.343932842e-01f},
    {7.071067812e-01f, -7.071067812e-01f}, {6.343932842e-01f, -7.730104534e-01f},
    {5.555702330e-01f, -8.314696123e-01f}, {4.713967368e-01f, -8.819212643e-01f},
    {3.826834324e-01f, -9.238795325e-01f}, {2.902846773e-01f, -9.569403357e-01f},
    {1.950903220e-01f, -9.807852804e-01f}, {9.801714033e-02f, -9.951847267e-01f},
    {6.123233996e-17f, -1.000000000e+00f}, {-9.801714033e-02f, -9.951847267e-01f},
    {-1.950903220e-01f, -9.807852804e-01f}, {-2.902846773e-01f, -9.569403357e-01f},
    {-3.826834324e-01f, -9.238795325e-01f}, {-4.713967368e-01f, -8.819212643e-01f},
    {-5.555702330e-01f, -8.314696123e-01f}, {-6.343932842e-01f, -7.730104534e-01f},
    {-7.071067812e-01f, -7.071067812e-01f}, {-7.730104534e-01f, -6.343932842e-01f},
    {-8.314696123e-01f, -5.555702330e-01f}, {-8.819212643e-01f, -4.713967368e-01f},
    {-9.238795325e-01f, -3.826834324e-01f}, {-9.569403357e-01f, -2.902846773e-01f},
    {-9.807852804e-01f, -1.950903220e-01f}, {-9.951847267e-01f, -9.801714033e-02f},
    {-1.000000000e+00f, -1.224646799e-16f}, {-9.951847267e-01f, 9.801714033e-02f},
    {-9.807852804e-01f, 1.950903220e-01f}, {-9.569403357e-01f, 2.902846773e-01f},
    {-9.238795325e-01f, 3.826834324e-01f}, {-8.819212643e-01f, 4.713967368e-01f},
    {-8.314696123e-01f, 5.555702330e-01f}, {-7.730104534e-01f, 6.343932842e-01f},
    {-7.071067812e-01f, 7.071067812e-01f}, {-6.343932842e-01f, 7.730104534e-01f},
    {-5.555702330e-01f, 8.314696123e-01f}, {-4.713967368e-01f, 8.819212643e-01f},
    {-3.826834324e-01f, 9.238795325e-01f}, {-2.902846773e-01f, 9.569403357e-01f},
    {-1.950903220e-01f, 9.807852804e-01f}, {-9.801714033e-02f, 9.951847267e-01f},
    {-1.836970199e-16f, 1.000000000e+00f}, {9.801714033e-02f, 9.951847267e-01f},
    {1.950903220e-01f, 9.807852804e-01f}, {2.902846773e-01f, 9.569403357e-01f},
    {3.826834324e-01f, 9.238795325e-01f}, {4.713967368e-01f, 8.819212643e-01f},
    {5.555702330e-01f, 8.314696123e-01f}, {6.343932842e-01f, 7.730104534e-01f},
    {7.071067812e-01f, 7.071067812e-01f}, {7.730104534e-01f, 6.343932842e-01f},
    {8.314696123e-01f, 5.555702330e-01f}, {8.819212643e-01f, 4.713967368e-01f},
    {9.238795325e-01f, 3.826834324e-01f}, {9.569403357e-01f, 2.902846773e-01f},
    {9.807852804e-01f, 1.950903220e-01f}, {9.951847267e-01f, 9.801714033e-02f},
};

template <typename R>
struct Tw64 {
    cx<R> w[7];
    __device__ __forceinline__ void init(int t) {
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            if constexpr (sizeof(R) == 4) {
                const float2 v = kTw64[(q * t) & 63];
                w[q - 1] = {v.x, v.y};
            } else {                 // f64 mode: the table has float digits only
                double sn, cs;
                sincospi(-(double)((q * t) & 63) / 32.0, &sn, &cs);
                w[q - 1] = {cs, sn};
            }
        }
    }
};

template <typename R>
__device__ __forceinline__ cx<R> conjf(cx<R> a) { return {a.x, -a.y}; }

// rows ra = 2 slot, rb = ra + 1 of a real image, x[e] = (row ra, row rb) at column t + 8 e (zero
// beyond the image) -> half spectra in F
template <typename R>
__device__ __forceinline__ void cf_rows_forward(const cx<R>* x, cx<R> (*F)[CFP],
                                                cx<R>* buf, const Tw64<R>& tw, int slot, int t) {
    const int ra = 2 * slot, rb = ra + 1;
    const cx<R>* res = fft_forward_regs<R, CF, true>(x, buf, buf, tw.w, t);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = t + 8 * e;
        const cx<R> zk = res[lds_out<CF, 8>(k)], zm = res[lds_out<CF, 8>((CF - k) % CF)];
        if (k == 0) {
            const cx<R> zn = res[lds_out<CF, 8>(CFH)];
            F[ra][0] = {zk.x, zn.x};      // (DC, Nyquist) of row ra, both real
            F[rb][0] = {zk.y, zn.y};
        } else {
            F[ra][k] = {(R)0.5 * (zk.x + zm.x), (R)0.5 * (zk.y - zm.y)};
            F[rb][k] = {(R)0.5 * (zk.y + zm.y), -(R)0.5 * (zk.x - zm.x)};
        }
    }
}

// column `slot` of F (rows < nrow_in non-zero) -> forward transform along the rows, in `buf`
template <typename R>
__device__ __forceinline__ const cx<R>* cf_col_forward(cx<R> (*F)[CFP], int nrow_in,
                                                       cx<R>* buf, const Tw64<R>& tw,
                                                       int slot, int t) {
    cx<R> x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = t + 8 * e;
        x[e] = r < nrow_in ? F[r][slot] : cx<R>{(R)0, (R)0};
    }
    return fft_forward_regs<R, CF, true>(x, buf, buf, tw.w, t);
}

// split the packed column 0 spectrum into the DC column and the Nyquist column at kx
template <typename R>
__device__ __forceinline__ void cf_split0(const cx<R>* res, int kx, cx<R>& a0,
                                          cx<R>& a32) {
    const cx<R> p = res[lds_out<CF, 8>(kx)], pm = res[lds_out<CF, 8>((CF - kx) % CF)];
    a0 = {(R)0.5 * (p.x + pm.x), (R)0.5 * (p.y - pm.y)};
    a32 = {(R)0.5 * (p.y + pm.y), -(R)0.5 * (p.x - pm.x)};
}

template <typename R>
constexpr size_t conv_smem_bytes(bool with_kernel) {
    return sizeof(cx<R>) * (CF * CFP + CFH * CFB) + (with_kernel ? sizeof(R) * KS * KS : 0);
}


// K_KHAT: spectrum of astropy's Moffat2DKernel(gamma, alpha, 41, 41) (psfrec.py:916, 927) on the
// 64x64 frame, transposed half plane khat[k][kx], with the 1/4096 of the inverse folded in.
// (device body: `kid` the kernel, `conv_smem` conv_smem_bytes<R>(true) bytes of LDS, 256 threads)
template <typename R>
__device__ __forceinline__ void khat_body(const double* __restrict__ gam, const double* __restrict__ alp,
                                          cx<R>* __restrict__ khat, unsigned char* conv_smem, int kid) {
    cx<R> (*F)[CFP] = reinterpret_cast<cx<R> (*)[CFP]>(conv_smem);
    cx<R> (*bufs)[CFB] = reinterpret_cast<cx<R> (*)[CFB]>(conv_smem + sizeof(cx<R>) * CF * CFP);
    R* ker = reinterpret_cast<R*>(conv_smem + sizeof(cx<R>) * (CF * CFP + CFH * CFB));
    __shared__ double part[4];
    __shared__ double tot;
    const int slot = threadIdx.x >> 3, t = threadIdx.x & 7;
    Tw64<R> tw;
    tw.init(t);
    const double g2 = gam[kid] * gam[kid], al = alp[kid];
    constexpr int NV = (KS * KS + 255) / 256;
    double vals[NV];
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int e = threadIdx.x + m * 256;
        double v = 0.0;
        if (e < KS * KS) {
            const int dy = e / KS - KS / 2, dx = e % KS - KS / 2;
            const double x1 = 1.0 + (double)(dx * dx + dy * dy) / g2;
            // the tip-tilt kernel has beta = 2 exactly (psfrec.py:879): no pow
            v = al == 2.0 ? 1.0 / (x1 * x1) : pow(x1, -al);
        }
        vals[m] = v;
        s += v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tot = (part[0] + part[1]) + (part[2] + part[3]);
    __syncthreads();
    const double inv = 1.0 / (tot * (double)(CF * CF));
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int e = threadIdx.x + m * 256;
        if (e < KS * KS) ker[e] = (R)(vals[m] * inv);
    }
    __syncthreads();
    // rows (41 of them, pitch 41): reuse the row-pair transform with a 41-wide image
    if (slot < (KS + 1) / 2) {
        const int ra = 2 * slot, rb = ra + 1;
        cx<R> x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = t + 8 * e;
            x[e] = {(c < KS) ? ker[ra * KS + c] : (R)0, (c < KS && rb < KS) ? ker[rb * KS + c] : (R)0};
        }
        const cx<R>* res = fft_forward_regs<R, CF, true>(x, bufs[slot], bufs[slot], tw.w, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = t + 8 * e;
            const cx<R> zk = res[lds_out<CF, 8>(k)], zm = res[lds_out<CF, 8>((CF - k) % CF)];
            if (k == 0) {
                const cx<R> zn = res[lds_out<CF, 8>(CFH)];
                F[ra][0] = {zk.x, zn.x};
                F[rb][0] = {zk.y, zn.y};
            } else {
                F[ra][k] = {(R)0.5 * (zk.x + zm.x), (R)0.5 * (zk.y - zm.y)};
                F[rb][k] = {(R)0.5 * (zk.y + zm.y), -(R)0.5 * (zk.x - zm.x)};
            }
        }
    }
    __syncthreads();
    const cx<R>* res = cf_col_forward(F, KS, bufs[slot], tw, slot, t);
    cx<R>* out = khat + (size_t)kid * (CFH + 1) * CF;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kx = t + 8 * e;
        if (slot == 0) {
            cx<R> a0, a32;
            cf_split0(res, kx, a0, a32);
            out[kx] = a0;
            out[CFH * CF + kx] = a32;
        } else {
            out[slot * CF + kx] = res[lds_out<CF, 8>(kx)];
        }
    }
}


}  // namespace
}  // namespace mpsfr
