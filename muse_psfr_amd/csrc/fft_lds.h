// LDS-resident Stockham FFTs for gfx950 (wave64), used by every transform on the hot path.
//
// One length-N line is transformed by TPR cooperating threads; a 256-thread workgroup carries
// 256/TPR independent lines ("slots") in lock-step, so the only synchronisation is one
// workgroup barrier per radix pass.  Radix plans: 128 = 8.4.4, 256 = 8.8.4, 512 = 8.8.8,
// 1024 = 8.8.4.4, 1280 = 8.8.4.5 (the reference-native grid of psfrec.py:954-955 needs radix 5).
// Forward sign convention: X[k] = sum_n x[n] exp(-2 pi i n k / N).
#pragma once
#include <hip/hip_runtime.h>

namespace mpsfr {

template <typename R>
struct cx {
    R x, y;
};

template <typename R>
__device__ __forceinline__ cx<R> cadd(cx<R> a, cx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R>
__device__ __forceinline__ cx<R> csub(cx<R> a, cx<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R>
__device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by -i
template <typename R>
__device__ __forceinline__ cx<R> cmulmi(cx<R> a) { return {a.y, -a.x}; }

template <typename R>
__device__ __forceinline__ void dft2(cx<R>& a, cx<R>& b) {
    cx<R> t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

template <typename R>
__device__ __forceinline__ void dft4(cx<R>& a0, cx<R>& a1, cx<R>& a2, cx<R>& a3) {
    cx<R> t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmulmi(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

template <typename R>
__device__ __forceinline__ void dft8(cx<R>* v) {
    const R h = (R)0.70710678118654752440;
    dft4(v[0], v[2], v[4], v[6]);
    dft4(v[1], v[3], v[5], v[7]);
    cx<R> o1 = {h * (v[3].x + v[3].y), h * (v[3].y - v[3].x)};      // * (1 - i)/sqrt2
    cx<R> o2 = cmulmi(v[5]);                                         // * -i
    cx<R> o3 = {h * (v[7].y - v[7].x), -h * (v[7].x + v[7].y)};     // * (-1 - i)/sqrt2
    cx<R> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
}

template <typename R>
__device__ __forceinline__ void dft5(cx<R>* v) {
    const R c1 = (R)0.30901699437494742410, c2 = (R)-0.80901699437494742410;
    const R s1 = (R)0.95105651629515357212, s2 = (R)0.58778525229247312917;
    cx<R> t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    cx<R> t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    cx<R> a0 = v[0];
    cx<R> m1 = {a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y};
    cx<R> m2 = {a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y};
    cx<R> n1 = {s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
    cx<R> n2 = {s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
    v[0] = {a0.x + t1.x + t2.x, a0.y + t1.y + t2.y};
    v[1] = {m1.x + n1.y, m1.y - n1.x};   // m1 - i n1
    v[4] = {m1.x - n1.y, m1.y + n1.x};   // m1 + i n1
    v[2] = {m2.x + n2.y, m2.y - n2.x};
    v[3] = {m2.x - n2.y, m2.y + n2.x};
}

template <typename R, int RADIX>
__device__ __forceinline__ void dftr(cx<R>* v) {
    if constexpr (RADIX == 2) dft2(v[0], v[1]);
    if constexpr (RADIX == 4) dft4(v[0], v[1], v[2], v[3]);
    if constexpr (RADIX == 8) dft8(v);
    if constexpr (RADIX == 5) dft5(v);
}

// One Stockham autosort pass.  NS = product of the radices of the previous passes.
// tw[m] = exp(-2 pi i m / N), m in [0, N).
template <typename R, int N, int RADIX, int NS, int TPR>
__device__ __forceinline__ void fft_pass(const cx<R>* __restrict__ in, cx<R>* __restrict__ out,
                                         const cx<R>* __restrict__ tw, int t) {
    constexpr int NB = N / RADIX;
    for (int j = t; j < NB; j += TPR) {
        const int k = j % NS;
        cx<R> v[RADIX];
#pragma unroll
        for (int q = 0; q < RADIX; ++q) v[q] = in[j + q * NB];
        if constexpr (NS > 1) {
            constexpr int TS = N / (NS * RADIX);
#pragma unroll
            for (int q = 1; q < RADIX; ++q) v[q] = cmul(v[q], tw[q * k * TS]);
        }
        dftr<R, RADIX>(v);
        const int base = (j - k) * RADIX + k;
#pragma unroll
        for (int q = 0; q < RADIX; ++q) out[base + q * NS] = v[q];
    }
}

// Full forward FFT of one line held in LDS buffer `a` (natural order), scratch `b`.
// Every thread of the workgroup must call this (it contains workgroup barriers).
// Returns the buffer that holds the result in natural order.
template <typename R, int N, int TPR>
__device__ __forceinline__ cx<R>* fft_forward(cx<R>* a, cx<R>* b, const cx<R>* tw, int t) {
    static_assert(N == 128 || N == 256 || N == 512 || N == 1024 || N == 1280, "unsupported N");
    if constexpr (N == 128) {
        fft_pass<R, N, 8, 1, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 4, 8, TPR>(b, a, tw, t);
        __syncthreads();
        fft_pass<R, N, 4, 32, TPR>(a, b, tw, t);
        __syncthreads();
        return b;
    } else if constexpr (N == 256) {
        fft_pass<R, N, 8, 1, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 8, 8, TPR>(b, a, tw, t);
        __syncthreads();
        fft_pass<R, N, 4, 64, TPR>(a, b, tw, t);
        __syncthreads();
        return b;
    } else if constexpr (N == 512) {
        fft_pass<R, N, 8, 1, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 8, 8, TPR>(b, a, tw, t);
        __syncthreads();
        fft_pass<R, N, 8, 64, TPR>(a, b, tw, t);
        __syncthreads();
        return b;
    } else if constexpr (N == 1024) {
        fft_pass<R, N, 8, 1, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 8, 8, TPR>(b, a, tw, t);
        __syncthreads();
        fft_pass<R, N, 4, 64, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 4, 256, TPR>(b, a, tw, t);
        __syncthreads();
        return a;
    } else {
        fft_pass<R, N, 8, 1, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 8, 8, TPR>(b, a, tw, t);
        __syncthreads();
        fft_pass<R, N, 4, 64, TPR>(a, b, tw, t);
        __syncthreads();
        fft_pass<R, N, 5, 256, TPR>(b, a, tw, t);
        __syncthreads();
        return a;
    }
}

// threads per line for a given N (8 or 10 points per thread)
template <int N>
struct LineCfg {
    static constexpr int TPR = (N == 1280) ? 128 : N / 8;
    static constexpr int SLOTS = 256 / TPR;
};

}  // namespace mpsfr
