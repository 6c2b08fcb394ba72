// LDS-resident Stockham FFTs for gfx950 (wave64), used by every transform on the hot path.
//
// One length-N line is transformed by TPR cooperating threads ("slot"); a workgroup carries
// SLOTS independent lines.  Radix plans: 128 = 8.4.4, 256 = 8.8.4, 512 = 8.8.8, 1024 = 8.8.4.4,
// 1280 = 20.4.4.4 (the reference-native grid of psfrec.py:954-955 needs a factor 5; with 64 threads
// per line only radices 2, 4, 5, 10, 20 divide the butterflies evenly, and 20 first saves a pass).
//
// gfx950 specifics:
//  * LDS images are padded (x -> x + x/8): the autosort writes of a pass have a lane stride of
//    `radix` elements, which without padding lands 8-16 lanes on one bank
//    (SQ_LDS_BANK_CONFLICT was 41 % of the LDS cycles of the first version).
//  * When a slot is at most one wavefront (TPR <= 64) the passes need no s_barrier at all: LDS
//    operations of one wave execute in order, so a wave-scope fence is enough.
//  * The pass twiddles of a thread depend only on its butterfly index, so kernels that run many
//    transforms per thread (one per wavelength) keep them in registers (TwRegs).
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

// 20 = 4 x 5 (Cooley-Tukey, n = 5 n1 + n2, k = k1 + 4 k2): five radix-4 butterflies over n1,
// the twiddles W20^(n2 k1), four radix-5 butterflies over n2.
template <typename R>
__device__ __forceinline__ void dft20(cx<R>* v) {
    // W20^m = exp(-2 pi i m / 20), m = 0..12
    constexpr double C20[13] = {1.0, 0.95105651629515357212, 0.80901699437494742410,
                                0.58778525229247312917, 0.30901699437494742410, 0.0,
                                -0.30901699437494742410, -0.58778525229247312917,
                                -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                -0.95105651629515357212, -0.80901699437494742410};
    constexpr double S20[13] = {0.0, -0.30901699437494742410, -0.58778525229247312917,
                                -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                -0.95105651629515357212, -0.80901699437494742410,
                                -0.58778525229247312917, -0.30901699437494742410, 0.0,
                                0.30901699437494742410, 0.58778525229247312917};
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) dft4(v[n2], v[5 + n2], v[10 + n2], v[15 + n2]);
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
        for (int n2 = 1; n2 < 5; ++n2) {
            const cx<R> w = {(R)C20[n2 * k1], (R)S20[n2 * k1]};
            v[5 * k1 + n2] = cmul(v[5 * k1 + n2], w);
        }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft5(v + 5 * k1);
    cx<R> o[20];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) o[k1 + 4 * k2] = v[5 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 20; ++i) v[i] = o[i];
}

template <typename R, int RADIX>
__device__ __forceinline__ void dftr(cx<R>* v) {
    if constexpr (RADIX == 2) dft2(v[0], v[1]);
    if constexpr (RADIX == 4) dft4(v[0], v[1], v[2], v[3]);
    if constexpr (RADIX == 8) dft8(v);
    if constexpr (RADIX == 5) dft5(v);
    if constexpr (RADIX == 20) dft20(v);
}

// ---- plan ---------------------------------------------------------------------------------
template <int N>
struct Plan;
template <>
struct Plan<64> {     // small frames of the FFT convolution (k_conv_fft): 8 threads per line
    static constexpr int NP = 2, TPR = 8, SLOTS = 32;
    static constexpr int radix[5] = {8, 8, 1, 1, 1};
};
template <>
struct Plan<128> {
    static constexpr int NP = 3, TPR = 16, SLOTS = 16;
    static constexpr int radix[5] = {8, 4, 4, 1, 1};
};
template <>
struct Plan<256> {
    static constexpr int NP = 3, TPR = 32, SLOTS = 8;
    static constexpr int radix[5] = {8, 8, 4, 1, 1};
};
#ifndef MPSFR_SLOTS512
#define MPSFR_SLOTS512 4
#endif
template <>
struct Plan<512> {
    static constexpr int NP = 3, TPR = 64, SLOTS = MPSFR_SLOTS512;
    static constexpr int radix[5] = {8, 8, 8, 1, 1};
};
template <>
struct Plan<1024> {
    static constexpr int NP = 4, TPR = 128, SLOTS = 2;
    static constexpr int radix[5] = {8, 8, 4, 4, 1};
};
template <>
struct Plan<1280> {
    static constexpr int NP = 4, TPR = 64, SLOTS = 2;
    static constexpr int radix[5] = {4, 4, 4, 20, 1};
};

template <int N>
struct LineCfg {
    static constexpr int TPR = Plan<N>::TPR;
    static constexpr int SLOTS = Plan<N>::SLOTS;
    static constexpr int THREADS = TPR * SLOTS;
    static constexpr int NPAD = N + N / 8;          // padded line length in LDS
    static constexpr bool WSYNC = TPR <= 64;        // a slot never spans two wavefronts
};

__device__ __forceinline__ int lds_pad(int x) { return x + (x >> 3); }

// LDS image of a line after pass B of the plan (B = -1: a line staged by the caller, padded).
// A pass reads the image of the pass before and writes its own, in place, so the layout may
// change from pass to pass.  Every map is x plus multiples of floor(x / 2^s) chosen so that the
// addresses of a pass stay "lane base + immediate offset".  The table below is the minimum of the
// bank model of MI355X_MICROARCH.md (ds_write_b64: 16-lane groups over 32 banks, ds_read_b64:
// 32-lane groups over 64 banks, the b128 forms for fp64) over that family, per plan, pass and
// element size ES; e.g. N = 512 fp32, LDS-array cycles per line (writes + reads of the next pass):
//   after pass 0: x + x/8 32 + 32;  after pass 1: x + 8 (x/64) 32 + 16 (padded 32 + 32);
//   after pass 2: identity 32 (padded 64).  1280-point fp32 lines go from 880 to 560 cycles.
template <int N, int B, int ES>
__device__ __forceinline__ int lds_map(int x) {
    if constexpr (B <= 0) {
        // first image: padded, except after a radix-4 first pass of 8-byte elements
        if constexpr (B == 0 && ES == 8 && (N == 1280 || N == 128)) return x + (x >> 4);
        return lds_pad(x);
    } else if constexpr (ES == 8 && B == 1 && N == 512) {
        return x + ((x >> 6) << 3);
    } else if constexpr (ES == 8 && B == 1 && (N == 256 || N == 1024 || N == 1280)) {
        return x + ((x >> 5) << 2);
    } else {
        return x;
    }
}

// index of element x of a finished transform
template <int N, int ES>
__device__ __forceinline__ int lds_out(int x) { return lds_map<N, Plan<N>::NP - 1, ES>(x); }

constexpr int plan_ns(const int* radix, int p) {
    int ns = 1;
    for (int i = 0; i < p; ++i) ns *= radix[i];
    return ns;
}

// number of twiddle factors one thread needs over all passes after the first
template <int N>
constexpr int tw_count() {
    int n = 0;
    for (int p = 1; p < Plan<N>::NP; ++p)
        n += (N / Plan<N>::radix[p] / Plan<N>::TPR) * (Plan<N>::radix[p] - 1);
    return n;
}

// Synchronisation between passes: workgroup barrier, or (slot within one wave) a wave-scope fence.
template <bool WSYNC>
__device__ __forceinline__ void fft_sync() {
    if constexpr (WSYNC) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// Per-thread twiddles of every pass, kept in registers.  twg[m] = exp(-2 pi i m / N) (global).
template <typename R, int N>
struct TwRegs {
    cx<R> w[tw_count<N>() > 0 ? tw_count<N>() : 1];
    template <typename RT>
    __device__ __forceinline__ void init(const cx<RT>* __restrict__ twg, int t) {
        using P = Plan<N>;
        int o = 0;
#pragma unroll
        for (int p = 1; p < P::NP; ++p) {
            const int RADIX = P::radix[p];
            const int NS_ = plan_ns(P::radix, p);
            const int NB = N / RADIX, TS = N / (NS_ * RADIX);
#pragma unroll
            for (int b = 0; b < NB / P::TPR; ++b) {
                const int k = (t + b * P::TPR) % NS_;
#pragma unroll
                for (int q = 1; q < RADIX; ++q) {
                    const cx<RT> v = twg[q * k * TS];
                    w[o++] = {(R)v.x, (R)v.y};
                }
            }
        }
    }
};

// One Stockham autosort pass p of the plan (padded LDS images).
//  REGTW  : `tw` is the thread's TwRegs array; otherwise the LDS table
//           tw[lds_pad(m)] = exp(-2 pi i m / N) (padded like the lines: the lanes of a pass read
//           it with strides q * TS).
//  FROMREG: pass 0 only -- the inputs are the thread's own line elements x[e] = line[t + e*TPR]
//           (a thread's first-pass butterflies read exactly the elements it owns), so the line
//           never has to be staged through LDS.
// All inputs of the thread are gathered before anything is written, so `in` may equal `out`
// when the slot is a single wavefront (in-order LDS) -- see fft_forward.
// TOREG: the pass writes nothing to LDS; its results stay in `out`, used as a register array:
//        out[b * RADIX + q] = element (t + b TPR) % NS + ((t + b TPR) - (t + b TPR) % NS) RADIX + q NS
//        of the pass's image -- for the LAST pass of a plan that is element t + b TPR + q N / RADIX of
//        the transform: for every q the lanes of a slot hold consecutive elements.
template <typename R, int N, int P_, bool REGTW, bool FROMREG, bool TOREG = false>
__device__ __forceinline__ void fft_pass(const cx<R>* in, cx<R>* out, const cx<R>* tw, int t) {
    using P = Plan<N>;
    constexpr int RADIX = P::radix[P_];
    constexpr int NS_ = plan_ns(P::radix, P_);
    constexpr int NB = N / RADIX;
    static_assert(NB % P::TPR == 0, "plan must give every thread the same number of butterflies");
    constexpr int NBT = NB / P::TPR;
    static_assert(!FROMREG || P_ == 0, "register input only for the first pass");
    constexpr int WOFF = [] {
        int n = 0;
        for (int p = 1; p < P_; ++p) n += (N / P::radix[p] / P::TPR) * (P::radix[p] - 1);
        return n;
    }();
    cx<R> v[NBT][RADIX];
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        const int j = t + b * P::TPR;
#pragma unroll
        for (int q = 0; q < RADIX; ++q) {
            if constexpr (FROMREG)
                v[b][q] = in[b + q * NBT];
            else
                v[b][q] = in[lds_map<N, P_ - 1, sizeof(cx<R>)>(j + q * NB)];
        }
    }
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        const int j = t + b * P::TPR;
        const int k = j % NS_;
        if constexpr (P_ > 0) {
            if constexpr (REGTW) {
#pragma unroll
                for (int q = 1; q < RADIX; ++q)
                    v[b][q] = cmul(v[b][q], tw[WOFF + b * (RADIX - 1) + q - 1]);
            } else {
                constexpr int TS = N / (NS_ * RADIX);
#pragma unroll
                for (int q = 1; q < RADIX; ++q)
                    v[b][q] = cmul(v[b][q], tw[lds_pad(q * k * TS)]);
            }
        }
        dftr<R, RADIX>(v[b]);
    }
    if constexpr (TOREG) {
#pragma unroll
        for (int b = 0; b < NBT; ++b)
#pragma unroll
            for (int q = 0; q < RADIX; ++q) out[b * RADIX + q] = v[b][q];
        return;
    }
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        const int j = t + b * P::TPR;
        const int k = j % NS_;
        const int base = (j - k) * RADIX + k;
#pragma unroll
        for (int q = 0; q < RADIX; ++q)
            out[lds_map<N, P_, sizeof(cx<R>)>(base + q * NS_)] = v[b][q];
    }
}

template <typename R, int N, bool REGTW, int P_>
__device__ __forceinline__ cx<R>* fft_rest(cx<R>* cur, cx<R>* other, const cx<R>* tw, int t) {
    using P = Plan<N>;
    constexpr bool WS = LineCfg<N>::WSYNC;
    if constexpr (P_ >= P::NP) {
        return cur;
    } else {
        cx<R>* dst = WS ? cur : other;       // single wave per slot: in place
        fft_pass<R, N, P_, REGTW, false>(cur, dst, tw, t);
        fft_sync<WS>();
        return fft_rest<R, N, REGTW, P_ + 1>(dst, WS ? other : cur, tw, t);
    }
}

// Forward FFT of one line whose elements are in registers: x[e] = line[t + e*TPR].
// `a` (and `b` when a slot spans two wavefronts) are padded LDS buffers of NPAD elements.
// Every thread of the slot (of the workgroup if !WSYNC) must call this.  No synchronisation is
// needed before the call beyond "nobody still reads a"; on return the result (natural order,
// index with lds_out<N, ES>) is visible to the slot.  Returns the buffer holding it.
template <typename R, int N, bool REGTW>
__device__ __forceinline__ cx<R>* fft_forward_regs(const cx<R>* x, cx<R>* a, cx<R>* b,
                                                   const cx<R>* tw, int t) {
    fft_pass<R, N, 0, REGTW, true>(x, a, tw, t);
    fft_sync<LineCfg<N>::WSYNC>();
    return fft_rest<R, N, REGTW, 1>(a, b, tw, t);
}

// Same with the line staged in LDS buffer `a` (natural order, padded); the caller must have made
// it visible (fft_sync) before the call.
template <typename R, int N, bool REGTW>
__device__ __forceinline__ cx<R>* fft_forward(cx<R>* a, cx<R>* b, const cx<R>* tw, int t) {
    return fft_rest<R, N, REGTW, 0>(a, b, tw, t);
}

// Passes P_ .. NP-2 through LDS, the last one into registers (fft_pass, TOREG): for a consumer that
// reads exactly the elements a thread produces, the last image never goes through LDS.
template <typename R, int N, bool REGTW, int P_>
__device__ __forceinline__ void fft_rest_lastreg(cx<R>* cur, cx<R>* other, const cx<R>* tw, int t, cx<R>* vout) {
    using P = Plan<N>;
    constexpr bool WS = LineCfg<N>::WSYNC;
    if constexpr (P_ == P::NP - 1) {
        fft_pass<R, N, P_, REGTW, false, true>(cur, vout, tw, t);
    } else {
        cx<R>* dst = WS ? cur : other;
        fft_pass<R, N, P_, REGTW, false>(cur, dst, tw, t);
        fft_sync<WS>();
        fft_rest_lastreg<R, N, REGTW, P_ + 1>(dst, WS ? other : cur, tw, t, vout);
    }
}
// elements per thread of the last pass's register image, and the transform index of entry (b, q)
template <int N>
constexpr int fft_last_radix() { return Plan<N>::radix[Plan<N>::NP - 1]; }
template <int N>
constexpr int fft_last_nbt() { return N / fft_last_radix<N>() / Plan<N>::TPR; }
// Line staged in LDS buffer `a` (as for fft_forward) -> vout[fft_last_nbt * fft_last_radix]:
// vout[b * RADIX + q] = X[t + b TPR + q N / RADIX].
template <typename R, int N, bool REGTW>
__device__ __forceinline__ void fft_forward_lastreg(cx<R>* a, cx<R>* b, const cx<R>* tw, int t, cx<R>* vout) {
    fft_rest_lastreg<R, N, REGTW, 0>(a, b, tw, t, vout);
}

// The same for a line whose elements are in registers (x[e] = line[t + e TPR], as for
// fft_forward_regs).
template <typename R, int N, bool REGTW>
__device__ __forceinline__ void fft_forward_regs_lastreg(const cx<R>* x, cx<R>* a, cx<R>* b, const cx<R>* tw,
                                                         int t, cx<R>* vout) {
    static_assert(Plan<N>::NP >= 2, "needs a pass through LDS before the last one");
    fft_pass<R, N, 0, REGTW, true>(x, a, tw, t);
    fft_sync<LineCfg<N>::WSYNC>();
    fft_rest_lastreg<R, N, REGTW, 1>(a, b, tw, t, vout);
}

// LDS buffers per slot: one when a slot is a single wavefront (in-place passes), else two
template <int N>
constexpr int fft_nbuf() { return LineCfg<N>::WSYNC ? 1 : 2; }

// whether a kernel that runs many transforms per thread should keep its twiddles in registers
template <int N>
constexpr bool use_reg_twiddles() { return tw_count<N>() <= 20; }

}  // namespace mpsfr
