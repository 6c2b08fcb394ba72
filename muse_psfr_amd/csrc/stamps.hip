// HIP kernels of the muse-psfr PSF-reconstruction hot path, written for gfx950 (MI355X, wave64).
// Reference citations are to /root/reference/muse_psfr/psfrec.py.  See DESIGN.md for the data
// layout and the derivation of the restructured algorithm.
//
// Stamp stage: Moffat kernels, the two convolutions (direct / 64-point FFT), Moffat fit, stamp sum.
#include "device_common.h"
#include "conv_frames.h"
#include "mf_common.h"

namespace mpsfr {

namespace {

// ------------------------------------------------------------------------------------------
// K_MOFFAT_KERNELS: astropy Moffat2DKernel(gamma, alpha, 41, 41) (psfrec.py:916, 927):
// (1 + r^2/gamma^2)^-alpha at integer offsets, normalised to sum 1.
// ------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(256)
k_moffat_kernels(const double* __restrict__ gam, const double* __restrict__ alp,
                 R* __restrict__ out) {
    __shared__ double part[4];
    __shared__ double tot;
    const int k = blockIdx.x;
    const double g2 = gam[k] * gam[k], al = alp[k];
    double vals[(KS * KS + 255) / 256];
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < (KS * KS + 255) / 256; ++m) {
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
    const double inv = 1.0 / tot;
#pragma unroll
    for (int m = 0; m < (KS * KS + 255) / 256; ++m) {
        const int e = threadIdx.x + m * 256;
        if (e < KS * KS) out[(size_t)k * KS * KS + e] = (R)(vals[m] * inv);
    }
}

// ------------------------------------------------------------------------------------------
// K_CONV: convolve_final_psf (psfrec.py:874-930): two zero-padded 'same' convolutions with
// 41x41 Moffat kernels (tip-tilt kernel of the task, instrument kernel of the wavelength).
// One workgroup per stamp.  The stamp sits at offset 20 inside an 80-row zero frame in LDS, so
// the tap loops need no bounds checks.  Each thread owns a 1x8 strip of outputs (200 strips, 50
// per wave); the frame pitch of 84 words and the strip map make the 16-byte row reads of a wave
// conflict-free.  The kernel is symmetric in its
// row index (K[a][b] = K[40-a][b]), so image rows i-a and i+a are added first and the tap
// count halves.  The taps K[a][b] are wave-uniform and arrive through scalar loads, so the
// inner loop is pure FMA with an SGPR operand.  Each 41-tap row sum is accumulated in R and
// then added to an fp64 accumulator.
// ------------------------------------------------------------------------------------------
template <typename R>
struct Vec4;
template <>
struct Vec4<float> { using type = float4; };
template <>
struct Vec4<double> { using type = double4; };

template <typename R>
__global__ void __launch_bounds__(256)
k_conv(int nl, const R* __restrict__ pre, const R* __restrict__ ktt,
       const R* __restrict__ kmuse, double* __restrict__ fin) {
    constexpr int PH = NS + KS - 1;   // 80 frame rows
    constexpr int PW = 84;            // frame pitch (80 used)
    constexpr int HK = KS / 2;        // 20
    constexpr int SW = 8;             // strip width
    constexpr int LW = SW + KS - 1;   // 48 frame columns feed one strip
    constexpr int NSTRIP = NS * NS / SW;
    using V4 = typename Vec4<R>::type;
    extern __shared__ __align__(32) unsigned char smem[];
    R* img = reinterpret_cast<R*>(smem);   // [PH][PW]
    const int l = blockIdx.x, task = blockIdx.y;
    const R* src = pre + ((size_t)task * nl + l) * NS * NS;
    for (int e = threadIdx.x; e < PH * PW; e += 256) {
        const int P = e / PW - HK, Q = e % PW - HK;
        img[e] = (P >= 0 && P < NS && Q >= 0 && Q < NS) ? (R)src[P * NS + Q] : (R)0;
    }
    // strip map: wave w owns output rows 10w .. 10w+9, lane l < 50 -> row 10w + l%10, column
    // block l/10.  With the 84-word pitch every 16-lane group of a ds_read_b128 then touches 16
    // distinct 4-bank slots (brute-forced against the gfx950 lane groups; the column-block-major
    // map cost 1.43x, SQ_LDS_BANK_CONFLICT = 53 % of LDS cycles).
    static_assert(NSTRIP == 200 && NS == 40, "strip map assumes 40x40 stamps");
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const bool active = ln < 50;
    const int i = 10 * wv + (active ? ln % 10 : 0), j0 = (active ? ln / 10 : 0) * SW;
    double acc[SW];
    for (int pass = 0; pass < 2; ++pass) {
        const R* __restrict__ kg = pass == 0 ? ktt + (size_t)task * KS * KS
                                             : kmuse + (size_t)l * KS * KS;
        __syncthreads();
#pragma unroll
        for (int o = 0; o < SW; ++o) acc[o] = 0.0;
        for (int a = 0; a <= HK; ++a) {
            // frame rows i-a+40 and (a < 20) i+a, columns j0 .. j0+47
            const V4* r1 = reinterpret_cast<const V4*>(
                __builtin_assume_aligned(img + (i - a + 2 * HK) * PW + j0, sizeof(V4)));
            const V4* r2 = reinterpret_cast<const V4*>(
                __builtin_assume_aligned(img + (i + a) * PW + j0, sizeof(V4)));
            R sv[LW];
#pragma unroll
            for (int c = 0; c < LW / 4; ++c) {
                const V4 q = r1[c];
                sv[4 * c] = q.x; sv[4 * c + 1] = q.y; sv[4 * c + 2] = q.z; sv[4 * c + 3] = q.w;
            }
            if (a < HK) {
#pragma unroll
                for (int c = 0; c < LW / 4; ++c) {
                    const V4 q = r2[c];
                    sv[4 * c] += q.x; sv[4 * c + 1] += q.y; sv[4 * c + 2] += q.z; sv[4 * c + 3] += q.w;
                }
            }
            R part[SW];
#pragma unroll
            for (int o = 0; o < SW; ++o) part[o] = (R)0;
            const R* __restrict__ krow = kg + a * KS;
#pragma unroll
            for (int b = 0; b < KS; ++b) {
                const R kv = krow[b];
#pragma unroll
                for (int o = 0; o < SW; ++o) part[o] += kv * sv[o - b + 2 * HK];
            }
#pragma unroll
            for (int o = 0; o < SW; ++o) acc[o] += (double)part[o];
        }
        __syncthreads();
        if (pass == 0 && active) {
#pragma unroll
            for (int o = 0; o < SW; ++o) img[(i + HK) * PW + j0 + HK + o] = (R)acc[o];
        }
    }
    if (active) {
        double* out = fin + ((size_t)task * nl + l) * NS * NS + i * NS + j0;
#pragma unroll
        for (int o = 0; o < SW; ++o) out[o] = acc[o];
    }
}

// ------------------------------------------------------------------------------------------
// K_CONV (mixed mode): the same two 'same' convolutions through 64-point FFTs in LDS.
// The outputs needed are indices [20, 60) of the 80-long linear convolution, which a circular
// convolution of length 64 leaves un-aliased, so 64x64 frames suffice (scipy pads to 80,
// psfrec.py:917).  Per convolution and stamp: 20 row-pair transforms (two real rows per complex
// FFT), 32 column transforms forth and back with the kernel spectrum multiplied in between (the
// real DC and Nyquist columns share one complex column), 20 row-pair inverse transforms --
// ~7x fewer instructions than the direct form.  One workgroup per stamp, 32 lines of 8 threads.
// khat[k][kx], k in [0, 33), kx in [0, 64): kernel spectrum (1/4096 folded in), from K_KHAT.
// ------------------------------------------------------------------------------------------
// TF: type of the final stamps -- float inside the pipeline (their values are float anyway: the fit
// and the stamp sum read half the bytes), double when they go straight into the caller's psf_out.
// R: arithmetic type (float: mixed mode; double: f64 mode, 74 KB of LDS).
template <typename R, typename TF>
#ifndef MPSFR_CONV_WAVES
#define MPSFR_CONV_WAVES 4
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(R) == 4 ? MPSFR_CONV_WAVES : 2)))
k_conv_fft(int nl, const R* __restrict__ pre, const cx<R>* __restrict__ khat_tt,
           const cx<R>* __restrict__ khat_muse, TF* __restrict__ fin, MfFinish mf) {
    extern __shared__ __align__(16) unsigned char conv_smem[];
    cx<R> (*F)[CFP] = reinterpret_cast<cx<R> (*)[CFP]>(conv_smem);
    cx<R> (*bufs)[CFB] = reinterpret_cast<cx<R> (*)[CFB]>(conv_smem + sizeof(cx<R>) * CF * CFP);
    const int l = blockIdx.x, task = blockIdx.y;
    const int slot = threadIdx.x >> 3, t = threadIdx.x & 7;
    Tw64<R> tw;
    tw.init(t);
    const R* src = pre + ((size_t)task * nl + l) * NS * NS;
    cx<R>* buf = bufs[slot];
    // The image never sits in LDS: a slot reads its row pair of the input from global memory, and
    // the rows it produces in the first convolution are exactly the ones it transforms in the
    // second (taken from its own line buffer).
    // Round 6 -- unless the stamp is one K_OTF_MFMA2 left as the partial tiles of several sweeps: then the first wave
    // adds them up and runs that kernel's epilogue into LDS (the frame F is free until the first barrier of the
    // pass loop), and the slots take their rows from there.  K_MF_FINISH, which did this through `pre` as a launch
    // of its own between the two kernels, was 7 us of a call's chain alone and 11-14 beside the other lane.
    cx<R> x[8];
    bool from_lds = false;
    if constexpr (sizeof(R) == 4) {
        if (mf.gsw != nullptr) {
            const int m = mf.gsw[task * mf.ngr + l / mf.per];
            if (__builtin_popcount(m) > 1) {           // (uniform over the workgroup)
                float* st = reinterpret_cast<float*>(conv_smem);
                if (threadIdx.x < 64) finish_stamp(mf, m, task, nl, l, (int)threadIdx.x, st);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = t + 8 * e;
                    const bool in = c < NS && slot < NS / 2;
                    x[e] = {in ? st[2 * slot * NS + c] : 0.f, in ? st[(2 * slot + 1) * NS + c] : 0.f};
                }
                from_lds = true;
            }
        }
    }
    if (!from_lds) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = t + 8 * e;
            const bool in = c < NS && slot < NS / 2;
            x[e] = {in ? src[2 * slot * NS + c] : (R)0, in ? src[(2 * slot + 1) * NS + c] : (R)0};
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        const cx<R>* __restrict__ kh = pass == 0 ? khat_tt + (size_t)task * (CFH + 1) * CF
                                                     : khat_muse + (size_t)l * (CFH + 1) * CF;
        // the thread's kernel-spectrum values, fetched now so that the row transforms hide the
        // latency (slot 0 also carries the Nyquist column)
        cx<R> khv[8], khn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            khv[e] = kh[slot * CF + t + 8 * e];
            khn[e] = slot == 0 ? kh[CFH * CF + t + 8 * e] : cx<R>{(R)0, (R)0};
        }
        __syncthreads();        // F is free: the inverse rows of the previous pass have read it
        if (slot < NS / 2) cf_rows_forward(x, F, buf, tw, slot, t);
        __syncthreads();
        {   // columns: forward, multiply by the kernel spectrum, inverse (conjugation trick).  A thread
            // consumes exactly the elements t + 8 e its last pass produces, so the results of the
            // transforms stay in registers (fft_pass, TOREG) -- except in slot 0, whose packed DC /
            // Nyquist column needs the mirrored elements of the forward transform.
            cx<R> y[8];
            if (slot == 0) {
                const cx<R>* res = cf_col_forward(F, NS, buf, tw, slot, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    cx<R> a0, a32;
                    cf_split0(res, t + 8 * e, a0, a32);
                    const cx<R> b0 = cmul(a0, khv[e]), b32 = cmul(a32, khn[e]);
                    y[e] = conjf<R>(cx<R>{b0.x - b32.y, b0.y + b32.x});
                }
                fft_sync<true>();           // the line buffer is rewritten by the inverse transform
            } else {
                cx<R> xin[8], v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = t + 8 * e;
                    xin[e] = r < NS ? F[r][slot] : cx<R>{(R)0, (R)0};
                }
                fft_forward_regs_lastreg<R, CF, true>(xin, buf, buf, tw.w, t, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = conjf<R>(cmul(v[e], khv[e]));
                fft_sync<true>();
            }
            cx<R> r2[8];
            fft_forward_regs_lastreg<R, CF, true>(y, buf, buf, tw.w, t, r2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = t + 8 * e;
                if (r >= KS / 2 && r < KS / 2 + NS) F[r][slot] = conjf<R>(r2[e]);
            }
        }
        __syncthreads();
        if (slot < NS / 2) {   // inverse rows, two output rows per complex transform
            const int ra = KS / 2 + 2 * slot, rb = ra + 1;
            cx<R> z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = t + 8 * e;
                const int kk = k <= CFH ? k : CF - k;
                cx<R> a, b;
                if (kk == 0) {
                    a = {F[ra][0].x, (R)0};
                    b = {F[rb][0].x, (R)0};
                } else if (kk == CFH) {
                    a = {F[ra][0].y, (R)0};
                    b = {F[rb][0].y, (R)0};
                } else {
                    a = F[ra][kk];
                    b = F[rb][kk];
                    if (k > CFH) { a = conjf<R>(a); b = conjf<R>(b); }
                }
                z[e] = {a.x - b.y, -(a.y + b.x)};      // conj(A + iB)
            }
            // y_a = Re conj(res) = res.x, y_b = Im conj(res) = -res.y at columns [20, 60)
            if (pass == 1) {    // the final stamp: a thread writes the columns its last pass produced
                cx<R> v[8];
                fft_forward_regs_lastreg<R, CF, true>(z, buf, buf, tw.w, t, v);
                TF* out = fin + ((size_t)task * nl + l) * NS * NS;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = t + 8 * e;
                    if (c >= KS / 2 && c < KS / 2 + NS) {
                        const int i = 2 * slot, j = c - KS / 2;
                        out[i * NS + j] = (TF)v[e].x;
                        out[(i + 1) * NS + j] = (TF)(-v[e].y);
                    }
                }
            } else {            // rows 2 slot, 2 slot + 1 of the intermediate image, columns t + 8 e
                                // (shifted by 20: another thread's results, through LDS)
                const cx<R>* res = fft_forward_regs<R, CF, true>(z, buf, buf, tw.w, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = t + 8 * e;
                    const cx<R> v = res[lds_out<CF, 8>(c < NS ? c + KS / 2 : 0)];
                    x[e] = c < NS ? cx<R>{v.x, -v.y} : cx<R>{(R)0, (R)0};
                }
            }
        }
    }
}

// K_KHAT: see khat_body (conv_frames.h)
template <typename R>
__global__ void __launch_bounds__(256)
k_khat(const double* __restrict__ gam, const double* __restrict__ alp, cx<R>* __restrict__ khat) {
    extern __shared__ __align__(16) unsigned char conv_smem[];
    khat_body<R>(gam, alp, khat, conv_smem, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// K_FIT: 5-parameter circular Moffat least-squares fit per stamp (fit_psf_cube psfrec.py:861-871
// -> mpdaf Image.moffat_fit(circular=True, fit_back=False)): I (1 + ((p-p0)^2+(q-q0)^2)/a^2)^-n,
// unweighted, all 1600 pixels.  One wavefront per stamp (25 pixels per lane), the stamp in LDS.
// Levenberg-Marquardt (Marquardt scaling, Nielsen's gain-ratio damping update) iterated in the
// better-conditioned variables (I, p0, q0, w = FWHM, n) -- the minimum is the same point.
// Per-lane sums run in the evaluation type RE (float in mixed mode, double in f64 mode); the
// 5x5 solves of the float phase run in float on the Marquardt-scaled matrix.
// ------------------------------------------------------------------------------------------

template <typename T>
struct NormEqT {
    T a[15];   // upper triangle of J^T J, row-major: (0,0)(0,1)..(0,4)(1,1)..(4,4)
    T g[5];    // J^T r
    T chi2;
};
using NormEq = NormEqT<double>;

// Wave-wide sums for the fit, on the DPP path instead of ds_bpermute shuffles (21 sums per model
// evaluation; a shuffle goes through the LDS crossbar and its latency sat on the critical path of
// the serial LM iterations).  Quad swaps, half-row and row mirrors give every lane its row-of-16
// sum; row_bcast15 / row_bcast31 fold the four rows into lane 63, which is read into a scalar
// register -- the LM state is wave-uniform and lives in SGPRs.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                         0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_term(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL,
                                                              ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK,
                                                              0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
__device__ __forceinline__ float lane63(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ double lane63(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
template <typename T>
__device__ __forceinline__ T wave_total(T v) {
    v += dpp_term<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v += dpp_term<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v += dpp_term<0x141, 0xf>(v);    // row_half_mirror
    v += dpp_term<0x140, 0xf>(v);    // row_mirror: every lane holds its row's sum
    // rows 1..3 add lane 15 of the row before, then rows 2, 3 add lane 31: lane 63 = s2 + s3 + (s0 + s1).
    // All rows enabled (the other lanes hold partial sums nobody reads): with a row mask the
    // compiler cannot fuse the move into the addition and each step is three instructions, not one
    v += dpp_term<0x142, 0xf>(v);    // row_bcast15
    v += dpp_term<0x143, 0xf>(v);    // row_bcast31: lane 63 holds the total
    return lane63(v);
}

template <typename RE>
__device__ __forceinline__ RE fit_log(RE x);
template <>
__device__ __forceinline__ float fit_log<float>(float x) { return __logf(x); }
template <>
__device__ __forceinline__ double fit_log<double>(double x) { return log(x); }
template <typename RE>
__device__ __forceinline__ RE fit_exp(RE x);
template <>
__device__ __forceinline__ float fit_exp<float>(float x) { return __expf(x); }
template <>
__device__ __forceinline__ double fit_exp<double>(double x) { return exp(x); }

template <typename S>
__device__ __forceinline__ S fit_rcp(S x);
template <>
__device__ __forceinline__ float fit_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <>
__device__ __forceinline__ double fit_rcp<double>(double x) { return 1.0 / x; }
template <typename S>
__device__ __forceinline__ S fit_rsqrt(S x);
template <>
__device__ __forceinline__ float fit_rsqrt<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <>
__device__ __forceinline__ double fit_rsqrt<double>(double x) {
    // hardware seed (v_rsq_f64, ~2^-26) + two Newton steps: 9 instructions where sqrt and the
    // division took ~40; the ten of a 5 x 5 factorisation were most of its cost
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}

// Lean fp64 exp / log for the polish (arguments are tame: z <= 0 for the model, x >= 1 for the
// logarithm), ~18 and ~27 instructions against ~55 and ~65 for the general library routines.
//   exp: z = k ln2 + r, |r| <= ln2 / 2, degree-10 Taylor in r (remainder r^11 / 11! < 2.2e-13 --
//        the residuals it serves only have to beat the 1e-6 of the float model), v_ldexp_f64.
//   log: l0 = hardware log2 in fp32 (error ~1e-7), then log x = l0 + log1p(d) with
//        d = x exp(-l0) - 1 ~ 1e-7, three terms of the series (remainder d^4 / 4).
__device__ __forceinline__ double lean_exp(double z) {
    z = fmax(z, -700.0);
    const double k = rint(z * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, z);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 3628800.0;
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
__device__ __forceinline__ double lean_log(double x) {
    const double l0 = (double)(__builtin_amdgcn_logf((float)x) * 0.69314718f);
    const double d = fma(x, lean_exp(-l0), -1.0);
    return l0 + d * fma(d, fma(d, 1.0 / 3.0, -0.5), 1.0);
}

// Normal equations of the Moffat model at v = (I, p0, q0, w, eta), eta = 1/n,
// 1/a^2 = 4 (2^eta - 1) / w^2,
// over the lane's pixels o = lane + 64 m of the stamp `pix` (LDS), summed over the wave.
// Cross-lane sums run in the evaluation type: the float phase only has to reach the basin of
// convergence (tol 1e-3); the f64 mode reduces in double.  Every lane ends up with the totals.
template <typename RE>
__device__ __forceinline__ RE fit_exp2m1(RE eta);           // 2^eta - 1, eta in (0, 100)
template <>
__device__ __forceinline__ float fit_exp2m1<float>(float eta) {
    return __builtin_amdgcn_exp2f(eta) - 1.0f;
}
template <>
__device__ __forceinline__ double fit_exp2m1<double>(double eta) {
    return lean_exp(0.69314718055994530942 * eta) - 1.0;
}

// log2 / exp2 of the model passes.  Float: the bare hardware instructions -- the argument of the
// logarithm is >= 1 and the power is in (0, 1], so the denormal scaling and the extended-precision
// ln of __logf (11 instructions per pixel of the 56 the pass had) buy nothing, and the float phase
// only has to reach the basin of the fp64 polish.
template <typename RE>
__device__ __forceinline__ RE fit_log2(RE x);
template <>
__device__ __forceinline__ float fit_log2<float>(float x) { return __builtin_amdgcn_logf(x); }
template <>
__device__ __forceinline__ double fit_log2<double>(double x) { return log(x) * 1.4426950408889634074; }
template <typename RE>
__device__ __forceinline__ RE fit_exp2(RE x);
template <>
__device__ __forceinline__ float fit_exp2<float>(float x) { return __builtin_amdgcn_exp2f(x); }
template <>
__device__ __forceinline__ double fit_exp2<double>(double x) { return exp(x * 0.69314718055994530942); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Accumulators of the normal equations over a lane's pixels.  Generic form: 15 + 5 + 1 multiply-adds
// per pixel.  Float form: the same 21 sums as 9 packed multiply-adds (v_pk_fma_f32: two fp32 lanes
// per instruction at the issue cost of one) + 3 plain ones -- with r as a sixth "column" the rows
// of the upper triangle are  J_x * (J_x .. J_4, r), and the pairs (J0,J1) (J2,J3) (J4,r) sit in
// aligned register pairs; the factor J_x is a half of one of those pairs, picked by op_sel.
template <typename RE>
struct NormAcc {
    RE a[15], g[5], chi2;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 15; ++k) a[k] = (RE)0;
#pragma unroll
        for (int k = 0; k < 5; ++k) g[k] = (RE)0;
        chi2 = (RE)0;
    }
    __device__ __forceinline__ void add(const RE* J, RE r) {
        chi2 += r * r;
        int k = 0;
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            g[x] += J[x] * r;
#pragma unroll
            for (int y = x; y < 5; ++y) a[k++] += J[x] * J[y];
        }
    }
    template <typename F>
    __device__ __forceinline__ void totals(NormEqT<RE>& ne, F total) const {
        ne.chi2 = total(chi2);
#pragma unroll
        for (int k = 0; k < 15; ++k) ne.a[k] = total(a[k]);
#pragma unroll
        for (int k = 0; k < 5; ++k) ne.g[k] = total(g[k]);
    }
};
template <>
struct NormAcc<float> {
    f32x2 r0a, r0b, r0c;      // (a00,a01) (a02,a03) (a04,g0)
    f32x2 r1b, r1c;           // (a12,a13) (a14,g1)
    f32x2 r2b, r2c;           // (a22,a23) (a24,g2)
    f32x2 r3c, r4c;           // (a34,g3)  (a44,g4)
    float a11, a33, chi2;
    __device__ __forceinline__ void zero() {
        const f32x2 z = {0.f, 0.f};
        r0a = r0b = r0c = r1b = r1c = r2b = r2c = r3c = r4c = z;
        a11 = a33 = chi2 = 0.f;
    }
    __device__ __forceinline__ void add(const float* J, float r) {
        const f32x2 p01 = {J[0], J[1]}, p23 = {J[2], J[3]}, p4r = {J[4], r};
        const f32x2 s0 = {J[0], J[0]}, s1 = {J[1], J[1]}, s2 = {J[2], J[2]}, s3 = {J[3], J[3]},
                    s4 = {J[4], J[4]};
        r0a += s0 * p01; r0b += s0 * p23; r0c += s0 * p4r;
        r1b += s1 * p23; r1c += s1 * p4r;
        r2b += s2 * p23; r2c += s2 * p4r;
        r3c += s3 * p4r;
        r4c += s4 * p4r;
        a11 += J[1] * J[1];
        a33 += J[3] * J[3];
        chi2 += r * r;
    }
    template <typename F>
    __device__ __forceinline__ void totals(NormEqT<float>& ne, F total) const {
        ne.chi2 = total(chi2);
        const float a[15] = {r0a.x, r0a.y, r0b.x, r0b.y, r0c.x, a11, r1b.x, r1b.y, r1c.x,
                             r2b.x, r2b.y, r2c.x, a33, r3c.x, r4c.x};
        const float g[5] = {r0c.y, r1c.y, r2c.y, r3c.y, r4c.y};
#pragma unroll
        for (int k = 0; k < 15; ++k) ne.a[k] = total(a[k]);
#pragma unroll
        for (int k = 0; k < 5; ++k) ne.g[k] = total(g[k]);
    }
};

template <typename RE>
__device__ __forceinline__ void moffat_accumulate(const RE* pix, int lane, const RE* v,
                                                  NormEqT<RE>& ne) {
    constexpr int NPX = NS * NS / 64;
    NormAcc<RE> acc;
    acc.zero();
    // wave-uniform scalars in the evaluation type: in the mixed mode the whole LM phase is float
    // (fp64 here put ~40 double instructions and four double divisions on the serial path of
    // every iteration); the fp64 polish of k_fit removes what that costs in accuracy
    const RE n = fit_rcp<RE>(v[4]);                                         // v[4] = eta = 1/n
    const RE s_ = fit_exp2m1<RE>(v[4]);                                     // 2^eta - 1
    const RE i3 = fit_rcp<RE>(v[3]);
    const RE K = (RE)4 * s_ * i3 * i3;
    // d(1/a^2)/d eta / (1/a^2) = s'/s with s' = 2^eta ln2
    const RE dKn = (s_ + (RE)1) * (RE)0.69314718055994530942 * fit_rcp<RE>(s_);
    const RE I = v[0], p0 = v[1], q0 = v[2];
    const RE nsq2 = n * n * (RE)0.69314718055994530942;      // n^2 ln2: the logarithm below is to base 2
    const RE c12 = (RE)2 * n * K, c3 = c12 * i3, c4 = n * K * dKn;
    // Pixel map of the model passes: the lane is a cell (lr, lc) of an 8 x 8 block, and the 25 blocks
    // of the stamp are walked as 5 x 5 -- pixel (8 mo + lr, 8 mi + lc).  The coordinates are then one
    // addition per block row and one per pixel (the row-major map lane + 64 m cost a
    // division and a remainder by 40 per pixel, a fifth of the pass); the LDS reads stay free of bank
    // conflicts (40 lr + lc is distinct modulo 64 over the wave).
    const RE lrf = (RE)(lane >> 3) - p0, lcf = (RE)(lane & 7) - q0;
    const RE* pl = pix + (lane >> 3) * NS + (lane & 7);
    static_assert(NPX == 25 && NS == 40, "5 x 5 blocks of 8 x 8 pixels");
#pragma unroll 1
    for (int mo = 0; mo < 5; ++mo) {
        const RE dp = (RE)(8 * mo) + lrf, dp2 = dp * dp;
#pragma unroll
        for (int mi = 0; mi < 5; ++mi) {
        const RE dq = (RE)(8 * mi) + lcf;
        const RE u = dq * dq + dp2;
        const RE gg = (RE)1 + u * K;
        const RE lg2 = fit_log2<RE>(gg);
        const RE e = fit_exp2<RE>(-n * lg2);
        const RE mo_ = I * e;
        const RE r = mo_ - pl[mo * 8 * NS + mi * 8];
        // with t = model / g:  d/dp0 = 2 n K t dp,  d/dq0 = 2 n K t dq,  d/dw = 2 n K t u / w,
        // d/d eta = n^2 model ln g - n K (s'/s) t u   (the wave-uniform factors folded into c12, c3, c4)
        const RE t = mo_ * fit_rcp<RE>(gg);
        const RE tu = t * u, tc = t * c12;
        RE J[5];
        J[0] = e;
        J[1] = tc * dp;
        J[2] = tc * dq;
        J[3] = tu * c3;
        J[4] = mo_ * (nsq2 * lg2) - tu * c4;
        acc.add(J, r);
        }
    }
    acc.totals(ne, [](RE x) { return wave_total(x); });
}

#ifndef MPSFR_FIT_PAIRS
#define MPSFR_FIT_PAIRS 1
#endif
// The float pass two pixels at a time: every quantity of the pass is a pair (pixel A, pixel B) in an
// aligned register pair, so the arithmetic between the three transcendental instructions of a pixel --
// coordinates, residual, Jacobian, and the 21 sums -- runs in packed instructions (v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32: 42 per pair where the pixel-at-a-time form has 62).  The wave-uniform
// factors are the low halves of pairs picked by op_sel.  Pairs: (mi = 0, 1) and (2, 3) of each of the
// five block rows, then column mi = 4 by rows (0, 1), (2, 3), and its last pixel with an empty partner.
// The accumulators hold one partial sum per half (42 registers instead of 21): the kernel's 156
// registers at three waves per SIMD have the room (k_fit, MPSFR_FIT_WAVES).
__device__ __forceinline__ void moffat_accumulate_pairs(const float* pix, int lane, const float* v,
                                                        NormEqT<float>& ne) {
    f32x2 A[15], G[5], C = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 15; ++k) A[k] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 5; ++k) G[k] = f32x2{0.f, 0.f};
    const float n = fit_rcp<float>(v[4]);
    const float s_ = fit_exp2m1<float>(v[4]);
    const float i3 = fit_rcp<float>(v[3]);
    const float K = 4.f * s_ * i3 * i3;
    const float dKn = (s_ + 1.f) * 0.69314718f * fit_rcp<float>(s_);
    const float I = v[0], p0 = v[1], q0 = v[2];
    const float nsq2 = n * n * 0.69314718f;
    const float c12 = 2.f * n * K, c3 = c12 * i3, c4 = n * K * dKn;
    const f32x2 K2 = {K, K}, I2 = {I, I}, nn2 = {-n, -n}, nsq22 = {nsq2, nsq2}, c122 = {c12, c12},
                c32 = {c3, c3}, c42 = {c4, c4}, one2 = {1.f, 1.f};
    const float lrf = (float)(lane >> 3) - p0, lcf = (float)(lane & 7) - q0;      // pixel map: moffat_accumulate
    const float* pl = pix + (lane >> 3) * NS + (lane & 7);
    auto pair = [&](f32x2 u, f32x2 dpv, f32x2 dqv, f32x2 px, bool last) {
        const f32x2 gg = u * K2 + one2;
        const f32x2 lg2 = {fit_log2<float>(gg.x), fit_log2<float>(gg.y)};
        const f32x2 arg = lg2 * nn2;
        f32x2 e = {fit_exp2<float>(arg.x), fit_exp2<float>(arg.y)};
        if (last) e.y = 0.f;                     // the empty partner: model, residual and Jacobian vanish
        const f32x2 m = e * I2;
        const f32x2 r = m - px;
        const f32x2 rg = {fit_rcp<float>(gg.x), fit_rcp<float>(gg.y)};
        const f32x2 t = m * rg;
        const f32x2 tu = t * u, tc = t * c122;
        f32x2 J[5];
        J[0] = e;
        J[1] = tc * dpv;
        J[2] = tc * dqv;
        J[3] = tu * c32;
        J[4] = m * (lg2 * nsq22) - tu * c42;
        C += r * r;
        int k = 0;
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            G[x] += J[x] * r;
#pragma unroll
            for (int y = x; y < 5; ++y) A[k++] += J[x] * J[y];
        }
    };
    const f32x2 dqa = {lcf, 8.f + lcf}, dqb = {16.f + lcf, 24.f + lcf};
    const f32x2 dqa2 = dqa * dqa, dqb2 = dqb * dqb;
    const float dq4 = 32.f + lcf, dq4s = dq4 * dq4;
#pragma unroll 1
    for (int mo = 0; mo < 5; ++mo) {
        const float dp = (float)(8 * mo) + lrf, dp2 = dp * dp;
        const f32x2 dpv = {dp, dp}, dp2v = {dp2, dp2};
        const float* row = pl + mo * 8 * NS;
        pair(dqa2 + dp2v, dpv, dqa, f32x2{row[0], row[8]}, false);
        pair(dqb2 + dp2v, dpv, dqb, f32x2{row[16], row[24]}, false);
    }
    {
        const f32x2 dq4v = {dq4, dq4}, dq4sv = {dq4s, dq4s};
        const f32x2 dpa = {lrf, 8.f + lrf}, dpb = {16.f + lrf, 24.f + lrf};
        const float dp4 = 32.f + lrf;
        pair(dpa * dpa + dq4sv, dpa, dq4v, f32x2{pl[32], pl[8 * NS + 32]}, false);
        pair(dpb * dpb + dq4sv, dpb, dq4v, f32x2{pl[16 * NS + 32], pl[24 * NS + 32]}, false);
        pair(f32x2{dp4 * dp4 + dq4s, 0.f}, f32x2{dp4, 0.f}, dq4v, f32x2{pl[32 * NS + 32], 0.f}, true);
    }
    ne.chi2 = wave_total(C.x + C.y);
#pragma unroll
    for (int k = 0; k < 15; ++k) ne.a[k] = wave_total(A[k].x + A[k].y);
#pragma unroll
    for (int k = 0; k < 5; ++k) ne.g[k] = wave_total(G[k].x + G[k].y);
}

// the float pass of the LM phase (pairs of pixels) / the generic pass
template <typename RE>
__device__ __forceinline__ void lm_accumulate(const RE* pix, int lane, const RE* v, NormEqT<RE>& ne) {
    if constexpr (sizeof(RE) == 4 && MPSFR_FIT_PAIRS) moffat_accumulate_pairs(pix, lane, v, ne);
    else moffat_accumulate<RE>(pix, lane, v, ne);
}

// chi2 alone at (I, p0, q0, a, n): the residual pass without the Jacobian (a third of the work)
template <typename RE, typename DT>
__device__ __forceinline__ RE moffat_chi2(const DT* pix, int lane, const double* va) {
    const RE I = (RE)va[0], p0 = (RE)va[1], q0 = (RE)va[2], n = (RE)va[4];
    const RE K = (RE)(1.0 / (va[3] * va[3]));
    RE c[5] = {(RE)0, (RE)0, (RE)0, (RE)0, (RE)0};
    const RE lrf = (RE)(lane >> 3) - p0, lcf = (RE)(lane & 7) - q0;       // pixel map: moffat_accumulate
    const DT* pl = pix + (lane >> 3) * NS + (lane & 7);
#pragma unroll 1
    for (int mo = 0; mo < 5; ++mo) {
        const RE dp = (RE)(8 * mo) + lrf;
#pragma unroll
        for (int mi = 0; mi < 5; ++mi) {
            const RE dq = (RE)(8 * mi) + lcf;
            const RE r = I * fit_exp<RE>(-n * fit_log<RE>((RE)1 + (dp * dp + dq * dq) * K)) -
                         (RE)pl[mo * 8 * NS + mi * 8];
            c[mi] += r * r;
        }
    }
    return wave_total(((c[0] + c[1]) + (c[2] + c[3])) + c[4]);
}

// fp64 gradient J^T r of the Moffat model in (I, p0, q0, w, n) over the stamp in memory, for the
// polish of the mixed mode: the fixed point of the iteration is where this vanishes, whatever
// matrix the step is solved with, so the polish keeps the float normal matrix of the last LM
// iteration (relative error ~1e-3: linear convergence at that rate) and only these five sums are
// fp64 -- 10 accumulator registers instead of 42, which is what lets four waves share a SIMD.
__device__ __forceinline__ double sgpr(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}

template <typename TS>
__device__ __forceinline__ void moffat_gradient(const TS* __restrict__ src, int lane,
                                                const double* v, double* gout, double* chi2out) {
    // Only the residual r = model - data needs fp64: a systematic 1e-6 error of the float
    // log/exp model is what biases the fit.  The Jacobian multiplies r, which is ~1e-3 of the
    // peak at the solution, so its float rounding (6e-8, unbiased) moves the fixed point by
    // ~1e-10: J, the products J r and the 25 per-lane partial sums run in float (a third of the
    // fp64 instructions of an all-fp64 gradient).
    const double n = sgpr(1.0 / v[4]);
    const double s = lean_exp(0.69314718055994530942 * v[4]) - 1.0;
    const double K = sgpr(4.0 * s / (v[3] * v[3]));
    const float dKn = (float)sgpr((s + 1.0) * 0.69314718055994530942 / s);
    const float nsq = (float)(n * n);
    const double I = sgpr(v[0]), p0 = sgpr(v[1]), q0 = sgpr(v[2]);
    const float i3 = (float)sgpr(1.0 / v[3]), nf = (float)n, K2 = 2.0f * (float)K, Kf = (float)K;
    float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, c2sum = 0.f;
    const double lrd = (double)(lane >> 3) - p0, lcd = (double)(lane & 7) - q0;     // pixel map: moffat_accumulate
    const TS* pl = src + (lane >> 3) * NS + (lane & 7);
#pragma unroll 1
    for (int mo = 0; mo < 5; ++mo) {
        const double dp = (double)(8 * mo) + lrd, dp2 = dp * dp;
#pragma unroll
        for (int mi = 0; mi < 5; ++mi) {
        const double dq = (double)(8 * mi) + lcd;
        const double u = fma(dq, dq, dp2);
        const double gg = 1.0 + u * K;
        const double lg = lean_log(gg);
        const double e = lean_exp(-n * lg);
        const double mo_ = I * e;
        const float r = (float)(mo_ - (double)pl[mo * 8 * NS + mi * 8]);
        c2sum += r * r;
        const float mof = (float)mo_, uf = (float)u;
        const float cm = mof * nf * __builtin_amdgcn_rcpf((float)gg);
        const float c2 = cm * K2 * r;
        g[0] += (float)e * r;
        g[1] += c2 * (float)dp;
        g[2] += c2 * (float)dq;
        g[3] += c2 * uf * i3;
        g[4] += (nsq * mof * (float)lg - cm * uf * Kf * dKn) * r;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) gout[k] = wave_total((double)g[k]);    // lanes cancel: fp64
    *chi2out = (double)wave_total(c2sum);
}

// Cholesky factor of the Marquardt-scaled normal matrix  A'_ij = A_ij / (d_i d_j) + mu delta_ij,
// d_i = sqrt(A_ii) -- the same system as (A + mu diag A) x = -g, but with a unit diagonal, which
// is what lets the float phase factor it in float.  Fully unrolled: the factor lives in registers
// (dynamic indexing put it in scratch).  Li holds 1 / L_ii.  Returns false if not positive definite.
template <typename S, typename T>
__device__ __forceinline__ bool chol5(const NormEqT<T>& ne, S mu, S L[5][5], S Li[5], S id[5]) {
    S A[5][5];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i; j < 5; ++j) {
                A[i][j] = (S)ne.a[k];
                A[j][i] = (S)ne.a[k];
                ++k;
            }
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        ok = ok && (A[i][i] > (S)0);
        id[i] = fit_rsqrt<S>(A[i][i]);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) L[i][j] = A[i][j] * id[i] * id[j];
#pragma unroll
    for (int i = 0; i < 5; ++i) L[i][i] = (S)1 + mu;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        S s = L[j][j];
#pragma unroll
        for (int q = 0; q < j; ++q) s -= L[j][q] * L[j][q];
        ok = ok && (s > (S)0);
        Li[j] = fit_rsqrt<S>(s);
        L[j][j] = s * Li[j];
#pragma unroll
        for (int i = j + 1; i < 5; ++i) {
            S t = L[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) t -= L[i][q] * L[j][q];
            L[i][j] = t * Li[j];
        }
    }
    return ok;
}

// x = A^-1 b through the factor of chol5 (b and x in unscaled units)
template <typename S, typename X>
__device__ __forceinline__ void chol5_solve(const S L[5][5], const S Li[5], const S id[5],
                                            const S b[5], X* x) {
    S y[5], z[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        S t = b[i] * id[i];
#pragma unroll
        for (int q = 0; q < i; ++q) t -= L[i][q] * y[q];
        y[i] = t * Li[i];
    }
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        S t = y[i];
#pragma unroll
        for (int q = i + 1; q < 5; ++q) t -= L[q][i] * z[q];
        z[i] = t * Li[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) x[i] = (X)(z[i] * id[i]);
}

// solve (A + mu diag(A)) x = -g; S = arithmetic type of the factorisation
template <typename S, typename T>
__device__ __forceinline__ bool lm_solve(const NormEqT<T>& ne, S mu, S* x) {
    S L[5][5], Li[5], id[5], b[5];
    if (!chol5<S, T>(ne, mu, L, Li, id)) return false;
#pragma unroll
    for (int i = 0; i < 5; ++i) b[i] = -(S)ne.g[i];
    chol5_solve<S, S>(L, Li, id, b, x);
    return true;
}

// inverse of the symmetric 5x5: one factorisation, five back-substitutions; false if singular
// (in the arithmetic of the normal matrix: the error columns of the mixed mode need no fp64)
template <typename T>
__device__ __forceinline__ bool spd_inverse(const NormEqT<T>& ne, double cov[5][5]) {
    T L[5][5], Li[5], id[5];
    if (!chol5<T, T>(ne, (T)0, L, Li, id)) return false;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        T b[5];
        double x[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) b[k] = (k == c) ? (T)1 : (T)0;
        chol5_solve<T, double>(L, Li, id, b, x);
#pragma unroll
        for (int k = 0; k < 5; ++k) cov[k][c] = x[k];
    }
    return true;
}

// A polish step of relative size `rel` leaves an error of about c * rel, c = the contraction
// factor of the iteration (error of the float Gauss-Newton matrix, <~ 1e-2): steps below 1e-4 end
// the polish without a further gradient pass (error <~ 1e-6, under what the fp32 stamps allow).
#ifndef MPSFR_POLISH_TOL
#define MPSFR_POLISH_TOL 1.0e-4
#endif
#ifndef MPSFR_FIT_WAVES
#define MPSFR_FIT_WAVES 3
#endif
// amdgpu_waves_per_eu: at least three waves per SIMD for the float fit (without the hint the
// scheduler's appetite for registers took 208 VGPRs -> 2 waves).  Four (128 registers: every stamp
// of the bench step resident at once) was the better setting until the instruction diet of round 3;
// the leaner kernel wants 156 registers, and squeezed into 128 it pays 100 B of scratch and ~30
// moves per iteration: 66.9 us at four waves against 60.7 at three.  The fp64 mode needs the
// registers (2 waves).
template <typename RE>
constexpr int fit_min_waves() { return sizeof(RE) == 4 ? MPSFR_FIT_WAVES : 2; }

constexpr double kFitIllCond = 100.0;       // MPSFR_FIT_ILL_CONDITIONED, include/mpsfr.h

#ifndef MPSFR_FIT_MOMENT_START
#define MPSFR_FIT_MOMENT_START 1
#endif
// the stamp sum riding in the fit's launch (nwg = 0: none)
struct SumArgs {
    int nwg, ntask, nl, accumulate;
    double* sum;
};

// Stamps (= waves) per workgroup.  The waves of a workgroup share nothing, but a workgroup's slots are
// released together: with four stamps per workgroup a slot waits for the slowest of four fits.
#ifndef MPSFR_FIT_WG
#define MPSFR_FIT_WG 1
#endif
template <typename RE, typename TS>
__global__ void __launch_bounds__(64 * MPSFR_FIT_WG) __attribute__((amdgpu_waves_per_eu(fit_min_waves<RE>())))
k_fit(int nstamp, const TS* __restrict__ stamps, double* __restrict__ fit, double polish_tol, SumArgs sa) {
    constexpr int NPX = NS * NS / 64;                     // 25 pixels per lane
    static_assert(NPX * 64 == NS * NS, "the lane map assumes 1600 pixels");
    using S = RE;                                         // type of the LM state
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x < sa.nwg) {
        // The stamp sum of the chunk (K_STAMP_SUM's job: PSF_MEAN numerator, psfrec.py:1104) as the first
        // workgroups of the fit's launch: it reads the same stamps, is independent of the fits, and as a
        // launch of its own it was 5 us of a queue that runs one kernel at a time (one lane 12.13 -> 12.38 M
        // PSFs/s).  A wave adds the tasks of 64 outputs in task order (eight loads in flight): deterministic.
        const size_t per = (size_t)sa.nl * NS * NS;
        const size_t e = ((size_t)blockIdx.x * MPSFR_FIT_WG + (threadIdx.x >> 6)) * 64 + lane;
        if (e >= per) return;
        double acc = 0.0;
        int t = 0;
        for (; t + 8 <= sa.ntask; t += 8) {
            TS v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = stamps[(size_t)(t + k) * per + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += (double)v[k];
        }
        for (; t < sa.ntask; ++t) acc += (double)stamps[(size_t)t * per + e];
        sa.sum[e] = sa.accumulate ? sa.sum[e] + acc : acc;
        return;
    }
    const int st = ((int)blockIdx.x - sa.nwg) * MPSFR_FIT_WG + (threadIdx.x >> 6);
    if (st >= nstamp) return;                             // the whole wave exits together
    const TS* src = stamps + (size_t)st * NS * NS;
    // the stamp in the evaluation type, LDS-resident for the LM evaluations (25 fewer VGPRs than
    // register-resident pixels: with the gradient-only polish this reaches 4 waves per SIMD, so
    // all 3500 stamps of the bench step are resident at once instead of in two rounds)
    __shared__ RE spix[MPSFR_FIT_WG][NS * NS];
    RE* sp = spix[threadIdx.x >> 6];
    // comparisons in the type the stamp is stored in (exact; in double they were a conversion and
    // a two-register select per pixel for float stamps)
    TS best = (TS)-3.0e38;
    int besto = 0;
#pragma unroll
    for (int m = 0; m < NPX; ++m) {
        const int o = lane + m * 64;
        const TS d = src[o];
        sp[o] = (RE)d;
        if (d > best) { best = d; besto = o; }
    }
    // argmax (first maximum in C order, as np.argmax) and the pixel count above half maximum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const TS ob = __shfl_xor(best, o, 64);
        const int oo = __shfl_xor(besto, o, 64);
        if (ob > best || (ob == best && oo < besto)) { best = ob; besto = oo; }
    }
    // Start values.  The LM phase costs one pass over the stamp per iteration, so a start inside the
    // basin of quadratic convergence is worth a few hundred instructions of setup.  Moments of the
    // stamp over the largest disc around the brightest pixel that fits the stamp, R = (distance to the
    // nearest edge) + 1/2: for a Moffat sampled at its centre
    //     S1 = sum d   = I pi a^2 / (n - 1)  T1,   T1 = 1 - (1 + R^2/a^2)^(1 - n)
    //     S2 = sum d^2 = I^2 pi a^2 / (2n - 1) T2,  T2 = 1 - (1 + R^2/a^2)^(1 - 2n)
    // so (S2/T2) / (I S1/T1) = (n - 1)/(2n - 1) gives n and then a; the truncation factors T1, T2 by
    // fixed-point iteration from T = 1 (five rounds: eta to ~0.01, FWHM to 1 % on the bench stamps, which
    // are Moffat-like but not Moffats).  Mean LM passes per stamp 3.07 -> 2.34 on the bench workload,
    // 3.69 -> 2.86 on the native-grid goldens (NumPy study with the iteration rules of this kernel); a
    // brightest pixel within six pixels of an edge (caller stamps) falls back to the half-maximum area
    // and n = 2.5.  The least-squares minimum is unique: the start only sets the
    // iteration count.
    const int p0i = besto / NS, q0i = besto % NS;
    const int rm = min(min(p0i, NS - 1 - p0i), min(q0i, NS - 1 - q0i));
    int cnt = 0;
    const TS half = (TS)0.5 * best;
    float ms1 = 0.f, ms2 = 0.f;
    {
        const float r2 = ((float)rm + 0.5f) * ((float)rm + 0.5f);
        const float lrf = (float)((lane >> 3) - p0i), lcf = (float)((lane & 7) - q0i);   // pixel map: moffat_accumulate
        const RE* pl = sp + (lane >> 3) * NS + (lane & 7);
#pragma unroll
        for (int mo = 0; mo < 5; ++mo) {
            const float dp = (float)(8 * mo) + lrf, dp2 = dp * dp;
#pragma unroll
            for (int mi = 0; mi < 5; ++mi) {
                const float dq = (float)(8 * mi) + lcf;
                const float d = (float)pl[mo * 8 * NS + mi * 8];
                cnt += (TS)pl[mo * 8 * NS + mi * 8] > half ? 1 : 0;
                const float din = fmaf(dq, dq, dp2) <= r2 ? d : 0.f;
                ms1 += din;
                ms2 = fmaf(din, din, ms2);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    ms1 = wave_total(ms1);
    ms2 = wave_total(ms2);
    double fw0 = 2.0 * sqrt((double)cnt / kPi);
    fw0 = fmin(fmax(fw0, 1.5), (double)NS);
    float eta0 = 0.4f;
    if (MPSFR_FIT_MOMENT_START && rm >= 6 && ms1 > 0.f && (float)best > 0.f) {
        const float bf = (float)best, r2 = ((float)rm + 0.5f) * ((float)rm + 0.5f);
        float t1 = 1.f, t2 = 1.f, nn = 2.5f, a2 = 1.f;
#pragma unroll 1
        for (int k = 0; k < 6; ++k) {
            float rho = (ms2 * t1) * __builtin_amdgcn_rcpf(bf * ms1 * t2);
            rho = fminf(fmaxf(rho, 0.05f), 0.47f);
            nn = (1.f - rho) * __builtin_amdgcn_rcpf(1.f - 2.f * rho);
            nn = fminf(fmaxf(nn, 1.1f), 15.f);
            a2 = ms1 * (nn - 1.f) * __builtin_amdgcn_rcpf(t1 * bf * 3.14159265f);
            const float lx = __builtin_amdgcn_logf(1.f + r2 * __builtin_amdgcn_rcpf(a2));
            t1 = 1.f - __builtin_amdgcn_exp2f((1.f - nn) * lx);
            t2 = 1.f - __builtin_amdgcn_exp2f((1.f - 2.f * nn) * lx);
        }
        eta0 = __builtin_amdgcn_rcpf(nn);
        const float w = 2.f * __builtin_amdgcn_sqrtf(a2 * (__builtin_amdgcn_exp2f(eta0) - 1.f));
        if (w == w) fw0 = fmin(fmax((double)w, 1.5), (double)NS);
    }
    // LM variables (I, p0, q0, w = FWHM, eta = 1/n): towards broad, Gaussian-like profiles the
    // model is nearly linear in 1/n, and the valley that n -> large opens in (w, n) stays short --
    // at most 4 iterations where the fit in n took up to 29
    S v[5] = {(S)best, (S)p0i, (S)q0i, (S)fw0, (S)eta0};
    // float evaluation only has to reach the basin of quadratic convergence: the fp64 polish
    // below finishes the job.  Every lane carries the same LM state (the totals of
    // moffat_accumulate are wave-uniform), so the control flow is uniform.
#ifndef MPSFR_FIT_TOL_F32
#define MPSFR_FIT_TOL_F32 1.0e-3
#endif
    const S tol = sizeof(RE) == 4 ? (S)MPSFR_FIT_TOL_F32 : (S)1.0e-10;
    NormEqT<RE> ne;
    lm_accumulate<RE>(sp, lane, v, ne);
    S mu = (S)1.0e-2, nu = (S)2;
    const S mu_max = sizeof(RE) == 4 ? (S)1.0e15f : (S)1.0e15;
    int it = 0, status = 1;
#ifndef MPSFR_FIT_MAXIT
#define MPSFR_FIT_MAXIT 200
#endif
    const int maxit = MPSFR_FIT_MAXIT;
    while (it < maxit) {
        ++it;
        S dx[5];
        if (!lm_solve<S, RE>(ne, mu, dx)) {
            mu *= nu;
            nu *= (S)2;
            if (mu > mu_max) { status = 2; break; }
            continue;
        }
        // A stamp narrower than the PSF core has no finite minimum in n: the valley runs to the
        // Gaussian limit eta -> 0.  A step may cut eta to a fifth at most (the whole step is
        // scaled), so the valley is descended geometrically instead of through rejected steps.
        if (v[4] + dx[4] < (S)0.2 * v[4]) {
            const S sc = (S)-0.8 * v[4] * fit_rcp<S>(dx[4]);
#pragma unroll
            for (int k = 0; k < 5; ++k) dx[k] *= sc;
        }
        S vn[5], rel = (S)0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            vn[k] = v[k] + dx[k];
            rel = fmax(rel, fabs(dx[k]) * fit_rcp<S>(fabs(vn[k]) + (S)1.0e-30));
        }
        const bool inside = vn[3] > (S)1.0e-3 && vn[4] > (S)1.0e-3 && vn[4] < (S)1.0e2;
        if (inside && rel < tol) {       // converged: take the last (tiny) Gauss-Newton step
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = vn[k];
            status = 0;
            break;
        }
        NormEqT<RE> nn;
        S rho = (S)-1;
        if (inside) {
            lm_accumulate<RE>(sp, lane, vn, nn);
            // predicted decrease of chi2: dx^T (mu D dx - g)
            S pred = (S)0;
            const int dg[5] = {0, 5, 9, 12, 14};
#pragma unroll
            for (int k = 0; k < 5; ++k)
                pred += dx[k] * (mu * (S)ne.a[dg[k]] * dx[k] - (S)ne.g[k]);
            rho = ((S)ne.chi2 - (S)nn.chi2) * fit_rcp<S>(pred);    // NaN -> rejected
        }
        if (rho > (S)0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = vn[k];
            ne = nn;
            if (v[4] < (S)1.5e-3) { status = 0; break; }      // n > 666: Gaussian to 1e-3, stop
            const S c = (S)2 * rho - (S)1;
            mu = fmax(mu * fmax((S)(1.0 / 3.0), (S)1 - c * c * c), (S)1.0e-14);
            nu = (S)2;
        } else {
            mu *= nu;
            nu *= (S)2;
            if (mu > mu_max) { status = 0; break; }   // no further descent: at the minimum
        }
    }
    double vd[5];                          // from here on in fp64
#pragma unroll
    for (int k = 0; k < 5; ++k) vd[k] = (double)v[k];
    double polish_chi2 = -1.0;
    if constexpr (sizeof(RE) == 4) {
        // The float evaluation has systematic errors of ~1e-6 in the wings (v_log/v_exp), enough
        // to move beta by a few 1e-4 on flat-topped stamps.  Polish from the float solution with
        // steps  -A^-1 g,  g the fp64 gradient (moffat_gradient), A the float normal matrix of the
        // last LM iteration: one or two steps suffice.
        NormEq np;
#pragma unroll
        for (int k = 0; k < 15; ++k) np.a[k] = (double)ne.a[k];
        np.chi2 = -1.0;
#ifndef MPSFR_POLISH_MAX
#define MPSFR_POLISH_MAX 8
#endif
        for (int pz = 0; pz < MPSFR_POLISH_MAX && status != 2; ++pz) {
            moffat_gradient(src, lane, vd, np.g, &np.chi2);
            double dx[5];
            if (!lm_solve<double, double>(np, 1.0e-10, dx)) break;
            float rel = 0.f;               // a size, compared with 0.1 / 1e-3 / polish_tol: float
#pragma unroll
            for (int k = 0; k < 5; ++k)
                rel = fmaxf(rel, fabsf((float)dx[k]) *
                                     __builtin_amdgcn_rcpf(fabsf((float)(vd[k] + dx[k])) + 1.0e-30f));
            const bool inside = vd[3] + dx[3] > 1.0e-3 && vd[4] + dx[4] > 1.0e-3 &&
                                vd[4] + dx[4] < 1.0e2 && rel < 0.1f;
            if (!inside) break;
#pragma unroll
            for (int k = 0; k < 5; ++k) vd[k] += dx[k];
            ++it;
            polish_chi2 = rel < 1.0e-3f ? np.chi2 : -1.0;
            if (rel < (float)polish_tol) break;            // error after this step ~ 1e-2 rel
        }
    }
    // Outputs in (a, n).  The covariance comes from the normal matrix of the last LM iteration (in
    // (w, eta), a point within ~1e-4 of the final one for well-posed stamps) -- no further
    // Jacobian pass.  I, p0, q0 are the same variables in both parametrisations, so their
    // variances carry over; FWHM = w directly; n = 1/eta and alpha = w / (2 sqrt(2^eta - 1))
    // through their partial derivatives.
    const double n = 1.0 / vd[4];
    const double p2 = exp2(1.0 / n), s2 = p2 - 1.0, sq = sqrt(s2);
    const double al = fabs(vd[3]) / (2.0 * sq);
    const double va[5] = {vd[0], vd[1], vd[2], al, n};
    // chi2: the residuals of the last polish pass (fp64 model; one step of relative size < 1e-3
    // before the final point, so equal to first order) or, without a polish, a residual pass
    double chi2 = polish_chi2;
    if (chi2 < 0.0) chi2 = (double)moffat_chi2<RE>(sp, lane, va);
    if (lane == 0) {
        double* o = fit + (size_t)st * NFIT;
        o[0] = vd[0]; o[1] = vd[1]; o[2] = vd[2]; o[3] = al; o[4] = n;
        o[5] = fabs(vd[3]);
        o[6] = chi2;
        o[7] = (double)it;
        double cov[5][5];
        const double dof = (double)(NS * NS - 5);
        if (spd_inverse(ne, cov)) {
            const double s = chi2 / dof;
            o[8] = sqrt(fmax(cov[0][0] * s, 0.0));
            o[9] = sqrt(fmax(cov[1][1] * s, 0.0));
            o[10] = sqrt(fmax(cov[2][2] * s, 0.0));
            const double aw = 1.0 / (2.0 * sq);
            const double an = -al * p2 * 0.69314718055994530942 / (2.0 * s2);     // d alpha / d eta
            const double var = aw * aw * cov[3][3] + 2.0 * aw * an * cov[3][4] + an * an * cov[4][4];
            o[11] = sqrt(fmax(var * s, 0.0));
            o[12] = n * n * sqrt(fmax(cov[4][4] * s, 0.0));                        // |dn/d eta| = n^2
            o[13] = sqrt(fmax(cov[3][3] * s, 0.0));
            // Ill-conditioned fits (status bit 4, round 6).  With iid pixel noise of standard deviation sigma the
            // least-squares n has the standard deviation n^2 sqrt(cov[eta][eta]) sigma: the number below is that
            // for sigma = the peak, i.e. the change of n per unit of relative pixel noise.  From kFitIllCond = 100
            // on, noise of 1e-6 of the peak moves n by 1e-4 -- the parity tolerance -- or more: the minimum is
            // still the minimum, but stamps known to a few 1e-7 of their peak (mixed precision) do not determine
            // (fwhm, n) to the tolerance.  Stamps narrower than the PSF core are the case: 256^2 / 128^2 grids
            // with the rescaled pixel scale (FWHM 30 px in a 40 px stamp: 180-200); the bench rows at 512^2 stay
            // below 8 (profiles/r06_small_grid_margin.txt).
            if (n * n * sqrt(fmax(cov[4][4], 0.0)) * fabs(vd[0]) >= kFitIllCond) status |= 4;
        } else {
            for (int k = 0; k < 6; ++k) o[8 + k] = 0.0;
            if (status == 0) status = 2;
        }
        o[14] = (double)status;
        o[15] = vd[0] * kPi * al * al / (n - 1.0);
    }
}

// K_STAMP_SUM: deterministic sum of the final stamps over the tasks of a chunk (PSF_MEAN numerator,
// psfrec.py:1104).  64 outputs per workgroup; wave w adds tasks w, w+4, ... and the four partial
// sums are combined in a fixed order.
template <typename TF>
__global__ void __launch_bounds__(256) k_stamp_sum(int ntask, int nl, const TF* __restrict__ fin,
                                                   double* __restrict__ sum, int accumulate) {
    __shared__ double part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    const size_t per = (size_t)nl * NS * NS;
    double s = 0.0;
    if (e < per) {
        // (the loads of eight tasks in flight together; the additions stay in task order)
        int t = wave;
        for (; t + 28 < ntask; t += 32) {
            TF v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fin[(size_t)(t + 4 * k) * per + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
        }
        for (; t < ntask; t += 4) s += (double)fin[(size_t)t * per + e];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && e < per) {
        const double tot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        sum[e] = accumulate ? sum[e] + tot : tot;
    }
}

// K_PARAM_COPY: see launch_param_copy.  One workgroup of 1024 threads; a thread issues ALL of its loads from the
// pinned host memory before its first store, so a blob of up to 64 KB is one PCIe round trip (round 5: 256 threads
// with a load -> store loop, five dependent round trips for 20 KB, 8.2 us).  When all of the blob has been read,
// thread 0 writes `seq` into the pinned host word `flag` (system scope): the host may refill the blob.
constexpr int kParamCopyThreads = 1024, kParamCopyPer = 4;
__global__ void __launch_bounds__(kParamCopyThreads) k_param_copy(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16,
                                                                  unsigned long long* flag, unsigned long long seq) {
    for (int base = 0; base < n16; base += kParamCopyThreads * kParamCopyPer) {
        uint4 v[kParamCopyPer];
#pragma unroll
        for (int q = 0; q < kParamCopyPer; ++q) {
            const int i = base + q * kParamCopyThreads + (int)threadIdx.x;
            if (i < n16) v[q] = src[i];
        }
#pragma unroll
        for (int q = 0; q < kParamCopyPer; ++q) {
            const int i = base + q * kParamCopyThreads + (int)threadIdx.x;
            if (i < n16) dst[i] = v[q];
        }
    }
    if (flag != nullptr) {
        __syncthreads();                     // (every thread's loads have returned: they fed its stores)
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

void launch_param_copy(hipStream_t s, void* d_dst, const void* h_src_pinned, size_t bytes,
                       unsigned long long* h_flag_pinned, unsigned long long seq) {
    const int n16 = (int)(bytes / 16);
    if (n16 <= 0) return;
    // (a small blob needs fewer waves: the launch of 16 waves is not free either)
    const int nthr = n16 >= kParamCopyThreads ? kParamCopyThreads : (n16 + 63) / 64 * 64;
    hipLaunchKernelGGL(k_param_copy, dim3(1), dim3(nthr), 0, s, (uint4*)d_dst, (const uint4*)h_src_pinned, n16,
                       h_flag_pinned, seq);
}

void launch_moffat_kernels(hipStream_t s, int nker, const double* d_gamma, const double* d_alpha,
                           void* d_out, bool f64) {
    if (nker <= 0) return;
    if (f64)
        hipLaunchKernelGGL(k_moffat_kernels<double>, dim3(nker), dim3(256), 0, s, d_gamma, d_alpha,
                           (double*)d_out);
    else
        hipLaunchKernelGGL(k_moffat_kernels<float>, dim3(nker), dim3(256), 0, s, d_gamma, d_alpha,
                           (float*)d_out);
}

void launch_conv(hipStream_t s, int ntask, int nl, const void* d_pre, const void* d_ktt,
                 const void* d_kmuse, double* d_fin, bool f64) {
    dim3 grid(nl, ntask);
    constexpr int PH = NS + KS - 1, PW = 84;
    if (f64) {
        const size_t sm = (size_t)(PH * PW) * sizeof(double);
        hipLaunchKernelGGL(k_conv<double>, grid, dim3(256), sm, s, nl, (const double*)d_pre,
                           (const double*)d_ktt,
                           (const double*)d_kmuse, d_fin);
    } else {
        const size_t sm = (size_t)(PH * PW) * sizeof(float);
        hipLaunchKernelGGL(k_conv<float>, grid, dim3(256), sm, s, nl, (const float*)d_pre,
                           (const float*)d_ktt,
                           (const float*)d_kmuse, d_fin);
    }
}

void launch_khat(hipStream_t s, int nker, const double* d_gamma, const double* d_alpha,
                 void* d_khat, bool f64) {
    if (nker <= 0) return;
    if (f64) {
        allow_smem(k_khat<double>, conv_smem_bytes<double>(true));
        hipLaunchKernelGGL(k_khat<double>, dim3(nker), dim3(256), conv_smem_bytes<double>(true), s, d_gamma,
                           d_alpha, (cx<double>*)d_khat);
    } else {
        hipLaunchKernelGGL(k_khat<float>, dim3(nker), dim3(256), conv_smem_bytes<float>(true), s, d_gamma,
                           d_alpha, (cx<float>*)d_khat);
    }
}

void launch_conv_fft(hipStream_t s, int ntask, int nl, const void* d_pre, const void* d_khat_tt,
                     const void* d_khat_muse, void* d_fin, bool fin_f32, bool f64, const MfFinishArgs& finish) {
    const dim3 grid(nl, ntask);
    MfFinish mf;
    mf.gsw = f64 ? nullptr : finish.gsw; mf.part = (const f4*)finish.part;
    mf.per = finish.per; mf.ngr = finish.ngr; mf.nsw = finish.nsw;
    if (f64) {          // double stamps in, double arithmetic, double stamps out
        allow_smem((k_conv_fft<double, double>), conv_smem_bytes<double>(false));
        hipLaunchKernelGGL((k_conv_fft<double, double>), grid, dim3(256), conv_smem_bytes<double>(false), s, nl,
                           (const double*)d_pre, (const cx<double>*)d_khat_tt, (const cx<double>*)d_khat_muse,
                           (double*)d_fin, mf);
    } else if (fin_f32) {
        hipLaunchKernelGGL((k_conv_fft<float, float>), grid, dim3(256), conv_smem_bytes<float>(false), s, nl,
                           (const float*)d_pre, (const cx<float>*)d_khat_tt, (const cx<float>*)d_khat_muse,
                           (float*)d_fin, mf);
    } else {
        hipLaunchKernelGGL((k_conv_fft<float, double>), grid, dim3(256), conv_smem_bytes<float>(false), s, nl,
                           (const float*)d_pre, (const cx<float>*)d_khat_tt, (const cx<float>*)d_khat_muse,
                           (double*)d_fin, mf);
    }
}

void launch_fit(hipStream_t s, int nstamp, const void* d_stamps, bool stamps_f32, double* d_fit,
                bool f64, int sum_ntask, int sum_nl, double* d_sum, int sum_accumulate) {
    if (nstamp <= 0) return;
    SumArgs sa = {0, sum_ntask, sum_nl, sum_accumulate, d_sum};
    if (d_sum != nullptr && sum_ntask > 0) {
        const size_t per = (size_t)sum_nl * NS * NS;
        sa.nwg = (int)((per + 64 * MPSFR_FIT_WG - 1) / (64 * MPSFR_FIT_WG));
    }
    // One wavefront per stamp and one stamp per workgroup: the waves share nothing, and a workgroup of
    // four gave its slots back only when the slowest of its four fits was done (alone, bracketed:
    // 56.1 us with one stamp per workgroup, 57.9 with two, 58.0 with four).  (A whole workgroup per
    // stamp with the wave sums meeting in LDS measured 1.8x slower at 3500 stamps: every wave repeats
    // the 5x5 solves and the iterations serialise on barriers.)
    const dim3 grid(sa.nwg + (nstamp + MPSFR_FIT_WG - 1) / MPSFR_FIT_WG), blk(64 * MPSFR_FIT_WG);
    // f64 mode: the same float Levenberg-Marquardt iterations find the basin (they cost a third of
    // fp64 ones), and the polish on the fp64 stamps runs on until its steps are below 1e-8 (or
    // MPSFR_POLISH_MAX passes).  Only the residual of that pass is fp64: moffat_gradient rounds it to
    // float and sums the gradient and the normal matrix per lane in fp32, so what the polish converges to
    // is the root of a gradient carrying ~1e-7 of relative rounding, and the err_* columns come from the
    // float normal matrix.  Delivered, against the oracle: |d fwhm| 6e-9 arcsec, |d beta| 7e-7 (the tests
    // hold the f64 fits at 1e-6); MPSFR_FIT_F64_LM=1 builds keep the all-fp64 iterations.
#ifndef MPSFR_FIT_F64_LM
#define MPSFR_FIT_F64_LM 0
#endif
    if (f64 && MPSFR_FIT_F64_LM)
        hipLaunchKernelGGL((k_fit<double, double>), grid, blk, 0, s, nstamp,
                           (const double*)d_stamps, d_fit, 0.0, sa);
    else if (stamps_f32)
        hipLaunchKernelGGL((k_fit<float, float>), grid, blk, 0, s, nstamp, (const float*)d_stamps,
                           d_fit, (double)MPSFR_POLISH_TOL, sa);
    else
        hipLaunchKernelGGL((k_fit<float, double>), grid, blk, 0, s, nstamp,
                           (const double*)d_stamps, d_fit, f64 ? 1.0e-8 : (double)MPSFR_POLISH_TOL, sa);
}

void launch_stamp_sum(hipStream_t s, int ntask, int nl, const void* d_fin, bool fin_f32, double* d_sum,
                      int accumulate) {
    const size_t per = (size_t)nl * NS * NS;
    const dim3 grid((unsigned)((per + 63) / 64));
    if (fin_f32)
        hipLaunchKernelGGL(k_stamp_sum<float>, grid, dim3(256), 0, s, ntask, nl, (const float*)d_fin,
                           d_sum, accumulate);
    else
        hipLaunchKernelGGL(k_stamp_sum<double>, grid, dim3(256), 0, s, ntask, nl, (const double*)d_fin,
                           d_sum, accumulate);
}


}  // namespace mpsfr
