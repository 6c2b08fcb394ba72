// HIP kernels of the muse-psfr PSF-reconstruction hot path, written for gfx950 (MI355X, wave64).
// Reference citations are to /root/reference/muse_psfr/psfrec.py.  See DESIGN.md for the data
// layout and the derivation of the restructured algorithm.
//
// Per-wavelength stage: sampling tables, OTF lines -> sampled first pass, second pass.
#include "device_common.h"
#include "fft_r16.h"

namespace mpsfr {

namespace {

// ------------------------------------------------------------------------------------------
// K_GTABLE: per-wavelength sampling tables.  Sample i of the 40-pixel stamp sits at i*npixc/40
// in the centred crop (psfrec.py:672-683); its left neighbour is native index
// p_i = (floor(i npixc/40) - npixc/2) mod N with weight 1-a_i, a_i = frac.
// G[l][v][j] = w_v ((1-a_j) W^-(v p_j) + a_j W^-(v (p_j+1))), W = exp(-2 pi i/N), w_v = 1 for
// v in {0, N/2} else 2: bilinear interpolation folded into the second (column) pass.
// ------------------------------------------------------------------------------------------
// Layout of G within one wavelength.  fp64: [v][j].  fp32 (consumed by the MFMA second pass): lines
// are padded to a multiple of 8 and stored as [v / 8][v % 4][j][(v / 4) % 2], so that the lane
// that supplies B[k = v % 4][j] to two consecutive k-steps fetches both elements with one 16-byte
// load (8-byte loads kept the vector memory pipe, not the matrix pipe, busy).
__host__ __device__ constexpr int g_lines(int N) { return (N / 2 + 1 + 7) / 8 * 8; }
template <typename R>
__device__ __forceinline__ size_t g_index(int v, int j) {
    if constexpr (sizeof(R) == 4)
        return ((size_t)((v >> 3) * 4 + (v & 3)) * NS + j) * 2 + ((v >> 2) & 1);
    else
        return (size_t)v * NS + j;
}

// Layout of the sampled first-pass lines Tq within one (task, wavelength) block: [v][i], i.e. the
// 21 samples of a line are one contiguous 168-byte run written by the wave that owns the line.
// (A paired-line layout [v / 8][i][v % 4][(v / 4) % 2], which gives the MFMA second pass 16-byte
// coalesced operand loads, made that pass 10 % faster but doubled the HBM write traffic of
// K_OTF_ROWFFT: four waves then fill each 32-byte piece at different times and the partly
// written lines leave the L2 before they are complete -- WRITE_SIZE 305 MB against 156 MB.)
template <typename R>
__host__ __device__ constexpr int tq_block(int N) {
    return (N / 2 + 1) * NSH;
}
template <typename R>
__device__ __forceinline__ size_t tq_index(int v, int i) {
    return (size_t)v * NSH + i;
}

template <typename R>
__global__ void __launch_bounds__(256)
k_gtable(int N, const LamPar* __restrict__ lp, const cx<double>* __restrict__ twg,
         int* __restrict__ samp_p, R* __restrict__ samp_a, cx<R>* __restrict__ G) {
    const int l = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int npixc = lp[l].npixc;
    if (idx < NS) {
        const int q = idx * npixc;
        samp_p[l * NS + idx] = ((q / NS - npixc / 2) % N + N) % N;
        samp_a[l * NS + idx] = (R)((double)(q % NS) / NS);
    }
    const int H1 = N / 2 + 1, HP = g_lines(N);
    if (idx >= HP * NS) return;
    const int v = idx / NS, j = idx % NS;
    const int q = j * npixc;
    const int p = ((q / NS - npixc / 2) % N + N) % N;
    const double a = (double)(q % NS) / NS;
    const cx<double> w0 = twg[(int)(((long)v * p) % N)];
    const cx<double> w1 = twg[(int)(((long)v * (p + 1)) % N)];
    const double wv = v >= H1 ? 0.0 : (v == 0 || v == N / 2) ? 1.0 : 2.0;   // padding lines: 0
    // conj(W^(v p)) = exp(+2 pi i v p / N)
    G[(size_t)l * HP * NS + g_index<R>(v, j)] = {(R)(wv * ((1.0 - a) * w0.x + a * w1.x)),
                                                 (R)(-wv * ((1.0 - a) * w0.y + a * w1.y))};
}

// ------------------------------------------------------------------------------------------
// K_OTF_ROWFFT (dominant kernel): for one task and one line v of the transposed half plane, for
// every wavelength: OTF line = tel * sum_dir exp(c_l * D0)  (psfrec.py:793-797; the mean over
// directions of psfrec.py:674 commutes with the FFT), forward FFT along the line, and the
// bilinear-weighted extraction of the NS sampled positions -> Tq[task][l][v][i].
// ------------------------------------------------------------------------------------------
// exp(c * d) for the OTF.  FAST (float only): the caller pre-multiplies c by log2(e) and the
// hardware exp2 is used directly: one multiply + v_exp_f32 per value.
template <typename R, bool FAST>
__device__ __forceinline__ R exp_scale(R c) {
    if constexpr (FAST && sizeof(R) == 4) return c * (R)1.44269504088896340736;
    else return c;
}
// exp(x) in fp64 for the OTF of the f64 mode (x = c D <= 0 up to rounding): n = rint(x log2 e),
// r = x - n ln 2 in two pieces (|r| <= 0.347), the Taylor polynomial of degree 12 (next term 1.7e-16),
// v_ldexp_f64.  21 instructions where the library's exp -- with its table, its special cases and its
// error-free reductions for the full range -- took half of K_OTF_ROWFFT<double> (71 % vector-pipe busy,
// two exponentials per OTF element and wavelength pair: profiles/r04_pmc_f64.txt).  Results below the
// normal range underflow through ldexp like the library's; NaN stays NaN.
__device__ __forceinline__ double exp_otf64(double x) {
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);      // ln 2, high part (the reduction is exact for |n| < 2^10)
    r = fma(n, -1.90821492927058770002e-10, r);             // low part
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
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
    return __builtin_ldexp(p, (int)fmax(n, -2000.0));
}

template <typename R, bool FAST>
__device__ __forceinline__ R exp_sel(R x) {
    if constexpr (sizeof(R) == 8) {
        return exp_otf64(x);
    } else if constexpr (FAST) {
        return __builtin_amdgcn_exp2f(x);
    } else {
        return expf(x);
    }
}

// LDS table of the extraction, one entry per (wavelength, sample i <= 20): the positions of
// Z[p], Z[-p], Z[p+1], Z[-p-1] in the finished line (lds_out indices) and the bilinear weight.
struct SampOff {
    unsigned short p, mp, q, mq;
};
constexpr int kOtfLdsTabMaxNl = 128;      // above this the table would crowd the line buffers
// Groups of wavelength pairs (grid.z) of the per-wavelength kernels.  With line pruning the work
// of a (task, line group) runs from 0 to all pairs, and the workgroups that need every pair set
// the length of the kernel (18 transforms in a row: 55-70 us).  vkeep grows with the pair, so
// with G groups only the line groups that the early pairs need are split.
#ifndef MPSFR_OTF_PAIR_GROUPS
#define MPSFR_OTF_PAIR_GROUPS 2
#endif
__host__ __device__ inline int otf_pairs_per_group(int nl, bool pruned) {
    const int npair = (nl + 1) / 2;
    const int g = pruned ? MPSFR_OTF_PAIR_GROUPS : 1;
    return (npair + g - 1) / g;
}

template <typename R, int N, int ND, bool FASTEXP, bool LTAB>
__global__ void __launch_bounds__(LineCfg<N>::THREADS)
k_otf_rowfft(int ndir, int nl, const R* __restrict__ D0t, const R* __restrict__ telT,
             const LamPar* __restrict__ lp, const int* __restrict__ samp_p,
             const R* __restrict__ samp_a, cx<R>* __restrict__ Tq,
             const cx<double>* __restrict__ twg, const int* __restrict__ vkeep) {
    using L = LineCfg<N>;
    constexpr int TPR = L::TPR, SLOTS = L::SLOTS, THREADS = L::THREADS, NPAD = L::NPAD;
    constexpr int EPT = N / TPR;
    constexpr bool REGTW = use_reg_twiddles<N>(), WS = L::WSYNC;
    extern __shared__ __align__(16) unsigned char smem[];
    cx<R>* twl = reinterpret_cast<cx<R>*>(smem);          // only used when !REGTW
    cx<R>* bufA = twl + (REGTW ? 0 : NPAD);
    cx<R>* bufB = bufA + SLOTS * NPAD;
    // line pruning (stage_a.hip): the workgroup's lines start at v0; a wavelength pair whose
    // vkeep is not above v0 needs none of them (decided per workgroup: barriers stay uniform)
    const int* vk_task = vkeep != nullptr ? vkeep + (size_t)blockIdx.y * ((nl + 1) / 2) : nullptr;
    const int v0 = blockIdx.x * SLOTS;
    const int ppg = otf_pairs_per_group(nl, vkeep != nullptr);
    const int l_beg = (int)blockIdx.z * 2 * ppg, l_end = min(nl, ((int)blockIdx.z + 1) * 2 * ppg);
    // vkeep grows with the pair: nothing to do if the group's last pair does not need v0
    if (vk_task != nullptr && v0 >= vk_task[(l_end - 1) >> 1]) return;
    // after the line buffers: [nl][NSH] SampOff, [nl][NSH] weights, [nl] exponent factors
    SampOff* stab = reinterpret_cast<SampOff*>(bufA + fft_nbuf<N>() * SLOTS * NPAD);
    R* swt = reinterpret_cast<R*>(stab + (LTAB ? nl * NSH : 0));
    R* scl = swt + (LTAB ? nl * NSH : 0);
    const int slot = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int v = blockIdx.x * SLOTS + slot;
    const int task = blockIdx.y;
    const bool valid = v <= N / 2;
    const int vv = valid ? v : N / 2;
    TwRegs<R, N> twr;
    const cx<R>* twp;
    if constexpr (REGTW) {
        twr.init(twg, t);
        twp = twr.w;
    } else {
        for (int i = threadIdx.x; i < N; i += THREADS)
            twl[lds_pad(i)] = {(R)twg[i].x, (R)twg[i].y};
        twp = twl;
    }
    if constexpr (LTAB) {
        // Nothing inside the wavelength loop reads global memory: every wait on the vector-memory
        // counter there would also wait for the Tq stores of the wavelength before (the counter
        // is in issue order), i.e. expose a store round trip per transform.
        for (int e = threadIdx.x; e < nl * NSH; e += THREADS) {
            const int ll = e / NSH, i = e - ll * NSH;
            const int p = samp_p[ll * NS + i];
            const int q = p + 1 == N ? 0 : p + 1;
            const int mp = p == 0 ? 0 : N - p, mq = q == 0 ? 0 : N - q;
            stab[e] = {(unsigned short)lds_out<N, sizeof(cx<R>)>(p), (unsigned short)lds_out<N, sizeof(cx<R>)>(mp),
                       (unsigned short)lds_out<N, sizeof(cx<R>)>(q), (unsigned short)lds_out<N, sizeof(cx<R>)>(mq)};
            swt[e] = samp_a[ll * NS + i];
        }
        for (int e = threadIdx.x; e < nl; e += THREADS) scl[e] = exp_scale<R, FASTEXP>((R)lp[e].c);
    }
    if constexpr (LTAB || !REGTW) __syncthreads();
    cx<R>* a = bufA + slot * NPAD;
    cx<R>* b = bufB + slot * NPAD;     // only used when a slot spans two wavefronts
    // Single direction, hardware exp2: the telescope OTF goes into the exponent,
    // tel * 2^(c d) = 2^(c d + log2 tel)  (tel = 0 -> 2^-inf = 0), one fma + v_exp_f32 per value.
    constexpr bool FOLD = FASTEXP && ND == 1 && sizeof(R) == 4;
    R tel[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        tel[e] = telT[(size_t)vv * N + t + e * TPR];
        if constexpr (FOLD) tel[e] = __builtin_amdgcn_logf(tel[e]);      // log2
    }
    const R* dline = D0t + ((size_t)task * ndir * (N / 2 + 1) + vv) * N;
    const size_t dstride = (size_t)(N / 2 + 1) * N;
    R dreg[ND == 1 ? EPT : 1];
    if constexpr (ND == 1) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) dreg[e] = dline[t + e * TPR];
    }
    // the line is in registers before the loop starts (vmcnt(0)): the compiler then places no
    // vector-memory wait inside the loop, where it would also wait for the stores in flight
    if constexpr (ND == 1 && LTAB) __builtin_amdgcn_s_waitcnt(0x0F70);
    cx<R>* tq_task = Tq + (size_t)task * nl * tq_block<R>(N);
    // two wavelengths per complex transform: z = otf(la) + i otf(lb), both real lines
    for (int l = l_beg; l < l_end; l += 2) {
        if (vk_task != nullptr && v0 >= vk_task[l >> 1]) continue;
        const bool two = l + 1 < nl;
        R ca, cb;
        if constexpr (LTAB) {
            ca = scl[l];
            cb = scl[two ? l + 1 : l];
        } else {
            ca = exp_scale<R, FASTEXP>((R)lp[l].c);
            cb = exp_scale<R, FASTEXP>((R)lp[two ? l + 1 : l].c);
        }
        cx<R> x[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            R ra = (R)0, rb = (R)0;
            if constexpr (FOLD) {
                x[e] = {exp_sel<R, true>(fmaf(ca, dreg[e], tel[e])),
                        two ? exp_sel<R, true>(fmaf(cb, dreg[e], tel[e])) : (R)0};
                continue;
            } else if constexpr (ND == 1) {
                ra = exp_sel<R, FASTEXP>(ca * dreg[e]);
                rb = exp_sel<R, FASTEXP>(cb * dreg[e]);
            } else {
                for (int d = 0; d < ndir; ++d) {
                    const R dv = dline[d * dstride + t + e * TPR];
                    ra += exp_sel<R, FASTEXP>(ca * dv);
                    rb += exp_sel<R, FASTEXP>(cb * dv);
                }
            }
            x[e] = {tel[e] * ra, two ? tel[e] * rb : (R)0};
        }
        const cx<R>* res = fft_forward_regs<R, N, REGTW>(x, a, b, twp, t);
        if (valid) {
            // F_a[p] = (Z[p] + conj Z[-p]) / 2,  F_b[p] = (Z[p] - conj Z[-p]) / 2i.
            // Only samples i = 0..20: the OTF is real, so A[v][-p] = conj A[v][p], and the sample
            // positions are symmetric about the centre (p_(40-i) + 1 = -p_i, weights swapped),
            // hence Tq[v][40-i] = conj Tq[v][i].
            for (int idx = t; idx < 2 * NSH; idx += TPR) {
                const int which = idx / NSH, i = idx - which * NSH;
                if (which == 1 && !two) continue;
                const int ll = l + which;
                cx<R> zp, zmp, zq, zmq;
                R w;
                if constexpr (LTAB) {
                    const SampOff o = stab[ll * NSH + i];
                    w = swt[ll * NSH + i];
                    zp = res[o.p]; zmp = res[o.mp]; zq = res[o.q]; zmq = res[o.mq];
                } else {
                    const int p = samp_p[ll * NS + i];
                    w = samp_a[ll * NS + i];
                    const int q = p + 1 == N ? 0 : p + 1;
                    const int mp = p == 0 ? 0 : N - p, mq = q == 0 ? 0 : N - q;
                    zp = res[lds_out<N, sizeof(cx<R>)>(p)]; zmp = res[lds_out<N, sizeof(cx<R>)>(mp)];
                    zq = res[lds_out<N, sizeof(cx<R>)>(q)]; zmq = res[lds_out<N, sizeof(cx<R>)>(mq)];
                }
                cx<R> f0, f1;
                const R h = (R)0.5;
                if (which == 0) {
                    f0 = {h * (zp.x + zmp.x), h * (zp.y - zmp.y)};
                    f1 = {h * (zq.x + zmq.x), h * (zq.y - zmq.y)};
                } else {
                    f0 = {h * (zp.y + zmp.y), -h * (zp.x - zmp.x)};
                    f1 = {h * (zq.y + zmq.y), -h * (zq.x - zmq.x)};
                }
                tq_task[(size_t)ll * tq_block<R>(N) + tq_index<R>(v, i)] = {
                    ((R)1 - w) * f0.x + w * f1.x, ((R)1 - w) * f0.y + w * f1.y};
            }
        }
        fft_sync<WS>();     // extraction reads done before the next transform overwrites
    }
}

// ------------------------------------------------------------------------------------------
// K_OTF_R16: K_OTF_ROWFFT with the radix-16 plans of fft_r16.h (fp32, N = 256 / 512 / 1024): two
// LDS crossings per transform instead of three (four), the last radix-RL pass (RL = N / 256)
// evaluated only for the outputs the extraction reads.  Same inputs, same Tq out.
//
// Extraction in units: a unit is (wavelength of the pair, sample i <= 20, half h), h = 0 for the
// bilinear neighbour p, h = 1 for p + 1.  With z the neighbour and -z its mirror the unit forms
//   t = c (Z[z] +- conj Z[-z])     (c = half the bilinear weight; the sign / real-imaginary swap
//                                   separates the two wavelengths of the complex transform)
// and the two halves, in adjacent lanes, are added on the DPP path.  The table (K_XTAB, cached
// per wavelength set) holds per unit: k = z mod 256 and (-z) mod 256 (positions in image 1), c,
// and the factors of the merged pass,  Z[z] = img1[k] + sum_{q=1}^{RL-1} W_N^(q z) img1[k + 256 q].
// 84 units per line: three rounds of a 32-lane line (87 % of the lanes busy; items of four outputs
// kept 66 % busy), and all table loads of a transform are issued before its second pass.
// ------------------------------------------------------------------------------------------
#ifndef MPSFR_R16_KNOCK
#define MPSFR_R16_KNOCK 0       // kernel experiments: 1 = no extraction, 2 = no exp
#endif
#ifndef MPSFR_R16_EARLY_TABLE
#define MPSFR_R16_EARLY_TABLE 0
#endif
template <int RL>
struct XUnit {
    unsigned short kz, kmz;
    float c;
    cx<float> B[RL > 1 ? 2 * (RL - 1) : 1];     // z: q = 1..RL-1, then -z: q = 1..RL-1
};
template <>
struct XUnit<1> {
    unsigned short kz, kmz;
    float c;
};
static_assert(sizeof(XUnit<1>) == 8 && sizeof(XUnit<2>) == 24 && sizeof(XUnit<4>) == 56, "8-byte pieces");

template <int RL>
__global__ void __launch_bounds__(256)
k_xtab(int N, int nl, const int* __restrict__ samp_p, const float* __restrict__ samp_a,
       const cx<double>* __restrict__ twg, XUnit<RL>* __restrict__ xtab) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nl * NSH * 2) return;
    const int h = e & 1, li = e >> 1;
    const int l = li / NSH, i = li - l * NSH;
    const int p = samp_p[l * NS + i];
    const int z = h ? (p + 1 == N ? 0 : p + 1) : p;
    const int mz = z == 0 ? 0 : N - z;
    const float w = samp_a[l * NS + i];
    XUnit<RL> x;
    x.kz = (unsigned short)(z & 255);
    x.kmz = (unsigned short)(mz & 255);
    x.c = 0.5f * (h ? w : 1.0f - w);
    if constexpr (RL > 1) {
        for (int q = 1; q < RL; ++q) {
            const cx<double> a = twg[(int)(((long)q * z) % N)], b = twg[(int)(((long)q * mz) % N)];
            x.B[q - 1] = {(float)a.x, (float)a.y};
            x.B[RL - 1 + q - 1] = {(float)b.x, (float)b.y};
        }
    }
    xtab[e] = x;
}

template <int N, int ND, bool FASTEXP>
#ifndef MPSFR_R16_WAVES
#define MPSFR_R16_WAVES 4
#endif
__global__ void __launch_bounds__(R16<N>::THREADS)
__attribute__((amdgpu_waves_per_eu(MPSFR_R16_WAVES, MPSFR_R16_WAVES)))
k_otf_r16(int ndir, int nl, const float* __restrict__ D0t, const float* __restrict__ telT,
          const LamPar* __restrict__ lp, const XUnit<R16<N>::RL>* __restrict__ xtab,
          cx<float>* __restrict__ Tq, const cx<double>* __restrict__ twg,
          const int* __restrict__ vkeep) {
    using P = R16<N>;
    using R = float;
    constexpr int TPR = P::TPR, LINES = P::LINES, NPAD = P::NPAD, RL = P::RL;
    // line pruning (stage_a.hip), decided per workgroup as in K_OTF_ROWFFT
    const int* vk_task = vkeep != nullptr ? vkeep + (size_t)blockIdx.y * ((nl + 1) / 2) : nullptr;
    const int v0 = blockIdx.x * LINES;
    const int ppg = otf_pairs_per_group(nl, vkeep != nullptr);
    const int l_beg = (int)blockIdx.z * 2 * ppg, l_end = min(nl, ((int)blockIdx.z + 1) * 2 * ppg);
    if (vk_task != nullptr && v0 >= vk_task[(l_end - 1) >> 1]) return;
    constexpr int NU = 4 * NSH;                          // units per line
    constexpr int ROUNDS = (NU + TPR - 1) / TPR;
    constexpr int XW = sizeof(XUnit<RL>) / 8;            // 8-byte pieces per table entry
    extern __shared__ __align__(16) unsigned char smem[];
    cx<R>* bufs = reinterpret_cast<cx<R>*>(smem);
    R* scl = reinterpret_cast<R*>(bufs + LINES * NPAD);           // [nl] exponent factors
    const int slot = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int v = blockIdx.x * LINES + slot;
    const int task = blockIdx.y;
    const bool valid = v <= N / 2;
    const int vv = valid ? v : N / 2;
    R16Tw<R, N> twr;
    twr.init(twg, t);
    for (int e = threadIdx.x; e < nl; e += P::THREADS) scl[e] = exp_scale<R, FASTEXP>((R)lp[e].c);
    __syncthreads();
    cx<R>* buf = bufs + slot * NPAD;
    constexpr bool FOLD = FASTEXP && ND == 1;
    R tel[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        tel[e] = telT[(size_t)vv * N + t + e * TPR];
        if constexpr (FOLD) tel[e] = __builtin_amdgcn_logf(tel[e]);      // log2
    }
    const R* dline = D0t + ((size_t)task * ndir * (N / 2 + 1) + vv) * N;
    const size_t dstride = (size_t)(N / 2 + 1) * N;
    R dreg[ND == 1 ? 16 : 1];
    if constexpr (ND == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dreg[e] = dline[t + e * TPR];
    }
    cx<R>* tq_line = Tq + (size_t)task * nl * tq_block<R>(N) + tq_index<R>(v, 0);
    // the thread's units: u = r TPR + t -> (which wavelength of the pair, sample i, half)
    int u_li[ROUNDS];                  // which * NSH + i, or -1 beyond the last unit
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = r * TPR + t;
        u_li[r] = u < NU ? (u >> 1) : -1;
    }
    const int uh = t & 1;
    const uint2* xt8 = reinterpret_cast<const uint2*>(xtab);
    // two wavelengths per complex transform: z = otf(la) + i otf(lb), both real lines
    for (int l = l_beg; l < l_end; l += 2) {
        if (vk_task != nullptr && v0 >= vk_task[l >> 1]) continue;
        const bool two = l + 1 < nl;
        const R ca = scl[l], cb = scl[two ? l + 1 : l];
        cx<R> x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            R ra = (R)0, rb = (R)0;
            if constexpr (FOLD) {
                if (MPSFR_R16_KNOCK == 2) {      // experiment: no exp
                    x[e] = {fmaf(ca, dreg[e], tel[e]), fmaf(cb, dreg[e], tel[e])};
                    continue;
                }
                x[e] = {exp_sel<R, true>(fmaf(ca, dreg[e], tel[e])),
                        two ? exp_sel<R, true>(fmaf(cb, dreg[e], tel[e])) : (R)0};
                continue;
            } else if constexpr (ND == 1) {
                ra = exp_sel<R, FASTEXP>(ca * dreg[e]);
                rb = exp_sel<R, FASTEXP>(cb * dreg[e]);
            } else {
                for (int d = 0; d < ndir; ++d) {
                    const R dv = dline[d * dstride + t + e * TPR];
                    ra += exp_sel<R, FASTEXP>(ca * dv);
                    rb += exp_sel<R, FASTEXP>(cb * dv);
                }
            }
            x[e] = {tel[e] * ra, two ? tel[e] * rb : (R)0};
        }
        r16_pass0<R, N>(x, buf, t);
#if MPSFR_R16_EARLY_TABLE
        // table entries of this transform's units, in flight behind the second pass
        uint2 xe[ROUNDS][XW];
        bool act[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int li = u_li[r];
            act[r] = valid && li >= 0 && (two || li < NSH) && MPSFR_R16_KNOCK != 1;
            const size_t e = ((size_t)l * NSH + (act[r] ? li : 0)) * 2 + uh;       // li runs into l + 1
#pragma unroll
            for (int k = 0; k < XW; ++k) xe[r][k] = xt8[e * XW + k];
        }
        r16_pass1<R, N>(buf, twr.w, t);
#else
        r16_pass1<R, N>(buf, twr.w, t);
        // table entries of this transform's units: all rounds in flight at once (issued before the
        // second pass they cost more in spilled registers than the hidden latency is worth)
        uint2 xe[ROUNDS][XW];
        bool act[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int li = u_li[r];
            act[r] = valid && li >= 0 && (two || li < NSH) && MPSFR_R16_KNOCK != 1;
            const size_t e = ((size_t)l * NSH + (act[r] ? li : 0)) * 2 + uh;       // li runs into l + 1
#pragma unroll
            for (int k = 0; k < XW; ++k) xe[r][k] = xt8[e * XW + k];
        }
#endif
        if (MPSFR_R16_KNOCK == 1) {      // experiment: no extraction
            if (valid && t < NSH) tq_line[(size_t)l * tq_block<R>(N) + t] = buf[t * 7];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int kz = xe[r][0].x & 0xffff, kmz = xe[r][0].x >> 16;
            const R c = __builtin_bit_cast(float, xe[r][0].y);
            cx<R> zz = buf[kz], zm = buf[kmz];
            auto mad = [](cx<R> e, uint2 b, cx<R> o) -> cx<R> {     // e + b o
                const R bx = __builtin_bit_cast(float, b.x), by = __builtin_bit_cast(float, b.y);
                return {fmaf(-by, o.y, fmaf(bx, o.x, e.x)), fmaf(by, o.x, fmaf(bx, o.y, e.y))};
            };
#pragma unroll
            for (int q = 1; q < RL; ++q) {
                zz = mad(zz, xe[r][q], buf[kz + 256 * q]);
                zm = mad(zm, xe[r][RL - 1 + q], buf[kmz + 256 * q]);
            }
            // first wavelength: F = (Z[z] + conj Z[-z]) / 2; second: (Z[z] - conj Z[-z]) / 2i
            const bool second = u_li[r] >= NSH;
            R tx = second ? c * (zz.y + zm.y) : c * (zz.x + zm.x);
            R ty = second ? -c * (zz.x - zm.x) : c * (zz.y - zm.y);
            // the other half of the sample sits in the neighbouring lane (quad_perm [1, 0, 3, 2])
            tx += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                                0, __builtin_bit_cast(int, tx), 0xB1, 0xf, 0xf, false));
            ty += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                                0, __builtin_bit_cast(int, ty), 0xB1, 0xf, 0xf, false));
            if (act[r] && uh == 0) {
                const int li = u_li[r];                 // which * NSH + i: runs into wavelength l + 1
                const int which = li >= NSH ? 1 : 0;
                tq_line[(size_t)(l + which) * tq_block<R>(N) + (li - which * NSH)] = {tx, ty};
            }
        }
        fft_sync<true>();     // extraction reads done before the next transform overwrites
    }
}

// ------------------------------------------------------------------------------------------
// K_COLPASS: second (column) pass restricted to the sampled positions,
// stamp[i][j] = sum_v Re(G[l][v][j] * conj(Tq[v][i])), then clamp >= 0 (psfrec.py:680) and
// normalise to sum 1 (:685).  With Tq[v][40-i] = conj Tq[v][i] only i = 0..20 is stored and
//   P[i][j] = sum_v Gx Tx,  Q[i][j] = sum_v Gy Ty,  stamp[i][j] = P + Q,  stamp[40-i][j] = P - Q,
// i.e. half the products of the plain form.  One workgroup per stamp; a lane holds a 3x5 tile
// of (P, Q) pairs (7 x 8 tiles = 56 lanes), the four waves split the v range (32 lines staged
// in LDS per step, 8 per wave) and the partial tiles are summed through LDS at the end.
// ------------------------------------------------------------------------------------------
template <typename R, int N>
__global__ void __launch_bounds__(256)
k_colpass(int nl, const cx<R>* __restrict__ Tq, const cx<R>* __restrict__ G,
          R* __restrict__ pre, const int* __restrict__ vkeep) {
    constexpr int VW = 8, VB = 4 * VW, TI = 3, TJ = 5, NV = N / 2 + 1;
    constexpr int NPQ = NSH * NS;                 // 840 (P, Q) pairs
    static_assert(NSH == 7 * TI && NS == 8 * TJ, "tile map");
    // one raw buffer: staging {T [VB][NSH], G [VB][NS]} complex during the loop, then the
    // partial tiles [4][NPQ] complex (P, Q)
    constexpr size_t STAGE = (size_t)VB * (NSH + NS) * sizeof(cx<R>);
    constexpr size_t RED = (size_t)4 * NPQ * sizeof(cx<R>);
    __shared__ __align__(16) unsigned char raw[RED > STAGE ? RED : STAGE];
    cx<R>(*sT)[NSH] = reinterpret_cast<cx<R>(*)[NSH]>(raw);
    cx<R>(*sG)[NS] = reinterpret_cast<cx<R>(*)[NS]>(raw + VB * NSH * sizeof(cx<R>));
    __shared__ double part[4];
    __shared__ double tot;
    const int l = blockIdx.x, task = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool act = lane < 56;
    const int i0 = TI * (act ? lane >> 3 : 0), j0 = TJ * (lane & 7);
    const cx<R>* Tp = Tq + ((size_t)task * nl + l) * tq_block<R>(N);
    const cx<R>* Gp = G + (size_t)l * g_lines(N) * NS;
    // line pruning (stage_a.hip): lines at and beyond vkeep[task][pair] were never written
    const int nv = vkeep != nullptr ? min(NV, vkeep[(size_t)task * ((nl + 1) / 2) + (l >> 1)]) : NV;
    R accP[TI][TJ], accQ[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) { accP[a][b] = (R)0; accQ[a][b] = (R)0; }
    constexpr int NET = (VB * NSH + 255) / 256, NEG = VB * NS / 256;
    cx<R> rt[NET], rg[NEG];
    auto fetch = [&](int v0) {
#pragma unroll
        for (int k = 0; k < NET; ++k) {
            const int e = threadIdx.x + k * 256;
            rt[k] = (e < VB * NSH && v0 * NSH + e < nv * NSH) ? Tp[(size_t)v0 * NSH + e]
                                                            : cx<R>{(R)0, (R)0};
        }
#pragma unroll
        for (int k = 0; k < NEG; ++k) {
            const int e = threadIdx.x + k * 256;
            rg[k] = v0 * NS + e < NV * NS ? Gp[(size_t)v0 * NS + e] : cx<R>{(R)0, (R)0};
        }
    };
    fetch(0);
    for (int v0 = 0; v0 < nv; v0 += VB) {
#pragma unroll
        for (int k = 0; k < NET; ++k) {
            const int e = threadIdx.x + k * 256;
            if (e < VB * NSH) (&sT[0][0])[e] = rt[k];
        }
#pragma unroll
        for (int k = 0; k < NEG; ++k) (&sG[0][0])[threadIdx.x + k * 256] = rg[k];
        __syncthreads();
        if (v0 + VB < nv) fetch(v0 + VB);      // prefetch the next block behind the FMAs
#pragma unroll
        for (int vb = 0; vb < VW; ++vb) {
            const int vr = wave * VW + vb;
            cx<R> t[TI], g[TJ];
#pragma unroll
            for (int k = 0; k < TI; ++k) t[k] = sT[vr][i0 + k];
#pragma unroll
            for (int k = 0; k < TJ; ++k) g[k] = sG[vr][j0 + k];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    accP[a][b] += g[b].x * t[a].x;
                    accQ[a][b] += g[b].y * t[a].y;
                }
        }
        __syncthreads();
    }
    // sum the four partial tiles
    cx<R>* red = reinterpret_cast<cx<R>*>(raw);
    if (act) {
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b)
                red[wave * NPQ + (i0 + a) * NS + j0 + b] = {accP[a][b], accQ[a][b]};
    }
    __syncthreads();
    constexpr int NO = (NS * NS + 255) / 256;
    R val[NO];
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < NO; ++m) {
        const int o = threadIdx.x + m * 256;
        R x = (R)0;
        if (o < NS * NS) {
            const int i = o / NS, j = o - i * NS;
            const int e = (i < NSH ? i : NS - i) * NS + j;
            const cx<R> a = red[e], b = red[NPQ + e], c = red[2 * NPQ + e], d = red[3 * NPQ + e];
            const R P = (a.x + b.x) + (c.x + d.x), Q = (a.y + b.y) + (c.y + d.y);
            x = i < NSH ? P + Q : P - Q;
            if (x < (R)0) x = (R)0;
            s += (double)x;
        }
        val[m] = x;
    }
    s = wave_sum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) tot = (part[0] + part[1]) + (part[2] + part[3]);
    __syncthreads();
    const double inv = 1.0 / tot;
    R* out = pre + ((size_t)task * nl + l) * NS * NS;
#pragma unroll
    for (int m = 0; m < NO; ++m) {
        const int o = threadIdx.x + m * 256;
        if (o < NS * NS) out[o] = (R)((double)val[m] * inv);
    }
}


// K_COLPASS, fp32 variant on the matrix cores.  The second pass is a dense contraction over the
// lines v,  P[(task,i)][j] = sum_v Tx[task][v][i] Gx[v][j]  (Q likewise with the imaginary parts),
// and what bounded the tiled variant above was not the FMAs but handing every lane its operands
// (8 ds_read_b64 per 30 FMAs: LDS pipe 67 % busy, VALU 33 %).  v_mfma_f32_16x16x4_f32 runs at the
// same 64 FLOP/clk/SIMD as v_fma_f32 with exact f32 products (a k-ordered fmaf chain), but takes
// each operand element from ONE lane: lane l supplies A[row l&15][k l>>4] and B[k l>>4][col l&15],
// both straight from global memory -- no LDS, no broadcast.
//   rows : three tasks x 21 samples = 63 of 64 (four 16-row tiles)
//   cols : 40 of 48 (three 16-column tiles)
//   k    : four lines per instruction; each of the four waves owns one row tile over all lines,
//          so there is no cross-wave reduction -- LDS only gathers the finished tiles.
// 82 % of the multiplies are useful; P and Q share their operand loads (T and G are complex).
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MPSFR_COLPASS_WAVES
#define MPSFR_COLPASS_WAVES 4
#endif
template <int N>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MPSFR_COLPASS_WAVES)))
k_colpass_m(int ntask, int nl, const cx<float>* __restrict__ Tq, const cx<float>* __restrict__ G,
            float* __restrict__ pre, const int* __restrict__ vkeep) {
    constexpr int NV = N / 2 + 1, TPG = 3, MT = 4, NT = 3, NCOL = NT * 16;
    static_assert(TPG * NSH <= MT * 16 && NS <= NCOL, "tile map");
    __shared__ float red[2][MT * 16][NCOL];  // P, Q
    __shared__ double part[4][TPG];
    const int l = blockIdx.x, tg = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    // wave w owns row tile w (all lines, all columns): no cross-wave reduction
    const int r = 16 * wave + lr, tl = r / NSH, i = r - tl * NSH;
    const bool aok = tl < TPG && tg * TPG + tl < ntask;
    // Line pruning (stage_a.hip): lines at and beyond vkeep[task][pair] were never written; a row
    // reads zeros there, and the loop ends at the largest vkeep of the group's tasks.
    const int npair = (nl + 1) / 2;
    const int nv_row = (vkeep != nullptr && aok) ? vkeep[(tg * TPG + tl) * npair + (l >> 1)] : NV;
    int nv_grp = NV;
    if (vkeep != nullptr) {
        nv_grp = 0;
        for (int k = 0; k < TPG; ++k)
            if (tg * TPG + k < ntask) nv_grp = max(nv_grp, vkeep[(tg * TPG + k) * npair + (l >> 1)]);
    }
    // rows without a task read the group's first task (valid memory); they are never used
    const cx<float>* ap = Tq + ((size_t)(aok ? tg * TPG + tl : tg * TPG) * nl + l) * tq_block<float>(N) +
                          (aok ? i : 0);
    // B: G in the paired-line layout of g_index<float>: one 16-byte load carries this lane's
    // element for k-steps 2 s and 2 s + 1 (lines 8 s + lk and 8 s + 4 + lk)
    constexpr int NG = g_lines(N) / 8;                   // double steps
    const f32x4* bp[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int j = 16 * ct + lr;
        bp[ct] = reinterpret_cast<const f32x4*>(G) + (size_t)l * NG * 4 * NS + (size_t)lk * NS +
                 (j < NS ? j : 0);                       // columns 40..47 are dropped
    }
    f32x4 accP[NT], accQ[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        accP[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        accQ[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    struct Ops {
        cx<float> a0, a1;       // T[8 s + lk][i], T[8 s + 4 + lk][i]
        f32x4 b[NT];
    };
    auto load = [&](int s, Ops& o) {
        const int sc = s < NG ? s : NG - 1;              // past the end: any valid address
        const int v0 = 8 * sc + lk, v1 = v0 + 4;
        const cx<float> t0 = ap[(size_t)(v0 < NV ? v0 : NV - 1) * NSH];
        const cx<float> t1 = ap[(size_t)(v1 < NV ? v1 : NV - 1) * NSH];
        o.a0 = (s < NG && v0 < nv_row) ? t0 : cx<float>{0.f, 0.f};     // padding / pruned: A = 0
        o.a1 = (s < NG && v1 < nv_row) ? t1 : cx<float>{0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) o.b[ct] = bp[ct][(size_t)sc * 4 * NS];
    };
    auto mma = [&](const Ops& o) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            accP[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0.x, o.b[ct][0], accP[ct], 0, 0, 0);
            accQ[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0.y, o.b[ct][1], accQ[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            accP[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1.x, o.b[ct][2], accP[ct], 0, 0, 0);
            accQ[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1.y, o.b[ct][3], accQ[ct], 0, 0, 0);
        }
    };
    // two operand sets, roles swapped inside the body (no register rotation at the back edge);
    // the loop is branch-free: double steps past the end load valid memory with A = 0.  (A ring
    // of four sets -- three double steps of loads in flight -- measured no faster.)
    Ops oa, ob;
    load(0, oa);
    const int ng_used = min(NG, (nv_grp + 7) / 8);
    for (int s = 0; s < ng_used; s += 2) {
        load(s + 1, ob);
        mma(oa);
        load(s + 2, oa);
        mma(ob);
    }
    // C layout of the 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            red[0][16 * wave + 4 * lk + r4][16 * ct + lr] = accP[ct][r4];
            red[1][16 * wave + 4 * lk + r4][16 * ct + lr] = accQ[ct][r4];
        }
    __syncthreads();
    const int ng = min(TPG, ntask - tg * TPG);
    constexpr int NO = (NS * NS + 255) / 256;
    float val[TPG][NO];
    double s[TPG];
#pragma unroll
    for (int k = 0; k < TPG; ++k) {
        s[k] = 0.0;
#pragma unroll
        for (int m = 0; m < NO; ++m) {
            const int o = threadIdx.x + m * 256;
            float x = 0.f;
            if (k < ng && o < NS * NS) {
                const int row = o / NS, j = o - row * NS;
                const int r = k * NSH + (row < NSH ? row : NS - row);
                const float P = red[0][r][j], Q = red[1][r][j];
                x = fmaxf(row < NSH ? P + Q : P - Q, 0.f);        // clamp: psfrec.py:680
                s[k] += (double)x;
            }
            val[k][m] = x;
        }
        s[k] = wave_sum(s[k]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < TPG; ++k) part[wave][k] = s[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TPG; ++k) {
        if (k >= ng) break;
        const double inv = 1.0 / ((part[0][k] + part[1][k]) + (part[2][k] + part[3][k]));   // :685
        float* out = pre + ((size_t)(tg * TPG + k) * nl + l) * NS * NS;
#pragma unroll
        for (int m = 0; m < NO; ++m) {
            const int o = threadIdx.x + m * 256;
            if (o < NS * NS) out[o] = (float)((double)val[k][m] * inv);
        }
    }
}

}  // namespace

void launch_gtable(hipStream_t s, int N, int nl, const LamPar* d_lp, const void* d_tw64,
                   int* d_samp_p, void* d_samp_a, void* d_G, bool f64) {
    dim3 grid((g_lines(N) * NS + 255) / 256, nl);
    if (f64)
        hipLaunchKernelGGL(k_gtable<double>, grid, dim3(256), 0, s, N, d_lp,
                           (const cx<double>*)d_tw64, d_samp_p, (double*)d_samp_a,
                           (cx<double>*)d_G);
    else
        hipLaunchKernelGGL(k_gtable<float>, grid, dim3(256), 0, s, N, d_lp,
                           (const cx<double>*)d_tw64, d_samp_p, (float*)d_samp_a, (cx<float>*)d_G);
}

template <typename R, int NN, int ND, bool FE, bool LTAB>
static void launch_otf_tt(hipStream_t s, int ntask, int ndir, int nl, const void* d_D0t,
                          const void* d_tel, const LamPar* d_lp, const int* d_samp_p,
                          const void* d_samp_a, void* d_Tq, const void* d_tw64, const int* d_vkeep) {
    constexpr int SL = LineCfg<NN>::SLOTS;
#ifndef MPSFR_OTF_EXTRA_LDS
#define MPSFR_OTF_EXTRA_LDS 0
#endif
    const size_t sm = fft_smem<R, NN>(!use_reg_twiddles<NN>(), fft_nbuf<NN>()) + MPSFR_OTF_EXTRA_LDS +
                      (LTAB ? (size_t)nl * (NSH * (sizeof(SampOff) + sizeof(R)) + sizeof(R)) : 0);
    allow_smem(k_otf_rowfft<R, NN, ND, FE, LTAB>, sm);
    const int ppg = otf_pairs_per_group(nl, d_vkeep != nullptr);
    dim3 grid((NN / 2 + 1 + SL - 1) / SL, ntask, ((nl + 1) / 2 + ppg - 1) / ppg);
    hipLaunchKernelGGL((k_otf_rowfft<R, NN, ND, FE, LTAB>), grid, dim3(LineCfg<NN>::THREADS), sm, s,
                       ndir, nl, (const R*)d_D0t, (const R*)d_tel, d_lp, d_samp_p,
                       (const R*)d_samp_a, (cx<R>*)d_Tq, (const cx<double>*)d_tw64, d_vkeep);
}

template <typename R, int NN, int ND, bool FE>
static void launch_otf_t(hipStream_t s, int ntask, int ndir, int nl, const void* d_D0t,
                         const void* d_tel, const LamPar* d_lp, const int* d_samp_p,
                         const void* d_samp_a, void* d_Tq, const void* d_tw64, const int* d_vkeep) {
#ifndef MPSFR_OTF_LDSTAB
#define MPSFR_OTF_LDSTAB 1
#endif
    if (MPSFR_OTF_LDSTAB && nl <= kOtfLdsTabMaxNl)
        launch_otf_tt<R, NN, ND, FE, true>(s, ntask, ndir, nl, d_D0t, d_tel, d_lp, d_samp_p,
                                           d_samp_a, d_Tq, d_tw64, d_vkeep);
    else
        launch_otf_tt<R, NN, ND, FE, false>(s, ntask, ndir, nl, d_D0t, d_tel, d_lp, d_samp_p,
                                            d_samp_a, d_Tq, d_tw64, d_vkeep);
}

template <int NN, int ND, bool FE>
static void launch_otf_r16_t(hipStream_t s, int ntask, int ndir, int nl, const void* d_D0t,
                             const void* d_tel, const LamPar* d_lp, const void* d_xtab, void* d_Tq,
                             const void* d_tw64, const int* d_vkeep) {
    using P = R16<NN>;
#ifndef MPSFR_R16_EXTRA_LDS
#define MPSFR_R16_EXTRA_LDS 0
#endif
    const size_t sm = (size_t)P::LINES * P::NPAD * sizeof(cx<float>) + (size_t)nl * sizeof(float) +
                      MPSFR_R16_EXTRA_LDS;
    allow_smem(k_otf_r16<NN, ND, FE>, sm);
    const int ppg = otf_pairs_per_group(nl, d_vkeep != nullptr);
    dim3 grid((NN / 2 + 1 + P::LINES - 1) / P::LINES, ntask, ((nl + 1) / 2 + ppg - 1) / ppg);
    hipLaunchKernelGGL((k_otf_r16<NN, ND, FE>), grid, dim3(P::THREADS), sm, s, ndir, nl,
                       (const float*)d_D0t, (const float*)d_tel, d_lp,
                       (const XUnit<R16<NN>::RL>*)d_xtab, (cx<float>*)d_Tq, (const cx<double>*)d_tw64,
                       d_vkeep);
}

#ifndef MPSFR_OTF_R16
#define MPSFR_OTF_R16 1
#endif
// Measured (MI355X, K_OTF time per 100 rows): 512^2 x 35 lambda 213 us against 221 us for the
// radix-8 plan; 1024^2 x 70 lambda 1.58 ms against 3.07 ms (8.8.4.4 needs two wavefronts per line
// and four crossings); 256^2 is slower with radix 16 (16 lanes per line: six extraction rounds) and
// keeps the 8.8.4 plan; so does 512^2 with several directions (the kernel is then bound by the
// nine exp per point, and the radix-8 kernel runs five waves per SIMD against four).
bool otf_uses_r16(int N, bool f64, int nl, int ndir) {
    return MPSFR_OTF_R16 && !f64 && ((N == 512 && ndir == 1) || N == 1024) && nl <= 4096;
}

// one table entry per (wavelength, sample, half) plus one wavelength of slack (the units of an
// odd last wavelength read ahead)
size_t xtab_bytes(int nl) { return (size_t)(nl + 1) * NSH * 2 * sizeof(XUnit<4>); }

void launch_xtab(hipStream_t s, int N, int nl, const int* d_samp_p, const void* d_samp_a,
                 const void* d_tw64, void* d_xtab) {
    const dim3 grid((nl * NSH * 2 + 255) / 256);
    if (N == 256)
        hipLaunchKernelGGL(k_xtab<1>, grid, dim3(256), 0, s, N, nl, d_samp_p, (const float*)d_samp_a,
                           (const cx<double>*)d_tw64, (XUnit<1>*)d_xtab);
    else if (N == 512)
        hipLaunchKernelGGL(k_xtab<2>, grid, dim3(256), 0, s, N, nl, d_samp_p, (const float*)d_samp_a,
                           (const cx<double>*)d_tw64, (XUnit<2>*)d_xtab);
    else
        hipLaunchKernelGGL(k_xtab<4>, grid, dim3(256), 0, s, N, nl, d_samp_p, (const float*)d_samp_a,
                           (const cx<double>*)d_tw64, (XUnit<4>*)d_xtab);
}

#define OTF_ARGS s, ntask, ndir, nl, d_D0t, d_tel, d_lp, d_samp_p, d_samp_a, d_Tq, d_tw64, d_vkeep
#define R16_ARGS s, ntask, ndir, nl, d_D0t, d_tel, d_lp, d_xtab, d_Tq, d_tw64, d_vkeep
void launch_otf_rowfft(hipStream_t s, int N, int ntask, int ndir, int nl, const void* d_D0t,
                       const void* d_tel, const LamPar* d_lp, const int* d_samp_p,
                       const void* d_samp_a, const void* d_xtab, void* d_Tq, const void* d_tw64,
                       bool f64, bool fast_exp, const int* d_vkeep) {
    if (otf_uses_r16(N, f64, nl, ndir)) {
#define R16_CASE(NN_)                                                                     \
    if (fast_exp) {                                                                       \
        if (ndir == 1) launch_otf_r16_t<NN_, 1, true>(R16_ARGS);                          \
        else launch_otf_r16_t<NN_, 0, true>(R16_ARGS);                                    \
    } else {                                                                              \
        if (ndir == 1) launch_otf_r16_t<NN_, 1, false>(R16_ARGS);                         \
        else launch_otf_r16_t<NN_, 0, false>(R16_ARGS);                                   \
    }
        if (N == 256) { R16_CASE(256) } else if (N == 512) { R16_CASE(512) } else { R16_CASE(1024) }
#undef R16_CASE
        return;
    }
    DISPATCH_N(N, {
        if (f64) {
            if (ndir == 1) launch_otf_t<double, NN, 1, false>(OTF_ARGS);
            else launch_otf_t<double, NN, 0, false>(OTF_ARGS);
        } else if (fast_exp) {
            if (ndir == 1) launch_otf_t<float, NN, 1, true>(OTF_ARGS);
            else launch_otf_t<float, NN, 0, true>(OTF_ARGS);
        } else {
            if (ndir == 1) launch_otf_t<float, NN, 1, false>(OTF_ARGS);
            else launch_otf_t<float, NN, 0, false>(OTF_ARGS);
        }
    })
}
#undef OTF_ARGS
#undef R16_ARGS

void launch_colpass(hipStream_t s, int N, int ntask, int nl, const void* d_Tq, const void* d_G,
                    void* d_pre, bool f64, const int* d_vkeep) {
    dim3 grid(nl, ntask);
    DISPATCH_N(N, {
        if (f64)
            hipLaunchKernelGGL((k_colpass<double, NN>), grid, dim3(256), 0, s, nl,
                               (const cx<double>*)d_Tq, (const cx<double>*)d_G, (double*)d_pre, d_vkeep);
        else
            hipLaunchKernelGGL((k_colpass_m<NN>), dim3(nl, (ntask + 2) / 3), dim3(256), 0, s, ntask,
                               nl, (const cx<float>*)d_Tq, (const cx<float>*)d_G, (float*)d_pre, d_vkeep);
    })
}


}  // namespace mpsfr
