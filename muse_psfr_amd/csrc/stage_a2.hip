// Stage A in its "series + patch" form (round 4): the structure function D_phi0 of a task without any
// full-size transform.  Reference: simul_psd_wfm (psfrec.py:36-151), psd_fit (:616-626) and the
// structure function of psd_to_psf (:717-722).
//
// The residual PSD of a task is  PSD = F + P  with
//   F = the fitting term  cfit r0^(-5/3) (f^2 + 1/L0^2)^(-11/6) [f >= fc]  on the whole half-pixel
//       grid (psfrec.py:616-626), and
//   P = max(F, AO) - F >= 0, which lives in the 80 x 80 corrected zone only (psfrec.py:148-149).
// The transform is linear, so  D = D_F + D_P.
//
//  * D_F.  With eps = 1/L0^2, eps0 = 1/128 and delta = eps - eps0 (|delta| <= 0.0126 for L0 >= 7 m,
//    against f^2 + eps0 >= 2.2578 on the support of F),
//        (f^2 + eps)^(-11/6) = sum_k binom(-11/6, k) delta^k (f^2 + eps0)^(-11/6 - k)
//    converges by a factor <= 0.0056 per term.  The structure functions Hd_k of the terms depend on the
//    grid alone: they are computed ONCE PER CONTEXT with the full-size fp64 transforms of stage_a.hip
//    ("basis" tasks), and a task's D_F = r0^(-5/3) sum_k delta^k Hd_k is a polynomial per pixel.  D_F is
//    1-2 % of D (the fitting error saturates at small separations), and every term is a structure
//    function of a non-negative PSD -- no cancellation is left at run time -- so the mixed mode
//    evaluates it in fp32 from fp32 tables (4 terms: truncation 4e-9 of D_F); the f64 mode keeps 8
//    fp64 terms (6e-18).
//  * D_P = 2 scale (sum P - Re FFT2(P)) in fp64, as a PRUNED transform: 80 x 80 inputs, (N/2+1) x N
//    outputs.  Row pass (K_PATCH_ROWS): T[su][y] = sum_sv P[su][sv] W^(sv y).  Column pass
//    (K_DPHI_SERIES): with x = 64 k1 + k2, Q = N / 64 and su = r + Q j,
//        X[64 k1 + k2] = sum_r W_Q^(r k1) S_r[k2],   S_r[k2] = W_N^(r k2) sum_j T[r + Q j] W_64^(j k2):
//    a lane (k2) folds the 80 inputs into Q sums and runs a Q-point transform in its own registers;
//    its Q outputs are x = k2, 64 + k2, ...: for every k1 the 64 lanes store one 256-byte piece of the
//    line.  No LDS pass, no barrier, no workspace of N^2 size: the 7 MB per task of row transforms
//    (1280^2) become 0.8 MB, and the kernel is bound by its fp64 multiply-adds (~30 per pixel).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "device_common.h"
#include "dpp_groups.h"
#include "psd_model.h"

namespace mpsfr {

namespace {

constexpr double kEps0 = 1.0 / 128.0;      // expansion point of 1/L0^2 (L0 = 11.3 m)

// W_Q^m = exp(-2 pi i m / Q), the constant twiddles of the in-lane transforms (m <= 3 (Q/4 - 1))
template <int Q> struct WQ;
template <> struct WQ<8> {
    static constexpr double c[4] = {1.0, 0.70710678118654752440, 0.0, -0.70710678118654752440};
    static constexpr double s[4] = {0.0, -0.70710678118654752440, -1.0, -0.70710678118654752440};
};
template <> struct WQ<16> {
    static constexpr double c[10] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
                                     0.0, -0.38268343236508977173, -0.70710678118654752440,
                                     -0.92387953251128675613, -1.0, -0.92387953251128675613};
    static constexpr double s[10] = {0.0, -0.38268343236508977173, -0.70710678118654752440,
                                     -0.92387953251128675613, -1.0, -0.92387953251128675613,
                                     -0.70710678118654752440, -0.38268343236508977173, 0.0,
                                     0.38268343236508977173};
};
template <> struct WQ<20> {
    static constexpr double c[13] = {1.0, 0.95105651629515357212, 0.80901699437494742410,
                                     0.58778525229247312917, 0.30901699437494742410, 0.0,
                                     -0.30901699437494742410, -0.58778525229247312917,
                                     -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                     -0.95105651629515357212, -0.80901699437494742410};
    static constexpr double s[13] = {0.0, -0.30901699437494742410, -0.58778525229247312917,
                                     -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                     -0.95105651629515357212, -0.80901699437494742410,
                                     -0.58778525229247312917, -0.30901699437494742410, 0.0,
                                     0.30901699437494742410, 0.58778525229247312917};
};
template <> struct WQ<4> {
    static constexpr double c[1] = {1.0};
    static constexpr double s[1] = {0.0};
};

template <int M>
__device__ __forceinline__ void dftm(cx<double>* v) {
    if constexpr (M == 2) dft2(v[0], v[1]);
    if constexpr (M == 4) dft4(v[0], v[1], v[2], v[3]);
    if constexpr (M == 5) dft5(v);
}

// terms of the fold: su = r + Q j in [-40, 40)
template <int Q> constexpr int fold_jmin() { return -((NAO / 2 + Q - 1) / Q); }
template <int Q> constexpr int fold_jmax() { return (NAO / 2 + Q - 1) / Q; }      // exclusive
template <int Q> constexpr int fold_nj() { return fold_jmax<Q>() - fold_jmin<Q>(); }
constexpr bool fold_valid(int Q, int r, int j) { return r + Q * j >= -NAO / 2 && r + Q * j < NAO / 2; }

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Re X[k1], X[k1] = sum_r S_r W_Q^(r k1), for k1 = 0..Q-1.  Q = 4 M: r = 4 r2 + r1, k1 = M a + b;
// M-point transforms over r2, the twiddles W_Q^(r1 b), and the last radix-4 stage on real parts only.
// G(r1, v) (r1 a std::integral_constant) fills v[r2] = S_(4 r2 + r1): the caller produces the sums of
// one r1 at a time, so that at Q = 20 only 5 of the 20 are alive beside the partial results.
template <int Q, typename F>
__device__ __forceinline__ void dftq_real_out(F&& G, double* out) {
    static_assert(Q % 4 == 0, "Q = 2 is handled by the caller");
    constexpr int M = Q / 4;
    double t0[M], t1[M], t2[M], t3[M];
    {
        cx<double> v[M];
        G(std::integral_constant<int, 0>{}, v);
        dftm<M>(v);
#pragma unroll
        for (int b = 0; b < M; ++b) t0[b] = v[b].x;
    }
    {
        cx<double> v[M];
        G(std::integral_constant<int, 2>{}, v);
        dftm<M>(v);
#pragma unroll
        for (int b = 0; b < M; ++b) {
            const double re = b == 0 ? v[b].x : v[b].x * WQ<Q>::c[2 * b] - v[b].y * WQ<Q>::s[2 * b];
            t1[b] = t0[b] - re;
            t0[b] = t0[b] + re;
        }
    }
    {
        cx<double> v[M];
        G(std::integral_constant<int, 1>{}, v);
        dftm<M>(v);
#pragma unroll
        for (int b = 0; b < M; ++b) {
            if (b == 0) {
                t2[b] = v[b].x;
                t3[b] = v[b].y;
            } else {
                t2[b] = v[b].x * WQ<Q>::c[b] - v[b].y * WQ<Q>::s[b];
                t3[b] = v[b].x * WQ<Q>::s[b] + v[b].y * WQ<Q>::c[b];
            }
        }
    }
    {
        cx<double> v[M];
        G(std::integral_constant<int, 3>{}, v);
        dftm<M>(v);
#pragma unroll
        for (int b = 0; b < M; ++b) {
            if (b == 0) {
                t2[b] += v[b].x;
                t3[b] -= v[b].y;
            } else {
                t2[b] += v[b].x * WQ<Q>::c[3 * b] - v[b].y * WQ<Q>::s[3 * b];
                t3[b] -= v[b].x * WQ<Q>::s[3 * b] + v[b].y * WQ<Q>::c[3 * b];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < M; ++b) {
        out[b] = t0[b] + t2[b];
        out[2 * M + b] = t0[b] - t2[b];
        out[M + b] = t1[b] + t3[b];
        out[3 * M + b] = t1[b] - t3[b];
    }
}

// 16-point transform in registers (natural order): 4 x 4
__device__ __forceinline__ void dft16(cx<double>* v) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
        for (int n2 = 1; n2 < 4; ++n2) {
            const cx<double> w = {WQ<16>::c[n2 * k1], WQ<16>::s[n2 * k1]};
            v[4 * k1 + n2] = cmul(v[4 * k1 + n2], w);
        }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    cx<double> o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[i];
}

template <int Q>
__device__ __forceinline__ void dftq(cx<double>* v) {
    if constexpr (Q == 16) dft16(v);
    else dftr<double, Q>(v);
}

// ------------------------------------------------------------------------------------------
// Broadcast operands.  Every lane of a wave needs the same 80 inputs of a fold, each multiplied by the
// lane's own twiddle.  They travel in 5 register pairs per component -- lane l holds input
// 16 a + (l mod 16) in pair a, the same in all four rows of 16 lanes -- and the multiply-add picks its
// lane with the DPP control row_newbcast (the one DPP control gfx90a+ has for 64-bit operations):
// no LDS read, no scalar load, no extra instruction per operand.  (Through the scalar cache, the
// first form of this kernel, the 1280 bytes per line and task were 26 s_load per 440 vector
// instructions and the waves waited 68 % of their cycles for them.)
// hipcc does not pad the wait states of an instruction inside an asm statement: a VGPR written by a
// vector instruction must not be read through DPP within 2 wait states.  The operands below come out
// of global loads; tools/isa_lint.py (R6) checks every DPP instruction of the build.
// ------------------------------------------------------------------------------------------
template <int LANE>
__device__ __forceinline__ double mov_bc(double x) {                             // x[LANE]
    double d;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(x), "n"(LANE));
    return d;
}

constexpr int kNX = NAO / 16;        // register pairs per component of a fold's 80 inputs

// The sums of a group of CNT residues r_i = R0 + RS i:  acc[i] = sum_j in[r_i + Q j] W_64^(j k2)
// (the factor W_N^(r k2) is the caller's).  xr / xi: the inputs as broadcast operands (index
// su + 40; xi unused for a real input), wj[j - JMIN] = W_64^(j k2).  Term by term over j, all
// residues of the group in one multiply-add block (dpp_groups.h): the accumulators of the group are
// the independent chains that cover the latency of the fp64 DPP multiply-add.
template <int Q, int R0, int RS, int CNT, int J>
struct FoldValid {
    int n = 0, idx[CNT > 0 ? CNT : 1] = {};
    constexpr FoldValid() {
        for (int i = 0; i < CNT; ++i)
            if (fold_valid(Q, R0 + RS * i, J)) idx[n++] = i;
    }
};

template <int Q, bool CPLX, int R0, int RS, int CNT, int J = fold_jmin<Q>()>
__device__ __forceinline__ void fold_group(cx<double>* acc, const double* xr, const double* xi,
                                           const cx<double>* wj) {
    if constexpr (J < fold_jmax<Q>()) {
        constexpr FoldValid<Q, R0, RS, CNT, J> V;
#define MPSFR_FN(k) (R0 + RS * V.idx[k] + Q * J + NAO / 2)      /* input index of the k-th valid residue */
#define MPSFR_FA(k) acc[V.idx[k]]
#define MPSFR_FX(k) xr[MPSFR_FN(k) / 16], xi[MPSFR_FN(k) / 16]
#define MPSFR_FR(k) xr[MPSFR_FN(k) / 16]
#define MPSFR_FL(k) MPSFR_FN(k) % 16
        if constexpr (J == 0) {          // the twiddle is 1
            static_for<0, V.n>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                acc[V.idx[k]].x += mov_bc<MPSFR_FL(k)>(xr[MPSFR_FN(k) / 16]);
                if constexpr (CPLX) acc[V.idx[k]].y += mov_bc<MPSFR_FL(k)>(xi[MPSFR_FN(k) / 16]);
            });
        } else {
            const cx<double> w = wj[J - fold_jmin<Q>()];
            constexpr int n = V.n;
            static_assert(n <= 5, "group sizes up to 5");
            if constexpr (CPLX) {
                if constexpr (n == 5)
                    cmac_group5<MPSFR_FL(0), MPSFR_FL(1), MPSFR_FL(2), MPSFR_FL(3), MPSFR_FL(4)>(
                        MPSFR_FA(0), MPSFR_FA(1), MPSFR_FA(2), MPSFR_FA(3), MPSFR_FA(4), MPSFR_FX(0), MPSFR_FX(1),
                        MPSFR_FX(2), MPSFR_FX(3), MPSFR_FX(4), w);
                if constexpr (n == 4)
                    cmac_group4<MPSFR_FL(0), MPSFR_FL(1), MPSFR_FL(2), MPSFR_FL(3)>(
                        MPSFR_FA(0), MPSFR_FA(1), MPSFR_FA(2), MPSFR_FA(3), MPSFR_FX(0), MPSFR_FX(1), MPSFR_FX(2),
                        MPSFR_FX(3), w);
                if constexpr (n == 2 || n == 3)
                    cmac_group2<MPSFR_FL(0), MPSFR_FL(1)>(MPSFR_FA(0), MPSFR_FA(1), MPSFR_FX(0), MPSFR_FX(1), w);
                if constexpr (n == 3) cmac_group1<MPSFR_FL(2)>(MPSFR_FA(2), MPSFR_FX(2), w);
                if constexpr (n == 1) cmac_group1<MPSFR_FL(0)>(MPSFR_FA(0), MPSFR_FX(0), w);
            } else {
                if constexpr (n == 5)
                    rmac_group5<MPSFR_FL(0), MPSFR_FL(1), MPSFR_FL(2), MPSFR_FL(3), MPSFR_FL(4)>(
                        MPSFR_FA(0), MPSFR_FA(1), MPSFR_FA(2), MPSFR_FA(3), MPSFR_FA(4), MPSFR_FR(0), MPSFR_FR(1),
                        MPSFR_FR(2), MPSFR_FR(3), MPSFR_FR(4), w);
                if constexpr (n == 4)
                    rmac_group4<MPSFR_FL(0), MPSFR_FL(1), MPSFR_FL(2), MPSFR_FL(3)>(
                        MPSFR_FA(0), MPSFR_FA(1), MPSFR_FA(2), MPSFR_FA(3), MPSFR_FR(0), MPSFR_FR(1), MPSFR_FR(2),
                        MPSFR_FR(3), w);
                if constexpr (n == 2 || n == 3)
                    rmac_group2<MPSFR_FL(0), MPSFR_FL(1)>(MPSFR_FA(0), MPSFR_FA(1), MPSFR_FR(0), MPSFR_FR(1), w);
                if constexpr (n == 3) rmac_group1<MPSFR_FL(2)>(MPSFR_FA(2), MPSFR_FR(2), w);
                if constexpr (n == 1) rmac_group1<MPSFR_FL(0)>(MPSFR_FA(0), MPSFR_FR(0), w);
            }
        }
#undef MPSFR_FN
#undef MPSFR_FA
#undef MPSFR_FX
#undef MPSFR_FR
#undef MPSFR_FL
        fold_group<Q, CPLX, R0, RS, CNT, J + 1>(acc, xr, xi, wj);
    }
}

// residues per group: the M = Q / 4 residues r = 4 r2 + r1 of one r1 (what the transform below asks for
// at a time) where that gives at least 4 chains; all of them otherwise
template <int Q> constexpr int group_cnt() { return Q >= 16 ? Q / 4 : (Q < 4 ? Q : 4); }

// v[i] = S_r for r = R0 + RS i, i < CNT, the factor W_N^(r k2) = wrf(r) included
template <int Q, bool CPLX, int R0, int RS, int CNT, typename WR>
__device__ __forceinline__ void fold_sums(cx<double>* v, const double* xr, const double* xi,
                                          const cx<double>* wj, WR&& wrf) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = {0.0, 0.0};
    fold_group<Q, CPLX, R0, RS, CNT>(v, xr, xi, wj);
#pragma unroll
    for (int i = 0; i < CNT; ++i)
        if (R0 + RS * i > 0) v[i] = cmul(v[i], wrf(R0 + RS * i));
}

// ------------------------------------------------------------------------------------------
// K_PATCH_GEN: P[td][su + 40][sv + 40] = max(F, AO) - F on the corrected zone (psfrec.py:148-149;
// F and AO exactly as K_PSD_ROWFFT evaluates them).
// ------------------------------------------------------------------------------------------
template <bool F64>
__global__ void __launch_bounds__(256) k_patch_gen(int ndir, const TaskPar* __restrict__ tp,
                                                   const double* __restrict__ aotab, double cfit,
                                                   double* __restrict__ P) {
    constexpr int NEWTON = F64 ? 2 : 1;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= NAO * NAO) return;
    const int td = blockIdx.y, task = td / ndir, d = td % ndir;
    const TaskPar p = tp[task];
    const double* tb = aotab + ((size_t)(p.geom * ndir + d) * 3) * (NAO * NAO);
    const int su = pix / NAO - NAO / 2, sv = pix % NAO - NAO / 2;
    const double fit = psd_fit_value<NEWTON>(su, sv, p, cfit);
    P[(size_t)td * (NAO * NAO) + pix] = psd_with_ao<NEWTON>(fit, su, sv, p, tb) - fit;
}

// ------------------------------------------------------------------------------------------
// K_PATCH_ROWS: T[td][su + 40][y] = sum_sv P[su][sv] exp(-2 pi i sv y / N), y in [0, N/2], and
// sp[td] = sum P.  A wave takes a row su: lane k2 folds the row's 80 values (broadcast operands)
// into Q sums with its own twiddles, transforms them in registers and owns y = k2, 64 + k2, ...
// ------------------------------------------------------------------------------------------
constexpr int kRowsPerWg = 16;
template <int N>
__global__ void __launch_bounds__(256) k_patch_rows(const double* __restrict__ P,
                                                    const cx<double>* __restrict__ twg,
                                                    cx<double>* __restrict__ T, double* __restrict__ sp) {
    constexpr int Q = N / 64, H1 = N / 2 + 1, NJ = fold_nj<Q>(), JMIN = fold_jmin<Q>();
    constexpr bool WJREG = NJ <= 10;
    __shared__ cx<double> swj[WJREG ? 1 : NJ][64];
    __shared__ double sred[256];
    const int td = blockIdx.y;
    const double* Pg = P + (size_t)td * (NAO * NAO);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (!WJREG) {
        for (int i = threadIdx.x; i < NJ * 64; i += 256)
            swj[i >> 6][i & 63] = twg[(((Q * ((i >> 6) + JMIN) * (i & 63)) % N) + N) % N];
        __syncthreads();
    }
    if (blockIdx.x == 0) {      // sum of the patch, in an order fixed by the launch geometry
        double a = 0.0;
        for (int i = threadIdx.x; i < NAO * NAO; i += 256) a += Pg[i];
        sred[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < 64) {
            double b = (sred[threadIdx.x] + sred[threadIdx.x + 64]) + (sred[threadIdx.x + 128] + sred[threadIdx.x + 192]);
            b = wave_sum(b);
            if (threadIdx.x == 0) sp[td] = b;
        }
    }
    cx<double> wjr[WJREG ? NJ : 1], wr[Q];
    if constexpr (WJREG) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wjr[j] = twg[(((Q * (j + JMIN) * lane) % N) + N) % N];
    }
#pragma unroll
    for (int r = 1; r < Q; ++r) wr[r] = twg[(r * lane) % N];
    for (int i = 0; i < kRowsPerWg / 4; ++i) {
        const int n = blockIdx.x * kRowsPerWg + wave + 4 * i;          // row su + 40
        double xr[kNX];
#pragma unroll
        for (int a = 0; a < kNX; ++a) xr[a] = Pg[n * NAO + 16 * a + (lane & 15)];
        cx<double> S[Q];
        {
            constexpr int GC = Q % 5 == 0 ? 5 : (Q < 4 ? Q : 4);
            cx<double> wl[WJREG ? 1 : NJ];
            if constexpr (!WJREG) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) wl[j] = swj[j][lane];
            }
            const cx<double>* wjp = WJREG ? wjr : wl;
            static_for<0, Q / GC>([&](auto gc) {
                constexpr int R0 = decltype(gc)::value * GC;
                fold_sums<Q, false, R0, 1, GC>(S + R0, xr, xr, wjp, [&](int r) { return wr[r]; });
            });
        }
        dftq<Q>(S);
        cx<double>* Tt = T + ((size_t)td * NAO + n) * H1 + lane;
#pragma unroll
        for (int k1 = 0; k1 <= Q / 2; ++k1)
            if (64 * k1 + lane <= N / 2) Tt[64 * k1] = S[k1];
    }
}

// ------------------------------------------------------------------------------------------
// K_DPHI_SERIES: D0t[td][y][x] = r0^(-5/3) sum_k delta^k Hd_k[y][x] + scale2 (sp - Re X[x]) with
// X[x] = sum_su T[td][su][y] exp(-2 pi i su x / N).  Workgroup = (line y, group of tasks): the
// coefficients of the line and the twiddles W_N^(r k2) go to LDS once; every wave then walks its
// tasks alone (no barrier).  The 80 complex inputs of a line are broadcast operands (above).
//   coef: [y][x][K] (K fp32 / fp64 terms of pixel (y, x) side by side).
// ------------------------------------------------------------------------------------------
template <typename RO> struct SeriesCfg;
template <> struct SeriesCfg<float> { static constexpr int K = 4, THREADS = 256; };
template <> struct SeriesCfg<double> { static constexpr int K = 8, THREADS = 256; };

template <int N, typename RO>
constexpr size_t series_smem() {
    return (size_t)N * SeriesCfg<RO>::K * sizeof(RO) + (size_t)(N / 64) * 64 * sizeof(cx<double>);
}

template <int N, typename RO>
__global__ void __launch_bounds__((SeriesCfg<RO>::THREADS), (sizeof(RO) == 4 ? 2 : 1))
k_dphi_series(const cx<double>* __restrict__ T, const double* __restrict__ sp,
              const TaskPar* __restrict__ tp, int ndir, int ntd, int tg, const RO* __restrict__ coef,
              const cx<double>* __restrict__ twg, double scale2, RO* __restrict__ D0t,
              int* __restrict__ zero17, int dbg) {
    constexpr int Q = N / 64, H1 = N / 2 + 1, K = SeriesCfg<RO>::K, THREADS = SeriesCfg<RO>::THREADS;
    constexpr int NJ = fold_nj<Q>(), JMIN = fold_jmin<Q>();
    constexpr bool WJREG = NJ <= 10;         // the twiddles W_64^(j k2) of a lane in registers
    extern __shared__ __align__(16) unsigned char smem[];
    RO* scoef = reinterpret_cast<RO*>(smem);                                   // [N][K]
    cx<double>* swr = reinterpret_cast<cx<double>*>(smem + (size_t)N * K * sizeof(RO));   // [Q][64]
    const int y = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int td_end = min(ntd, (g + 1) * tg);
    int td = g * tg + wave;
    // the inputs of the wave's first line are requested before anything else
    // (two register sets: the inputs of a line are requested a whole line ahead -- T comes from the
    // Infinity Cache or from memory, and the polynomial phase alone did not cover that: 1280^2 172 -> us)
    cx<double> xva[kNX], xvb[kNX];
    auto fetch = [&](int t, cx<double>* xv) {
        const cx<double>* src = T + ((size_t)t * NAO + (lane & 15)) * H1 + y;
#pragma unroll
        for (int a = 0; a < kNX; ++a) xv[a] = src[(size_t)16 * a * H1];
    };
    if (td < td_end) fetch(td, xva);
    if (!(dbg & 8)) {
        // the line's coefficients: one contiguous block of N K values
        using V4 = typename std::conditional<sizeof(RO) == 4, float4, double2>::type;
        constexpr int NV = (int)((size_t)N * K * sizeof(RO) / 16);
        const V4* src = reinterpret_cast<const V4*>(coef + (size_t)y * N * K);
        V4* dst = reinterpret_cast<V4*>(scoef);
        for (int i = threadIdx.x; i < NV; i += THREADS) dst[i] = src[i];
        for (int i = threadIdx.x; i < Q * 64; i += THREADS) swr[i] = twg[((i >> 6) * (i & 63)) % N];
    }
    if (zero17 != nullptr && y == 0 && g == 0 && threadIdx.x < 17) zero17[threadIdx.x] = 0;
    cx<double> wjr[WJREG ? NJ : 1];
    if constexpr (WJREG) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wjr[j] = twg[(((Q * (j + JMIN) * lane) % N) + N) % N];
    }
    __syncthreads();
    auto line = [&](int td, const cx<double>* xv) {
        const int task = td / ndir;
        const double r0m53 = tp[task].r0m53, delta = tp[task].inv_l0sq - kEps0;
        const double spv = sp[td];
        double xr[kNX], xi[kNX];
#pragma unroll
        for (int a = 0; a < kNX; ++a) {
            xr[a] = xv[a].x;
            xi[a] = xv[a].y;
        }
        double out[Q];
        if (dbg & 2) {
#pragma unroll
            for (int k1 = 0; k1 < Q; ++k1) out[k1] = xr[k1 % kNX] + k1;
        } else {
            cx<double> wl[WJREG ? 1 : NJ];
            if constexpr (!WJREG) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) wl[j] = twg[(((Q * (j + JMIN) * lane) % N) + N) % N];
            }
            const cx<double>* wjp = WJREG ? wjr : wl;
            auto wrf = [&](int r) { return swr[r * 64 + lane]; };
            if constexpr (Q == 2) {
                cx<double> S[2];
                fold_sums<Q, true, 0, 1, 2>(S, xr, xi, wjp, wrf);
                out[0] = S[0].x + S[1].x;
                out[1] = S[0].x - S[1].x;
            } else if constexpr (Q >= 16) {
                // the Q / 4 residues of one r1 at a time: 4 or 5 independent accumulator pairs
                dftq_real_out<Q>(
                    [&](auto r1c, cx<double>* v) {
                        fold_sums<Q, true, decltype(r1c)::value, 4, Q / 4>(v, xr, xi, wjp, wrf);
                    },
                    out);
            } else {
                // Q = 4, 8: all sums first, four residues to a group
                cx<double> S[Q];
                static_for<0, Q / 4>([&](auto gc) {
                    constexpr int R0 = decltype(gc)::value * 4;
                    fold_sums<Q, true, R0, 1, 4>(S + R0, xr, xi, wjp, wrf);
                });
                dftq_real_out<Q>(
                    [&](auto r1c, cx<double>* v) {
#pragma unroll
                        for (int r2 = 0; r2 < Q / 4; ++r2) v[r2] = S[4 * r2 + decltype(r1c)::value];
                    },
                    out);
            }
        }
        if (dbg & 4) {
            if (out[0] == 1.2345) D0t[td] = 0;
            return;
        }
        RO* dst = D0t + ((size_t)td * H1 + y) * N + lane;
        if constexpr (sizeof(RO) == 4) {
            const float df = (float)delta, rf = (float)r0m53;
#pragma unroll
            for (int k1 = 0; k1 < Q; ++k1) {
                const float4 h = *reinterpret_cast<const float4*>(scoef + (size_t)(64 * k1 + lane) * K);
                const float dF = rf * fmaf(fmaf(fmaf(h.w, df, h.z), df, h.y), df, h.x);
                dst[64 * k1] = (float)(fma(scale2, spv - out[k1], (double)dF));
            }
        } else {
#pragma unroll
            for (int k1 = 0; k1 < Q; ++k1) {
                const double* h = scoef + (size_t)(64 * k1 + lane) * K;
                double a = h[K - 1];
#pragma unroll
                for (int k = K - 2; k >= 0; --k) a = fma(a, delta, h[k]);
                dst[64 * k1] = fma(scale2, spv - out[k1], r0m53 * a);
            }
        }
    };
    constexpr int STEP = THREADS / 64;
    while (td < td_end) {
        if (td + STEP < td_end && !(dbg & 1)) fetch(td + STEP, xvb);
        line(td, xva);
        td += STEP;
        if (td >= td_end) break;
        if (td + STEP < td_end && !(dbg & 1)) fetch(td + STEP, xva);
        line(td, xvb);
        td += STEP;
    }
}

// Hd planes [K][H1][N] (fp64, the output of K_COLFFT_DPHI for the basis tasks) -> coef[y][x][K]
template <typename RO>
__global__ void __launch_bounds__(256) k_series_coef(int n, int K, const double* __restrict__ planes,
                                                     RO* __restrict__ coef) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < K; ++k) coef[(size_t)i * K + k] = (RO)planes[(size_t)k * n + i];
}

}  // namespace

int series_terms(bool f64) { return f64 ? SeriesCfg<double>::K : SeriesCfg<float>::K; }
double series_eps0() { return kEps0; }

void launch_series_coef(hipStream_t s, int N, const double* d_planes, void* d_coef, bool f64) {
    const int n = (N / 2 + 1) * N;
    if (f64)
        hipLaunchKernelGGL(k_series_coef<double>, dim3((n + 255) / 256), dim3(256), 0, s, n,
                           SeriesCfg<double>::K, d_planes, (double*)d_coef);
    else
        hipLaunchKernelGGL(k_series_coef<float>, dim3((n + 255) / 256), dim3(256), 0, s, n,
                           SeriesCfg<float>::K, d_planes, (float*)d_coef);
}

void launch_patch(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp, const double* d_aotab,
                  double cfit, const void* d_tw64, double* d_P, void* d_T, double* d_sp, bool f64) {
    const dim3 ggrid((NAO * NAO + 255) / 256, ntd);
    if (f64)
        hipLaunchKernelGGL(k_patch_gen<true>, ggrid, dim3(256), 0, s, ndir, d_tp, d_aotab, cfit, d_P);
    else
        hipLaunchKernelGGL(k_patch_gen<false>, ggrid, dim3(256), 0, s, ndir, d_tp, d_aotab, cfit, d_P);
    DISPATCH_N(N, {
        hipLaunchKernelGGL((k_patch_rows<NN>), dim3(NAO / kRowsPerWg, ntd), dim3(256), 0, s, (const double*)d_P,
                           (const cx<double>*)d_tw64, (cx<double>*)d_T, d_sp);
    })
}

void launch_dphi_series(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp, const void* d_T,
                        const double* d_sp, const void* d_coef, const void* d_tw64, double scale2,
                        void* d_D0t, bool f64out, int* d_zero) {
    const int H1 = N / 2 + 1;
    // task groups: enough workgroups to fill the GPU several times over, but every workgroup's table
    // load (the line's coefficients) shared by as many tasks as that allows
    const int waves = (f64out ? SeriesCfg<double>::THREADS : SeriesCfg<float>::THREADS) / 64;
    int ngr = (2048 + H1 - 1) / H1;
    if (ngr * waves > ntd) ngr = (ntd + waves - 1) / waves;
    if (ngr < 1) ngr = 1;
    int tg = (ntd + ngr - 1) / ngr;
    tg = (tg + waves - 1) / waves * waves;      // every wave of a workgroup the same number of lines
    static const int env_tg = getenv("MPSFR_SERIES_TG") ? atoi(getenv("MPSFR_SERIES_TG")) : 0;   // experiments
    if (env_tg > 0) tg = env_tg;
    static const int env_dbg = getenv("MPSFR_SERIES_DBG") ? atoi(getenv("MPSFR_SERIES_DBG")) : 0;    // experiments
    ngr = (ntd + tg - 1) / tg;
    const dim3 grid(H1, ngr);
    DISPATCH_N(N, {
        if (f64out) {
            constexpr size_t sm = series_smem<NN, double>();
            allow_smem((k_dphi_series<NN, double>), sm);
            hipLaunchKernelGGL((k_dphi_series<NN, double>), grid, dim3(SeriesCfg<double>::THREADS), sm, s,
                               (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, tg, (const double*)d_coef,
                               (const cx<double>*)d_tw64, scale2, (double*)d_D0t, d_zero, env_dbg);
        } else {
            constexpr size_t sm = series_smem<NN, float>();
            allow_smem((k_dphi_series<NN, float>), sm);
            hipLaunchKernelGGL((k_dphi_series<NN, float>), grid, dim3(SeriesCfg<float>::THREADS), sm, s,
                               (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, tg, (const float*)d_coef,
                               (const cx<double>*)d_tw64, scale2, (float*)d_D0t, d_zero, env_dbg);
        }
    })
}

}  // namespace mpsfr
