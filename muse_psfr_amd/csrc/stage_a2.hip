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
#include "conv_frames.h"
#include "mf_common.h"
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

// Minima over the lanes 0-31 and 32-63 of four values at a time: five v_min_f32_dpp steps each (quad
// swaps, half-row and row mirrors: every lane holds the minimum of its row of 16; row_bcast:15: rows 1
// and 3 hold the minima of lanes 0-31 and 32-63 -- row 0 gets min(v, 0) and row 2 the minimum of rows
// 1 | 2, which nobody reads).  One statement: a DPP operand must not have been written in the two wait
// states before, which the three other values' instructions provide between the steps of one value
// (and the opening s_nop for whatever computed the inputs); hipcc does not form v_min_f32_dpp from
// __builtin_amdgcn_update_dpp + fminf with an identity of +inf (mov, mov_dpp, two wait states, min:
// the fused minima took a third of the kernel that way).
#define MPSFR_MIN4(ctrl)                                                        \
    "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_min_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_min_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void min32_x4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        MPSFR_MIN4("quad_perm:[1,0,3,2]")
        MPSFR_MIN4("quad_perm:[2,3,0,1]")
        MPSFR_MIN4("row_half_mirror")
        MPSFR_MIN4("row_mirror")
        MPSFR_MIN4("row_bcast:15")
        "s_nop 0"       /* (the next instruction may read d: written by a DPP instruction one state before) */
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// the same over the rows of 16 lanes alone (no row_bcast step): every lane holds its row's minimum
__device__ __forceinline__ void min16_x4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        MPSFR_MIN4("quad_perm:[1,0,3,2]")
        MPSFR_MIN4("quad_perm:[2,3,0,1]")
        MPSFR_MIN4("row_half_mirror")
        MPSFR_MIN4("row_mirror")
        "s_nop 0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef MPSFR_MIN4

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
// The twiddles of a lane, laid out per lane: twk[j - JMIN][k2] = W_64^(j k2) for the NJ fold terms,
// then twk[NJ + r][k2] = W_N^(r k2) for r < Q.  Built once per context: a wave reads a row of the
// table as 1 KB of consecutive bytes, where twg[(r k2) mod N] straight from the table of N-th roots
// is a gather of 64 cache lines per instruction -- the 16 such gathers at the head of every wave of
// K_PATCH_ROWS kept its vector-memory pipe full and the waves at 40 cycles per instruction.
// ------------------------------------------------------------------------------------------
// L: lanes per line (64 in K_PATCH_ROWS; series_lanes<N>() in K_DPHI_SERIES), Q = N / L
template <int N, int L>
__global__ void __launch_bounds__(64) k_series_twiddles(const cx<double>* __restrict__ twg,
                                                        cx<double>* __restrict__ twk) {
    constexpr int Q = N / L, NJ = fold_nj<Q>(), JMIN = fold_jmin<Q>();
    const int row = blockIdx.x, k2 = threadIdx.x;
    if (k2 >= L) return;
    if (row < NJ) twk[row * L + k2] = twg[(((Q * (row + JMIN) * k2) % N) + N) % N];
    else twk[row * L + k2] = twg[((row - NJ) * k2) % N];
}

// Lanes per line of K_DPHI_SERIES.  A wave carries 64 / L lines (the same y, consecutive tasks): the
// fold costs 320 multiply-adds per WAVE whatever L is -- each row of 16 lanes holds its own line's
// inputs as broadcast operands -- while the in-lane transform grows with Q = N / L.  Q = 16 wherever
// the grid allows it (512^2: two lines per wave, 560 -> 330 instructions per line; 256^2: four).
template <int N>
constexpr int series_lanes() { return N <= 256 ? 16 : (N == 512 ? 32 : 64); }
template <int N, int L>
constexpr size_t twiddle_entries() { return (size_t)(fold_nj<N / L>() + N / L) * L; }

// ------------------------------------------------------------------------------------------
// K_PATCH_GEN: P[td][su + 40][sv + 40] = max(F, AO) - F on the corrected zone (psfrec.py:148-149;
// F and AO exactly as K_PSD_ROWFFT evaluates them).  A thread is (task, pixel) and walks the task's
// directions: the fitting term and the von Karman factor of the pixel -- two x^(-11/6), nearly all of
// the arithmetic -- do not depend on the direction, only the tables do (nine directions: 21.6 -> us).
// ------------------------------------------------------------------------------------------
// Round 6: the call's parameter blob rides in this launch (K_PARAM_COPY, the kernel that fetched it on its own at the
// head of every call, cost the call's queue 8 us in round 5 and 5 us as one PCIe round trip -- an empty kernel
// costs 4, profiles/r06_ubench_pcopy.txt).  With pc.n16 > 0 the workgroups of the row blockIdx.y == gridDim.y - 1
// copy the blob from pinned host memory to the device for the kernels BEHIND this one, every load issued before
// the first store; the task workgroups read their task's 40 bytes from `tp` = the TaskPar array INSIDE THE PINNED
// BLOB: five lanes of a workgroup, one PCIe read, the rest through LDS (2500 such reads beside a launch cost it
// ~1 us, 10^4 -- one per wave -- 2.5 us: same file).  No fence, no flag between workgroups: an agent-scope acquire
// inside a kernel invalidates the L2 of its XCD (a first version in which the task workgroups waited for the copy
// and read the device blob made a call 60 us longer).  The host may refill the pinned blob once every workgroup here
// has read its task: it is told by the first workgroup of the NEXT kernel of the queue (K_PATCH_ROWS, ParamFlag).
struct ParamCopy {
    uint4* dst;
    const uint4* src;
    int n16;                        // 0: no copy in this launch (tp is device memory)
};
struct ParamFlag {
    unsigned long long* flag;
    unsigned long long seq;
    int* queue_zero;
};

template <bool F64>
__global__ void __launch_bounds__(256) k_patch_gen(int ndir, const TaskPar* __restrict__ tp,
                                                   const double* __restrict__ aotab, double cfit,
                                                   double* __restrict__ P, ParamCopy pc) {
    constexpr int NEWTON = F64 ? 2 : 1;
    constexpr int TPW = sizeof(TaskPar) / 8;
    static_assert(sizeof(TaskPar) % 8 == 0, "TaskPar is read in 8-byte words");
    __shared__ unsigned long long tps[TPW];
    const int task = blockIdx.y;
    TaskPar p;
    if (pc.n16 > 0) {
        if (blockIdx.y == gridDim.y - 1) {
            constexpr int PER = 4;
            for (int base = blockIdx.x * 256 * PER; base < pc.n16; base += gridDim.x * 256 * PER) {
                uint4 v[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int i = base + q * 256 + (int)threadIdx.x;
                    if (i < pc.n16) v[q] = pc.src[i];
                }
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int i = base + q * 256 + (int)threadIdx.x;
                    if (i < pc.n16) pc.dst[i] = v[q];
                }
            }
            return;
        }
        if (threadIdx.x < TPW)
            tps[threadIdx.x] = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(tp + task) + threadIdx.x);
        __syncthreads();
        p = *reinterpret_cast<const TaskPar*>(tps);
    } else {
        p = tp[task];
    }
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= NAO * NAO) return;
    const int su = pix / NAO - NAO / 2, sv = pix % NAO - NAO / 2;
    const double fit = psd_fit_value<NEWTON>(su, sv, p, cfit);
    const int ia = su < 0 ? su + NAO : su, ib = sv < 0 ? sv + NAO : sv;
    const double g2 = (double)(su * su + sv * sv) * (1.0 / 256.0);
    const double vk = 0.0229 * p.r0m53 * pow_m11_6<NEWTON>(g2 + p.inv_l0sq);        // :569-571
    const int o = ia * NAO + ib;
    if (ndir == 1) {
        const double* tb = aotab + ((size_t)p.geom * 3) * (NAO * NAO);
        const double ao = vk * (p.cn2_0 * tb[o] + p.cn2_1 * tb[NAO * NAO + o]) + tb[2 * NAO * NAO + o];
        P[(size_t)task * (NAO * NAO) + pix] = fmax(fit, ao) - fit;                           // :149
        return;
    }
    // (nine directions at a time with their table loads in flight together: one at a time, the loop was
    // nine memory latencies long)
    for (int d0 = 0; d0 < ndir; d0 += 9) {
        double t0[9], t1[9], t2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int d = min(d0 + i, ndir - 1);
            const double* tb = aotab + ((size_t)(p.geom * ndir + d) * 3) * (NAO * NAO);
            t0[i] = tb[o];
            t1[i] = tb[NAO * NAO + o];
            t2[i] = tb[2 * NAO * NAO + o];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (d0 + i >= ndir) break;
            const double ao = vk * (p.cn2_0 * t0[i] + p.cn2_1 * t1[i]) + t2[i];
            P[((size_t)task * ndir + d0 + i) * (NAO * NAO) + pix] = fmax(fit, ao) - fit;     // :149
        }
    }
}

// ------------------------------------------------------------------------------------------
// K_PATCH_ROWS: T[td][y][su + 40] = sum_sv P[su][sv] exp(-2 pi i sv y / N), y in [0, N/2], and
// sp[td] = sum P.  A wave takes one, two or four adjacent rows su at a time: lane k2 folds a row's 80 values (broadcast
// operands) into Q sums with its own twiddles, transforms them in registers and owns y = k2, 64 + k2, ...
// K_DPHI_SERIES reads the 80 values of one (td, y) as 1280 contiguous bytes (with T[td][su][y] every
// line of it gathered 80 cache lines: 12 of its 50 us at 512^2), so a lane stores its rows' values
// of one y as one 32- or 64-byte piece; the waves are independent (a transposition of 8 rows through LDS
// for 128-byte pieces cost two barriers per workgroup and, at 1280^2, all of a CU's LDS: 31 -> 46 us).
// ------------------------------------------------------------------------------------------
// L = series_lanes<N>() lanes per row, R = 64 / L rows per wave and pass (one per group of L lanes),
// Q = N / L: the same fold and in-lane transform as K_DPHI_SERIES, with a real input and a complex
// output of which y <= N/2 is kept.  A workgroup takes `qb` groups of 4 R adjacent rows of one td, one
// group per pass, a row set per wave.  The results of a pass meet in an LDS tile [y][4 R rows] and leave
// as pieces of 4 R x 16 bytes (64 bytes at 1280^2, 256 at 256^2): a lane storing its own y values wrote
// 16-byte pieces 1280 bytes apart, and at 1280^2 that alone took 25 of the kernel's 45 us.
template <int N>
constexpr size_t patch_rows_smem() { return (size_t)(N / 2 + 1) * (4 * (64 / series_lanes<N>()) + 1) * sizeof(cx<double>); }

// Round 6: the workgroups with blockIdx.y >= ntd are not rows of the patch: they compute the spectra of the chunk's
// tip-tilt Moffat kernels (the body of K_KHAT, conv_frames.h; KR = float / double, void: none) -- a kernel of its
// own at the head of every call until now (8 us alone, 16-48 us parked behind the other lane's K_OTF_MFMA2 in the
// pipelined run) whose LDS footprint is this kernel's.  Workgroup (0, 0) hands the pinned parameter blob back to the
// host (ParamFlag above: the kernel in front of this one was its last reader).
struct KhatArgs {
    int nker;
    const double* gam;
    const double* alp;
    void* khat;
};

template <int N, typename KR>
constexpr size_t patch_rows_smem_all() {
    if constexpr (std::is_void<KR>::value) return patch_rows_smem<N>();
    else return patch_rows_smem<N>() > conv_smem_bytes<KR>(true) ? patch_rows_smem<N>() : conv_smem_bytes<KR>(true);
}

template <int N, typename KR>
__global__ void __launch_bounds__(256) k_patch_rows(const double* __restrict__ P,
                                                    const cx<double>* __restrict__ twk,
                                                    cx<double>* __restrict__ T, double* __restrict__ sp, int qb,
                                                    int ntd, KhatArgs kh, ParamFlag pf) {
    constexpr int L = series_lanes<N>(), R = 64 / L, Q = N / L, H1 = N / 2 + 1, NJ = fold_nj<Q>();
    if (pf.flag != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(pf.flag, pf.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (the block queue of K_DPHI_SERIES_Q, the next kernel of this queue, starts at zero)
    if (pf.queue_zero != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *pf.queue_zero = 0;
    if constexpr (!std::is_void<KR>::value) {
        if ((int)blockIdx.y >= ntd) {
            extern __shared__ __align__(16) unsigned char smem_k[];
            const int kid = ((int)blockIdx.y - ntd) * (int)gridDim.x + (int)blockIdx.x;
            if (kid < kh.nker) khat_body<KR>(kh.gam, kh.alp, (cx<KR>*)kh.khat, smem_k, kid);
            return;
        }
    }
    constexpr bool WJREG = NJ <= 10;
    constexpr int NY = Q / 2 + 1;                        // values y = L k1 + k2 <= N/2 of a lane
    constexpr int RG = 4 * R, RS = RG + 1;               // rows per pass of the workgroup; padded tile row
    constexpr int NG = NAO / RG;                         // passes per td
    extern __shared__ __align__(16) unsigned char smem[];
    cx<double>* tile = reinterpret_cast<cx<double>*>(smem);          // [H1][RS]
    __shared__ double sred[256];
    const int td = blockIdx.y;
    const double* Pg = P + (size_t)td * (NAO * NAO);
    const int lane = threadIdx.x & 63, rho = lane / L, k2 = lane & (L - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g_end = min(NG, ((int)blockIdx.x + 1) * qb);
    int g = blockIdx.x * qb;
    double xa[kNX], xb[kNX];
    auto fetch = [&](int gg, double* xr) {
#pragma unroll
        for (int a = 0; a < kNX; ++a) xr[a] = Pg[(RG * gg + R * wave + rho) * NAO + 16 * a + (lane & 15)];
    };
    if (g < g_end) fetch(g, xa);
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
        for (int j = 0; j < NJ; ++j) wjr[j] = twk[j * L + k2];
    }
#pragma unroll
    for (int r = 1; r < Q; ++r) wr[r] = twk[(NJ + r) * L + k2];
    auto pass = [&](int gg, const double* xr) {
        cx<double> S[Q];
        {
            constexpr int GC = Q % 5 == 0 ? 5 : (Q < 4 ? Q : 4);
            cx<double> wl[WJREG ? 1 : NJ];
            if constexpr (!WJREG) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) wl[j] = twk[j * L + k2];
            }
            const cx<double>* wjp = WJREG ? wjr : wl;
            static_for<0, Q / GC>([&](auto gc) {
                constexpr int R0 = decltype(gc)::value * GC;
                fold_sums<Q, false, R0, 1, GC>(S + R0, xr, xr, wjp, [&](int r) { return wr[r]; });
            });
        }
        dftq<Q>(S);
#pragma unroll
        for (int k1 = 0; k1 < NY; ++k1)
            if (L * k1 + k2 <= N / 2) tile[(L * k1 + k2) * RS + R * wave + rho] = S[k1];
        __syncthreads();
        cx<double>* Tt = T + (size_t)td * H1 * NAO + RG * gg;
        for (int e = threadIdx.x; e < H1 * RG; e += 256) {
            const int y = e / RG, i = e % RG;
            Tt[(size_t)y * NAO + i] = tile[y * RS + i];
        }
        __syncthreads();
    };
    while (g < g_end) {
        if (g + 1 < g_end) fetch(g + 1, xb);
        pass(g, xa);
        g += 1;
        if (g >= g_end) break;
        if (g + 1 < g_end) fetch(g + 1, xa);
        pass(g, xb);
        g += 1;
    }
}

// ------------------------------------------------------------------------------------------
// K_DPHI_SERIES: D0t[td][y][x] = r0^(-5/3) sum_k delta^k Hd_k[y][x] + scale2 (sp - Re X[x]) with
// X[x] = sum_su T[td][y][su] exp(-2 pi i su x / N).  A wave takes one (td, y) at a time ("a line"):
// the 80 complex inputs are broadcast operands (above), lane k2 owns x = k2, 64 + k2, ...
//   coef: [y][x][K] (K fp32 / fp64 terms of pixel (y, x) side by side), staged in LDS per line y.
// ------------------------------------------------------------------------------------------
template <typename RO> struct SeriesCfg;
template <> struct SeriesCfg<float> { static constexpr int K = 4; };
template <> struct SeriesCfg<double> { static constexpr int K = 8; };

// The lines of a wave: xv = their inputs (lane l holds input 16 a + l % 16 of ITS line in xv[a]),
// wjp[j - JMIN] = W_L^(j k2) of the lane, swr[r * L + k2] = W_N^(r k2) (LDS), scoef = the coefficients
// of line y (LDS, [x][K]); r0m53, delta, spv, dst, dlin: the lane's line (td); k2 = lane % L.
template <int N, typename RO, int L>
__device__ __forceinline__ void series_line(const cx<double>* xv, const cx<double>* wjp,
                                            const cx<double>* swr, const RO* scoef, double r0m53,
                                            double delta, double spv, double scale2, RO* dst, float* dlin,
                                            bool valid, int lane, unsigned kmask) {
    // kmask (wave-uniform): bit k1 set = some column x in [L k1, L k1 + L) of this line lies inside the support of
    // the telescope OTF (K_SERIES_SUPPORT).  Outside it the OTF is identically zero whatever the structure function
    // (psfrec.py:784-797): the polynomial, the store and the minimum of such a piece are skipped -- 21 % of the half
    // plane; what the per-wavelength stage reads there is the zero the buffer was allocated with.
    constexpr int Q = N / L, K = SeriesCfg<RO>::K;
    constexpr float kSkipped = 3.0e38f;
    const int k2 = lane & (L - 1);
    double xr[kNX], xi[kNX];
#pragma unroll
    for (int a = 0; a < kNX; ++a) {
        xr[a] = xv[a].x;
        xi[a] = xv[a].y;
    }
    double out[Q];
    // A branch hipcc cannot fold (the flag comes out of an asm statement and is always 1): the join
    // behind it pins `out` in registers and keeps the fold and the transform apart from the
    // polynomial phase.  As one basic block the 1280^2 kernel is allocated 256 registers + 124 bytes
    // of scratch instead of 238 + 0 and takes 246 us instead of 180; scheduling fences alone
    // (__builtin_amdgcn_sched_barrier) do not change that.
    int always;
    asm volatile("s_mov_b32 %0, 1" : "=s"(always));
    if (!always) {
#pragma unroll
        for (int k1 = 0; k1 < Q; ++k1) out[k1] = xr[k1 % kNX];
    } else {
        auto wrf = [&](int r) { return swr[r * L + k2]; };
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
    // Block minima for the pruning of the per-wavelength stage, while the values are in registers
    // (K_DMIN read all of D back for them: 12 us at 512^2, 69 us at 1280^2): the minimum of max(D, 0)
    // over the 32 columns [32 kb, 32 kb + 32) of the line goes to dlin[kb]; K_DMIN16 takes the minima
    // over 16 lines.  A lane's values are x = L k1 + k2:
    //   L = 64: kb = 2 k1 + (lane >= 32); min32_x4 leaves the minima in lanes 16-31 and 48-63, lane
    //           16 + k1 % 16 (48 + k1 % 16) keeps the result of k1;
    //   L = 32: kb = k1, the 32 lanes of a line; the same, lanes 16-31 / 48-63 are the two lines;
    //   L = 16: kb = k1 / 2: the two values of a lane first, then the row of 16 lanes (min16_x4); lane
    //           kb % 16 of the line's row keeps kb.
    constexpr int NB = L == 16 ? Q / 2 : Q;              // values per lane that go into the lane minima
    float keep[(NB + 15) / 16], dq[Q < 4 ? 4 : Q];
#pragma unroll
    for (int i = 0; i < (NB + 15) / 16; ++i) keep[i] = 0.f;
    if constexpr (sizeof(RO) == 4) {
        const float df = (float)delta, rf = (float)r0m53;
#pragma unroll
        for (int k1 = 0; k1 < Q; ++k1) {
            if (!((kmask >> k1) & 1u)) {
                dq[k1] = kSkipped;
                continue;
            }
            const float4 h = *reinterpret_cast<const float4*>(scoef + (size_t)(L * k1 + k2) * K);
            const float dF = rf * fmaf(fmaf(fmaf(h.w, df, h.z), df, h.y), df, h.x);
            const float d = (float)(fma(scale2, spv - out[k1], (double)dF));
            if (valid) dst[L * k1] = d;
            dq[k1] = fmaxf(d, 0.f);
        }
    } else {
#pragma unroll
        for (int k1 = 0; k1 < Q; ++k1) {
            if (!((kmask >> k1) & 1u)) {
                dq[k1] = kSkipped;
                continue;
            }
            const double* h = scoef + (size_t)(L * k1 + k2) * K;
            double a = h[K - 1];
#pragma unroll
            for (int k = K - 2; k >= 0; --k) a = fma(a, delta, h[k]);
            const double d = fma(scale2, spv - out[k1], r0m53 * a);
            if (valid) dst[L * k1] = d;
            dq[k1] = fmaxf(__double2float_rd(d), 0.f);      // (rounded down: the bound stays a bound)
        }
    }
    if (dlin != nullptr) {
        if constexpr (Q < 4) {
            dq[2] = dq[0];
            dq[3] = dq[1];
        }
        if constexpr (L == 16) {
            static_assert(Q % 8 == 0, "pairs of values, four pairs to a statement");
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) dq[kb] = fminf(dq[2 * kb], dq[2 * kb + 1]);
#pragma unroll
            for (int kb = 0; kb < NB; kb += 4) {
                min16_x4(dq[kb], dq[kb + 1], dq[kb + 2], dq[kb + 3]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((lane & 15) == ((kb + i) & 15)) keep[(kb + i) >> 4] = dq[kb + i];
            }
            if (valid) {
#pragma unroll
                for (int i = 0; i < (NB + 15) / 16; ++i) {
                    const int kb = 16 * i + (lane & 15);
                    if (kb < NB) dlin[kb] = keep[i];
                }
            }
        } else {
#pragma unroll
            for (int k1 = 0; k1 < (Q < 4 ? 4 : Q); k1 += 4) {
                min32_x4(dq[k1], dq[k1 + 1], dq[k1 + 2], dq[k1 + 3]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (k1 + i < Q && (lane & 15) == ((k1 + i) & 15)) keep[(k1 + i) >> 4] = dq[k1 + i];
            }
            if (valid && (lane & 16)) {
#pragma unroll
                for (int i = 0; i < (Q + 15) / 16; ++i) {
                    const int k1 = 16 * i + (lane & 15);
                    if (k1 < Q) dlin[L == 64 ? 2 * k1 + (lane >> 5) : k1] = keep[i];
                }
            }
        }
    }
}

// Persistent form: one workgroup per CU, the C = (N/2+1) ntd lines in y-major order cut into equal
// contiguous shares; the waves of a workgroup take the lines of its share in turn (wave w: lines
// c0 + w, c0 + w + NW, ...), so every wave of the launch does the same number of lines +- 1 and the
// set-up (twiddles, first coefficients) is paid once.  At any time the waves of a workgroup are within
// NW lines of each other, i.e. in at most two consecutive y: the coefficients of y_lo and y_lo + 1 sit
// in two LDS slots, and when wave 0 moves on to a new y_lo the workgroup meets at a barrier and loads
// the next line's into the slot that fell free (a share is 1 - 3 lines y long: a handful of barriers
// per launch).  [first form: one workgroup per (y, group of tasks) -- 2056 workgroups at 512^2, whose
// set-up and +-1 task imbalance cost 20 of its 50 us]
#ifndef MPSFR_SERIES_THREADS
#define MPSFR_SERIES_THREADS 512
#endif
template <int N, typename RO>
constexpr int series_threads() { return N <= 128 ? 256 : MPSFR_SERIES_THREADS; }
template <int N, typename RO>
constexpr size_t series_smem() {
    return 2 * (size_t)N * SeriesCfg<RO>::K * sizeof(RO) + (size_t)N * sizeof(cx<double>);
}
template <int N, typename RO>
constexpr bool series_fits() { return series_smem<N, RO>() <= 160 * 1024; }

// A "unit" is what a wave takes at a time: the R = 64 / L lines (y; td = R q + rho, rho < R) of one y
// and R consecutive tasks; units in y-major order, c = y nq + q with nq = ceil(ntd / R).
template <int N, typename RO>
__global__ void __launch_bounds__((series_threads<N, RO>()), (512 / series_threads<N, RO>() > 1 ? 2 : 1))
k_dphi_series(const cx<double>* __restrict__ T, const double* __restrict__ sp,
              const TaskPar* __restrict__ tp, int ndir, int ntd, const RO* __restrict__ coef,
              const cx<double>* __restrict__ twk, double scale2, RO* __restrict__ D0t,
              float* __restrict__ dlin, int* __restrict__ zero17, const unsigned* __restrict__ support) {
    constexpr int L = series_lanes<N>(), R = 64 / L, Q = N / L, H1 = N / 2 + 1, K = SeriesCfg<RO>::K;
    constexpr int THREADS = series_threads<N, RO>(), NW = THREADS / 64, NJ = fold_nj<Q>();
    constexpr bool WJREG = NJ <= 10;         // the twiddles W_L^(j k2) of a lane in registers
    constexpr int LINE = N * K;              // coefficients of a line
    extern __shared__ __align__(16) unsigned char smem[];
    RO* scoef = reinterpret_cast<RO*>(smem);                                       // [2][N][K]
    cx<double>* swr = reinterpret_cast<cx<double>*>(smem + 2 * (size_t)LINE * sizeof(RO));   // [Q][L]
    const int lane = threadIdx.x & 63, rho = lane / L, k2 = lane & (L - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nq = (ntd + R - 1) / R;
    const long C = (long)H1 * nq;
    const int c0 = (int)(C * blockIdx.x / gridDim.x), c1 = (int)(C * (blockIdx.x + 1) / gridDim.x);
    if (zero17 != nullptr && blockIdx.x == 0 && threadIdx.x < kMfSchedInts) zero17[threadIdx.x] = 0;
    if (c0 >= c1) return;
    // (two register sets: the inputs of a unit are requested a whole unit ahead)
    cx<double> xva[kNX], xvb[kNX];
    auto fetch = [&](int c, cx<double>* xv) {
        const int y = c / nq, td = min(R * (c - y * nq) + rho, ntd - 1);
        const cx<double>* src = T + ((size_t)td * H1 + y) * NAO + (lane & 15);
#pragma unroll
        for (int a = 0; a < kNX; ++a) xv[a] = src[16 * a];
    };
    const int nwe = min(NW, nq);
    const bool active = wave < nwe;
    if (active && c0 + wave < c1) fetch(c0 + wave, xva);
    using V4 = typename std::conditional<sizeof(RO) == 4, float4, double2>::type;
    constexpr int NV = (int)((size_t)LINE * sizeof(RO) / 16);
    auto load_line = [&](int y) {          // (all threads) coefficients of line y -> slot y & 1
        if (y > N / 2) return;
        const V4* src = reinterpret_cast<const V4*>(coef + (size_t)y * LINE);
        V4* dst = reinterpret_cast<V4*>(scoef + (size_t)(y & 1) * LINE);
        for (int i = threadIdx.x; i < NV; i += THREADS) dst[i] = src[i];
    };
    int ylo = c0 / nq;
    load_line(ylo);
    load_line(ylo + 1);
    for (int i = threadIdx.x; i < Q * L; i += THREADS) swr[i] = twk[NJ * L + i];
    cx<double> wjr[WJREG ? NJ : 1];
    if constexpr (WJREG) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wjr[j] = twk[j * L + k2];
    }
    __syncthreads();
    auto unit = [&](int c, const cx<double>* xv) {
        const int y = c / nq, tdr = R * (c - y * nq) + rho;
        const bool valid = tdr < ntd;
        const int td = valid ? tdr : ntd - 1;
        const int task = td / ndir;
        const double r0m53 = tp[task].r0m53, delta = tp[task].inv_l0sq - kEps0;
        const double spv = sp[td];
        cx<double> wl[WJREG ? 1 : NJ];
        if constexpr (!WJREG) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) wl[j] = twk[j * L + k2];
        }
        const unsigned kmask = support != nullptr ? (unsigned)__builtin_amdgcn_readfirstlane((int)support[y]) : 0xffffffffu;
        series_line<N, RO, L>(xv, WJREG ? wjr : wl, swr, scoef + (size_t)(y & 1) * LINE, r0m53, delta, spv, scale2,
                              D0t + ((size_t)td * H1 + y) * N + k2,
                              dlin != nullptr ? dlin + ((size_t)td * H1 + y) * (N / 32) : nullptr, valid, lane, kmask);
    };
    // every wave runs the same number of rounds (the barriers below are met by all of them); a round
    // is one unit per active wave: c = cb + wave.  (With fewer units per y than waves a round would
    // span more than two y: only nq waves work then.)
    auto new_y = [&](int cb) {
        const int yb = cb / nq;                  // the y of wave 0's unit: uniform over the workgroup
        if (yb != ylo) {                         // (yb = ylo + 1: a round is at most nq units long)
            __syncthreads();                     // nobody reads line ylo any more
            for (int yy = ylo + 2; yy <= yb + 1; ++yy) load_line(yy);
            ylo = yb;
            __syncthreads();
        }
    };
    for (int cb = c0; cb < c1; cb += 2 * nwe) {
        new_y(cb);
        {
            const int c = cb + wave;
            if (active && c + nwe < c1) fetch(c + nwe, xvb);
            if (active && c < c1) unit(c, xva);
        }
        if (cb + nwe < c1) {
            new_y(cb + nwe);
            const int c = cb + nwe + wave;
            if (active && c + nwe < c1) fetch(c + nwe, xva);
            if (active && c < c1) unit(c, xvb);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K_DPHI_SERIES_Q (round 6): the same lines, dealt in BLOCKS from a queue instead of in equal contiguous shares.
// A block is NW units (a unit = what a wave takes at a time: R lines of one y and R consecutive tasks IN
// THE ORDER `perm` -- the host sorts the tasks by how much uncorrected turbulence they carry, so that the units of a
// block behave alike); wave w takes unit w of it.  The workgroups draw blocks from one counter in
// y-major order (the first gridDim.x without a draw), the coefficients of the block's line arrive by LDS-DMA in the
// slot the previous block does not use while that one is computed, and the block ends at one barrier.
// Why: the lines a task can do without (SeriesSkip below) all lie at large y, and a contiguous share of the y-major
// order would leave the workgroups with the small y the whole work.  With nothing skipped the kernel takes what
// K_DPHI_SERIES takes (a wave's lines and their arithmetic are the same, bit for bit).
//
// SeriesSkip -- lines stage B provably drops.  With T[y][su] the row transforms of the patch (this kernel's input),
//     Re X_y[x] = Re sum_su T[y][su] W^(su x) <= sum_su |T[y][su]| =: B(y)
// for every x, so D(x, y) >= D_P(x, y) >= scale2 (sum P - B(y)) =: Lb(y) on the whole line (D_F >= 0: it is the
// structure function of a non-negative PSD): 80 magnitudes against the column transform of the line.  The OTF of
// the line is below tel 2^(c' D) <= 2^(tlmax(y) + c' Lb(y)) at every wavelength when c' is that of the LONGEST one.
// A line is skipped -- D = 1e30 stored (an OTF of exactly zero), Lb in its `dlin` entries, a valid lower bound for
// the block minima of K_MF_PREP -- when
//   (eps rule)  tlmax(y) + c'max Lb(y) < thr_elem: every element is one the eps rule of the pruning drops anyway, or
//   (mass rule) log2(2 N) + tlmax(y) + c'max Lb(y) < thr_mass: the line's whole mass in both half planes is below
//               the share tier_eps / (8 (N/2+1)) of the tier budget that K_MF_PREP sets aside for the skipped lines
//               (it works with 3/4 of its floor budget then): all skipped lines together stay below tier_eps / 8 of
//               OTF[0][0] = 1 <= the PSF peak.
// Both are budgets the per-wavelength stage already documents (include/mpsfr.h: prune_eps, tier_eps): nothing new is
// given away.  y = 0 is never skipped (B(0) = sum P).
// ------------------------------------------------------------------------------------------
struct SeriesSkip {
    const float* tlmax;       // [N/2+1] log2 of the line maxima of the telescope OTF; nullptr: nothing is skipped
    float c2max;              // log2(e) c of the longest wavelength (the least negative)
    float thr_elem;           // log2, eps rule
    float thr_mass;           // log2, mass rule (-inf: off)
};

template <int N, typename RO>
constexpr size_t series_q_smem() {
    return 4 * (size_t)N * SeriesCfg<RO>::K * sizeof(RO) + (size_t)N * sizeof(cx<double>);
}
template <int N, typename RO>
constexpr bool series_q_fits() { return series_q_smem<N, RO>() <= 160 * 1024; }

// (a block is NW consecutive units of the y-major order c = y nq + q, one per wave: it lies in one y or straddles
// two, so a block owns two line slots and the kernel four; a launch with fewer units per y than a block takes
// K_DPHI_SERIES.  Blocks of 2 NW units were 15 % slower at 512^2 x 100 rows: 804 blocks on 256 workgroups are 3 or 4
// each, where a wave's 6.3 lines are 6 or 7.)
template <int N, typename RO>
__global__ void __launch_bounds__((series_threads<N, RO>()), (512 / series_threads<N, RO>() > 1 ? 2 : 1))
k_dphi_series_q(const cx<double>* __restrict__ T, const double* __restrict__ sp,
                const TaskPar* __restrict__ tp, int ndir, int ntd, const RO* __restrict__ coef,
                const cx<double>* __restrict__ twk, double scale2, RO* __restrict__ D0t,
                float* __restrict__ dlin, int* __restrict__ zero17, const unsigned* __restrict__ support,
                const int* __restrict__ perm, int* __restrict__ queue, SeriesSkip skip) {
    constexpr int L = series_lanes<N>(), R = 64 / L, Q = N / L, H1 = N / 2 + 1, K = SeriesCfg<RO>::K;
    constexpr int THREADS = series_threads<N, RO>(), NW = THREADS / 64, NJ = fold_nj<Q>();
    constexpr bool WJREG = NJ <= 10;
    constexpr int LINE = N * K;                               // coefficients of a line
    constexpr int LINE_BYTES = LINE * (int)sizeof(RO), CHUNKS = LINE_BYTES / 1024;
    static_assert(LINE_BYTES % 1024 == 0, "a line is whole 1 KB pieces of LDS-DMA");
    constexpr int GB = NW;                                    // units per block: one per wave
    extern __shared__ __align__(16) unsigned char smem[];
    RO* scoef = reinterpret_cast<RO*>(smem);                                       // [2 sets][2 lines][N][K]
    cx<double>* swr = reinterpret_cast<cx<double>*>(smem + 4 * (size_t)LINE * sizeof(RO));   // [Q][L]
    __shared__ int s_next[2];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63, rho = lane / L, k2 = lane & (L - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nq = (ntd + R - 1) / R;                         // units per y (>= GB: the host's choice of kernel)
    const int C = H1 * nq, nblk = (C + GB - 1) / GB;
    if (zero17 != nullptr && blockIdx.x == 0 && threadIdx.x < kMfSchedInts) zero17[threadIdx.x] = 0;
    int blk = blockIdx.x;
    if (blk >= nblk) return;
    // the lines of block b into set s: wave w brings pieces w, w + NW, ... of the first and, when the block
    // straddles two y, of the second line
    auto issue_lines = [&](int b, int set) {
        const int y0 = (b * GB) / nq, y1 = min(C - 1, b * GB + GB - 1) / nq;
        for (int ln = 0; ln <= y1 - y0; ++ln) {
            const char* src = reinterpret_cast<const char*>(coef + (size_t)(y0 + ln) * LINE);
#pragma unroll
            for (int c0 = 0; c0 < CHUNKS; c0 += NW) {
                const int c = c0 + wave;
                if (c < CHUNKS)
                    glds16s(src + (size_t)c * 1024, (unsigned)lane * 16,
                            lds0 + (unsigned)(2 * set + ln) * LINE_BYTES + (unsigned)c * 1024);
            }
        }
    };
    // unit u (0 .. 2 NW - 1) of block b: y and the td of this lane's row of lanes (-1: no such unit; -2: the unit
    // exists, this row's line does not)
    auto unit_of = [&](int b, int u, int& y) -> int {
        const int c = b * GB + u;
        if (c >= C) return -1;
        y = c / nq;
        const int r = R * (c - y * nq) + rho;
        return r < ntd ? (perm != nullptr ? perm[r] : r) : -2;
    };
    const int td_last = perm != nullptr ? perm[ntd - 1] : ntd - 1;
    cx<double> xva[kNX], xvb[kNX];
    auto fetch = [&](int b, int u, cx<double>* xv) {
        int y = 0;
        const int t = unit_of(b, u, y);
        if (t == -1) return;
        const cx<double>* src = T + ((size_t)(t >= 0 ? t : td_last) * H1 + y) * NAO + (lane & 15);
#pragma unroll
        for (int a = 0; a < kNX; ++a) xv[a] = src[16 * a];
    };
    issue_lines(blk, 0);
    fetch(blk, wave, xva);
    for (int i = threadIdx.x; i < Q * L; i += THREADS) swr[i] = twk[NJ * L + i];
    cx<double> wjr[WJREG ? NJ : 1];
    if constexpr (WJREG) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wjr[j] = twk[j * L + k2];
    }
    if (threadIdx.x == 0) s_next[0] = (int)gridDim.x + atomicAdd(queue, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto unit = [&](int b, int u, const cx<double>* xv, int set) {
        int y = 0;
        const int t = unit_of(b, u, y);
        if (t == -1) return;                                    // (wave-uniform: c does not depend on the lane)
        const bool valid = t >= 0;
        const int td = valid ? t : td_last;
        const int task = td / ndir;
        const double r0m53 = tp[task].r0m53, delta = tp[task].inv_l0sq - kEps0;
        const double spv = sp[td];
        cx<double> wl[WJREG ? 1 : NJ];
        if constexpr (!WJREG) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) wl[j] = twk[j * L + k2];
        }
        const unsigned kmask = support != nullptr ? (unsigned)__builtin_amdgcn_readfirstlane((int)support[y]) : 0xffffffffu;
        if (skip.tlmax != nullptr && y >= 4) {                  // (the first lines carry the peak's lower bound: never)
            // B(y) = sum of the 80 magnitudes of the line: lane l holds inputs 16 a + l % 16 of ITS row's line
            float m = 0.f;
#pragma unroll
            for (int a = 0; a < kNX; ++a) {
                const float re = (float)xv[a].x, im = (float)xv[a].y;
                m += __builtin_sqrtf(fmaf(re, re, im * im));
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m += __shfl_xor(m, o, 64);
            // (float magnitudes and sums carry ~1e-6 of B: 1e-5 more of it keeps Lb a LOWER bound)
            const float lb = fmaxf((float)(scale2 * (spv - (double)m * 1.00001)), 0.f);
            const float e = fmaf(skip.c2max, lb, skip.tlmax[y]);
            const bool drop = e < skip.thr_elem || e + (float)(1 + __builtin_ctz(N & -N) + (N == 1280 ? 3 : 0)) < skip.thr_mass;
            if (__all(drop || !valid)) {
                if (valid) {
                    RO* dst = D0t + ((size_t)td * H1 + y) * N + k2;
#pragma unroll
                    for (int k1 = 0; k1 < Q; ++k1)
                        if ((kmask >> k1) & 1u) dst[L * k1] = (RO)1.0e30f;
                    if (dlin != nullptr && k2 < N / 32) dlin[((size_t)td * H1 + y) * (N / 32) + k2] = lb;
                }
                return;
            }
        }
        const int ln = y - (b * GB) / nq;                       // 0 or 1: which line of the block's set
        series_line<N, RO, L>(xv, WJREG ? wjr : wl, swr, scoef + (size_t)(2 * set + ln) * LINE, r0m53, delta, spv, scale2,
                              D0t + ((size_t)td * H1 + y) * N + k2,
                              dlin != nullptr ? dlin + ((size_t)td * H1 + y) * (N / 32) : nullptr, valid, lane, kmask);
    };
    // (two blocks per turn of the loop: the register sets of the inputs alternate at compile time.  The draw for the
    // block after next stands behind the unit: issued in front of it, the wave of thread 0 waits for the atomic's
    // round trip before its unit's own loads come back -- vmcnt retires in order -- and the block for that wave:
    // 42 us against 39.6 at 512^2, 205 against 193 at 1280^2.)
    while (true) {
        int nxt = s_next[0];
        bool more = nxt < nblk;
        if (more) {
            issue_lines(nxt, 1);
            fetch(nxt, wave, xvb);
        }
        unit(blk, wave, xva, 0);
        if (more && threadIdx.x == 0) s_next[1] = (int)gridDim.x + atomicAdd(queue, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!more) break;
        blk = nxt;
        nxt = s_next[1];
        more = nxt < nblk;
        if (more) {
            issue_lines(nxt, 0);
            fetch(nxt, wave, xva);
        }
        unit(blk, wave, xvb, 1);
        if (more && threadIdx.x == 0) s_next[0] = (int)gridDim.x + atomicAdd(queue, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!more) break;
        blk = nxt;
    }
}

// The first form, kept for what does not fit two coefficient slots in LDS (f64 mode at 1280^2):
// workgroup = (line y, group of tasks), the coefficients of the line in LDS once, every wave then walks
// its tasks alone.
template <int N, typename RO>
constexpr size_t series1_smem() {
    return (size_t)N * SeriesCfg<RO>::K * sizeof(RO) + (size_t)(N / 64) * 64 * sizeof(cx<double>);
}

template <int N, typename RO>
__global__ void __launch_bounds__(256)
k_dphi_series1(const cx<double>* __restrict__ T, const double* __restrict__ sp,
               const TaskPar* __restrict__ tp, int ndir, int ntd, int tg, const RO* __restrict__ coef,
               const cx<double>* __restrict__ twk, double scale2, RO* __restrict__ D0t,
               float* __restrict__ dlin, int* __restrict__ zero17, const unsigned* __restrict__ support) {
    constexpr int Q = N / 64, H1 = N / 2 + 1, K = SeriesCfg<RO>::K, THREADS = 256;
    constexpr int NJ = fold_nj<Q>(), JMIN = fold_jmin<Q>();
    constexpr bool WJREG = NJ <= 10;
    extern __shared__ __align__(16) unsigned char smem[];
    RO* scoef = reinterpret_cast<RO*>(smem);                                   // [N][K]
    cx<double>* swr = reinterpret_cast<cx<double>*>(smem + (size_t)N * K * sizeof(RO));   // [Q][64]
    const int y = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int td_end = min(ntd, (g + 1) * tg);
    int td = g * tg + wave;
    cx<double> xva[kNX], xvb[kNX];
    auto fetch = [&](int t, cx<double>* xv) {
        const cx<double>* src = T + ((size_t)t * H1 + y) * NAO + (lane & 15);
#pragma unroll
        for (int a = 0; a < kNX; ++a) xv[a] = src[16 * a];
    };
    if (td < td_end) fetch(td, xva);
    {
        using V4 = typename std::conditional<sizeof(RO) == 4, float4, double2>::type;
        constexpr int NV = (int)((size_t)N * K * sizeof(RO) / 16);
        const V4* src = reinterpret_cast<const V4*>(coef + (size_t)y * N * K);
        V4* dst = reinterpret_cast<V4*>(scoef);
        for (int i = threadIdx.x; i < NV; i += THREADS) dst[i] = src[i];
        for (int i = threadIdx.x; i < Q * 64; i += THREADS) swr[i] = twk[NJ * 64 + i];
    }
    if (zero17 != nullptr && y == 0 && g == 0 && threadIdx.x < kMfSchedInts) zero17[threadIdx.x] = 0;
    cx<double> wjr[WJREG ? NJ : 1];
    if constexpr (WJREG) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wjr[j] = twk[j * 64 + lane];
    }
    __syncthreads();
    const unsigned kmask = support != nullptr ? (unsigned)__builtin_amdgcn_readfirstlane((int)support[y]) : 0xffffffffu;
    auto line = [&](int td, const cx<double>* xv) {
        const int task = td / ndir;
        const double r0m53 = tp[task].r0m53, delta = tp[task].inv_l0sq - kEps0;
        const double spv = sp[td];
        cx<double> wl[WJREG ? 1 : NJ];
        if constexpr (!WJREG) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) wl[j] = twk[j * 64 + lane];
        }
        series_line<N, RO, 64>(xv, WJREG ? wjr : wl, swr, scoef, r0m53, delta, spv, scale2,
                               D0t + ((size_t)td * H1 + y) * N + lane,
                               dlin != nullptr ? dlin + ((size_t)td * H1 + y) * (N / 32) : nullptr, true, lane, kmask);
    };
    constexpr int STEP = THREADS / 64;
    while (td < td_end) {
        if (td + STEP < td_end) fetch(td + STEP, xvb);
        line(td, xva);
        td += STEP;
        if (td >= td_end) break;
        if (td + STEP < td_end) fetch(td + STEP, xva);
        line(td, xvb);
        td += STEP;
    }
}

// K_SERIES_SUPPORT: support[y] bit k1 = some column x in [L k1, L k1 + L) of line y has a non-zero telescope OTF
// (L = series_lanes<N>(): the pieces a lane of K_DPHI_SERIES owns).  Once per context.
template <int N, typename RT>
__global__ void __launch_bounds__(64) k_series_support(const RT* __restrict__ tel, unsigned* __restrict__ support) {
    constexpr int L = series_lanes<N>(), Q = N / L;
    const int y = blockIdx.x, lane = threadIdx.x;
    unsigned m = 0;
    for (int k1 = 0; k1 < Q; ++k1) {
        bool any = false;
        for (int x = L * k1 + lane; x < L * k1 + L; x += 64) any = any || tel[(size_t)y * N + x] > (RT)0;
        if (__ballot(any) != 0ull) m |= 1u << k1;
    }
    if (lane == 0) support[y] = m;
}

// Hd planes [K][H1][N] (fp64, the output of K_COLFFT_DPHI for the basis tasks) -> coef[y][x][K]
template <typename RO>
__global__ void __launch_bounds__(256) k_series_coef(int n, int K, const double* __restrict__ planes,
                                                     RO* __restrict__ coef) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < K; ++k) coef[(size_t)i * K + k] = (RO)planes[(size_t)k * n + i];
}

// K_DMIN16: dlin[td][y][kb] (the per-line block minima of K_DPHI_SERIES) -> dblk[td][y / 16][kb], the
// minima over the blocks of 16 lines x 32 columns, and dline[td][y], the minima per line (what K_DMIN
// computes from D itself).
__global__ void __launch_bounds__(128) k_dmin16(int H1, int nks, const float* __restrict__ dlin,
                                                float* __restrict__ dline, float* __restrict__ dblk) {
    const int mt = blockIdx.x, td = blockIdx.y, nmt = gridDim.x;
    const int nline = min(16, H1 - 16 * mt);
    const float* src = dlin + ((size_t)td * H1 + 16 * mt) * nks;
    if ((int)threadIdx.x < nks) {
        float m = __builtin_inff();
        for (int i = 0; i < nline; ++i) m = fminf(m, src[i * nks + threadIdx.x]);
        dblk[((size_t)td * nmt + mt) * nks + threadIdx.x] = m;
    } else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + nline) {
        const int i = threadIdx.x - 64;
        float m = __builtin_inff();
        for (int kb = 0; kb < nks; ++kb) m = fminf(m, src[i * nks + kb]);
        dline[(size_t)td * H1 + 16 * mt + i] = m;
    }
}

}  // namespace

void launch_dmin16(hipStream_t s, int N, int ntd, const float* d_dlin, float* d_dline, float* d_dblk) {
    const int H1 = N / 2 + 1;
    hipLaunchKernelGGL(k_dmin16, dim3((H1 + 15) / 16, ntd), dim3(128), 0, s, H1, N / 32, d_dlin, d_dline, d_dblk);
}

// the twiddle buffer: the table for 64 lanes per line (K_PATCH_ROWS, K_DPHI_SERIES1), then the one for
// series_lanes<N>() lanes per line (K_DPHI_SERIES)
size_t series_twiddle_bytes(int N) {
    size_t n = 0;
    DISPATCH_N(N, { n = (twiddle_entries<NN, 64>() + twiddle_entries<NN, series_lanes<NN>()>()) * sizeof(cx<double>); })
    return n;
}

void launch_series_twiddles(hipStream_t s, int N, const void* d_tw64, void* d_twk) {
    DISPATCH_N(N, {
        constexpr int L = series_lanes<NN>();
        hipLaunchKernelGGL((k_series_twiddles<NN, 64>), dim3(fold_nj<NN / 64>() + NN / 64), dim3(64), 0, s,
                           (const cx<double>*)d_tw64, (cx<double>*)d_twk);
        hipLaunchKernelGGL((k_series_twiddles<NN, L>), dim3(fold_nj<NN / L>() + NN / L), dim3(64), 0, s,
                           (const cx<double>*)d_tw64, (cx<double>*)d_twk + twiddle_entries<NN, 64>());
    })
}

int series_terms(bool f64) { return f64 ? SeriesCfg<double>::K : SeriesCfg<float>::K; }
double series_eps0() { return kEps0; }

void launch_series_support(hipStream_t s, int N, const void* d_tel, bool f64, unsigned* d_support) {
    DISPATCH_N(N, {
        if (f64)
            hipLaunchKernelGGL((k_series_support<NN, double>), dim3(NN / 2 + 1), dim3(64), 0, s, (const double*)d_tel, d_support);
        else
            hipLaunchKernelGGL((k_series_support<NN, float>), dim3(NN / 2 + 1), dim3(64), 0, s, (const float*)d_tel, d_support);
    })
}

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
                  double cfit, const void* d_twk, double* d_P, void* d_T, double* d_sp, bool f64,
                  const PatchExtras& x) {
    const int ntask = ntd / ndir;
    const bool copy = x.blob_src != nullptr;
    const dim3 ggrid((NAO * NAO + 255) / 256, ntask + (copy ? 1 : 0));
    ParamCopy pc;
    pc.dst = (uint4*)x.blob_dst; pc.src = (const uint4*)x.blob_src; pc.n16 = copy ? (int)(x.blob_bytes / 16) : 0;
    // (with the blob in this launch the task parameters are read where the host wrote them)
    if (copy) d_tp = x.tp_host;
    if (f64)
        hipLaunchKernelGGL(k_patch_gen<true>, ggrid, dim3(256), 0, s, ndir, d_tp, d_aotab, cfit, d_P, pc);
    else
        hipLaunchKernelGGL(k_patch_gen<false>, ggrid, dim3(256), 0, s, ndir, d_tp, d_aotab, cfit, d_P, pc);
    ParamFlag pf;
    pf.flag = copy ? x.flag : nullptr; pf.seq = x.seq; pf.queue_zero = x.queue_zero;
    KhatArgs kh;
    kh.nker = x.khat_n; kh.gam = x.khat_gam; kh.alp = x.khat_alp; kh.khat = x.khat_out;
    DISPATCH_N(N, {
        constexpr int NG = NAO * series_lanes<NN>() / 256;       // passes (of 4 x 64 / L rows) per td
        // passes per workgroup: around 512 workgroups in all (1024: +2 us at 512^2, +4 at 1280^2; 256: +2)
        int nb = (512 + ntd - 1) / ntd;
        if (nb > NG) nb = NG;
        if (nb < 1) nb = 1;
        const int qb = (NG + nb - 1) / nb;
        nb = (NG + qb - 1) / qb;
        const int ky = kh.nker > 0 ? (kh.nker + nb - 1) / nb : 0;       // rows of workgroups for the kernel spectra
        if (kh.nker <= 0) {
            constexpr size_t sm = patch_rows_smem_all<NN, void>();
            allow_smem((k_patch_rows<NN, void>), sm);
            hipLaunchKernelGGL((k_patch_rows<NN, void>), dim3(nb, ntd), dim3(256), sm, s, (const double*)d_P,
                               (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), (cx<double>*)d_T, d_sp, qb, ntd, kh, pf);
        } else if (x.khat_f64) {
            constexpr size_t sm = patch_rows_smem_all<NN, double>();
            allow_smem((k_patch_rows<NN, double>), sm);
            hipLaunchKernelGGL((k_patch_rows<NN, double>), dim3(nb, ntd + ky), dim3(256), sm, s, (const double*)d_P,
                               (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), (cx<double>*)d_T, d_sp, qb, ntd, kh, pf);
        } else {
            constexpr size_t sm = patch_rows_smem_all<NN, float>();
            allow_smem((k_patch_rows<NN, float>), sm);
            hipLaunchKernelGGL((k_patch_rows<NN, float>), dim3(nb, ntd + ky), dim3(256), sm, s, (const double*)d_P,
                               (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), (cx<double>*)d_T, d_sp, qb, ntd, kh, pf);
        }
    })
}

// the queue-fed form needs four line slots in LDS and at least a block of units per y
template <int N, typename RO>
bool series_q_usable(int ntd) {
    constexpr int R = 64 / series_lanes<N>(), GB = series_threads<N, RO>() / 64;
    return series_q_fits<N, RO>() && (ntd + R - 1) / R >= GB;
}

void launch_dphi_series(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp, const void* d_T,
                        const double* d_sp, const void* d_coef, const void* d_twk, double scale2,
                        void* d_D0t, float* d_dlin, bool f64out, int* d_zero, int ncu, const unsigned* d_support,
                        const SeriesQueue& qx) {
    const int H1 = N / 2 + 1;
    SeriesSkip sk;
    sk.tlmax = qx.tlmax; sk.c2max = qx.c2max; sk.thr_elem = qx.thr_elem; sk.thr_mass = qx.thr_mass;
    auto first_form = [&](auto kernel, size_t sm) {
        // task groups: enough workgroups to fill the GPU several times over, every workgroup's table load
        // shared by as many tasks as that allows, every wave of a workgroup the same number of lines
        const int waves = 4;
        int ngr = (2048 + H1 - 1) / H1;
        if (ngr * waves > ntd) ngr = (ntd + waves - 1) / waves;
        if (ngr < 1) ngr = 1;
        int tg = (ntd + ngr - 1) / ngr;
        tg = (tg + waves - 1) / waves * waves;
        ngr = (ntd + tg - 1) / tg;
        return std::make_pair(dim3(H1, ngr), tg);
    };
    DISPATCH_N(N, {
        if (f64out) {
            if constexpr (series_fits<NN, double>()) {
                constexpr size_t sm = series_smem<NN, double>();
                if (qx.queue != nullptr && series_q_usable<NN, double>(ntd)) {
                    constexpr size_t smq = series_q_smem<NN, double>() <= 160 * 1024 ? series_q_smem<NN, double>() : 0;
                    allow_smem((k_dphi_series_q<NN, double>), smq);
                    hipLaunchKernelGGL((k_dphi_series_q<NN, double>), dim3(ncu), dim3(series_threads<NN, double>()), smq, s,
                                       (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, (const double*)d_coef,
                                       (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), scale2, (double*)d_D0t, d_dlin, d_zero,
                                       d_support, qx.perm, qx.queue, sk);
                    return;
                }
                allow_smem((k_dphi_series<NN, double>), sm);
                hipLaunchKernelGGL((k_dphi_series<NN, double>), dim3(ncu), dim3(series_threads<NN, double>()), sm, s,
                                   (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, (const double*)d_coef,
                                   (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), scale2, (double*)d_D0t, d_dlin, d_zero, d_support);
            } else {
                constexpr size_t sm = series1_smem<NN, double>();
                allow_smem((k_dphi_series1<NN, double>), sm);
                const auto gt = first_form(0, sm);
                hipLaunchKernelGGL((k_dphi_series1<NN, double>), gt.first, dim3(256), sm, s,
                                   (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, gt.second, (const double*)d_coef,
                                   (const cx<double>*)d_twk, scale2, (double*)d_D0t, d_dlin, d_zero, d_support);
            }
        } else {
            constexpr size_t sm = series_smem<NN, float>();
            if (qx.queue != nullptr && series_q_usable<NN, float>(ntd)) {
                constexpr size_t smq = series_q_smem<NN, float>();
                allow_smem((k_dphi_series_q<NN, float>), smq);
                hipLaunchKernelGGL((k_dphi_series_q<NN, float>), dim3(ncu), dim3(series_threads<NN, float>()), smq, s,
                                   (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, (const float*)d_coef,
                                   (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), scale2, (float*)d_D0t, d_dlin, d_zero,
                                   d_support, qx.perm, qx.queue, sk);
                return;
            }
            allow_smem((k_dphi_series<NN, float>), sm);
            hipLaunchKernelGGL((k_dphi_series<NN, float>), dim3(ncu), dim3(series_threads<NN, float>()), sm, s,
                               (const cx<double>*)d_T, d_sp, d_tp, ndir, ntd, (const float*)d_coef,
                               (const cx<double>*)d_twk + twiddle_entries<NN, 64>(), scale2, (float*)d_D0t, d_dlin, d_zero, d_support);
        }
    })
}

}  // namespace mpsfr
