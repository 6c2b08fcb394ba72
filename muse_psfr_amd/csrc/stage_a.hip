// HIP kernels of the muse-psfr PSF-reconstruction hot path, written for gfx950 (MI355X, wave64).
// Reference citations are to /root/reference/muse_psfr/psfrec.py.  See DESIGN.md for the data
// layout and the derivation of the restructured algorithm.
//
// Stage A, once per (task, direction): AO tables, telescope OTF, PSD -> structure function.
#include "device_common.h"
#include "psd_model.h"

namespace mpsfr {

namespace {

// ------------------------------------------------------------------------------------------
// K_AO_TABLES: row-independent tables of the AO-corrected zone (dsp4muse, psfrec.py:531-613 with
// calc_mat_rec_glao_finale :218-364 (LSE, one DM layer) and calc_dsp_res_glao_finale :367-528).
// tab[geom][dir][{T0,T1,noise}][a][b] with (a, b) = (fy index, fx index), i.e. already
// transposed as psfrec.py:613 does, so that  PSD_AO[a][b] = VK * (cn2_0 T0 + cn2_1 T1) + noise.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ao_tables(AoGeom g, const uint8_t* __restrict__ mrec,
                                                   const uint8_t* __restrict__ mres,
                                                   double* __restrict__ tab) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= NAO * NAO) return;
    const int d = blockIdx.y, geom = blockIdx.z;
    const int i = pix / NAO, j = pix % NAO;              // i <-> fx, j <-> fy (reference layout)
    const int ki = i < NAO / 2 ? i : i - NAO, kj = j < NAO / 2 ? j : j - NAO;
    const double fx = ki / 16.0, fy = kj / 16.0;         // fftfreq(80, 0.2), psfrec.py:548
    const double f = sqrt(fx * fx + fy * fy);
    // psfrec.py:552-554 + :241-242: arg = arctan(fy/fx) folds the grid onto fx >= 0
    const double gx = fabs(fx), gy = ki < 0 ? -fy : fy;
    bool mr, ms;
    if (mrec != nullptr) {
        mr = mrec[pix] != 0;
        ms = mres[pix] != 0;
    } else {   // exact rule on the integer grid: fc = 1.5 = 24/16 (psfrec.py:254-257, 432-435)
        const int ai = ki < 0 ? -ki : ki, aj = kj < 0 ? -kj : kj;
        mr = ai >= 24 || aj >= 24;
        ms = ai > 24 || aj > 24;
    }
    const double pitch = 8.0 / 24.0;
    const double wamp = 2.0 * kPi * f * sinc_pi(pitch * gx) * sinc_pi(pitch * gy);  // |wfs|, :252
    const int n = g.nlgs[geom];
    const bool haveW = !mr && wamp != 0.0 && pix != 0;     // psfrec.py:339, 351-352
    const double b0 = g.dir[0][d], b1 = g.dir[1][d];
    const double bf = b0 * gx + b1 * gy;
    const double theta_dm = 2.0 * kPi * 1.0 * kArcminH * bf;                       // :464
    double T[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const double ph = 2.0 * kPi * (g.h[l] * kArcminH * bf -
                                       (g.wind[0][l] * kDeltaT * gx + g.wind[1][l] * kDeltaT * gy));
        double pr, pi_;
        sincos(ph, &pi_, &pr);                                                      // :454-457
        if (haveW && !ms) {
            // sum_g (PbetaDM W_g) Mv[l,g] = (www/n) sum_g exp(i (theta_dm + psi_lg - phi_g))
            const double www = sinc_pi(g.wind[0][l] * kTi * gx + g.wind[1][l] * kTi * gy);  // :442
            double sr = 0.0, si = 0.0;
            for (int q = 0; q < n; ++q) {
                const double pf = gx * g.poslgs[geom][0][q] + gy * g.poslgs[geom][1][q];
                const double phi = 2.0 * kPi * pf * 1.0 * kArcminH;                 // :279-281
                const double psi = 2.0 * kPi * pf * g.h[l] * kArcminH;              // :440-443
                double s_, c_;
                sincos(theta_dm + psi - phi, &s_, &c_);
                sr += c_;
                si += s_;
            }
            pr -= www / n * sr;
            pi_ -= www / n * si;
        }
        T[l] = pr * pr + pi_ * pi_;                                                 // :489
    }
    double noise = haveW ? 1.0 / (n * wamp * wamp) : 0.0;                           // :515
    if (pix == 0) { T[0] = 0.0; T[1] = 0.0; noise = 0.0; }                          // :490, :516
    double* o = tab + ((size_t)(geom * g.ndir + d) * 3) * (NAO * NAO) + j * NAO + i;  // transpose :613
    o[0] = T[0];
    o[NAO * NAO] = T[1];
    o[2 * NAO * NAO] = noise;
}

// ------------------------------------------------------------------------------------------
// K_TEL_OTF: telescope OTF (psfrec.py:784-790) as the exact integer autocorrelation of the pupil
// mask: fft2(|ifft2(tab)|^2) = (tab (*) tab) / N^2 for a real 0/1 array, so
// dlFTO[u][v] * N^2 = count(u, v) / sum(pup).  Rows of the pupil are bit masks.
// telT[v][u], v in [0, N/2], u in [0, N) (transposed half plane; the OTF is even and symmetric).
// ------------------------------------------------------------------------------------------
template <typename RO>
__global__ void __launch_bounds__(256) k_tel_otf(int N, const uint64_t* __restrict__ rows,
                                                 int words, double pupsum, RO* __restrict__ telT) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y;
    if (u >= N) return;
    const int H = N / 2;
    const int su = u < H ? u : u - N;
    const int wpad = 2 * words + 1;
    const int ws = v >> 6, bs = v & 63;
    long count = 0;
    const int p0 = su < 0 ? -su : 0, p1 = su < 0 ? H : H - su;
    for (int p = p0; p < p1; ++p) {
        const uint64_t* A = rows + (size_t)p * wpad;
        const uint64_t* B = rows + (size_t)(p + su) * wpad;
        for (int w = 0; w < words; ++w) {
            uint64_t sh = B[w + ws] >> bs;
            if (bs) sh |= B[w + ws + 1] << (64 - bs);
            count += __popcll(A[w] & sh);
        }
    }
    telT[(size_t)v * N + u] = (RO)((double)count / pupsum);
}

// ------------------------------------------------------------------------------------------
// K_PSD_ROWFFT: residual phase PSD (simul_psd_wfm psfrec.py:36-151: psd_fit :616-626 outside /
// max(fit, AO) inside the 80x80 corrected zone :148-149) generated on the fly in FFT-native
// layout, and its forward FFT along the row.
//
// Only N/2 + 40 of the N rows are distinct: outside the corrected zone the PSD depends on the row
// only through (su + 1/2)^2 (half-pixel grid of psfrec.py:618), so rows su and -1-su are equal.
// The distinct rows are su in [-40, N/2), stored compactly and transposed as
// Ct[td][y][su + 40], y in [0, N/2];
// K_COLFFT_DPHI mirrors them back.  Two real rows share one complex transform
// (z = row_a + i row_b; C_a = (Z[y] + conj Z[-y])/2, C_b = (Z[y] - conj Z[-y])/2i).
// ------------------------------------------------------------------------------------------
template <int N>
constexpr int psd_rows() { return N / 2 + NAO / 2; }

// Lines per workgroup of the two fp64 line-FFT kernels of stage A.  A 1280-point fp64 line is 22.5 KB
// of LDS (padded) and a wave: with the plan's two lines per workgroup (plus the twiddle table, 68 KB)
// two workgroups = FOUR waves fit a CU, and the kernels ran at a third of the bytes per second they
// reach at 512^2.  Six lines and the table are 157.5 KB -- one workgroup of six waves per CU, and the
// transposed stores of K_PSD_ROWFFT come in 192-byte pieces instead of 64.
#ifndef MPSFR_A_SLOTS_ROW
#define MPSFR_A_SLOTS_ROW 6
#endif
#ifndef MPSFR_A_SLOTS_COL
#define MPSFR_A_SLOTS_COL 6
#endif
// (COL: K_COLFFT_DPHI; otherwise K_PSD_ROWFFT)
template <int N, bool COL>
constexpr int a_slots() { return N == 1280 ? (COL ? MPSFR_A_SLOTS_COL : MPSFR_A_SLOTS_ROW) : Plan<N>::SLOTS; }
template <int N, bool COL>
constexpr int a_threads() { return a_slots<N, COL>() * Plan<N>::TPR; }
// The pass twiddles of a thread live in registers where the plan needs at most 20 of them (every grid
// but 1280): no table in LDS (a quarter of the LDS reads of a transform, and a fifth of the workgroup's
// LDS -- one more workgroup per CU), and a thread fetches its 12-19 values straight from the table
// in memory instead of the workgroup copying all N.
template <int N>
constexpr bool a_regtw() { return use_reg_twiddles<N>(); }
template <int N, bool COL>
constexpr size_t a_smem() {
    return (size_t)((a_regtw<N>() ? 0 : 1) + fft_nbuf<N>() * a_slots<N, COL>()) * LineCfg<N>::NPAD * sizeof(cx<double>);
}

// F64: the f64 mode (two Newton steps in x^(-11/6)).  dcpart[td][workgroup]: the workgroup's share of
// S00 = sum of the PSD = Re sum_r C[td][r][0] (bg[0,0], psfrec.py:721; compact rows with su >= 40
// stand for two rows) -- K_COLFFT_DPHI adds the shares in a fixed order.
template <int N, bool F64>
__global__ void __launch_bounds__((a_threads<N, false>()))
k_psd_rowfft(int ndir, const TaskPar* __restrict__ tp, const double* __restrict__ aotab,
             double cfit, cx<double>* __restrict__ C, const cx<double>* __restrict__ twg,
             double* __restrict__ dcpart) {
    using L = LineCfg<N>;
    constexpr int TPR = L::TPR, SLOTS = a_slots<N, false>(), THREADS = a_threads<N, false>(), NPAD = L::NPAD;
    constexpr int EPT = N / TPR, NR = psd_rows<N>();
    constexpr int NEWTON = F64 ? 2 : 1;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool REGTW = a_regtw<N>();
    cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
    cx<double>* bufA = tw + (REGTW ? 0 : NPAD);
    cx<double>* bufB = bufA + SLOTS * NPAD;    // only used when a slot spans two wavefronts
    const int slot = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int pair = blockIdx.x * SLOTS + slot;             // rows 2 pair, 2 pair + 1 (compact)
    const int td = blockIdx.y;
    const int task = td / ndir, d = td % ndir;
    TwRegs<double, N> twr;
    const cx<double>* twp = tw;
    if constexpr (REGTW) {
        twr.init(twg, t);
        twp = twr.w;
    } else {
        for (int i = threadIdx.x; i < N; i += THREADS) tw[lds_pad(i)] = twg[i];
    }
    const TaskPar p = tp[task];
    const bool valid = 2 * pair < NR;
    const int ca = valid ? 2 * pair : 0, cb = ca + 1;       // NR is even
    const int sua = ca - NAO / 2, sub = cb - NAO / 2;
    const double* tb = aotab + ((size_t)(p.geom * ndir + d) * 3) * (NAO * NAO);
    cx<double> x[EPT];
    if constexpr (TPR <= 64 && EPT % 2 == 0) {
        // The fitting term depends on the column through (sv + 1/2)^2 only: columns sv and -1-sv are
        // equal, bit for bit.  Column c = t + e TPR has its mirror N-1-c = (TPR-1-t) + (EPT-1-e) TPR:
        // a lane evaluates the first half of its elements and takes the others from lane TPR-1-t of
        // its slot (x^(-11/6) is what this kernel spends its vector time on).
#pragma unroll
        for (int e = 0; e < EPT / 2; ++e) {
            const int sv = t + e * TPR;                          // < N/2
            x[e] = {psd_fit_value<NEWTON>(sua, sv, p, cfit), psd_fit_value<NEWTON>(sub, sv, p, cfit)};
        }
#pragma unroll
        for (int e = 0; e < EPT / 2; ++e) {
            x[EPT - 1 - e].x = __shfl_xor(x[e].x, TPR - 1, 64);
            x[EPT - 1 - e].y = __shfl_xor(x[e].y, TPR - 1, 64);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = t + e * TPR;
            const int sv = c < N / 2 ? c : c - N;
            x[e] = {psd_fit_value<NEWTON>(sua, sv, p, cfit), psd_fit_value<NEWTON>(sub, sv, p, cfit)};
        }
    }
    if (sua < NAO / 2 && p.basis == 0) {                        // (wave-uniform when TPR >= 64)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int c = t + e * TPR;
            const int sv = c < N / 2 ? c : c - N;
            x[e] = {psd_with_ao<NEWTON>(x[e].x, sua, sv, p, tb), psd_with_ao<NEWTON>(x[e].y, sub, sv, p, tb)};
        }
    }
    if constexpr (!REGTW) __syncthreads();      // twiddle table
    const cx<double>* res = fft_forward_regs<double, N, REGTW>(x, bufA + slot * NPAD,
                                                               bufB + slot * NPAD, twp, t);
    // Store transposed, Ct[td][y][compact row], so that K_COLFFT_DPHI reads whole columns
    // contiguously: the workgroup's 2*SLOTS rows of one y form a 32*SLOTS-byte segment; lanes run
    // over (y, row) with the row fastest, each unpacking its value from the slot buffers.
    __syncthreads();
    constexpr int RW = 2 * SLOTS;                       // rows per workgroup
    static_assert(RW <= 64, "the DC share is summed by the first RW lanes");
    const int row0 = blockIdx.x * RW;
    cx<double>* Ct = C + (size_t)td * (N / 2 + 1) * NR;
    auto slot_buf = [&](int rr) {
        return (L::WSYNC ? bufA : (res == bufA + slot * NPAD ? bufA : bufB)) + (rr >> 1) * NPAD;
    };
    if (threadIdx.x < 64) {        // DC share: Z[0] = (sum of row a) + i (sum of row b)
        const int rr = threadIdx.x;
        double v = 0.0;
        if (rr < RW && row0 + rr < NR) {
            const cx<double> z = slot_buf(rr)[lds_out<N, 16>(0)];
            v = (row0 + rr >= NAO ? 2.0 : 1.0) * ((rr & 1) == 0 ? z.x : z.y);
        }
        // (a fixed order: the lanes beyond RW hold zeros)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) dcpart[(size_t)td * gridDim.x + blockIdx.x] = v;
    }
    for (int idx = threadIdx.x; idx < (N / 2 + 1) * RW; idx += THREADS) {
        const int y = idx / RW, rr = idx - y * RW;
        if (row0 + rr >= NR) continue;
        const cx<double>* rs = slot_buf(rr);
        const cx<double> z = rs[lds_out<N, 16>(y)], zm = rs[lds_out<N, 16>(y == 0 ? 0 : N - y)];
        cx<double> o;
        if ((rr & 1) == 0) o = {0.5 * (z.x + zm.x), 0.5 * (z.y - zm.y)};
        else o = {0.5 * (z.y + zm.y), -0.5 * (z.x - zm.x)};
        Ct[(size_t)y * NR + row0 + rr] = o;
    }
}

// ------------------------------------------------------------------------------------------
// K_COLFFT_DPHI: column FFTs of C and the structure function (psfrec.py:717-722 without the
// wavelength factor): D0t[td][y][x] = 2 scale (S00 - Re S[x][y]), y in [0, N/2], x in [0, N).
// ------------------------------------------------------------------------------------------
template <int N, typename RO>
__global__ void __launch_bounds__((a_threads<N, true>()))
k_colfft_dphi(const cx<double>* __restrict__ C, const double* __restrict__ dcpart, int nparts,
              double scale2, RO* __restrict__ D0t, const cx<double>* __restrict__ twg,
              int* __restrict__ zero17) {
    using L = LineCfg<N>;
    constexpr int TPR = L::TPR, SLOTS = a_slots<N, true>(), THREADS = a_threads<N, true>(), NPAD = L::NPAD;
    constexpr int NR = psd_rows<N>(), NLD = (NR + TPR - 1) / TPR;
    constexpr int NYG = (N / 2 + 1 + SLOTS - 1) / SLOTS;          // groups of SLOTS columns
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool REGTW = a_regtw<N>();
    cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
    cx<double>* bufA = tw + (REGTW ? 0 : NPAD);
    cx<double>* bufB = bufA + SLOTS * NPAD;
    const int slot = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int td = blockIdx.y;
    const cx<double>* Ct = C + (size_t)td * (N / 2 + 1) * NR;
    // A workgroup walks the column groups blockIdx.x, blockIdx.x + gridDim.x, ...: the values of the
    // next column are in flight (registers) behind the transform of the current one -- one workgroup
    // per column group spent 47 % of its wave cycles waiting for its loads of C.
    // DIRECT (every grid but 1280): a thread loads the column in the layout of the FIRST PASS, element
    // n = t + e TPR of the mirrored line straight from its compact row (su >= -40: row su; below: the
    // row -1-su that stands for it), so the line is never staged in LDS: together with the last pass
    // kept in registers, a 512-point column costs 32 LDS accesses per lane instead of 66.  (At 1280 the
    // 20 elements per lane do not fit beside the radix-20 pass: the column is staged.)
    // (and at 256, two lines per wave, the staged form measured 7 % faster)
    constexpr bool DIRECT = N != 1280 && N != 256;
    constexpr int EPT = N / TPR;
    cx<double> pre[DIRECT ? EPT : NLD];
    auto fetch = [&](int yg) {
        const int y = yg * SLOTS + slot;
        const cx<double>* col = Ct + (size_t)(y <= N / 2 ? y : 0) * NR;
        if constexpr (DIRECT) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int n = t + e * TPR;
                const int su = n < N / 2 ? n : n - N;
                const int ci = (su >= -NAO / 2 ? su : -1 - su) + NAO / 2;
                pre[e] = y <= N / 2 ? col[ci] : cx<double>{0.0, 0.0};
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int ci = t + k * TPR;
                pre[k] = (ci < NR && y <= N / 2) ? col[ci] : cx<double>{0.0, 0.0};
            }
        }
    };
    fetch(blockIdx.x);
    TwRegs<double, N> twr;
    const cx<double>* twp = tw;
    if constexpr (REGTW) {
        twr.init(twg, t);
        twp = twr.w;
    } else {
        for (int i = threadIdx.x; i < N; i += THREADS) tw[lds_pad(i)] = twg[i];
        __syncthreads();        // the table is the one thing the waves of the workgroup share
    }
    // the counters of the matrix-core stage's work lists (K_MF_PREP, K_OTF_MFMA2) start from zero
    if (zero17 != nullptr && td == 0 && blockIdx.x == 0 && threadIdx.x < kMfSchedInts) zero17[threadIdx.x] = 0;
    // S00 = the shares of K_PSD_ROWFFT's workgroups, added in an order fixed by N alone (every wave
    // of every workgroup gets the same bits)
    double dc = 0.0;
    for (int b = threadIdx.x & 63; b < nparts; b += 64) dc += dcpart[(size_t)td * nparts + b];
    dc = wave_sum(dc);
    for (int yg = blockIdx.x; yg < NYG; yg += gridDim.x) {
        const int y = yg * SLOTS + slot;
        if constexpr (DIRECT) {
            // first pass from the registers, the next column requested behind it, the last pass into
            // registers: entry (b, q) is element t + b TPR + q N / RADIX of the transform, so for every
            // (b, q) the lanes of the slot hold consecutive x and the stores of D are whole cache lines
            constexpr int LR = fft_last_radix<N>(), LB = fft_last_nbt<N>();
            cx<double> vout[LB * LR];
            fft_pass<double, N, 0, REGTW, true>(pre, bufA + slot * NPAD, twp, t);
            fft_sync<L::WSYNC>();
            if (yg + (int)gridDim.x < NYG) fetch(yg + gridDim.x);
            fft_rest_lastreg<double, N, REGTW, 1>(bufA + slot * NPAD, bufB + slot * NPAD, twp, t, vout);
            if (y <= N / 2) {
                RO* out = D0t + ((size_t)td * (N / 2 + 1) + y) * N + t;
#pragma unroll
                for (int b = 0; b < LB; ++b)
#pragma unroll
                    for (int q = 0; q < LR; ++q)
                        out[b * TPR + q * (N / LR)] = (RO)(scale2 * (dc - vout[b * LR + q].x));
            }
        } else {
            // column y: NR contiguous compact rows; rows su >= 40 also stand for row -1-su
            cx<double>* dst = bufA + slot * NPAD;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int ci = t + k * TPR;
                if (ci < NR) {
                    const int su = ci - NAO / 2;
                    dst[lds_pad(su < 0 ? su + N : su)] = pre[k];
                    if (su >= NAO / 2) dst[lds_pad(N - 1 - su)] = pre[k];
                }
            }
            fft_sync<L::WSYNC>();
            if (yg + (int)gridDim.x < NYG) fetch(yg + gridDim.x);
            const cx<double>* res =
                fft_forward<double, N, REGTW>(bufA + slot * NPAD, bufB + slot * NPAD, twp, t);
            if (y <= N / 2) {
                RO* out = D0t + ((size_t)td * (N / 2 + 1) + y) * N;
                for (int x = t; x < N; x += TPR) out[x] = (RO)(scale2 * (dc - res[lds_out<N, 16>(x)].x));
            }
        }
        // (a slot of at most one wave transforms and reads its own line: no workgroup barrier, the
        // waves of a workgroup run free of each other)
        fft_sync<L::WSYNC>();       // all reads of the line are done: the buffer may be rewritten
    }
}

// ------------------------------------------------------------------------------------------
// Stage-level entry points (mpsfr_simul_psd / mpsfr_psf_from_psd; psfrec.py:36-151, :644-686 with
// :717-722).  Not on the hot path: a caller's PSD is an arbitrary real image, so none of the
// symmetries of the model hold -- every row is transformed, the columns are gathered with their
// natural stride.
// K_PSD_IMAGE: the PSD of a task as the reference returns it: centred (DC at [N/2][N/2]), times
// (lambda_ref 1000 / 2 pi)^2 (psfrec.py:151), psd[d][row][col].
// K_PSDMEM_ROWS: Cm[d][r][y] = sum_c psd(r, c) W^(c y), r, c in FFT layout, y in [0, N/2].
// K_PSDMEM_COLS: D0t[d][y][x] = scale (S00 - Re sum_r Cm[d][r][y] W^(r x)), S00 = Re sum_r Cm[d][r][0].
// ------------------------------------------------------------------------------------------
template <int NEWTON>
__global__ void __launch_bounds__(256) k_psd_image(int N, int ndir, TaskPar p, const double* __restrict__ aotab,
                                                   double cfit, double unit, double* __restrict__ psd) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, d = blockIdx.z;
    if (col >= N) return;
    const int su = row - N / 2, sv = col - N / 2;           // centred image -> signed frequency index
    const double* tb = aotab + ((size_t)(p.geom * ndir + d) * 3) * (NAO * NAO);
    const double fit = psd_fit_value<NEWTON>(su, sv, p, cfit);
    psd[((size_t)d * N + row) * N + col] = unit * psd_with_ao<NEWTON>(fit, su, sv, p, tb);
}

template <int N>
constexpr size_t psdmem_smem() { return (size_t)(1 + 2 * LineCfg<N>::SLOTS) * LineCfg<N>::NPAD * sizeof(cx<double>); }

template <int N>
__global__ void __launch_bounds__((LineCfg<N>::THREADS))
k_psdmem_rows(const double* __restrict__ psd, cx<double>* __restrict__ Cm, const cx<double>* __restrict__ twg) {
    using L = LineCfg<N>;
    constexpr int TPR = L::TPR, SLOTS = L::SLOTS, THREADS = L::THREADS, NPAD = L::NPAD, EPT = N / TPR;
    extern __shared__ __align__(16) unsigned char smem[];
    cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
    cx<double>* bufA = tw + NPAD;
    cx<double>* bufB = bufA + SLOTS * NPAD;
    const int slot = threadIdx.x / TPR, t = threadIdx.x % TPR, d = blockIdx.y;
    const int r = blockIdx.x * SLOTS + slot;               // FFT-layout row (N is a multiple of SLOTS)
    for (int i = threadIdx.x; i < N; i += THREADS) tw[lds_pad(i)] = twg[i];
    const double* src = psd + ((size_t)d * N + (r + N / 2) % N) * N;
    cx<double> x[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = {src[(t + e * TPR + N / 2) % N], 0.0};
    __syncthreads();
    const cx<double>* res = fft_forward_regs<double, N, false>(x, bufA + slot * NPAD, bufB + slot * NPAD, tw, t);
    cx<double>* dst = Cm + ((size_t)d * N + r) * (N / 2 + 1);
    for (int y = t; y <= N / 2; y += TPR) dst[y] = res[lds_out<N, 16>(y)];
}

template <int N, typename RO>
__global__ void __launch_bounds__((LineCfg<N>::THREADS))
k_psdmem_cols(const cx<double>* __restrict__ Cm, double scale, RO* __restrict__ D0t,
              const cx<double>* __restrict__ twg) {
    using L = LineCfg<N>;
    constexpr int TPR = L::TPR, SLOTS = L::SLOTS, THREADS = L::THREADS, NPAD = L::NPAD, EPT = N / TPR, H1 = N / 2 + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double sdc[256];
    cx<double>* tw = reinterpret_cast<cx<double>*>(smem);
    cx<double>* bufA = tw + NPAD;
    cx<double>* bufB = bufA + SLOTS * NPAD;
    const int slot = threadIdx.x / TPR, t = threadIdx.x % TPR, d = blockIdx.y;
    const int y = blockIdx.x * SLOTS + slot;
    const bool on = y < H1;
    const cx<double>* Cd = Cm + (size_t)d * N * H1;
    for (int i = threadIdx.x; i < N; i += THREADS) tw[lds_pad(i)] = twg[i];
    // S00: the column y = 0 summed in an order fixed by the launch geometry (every workgroup the same bits)
    double a = 0.0;
    for (int r = threadIdx.x; r < N; r += THREADS) a += Cd[(size_t)r * H1].x;
    sdc[threadIdx.x] = a;
    cx<double> x[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = on ? Cd[(size_t)(t + e * TPR) * H1 + y] : cx<double>{0.0, 0.0};
    __syncthreads();
    double dc = 0.0;
    for (int i = 0; i < THREADS; ++i) dc += sdc[i];
    const cx<double>* res = fft_forward_regs<double, N, false>(x, bufA + slot * NPAD, bufB + slot * NPAD, tw, t);
    if (on) {
        RO* out = D0t + ((size_t)d * H1 + y) * N;
        for (int xx = t; xx < N; xx += TPR) out[xx] = (RO)(scale * (dc - res[lds_out<N, 16>(xx)].x));
    }
}

// ------------------------------------------------------------------------------------------
// K_DMIN: minima of D for the pruning of the per-wavelength stage, one workgroup per (td, group of
// 16 lines): dline[td][v] = min_u max(D[v][u], 0) and dblk[td][v / 16][u / 32] = the minimum over
// the 16 x 32 block (the blocks of K_OTF_MFMA, otf_mfma.hip).  A separate pass over D (just
// written, read back from the cache hierarchy at ~10 us per 52 MB) costs a third of what the same
// minima cost inside K_COLFFT_DPHI, whose column transforms then wait for the reductions.
// ------------------------------------------------------------------------------------------
// (T = float: mixed mode; T = double: f64 mode -- the minima are rounded DOWN to float, so that the
// bounds built from them stay upper bounds)
template <typename T> struct DminVec;
template <> struct DminVec<float> {
    typedef float4 V;
    static constexpr int W = 4;
    static __device__ __forceinline__ float vmin(const V& d) {
        return fminf(fminf(d.x, d.y), fminf(d.z, d.w));
    }
};
template <> struct DminVec<double> {
    typedef double2 V;
    static constexpr int W = 2;
    static __device__ __forceinline__ float vmin(const V& d) { return __double2float_rd(fmin(d.x, d.y)); }
};

template <typename T>
__global__ void __launch_bounds__(256) k_dmin(int N, const T* __restrict__ D0t,
                                              float* __restrict__ dline, float* __restrict__ dblk) {
    using DV = DminVec<T>;
    constexpr int VPB = 32 / DV::W;                  // consecutive vectors of one block of 32 columns
    constexpr int MAXKS = 1280 / 32;
    __shared__ int sblk[MAXKS], sline[16];          // bits of non-negative floats: integer order
    const int H1 = N / 2 + 1, nks = N / 32, nmt = (H1 + 15) / 16;
    const int mt = blockIdx.x, td = blockIdx.y;
    if (threadIdx.x < MAXKS) sblk[threadIdx.x] = 0x7f800000;
    if (threadIdx.x < 16) sline[threadIdx.x] = 0x7f800000;
    __syncthreads();
    const int nq = N / DV::W;                        // vectors per line
    const typename DV::V* src = reinterpret_cast<const typename DV::V*>(D0t + ((size_t)td * H1 + 16 * mt) * N);
    const int nline = min(16, H1 - 16 * mt);
    for (int f = threadIdx.x; f < nline * nq; f += 256) {
        float m = fmaxf(DV::vmin(src[f]), 0.f);
        // VPB consecutive vectors are one block of 32 columns (nq is a multiple of VPB)
#pragma unroll
        for (int o = 1; o < VPB; o <<= 1) m = fminf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & (VPB - 1)) == 0) {
            const int line = f / nq, kb = (f - line * nq) / VPB;
            atomicMin(&sblk[kb], __float_as_int(m));
            atomicMin(&sline[line], __float_as_int(m));
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nks) dblk[((size_t)td * nmt + mt) * nks + threadIdx.x] = __int_as_float(sblk[threadIdx.x]);
    if ((int)threadIdx.x < nline) dline[(size_t)td * H1 + 16 * mt + threadIdx.x] = __int_as_float(sline[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// Line pruning of the per-wavelength stage.  The OTF of a line is tel * sum_dir exp(c D) with
// c < 0 and D >= 0, so with  A_v = min over directions and u of D[v][u]  (K_COLFFT_DPHI) and
// B_v = log2 max_u tel[v][u]  (K_TEL_LINEMAX, once per context) every element of line v is at most
// 2^(c' A_v + B_v)  (c' = c log2 e), and the line (both half planes, all directions) weighs at
// most 2 N ndir times that.  The PSF peak is the sum of the whole OTF >= OTF[0][0] = 1, so dropping
// the lines v >= vkeep with  sum_{v >= vkeep} 2 N ndir 2^(c' A_v + B_v) <= eps  changes no stamp
// pixel by more than eps of the peak.  The bound grows with the wavelength, so a pair of
// wavelengths uses its longer one.  K_VKEEP: vkeep[task][pair] = the number of lines to transform;
// with seeing-limited PSFs most of the half plane is far below fp32 resolution (bench workload:
// 46 % of the lines survive eps = 1e-9; 81 % are not identically zero in fp32).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_tel_linemax(int N, const T* __restrict__ telT,
                                                     float* __restrict__ tlmax) {
    __shared__ float part[4];
    const int v = blockIdx.x;
    float m = 0.f;
    // (double: rounded UP to float -- the bound stays an upper bound)
    for (int u = threadIdx.x; u < N; u += 256) m = fmaxf(m, __double2float_ru((double)telT[(size_t)v * N + u]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        tlmax[v] = __builtin_amdgcn_logf(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
}

// dline / dblk: the minima of K_DMIN per (task, direction).  dminb (optional): [ntask][nmt][nks]
// block minima over the directions (block pruning of K_OTF_MFMA, otf_mfma.hip).
__global__ void __launch_bounds__(256) k_vkeep(int H1, int nks, int ndir, int nl,
                                               const LamPar* __restrict__ lp,
                                               const float* __restrict__ dline,
                                               const float* __restrict__ dblk,
                                               const float* __restrict__ tlmax, float thr_sum,
                                               int* __restrict__ vkeep, int fixed,
                                               float* __restrict__ dminb) {
    constexpr int MAXH = 1280 / 2 + 1;
    __shared__ float sa[MAXH], sb[MAXH];
    const int task = blockIdx.x, npair = (nl + 1) / 2;
    for (int v = threadIdx.x; v < H1; v += 256) {
        float a = __builtin_inff();
        for (int d = 0; d < ndir; ++d) a = fminf(a, dline[((size_t)task * ndir + d) * H1 + v]);
        sa[v] = a;
        sb[v] = tlmax[v];
    }
    if (dminb != nullptr) {
        const int nb = ((H1 + 15) / 16) * nks;
        for (int e = threadIdx.x; e < nb; e += 256) {
            float a = __builtin_inff();
            for (int d = 0; d < ndir; ++d) a = fminf(a, dblk[((size_t)task * ndir + d) * nb + e]);
            dminb[(size_t)task * nb + e] = a;
        }
    }
    __syncthreads();
    // Per wavelength pair: walk the lines from the top and stop where the summed bounds of
    // everything above would exceed the allowance.  Eight threads per pair first sum eight
    // segments of the half plane, so that the serial walk is one pass over the segment sums and
    // one inside a segment (a single thread over all lines put 257 dependent steps on the chain
    // of every call).
    constexpr int SEG = 8, MAXP = 2048;                  // nl <= 4096
    __shared__ float seg[32][SEG];                       // 32 pairs per sweep
    __shared__ int svk[MAXP];
    const int slen = (H1 + SEG - 1) / SEG;
    for (int p0 = 0; p0 < npair; p0 += 32) {
        const int pr = p0 + (int)(threadIdx.x >> 3), sg = threadIdx.x & 7;
        const bool on = pr < npair;
        // the pair is bounded by its LONGER wavelength, whichever of the two that is (c < 0 rises
        // with the wavelength; the caller's wavelengths come in any order)
        const int la = on ? 2 * pr : 0, lb = on && 2 * pr + 1 < nl ? 2 * pr + 1 : la;
        const float c2 = (float)fmax(lp[la].c, lp[lb].c) * 1.44269504088896340736f;
        float part = 0.f;
        for (int v = min(H1, (sg + 1) * slen) - 1; v >= sg * slen; --v)
            part += __builtin_amdgcn_exp2f(fmaf(c2, sa[v], sb[v]));
        seg[threadIdx.x >> 3][sg] = part;
        __syncthreads();
        if (on && sg == 0) {
            float sum = 0.f;
            int vk = 0;
            for (int k = SEG - 1; k >= 0 && vk == 0; --k) {
                const float next = sum + seg[threadIdx.x >> 3][k];
                if (next > thr_sum) {                   // the boundary is inside segment k
                    for (int v = min(H1, (k + 1) * slen) - 1; v >= k * slen; --v) {
                        sum += __builtin_amdgcn_exp2f(fmaf(c2, sa[v], sb[v]));
                        if (sum > thr_sum) { vk = v + 1; break; }
                    }
                    if (vk == 0) vk = k * slen + 1;      // rounding of the two summation orders
                }
                sum = next;
            }
            svk[pr] = fixed > 0 ? min(fixed, H1) : vk;   // fixed: experiments
        }
        __syncthreads();
    }
    // The consumers (per_lambda.hip, otf_mfma.hip) stage and loop to the LAST pair of a wavelength
    // group and rely on vkeep growing with the pair index.  With ascending wavelengths it does (c2
    // rises with the wavelength, so every term of the sum does); for any other order the running
    // maximum makes it so -- keeping more lines than needed is always safe.
    if (threadIdx.x == 0)
        for (int pr = 1; pr < npair; ++pr) svk[pr] = max(svk[pr], svk[pr - 1]);
    __syncthreads();
    for (int pr = threadIdx.x; pr < npair; pr += 256) vkeep[task * npair + pr] = svk[pr];
}

// K_TASK_ORDER: the tasks of a chunk by descending lines kept at the longest wavelength (ties by
// index).  The per-wavelength kernel dispatches its workgroups in this order: the tasks with the
// sharpest PSFs keep four times the tiles of the broadest, and a launch that meets them last ends
// on them (longest-processing-time-first list scheduling).
__global__ void __launch_bounds__(256) k_task_order(int ntask, int npair, const int* __restrict__ vkeep,
                                                    int* __restrict__ order) {
    constexpr int MAXT = 4096;                       // chunk_tasks is at most 4096
    __shared__ int key[MAXT];
    for (int t = threadIdx.x; t < ntask; t += 256) key[t] = vkeep[(size_t)t * npair + npair - 1];
    __syncthreads();
    for (int t = threadIdx.x; t < ntask; t += 256) {
        const int k = key[t];
        int rank = 0;
        for (int o = 0; o < ntask; ++o) rank += (key[o] > k) || (key[o] == k && o < t);
        order[rank] = t;
    }
}

}  // namespace

void launch_ao_tables(hipStream_t s, const AoGeom& g, const uint8_t* d_mask_rec,
                      const uint8_t* d_mask_res, double* d_tab) {
    dim3 grid((NAO * NAO + 255) / 256, g.ndir, 2);
    hipLaunchKernelGGL(k_ao_tables, grid, dim3(256), 0, s, g, d_mask_rec, d_mask_res, d_tab);
}

void launch_tel_otf(hipStream_t s, int N, const uint64_t* d_rows, int words, double pupsum,
                    void* d_tel, bool f64out) {
    dim3 grid((N + 255) / 256, N / 2 + 1);
    if (f64out)
        hipLaunchKernelGGL(k_tel_otf<double>, grid, dim3(256), 0, s, N, d_rows, words, pupsum,
                           (double*)d_tel);
    else
        hipLaunchKernelGGL(k_tel_otf<float>, grid, dim3(256), 0, s, N, d_rows, words, pupsum,
                           (float*)d_tel);
}

int psd_rowfft_groups(int N) {
    int n = 0;
    DISPATCH_N(N, { n = (psd_rows<NN>() / 2 + a_slots<NN, false>() - 1) / a_slots<NN, false>(); })
    return n;
}

void launch_psd_rowfft(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp,
                       const double* d_aotab, double cfit, void* d_C, const void* d_tw64,
                       double* d_dcpart, bool f64) {
    DISPATCH_N(N, {
        constexpr size_t sm = a_smem<NN, false>();
        constexpr int NPAIR = psd_rows<NN>() / 2, SL = a_slots<NN, false>();
        dim3 grid((NPAIR + SL - 1) / SL, ntd);
        if (f64) {
            allow_smem((k_psd_rowfft<NN, true>), sm);
            hipLaunchKernelGGL((k_psd_rowfft<NN, true>), grid, dim3(a_threads<NN, false>()), sm, s, ndir, d_tp,
                               d_aotab, cfit, (cx<double>*)d_C, (const cx<double>*)d_tw64, d_dcpart);
        } else {
            allow_smem((k_psd_rowfft<NN, false>), sm);
            hipLaunchKernelGGL((k_psd_rowfft<NN, false>), grid, dim3(a_threads<NN, false>()), sm, s, ndir, d_tp,
                               d_aotab, cfit, (cx<double>*)d_C, (const cx<double>*)d_tw64, d_dcpart);
        }
    })
}

void launch_tel_linemax(hipStream_t s, int N, const void* d_tel, float* d_tlmax, bool f64) {
    if (f64)
        hipLaunchKernelGGL(k_tel_linemax<double>, dim3(N / 2 + 1), dim3(256), 0, s, N, (const double*)d_tel, d_tlmax);
    else
        hipLaunchKernelGGL(k_tel_linemax<float>, dim3(N / 2 + 1), dim3(256), 0, s, N, (const float*)d_tel, d_tlmax);
}

void launch_dmin(hipStream_t s, int N, int ntd, const void* d_D0t, float* d_dline, float* d_dblk, bool f64) {
    const dim3 grid((N / 2 + 1 + 15) / 16, ntd);
    if (f64)
        hipLaunchKernelGGL(k_dmin<double>, grid, dim3(256), 0, s, N, (const double*)d_D0t, d_dline, d_dblk);
    else
        hipLaunchKernelGGL(k_dmin<float>, grid, dim3(256), 0, s, N, (const float*)d_D0t, d_dline, d_dblk);
}

void launch_task_order(hipStream_t s, int ntask, int nl, const int* d_vkeep, int* d_order) {
    hipLaunchKernelGGL(k_task_order, dim3(1), dim3(256), 0, s, ntask, (nl + 1) / 2, d_vkeep, d_order);
}

void launch_vkeep(hipStream_t s, int N, int ntask, int ndir, int nl, const LamPar* d_lp,
                  const float* d_dline, const float* d_dblk, const float* d_tlmax, float thr_sum,
                  int* d_vkeep, int fixed, float* d_dminb) {
    hipLaunchKernelGGL(k_vkeep, dim3(ntask), dim3(256), 0, s, N / 2 + 1, N / 32, ndir, nl, d_lp, d_dline,
                       d_dblk, d_tlmax, thr_sum, d_vkeep, fixed, d_dminb);
}

void launch_colfft_dphi(hipStream_t s, int N, int ntd, const void* d_C, const double* d_dcpart,
                        double scale2, void* d_D0t, bool f64out, const void* d_tw64, int* d_zero) {
    const int nparts = psd_rowfft_groups(N);
    DISPATCH_N(N, {
        constexpr int SL = a_slots<NN, true>();
        constexpr size_t sm = a_smem<NN, true>();
#ifndef MPSFR_COLFFT_ITERS
#define MPSFR_COLFFT_ITERS 5
#endif
        constexpr int NYG = (NN / 2 + 1 + SL - 1) / SL;
        dim3 grid((NYG + MPSFR_COLFFT_ITERS - 1) / MPSFR_COLFFT_ITERS, ntd);     // column groups per workgroup
        if (f64out) {
            allow_smem(k_colfft_dphi<NN, double>, sm);
            hipLaunchKernelGGL((k_colfft_dphi<NN, double>), grid, dim3(a_threads<NN, true>()), sm, s,
                               (const cx<double>*)d_C, d_dcpart, nparts, scale2, (double*)d_D0t,
                               (const cx<double>*)d_tw64, d_zero);
        } else {
            allow_smem(k_colfft_dphi<NN, float>, sm);
            hipLaunchKernelGGL((k_colfft_dphi<NN, float>), grid, dim3(a_threads<NN, true>()), sm, s,
                               (const cx<double>*)d_C, d_dcpart, nparts, scale2, (float*)d_D0t,
                               (const cx<double>*)d_tw64, d_zero);
        }
    })
}


void launch_psd_image(hipStream_t s, int N, int ndir, const TaskPar& p, const double* d_aotab, double cfit,
                      double unit, double* d_psd) {
    const dim3 grid((N + 255) / 256, N, ndir);
    hipLaunchKernelGGL(k_psd_image<2>, grid, dim3(256), 0, s, N, ndir, p, d_aotab, cfit, unit, d_psd);
}

void launch_dphi_from_psd(hipStream_t s, int N, int ndir, const double* d_psd, void* d_Cm, double scale,
                          void* d_D0t, bool f64out, const void* d_tw64) {
    DISPATCH_N(N, {
        constexpr size_t sm = psdmem_smem<NN>();
        constexpr int SL = LineCfg<NN>::SLOTS;
        allow_smem((k_psdmem_rows<NN>), sm);
        hipLaunchKernelGGL((k_psdmem_rows<NN>), dim3(NN / SL, ndir), dim3(LineCfg<NN>::THREADS), sm, s, d_psd,
                           (cx<double>*)d_Cm, (const cx<double>*)d_tw64);
        const dim3 grid((NN / 2 + 1 + SL - 1) / SL, ndir);
        if (f64out) {
            allow_smem((k_psdmem_cols<NN, double>), sm);
            hipLaunchKernelGGL((k_psdmem_cols<NN, double>), grid, dim3(LineCfg<NN>::THREADS), sm, s,
                               (const cx<double>*)d_Cm, scale, (double*)d_D0t, (const cx<double>*)d_tw64);
        } else {
            allow_smem((k_psdmem_cols<NN, float>), sm);
            hipLaunchKernelGGL((k_psdmem_cols<NN, float>), grid, dim3(LineCfg<NN>::THREADS), sm, s,
                               (const cx<double>*)d_Cm, scale, (float*)d_D0t, (const cx<double>*)d_tw64);
        }
    })
}

}  // namespace mpsfr
