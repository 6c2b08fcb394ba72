// Per-wavelength stage of the mixed-precision path on the matrix cores (gfx950, wave64).
// Reference citations are to /root/reference/muse_psfr/psfrec.py.
//
// What the stage computes: for every (task, wavelength) the 40 x 40 stamp
//     stamp[i][j] = sum_v sum_u  G[v][j] * OTF[v][u] * E[u][i]              (psfrec.py:672-685, 793-801)
// with OTF[v][u] = tel[v][u] sum_dir 2^(c D[v][u]) on the transposed half plane, E the bilinear-
// weighted DFT kernel of the 21 distinct sample positions along a line and G the same along the
// columns.  psf_muse only ever reads the four bilinear neighbours of 40 x 40 sample points of the
// N x N transform, so the "2-D FFT" is two small dense contractions with the OTF in the middle:
//     first pass  Tq[v][c] = sum_u OTF[v][u] E[u][c]       (lines x 42 real columns, K = N)
//     second pass P/Q[i][j] = sum_v Tq[v][i] G[v][j]        (21 x 21, K = lines kept)
// Both run on v_mfma_f32_16x16x32_f16 / 16x16x16_f16 with every operand split into two halves
// (x = hi + lo, three products hi*hi + lo*hi + hi*lo, fp32 accumulation): 22 significant bits per
// operand, i.e. the rounding of an fp32 product, at 16/3 of the fp32 matrix (= vector) rate.  The
// LDS-resident line FFTs this replaces ran at 0.3 of the fp32 vector peak, bound by the LDS
// store path; a dense contraction also prunes in BOTH directions (blocks of 16 lines x 32 columns
// whose elements are all below the bound are never generated), which an FFT cannot.
//
// One wavefront owns one stamp from the OTF to the normalised 40 x 40 pixels: no inter-wave
// reduction, results bit-identical for any chunking or lane count.  The OTF tile is generated in
// registers in the A-operand layout (one fma + v_exp_f32 per element and direction), the E / G
// tables are pre-split fp16 in the B-operand layout (cached per wavelength set), and the
// accumulator tile of the first pass is, as it stands, the A operand of the second (its rows are
// the contraction index).  A workgroup is one task and up to eight wavelengths, which share the
// D | log2 tel tiles through LDS (K_OTF_MFMA1 below).
#include "mf_common.h"

namespace mpsfr {

namespace {

// Column c of the first pass -> (sample i, real / imaginary part); -1 = padding.
//   tile 0: Re i = 0..15      tile 1: Im i = 0..15      tile 2: c-32 = 0..4 Re 16..20, 8..12 Im 16..20
__host__ __device__ inline int col_sample(int c, bool* imag) {
    if (c < 16) { *imag = false; return c; }
    if (c < 32) { *imag = true; return c - 16; }
    const int r = c - 32;
    if (r < 5) { *imag = false; return 16 + r; }
    if (r >= 8 && r < 13) { *imag = true; return 16 + r - 8; }
    return -1;
}

// ------------------------------------------------------------------------------------------
// K_MF_TABLES: per-wavelength operand tables, cached per wavelength set.
//   E[l][ks][ct][hl][lane] (8 fp16): B operand of the first pass.  Lane supplies column
//     c = 16 ct + (lane & 15) for u = 32 ks + 8 (lane >> 4) + 0..7;
//     E_i(u) = (1-a_i) W^(u p_i) + a_i W^(u (p_i+1)),  W = exp(-2 pi i / N)   (forward transform
//     of the line sampled at the bilinear neighbours p_i, p_i + 1 of psfrec.py:672-683).
//   G[l][mt][jt][xy][hl][lane] (4 fp16): B operand of the second pass.  Lane supplies column
//     j = 16 jt + (lane & 15) for v = 16 mt + 4 (lane >> 4) + 0..3;
//     G[v][j] = w_v conj((1-a_j) W^(v p_j) + a_j W^(v (p_j+1))),  w_v = 1 for v in {0, N/2}, 2
//     otherwise (the other half plane), 0 for padding lines.
// Sample i of the 40-pixel stamp sits at i npixc / 40 of the centred crop: left neighbour
// p_i = (floor(i npixc / 40) - npixc / 2) mod N with weight 1 - a_i, a_i = frac.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mf_tables(int N, int nl, const LamPar* __restrict__ lp, const cx<double>* __restrict__ twg,
            h8* __restrict__ E, h4* __restrict__ G) {
    const int l = blockIdx.y;
    const int npixc = lp[l].npixc;
    const int nks = mf_nks(N), nmt = mf_nmt(N);
    const int nE = nks * NCT * 64, nG = nmt * NJT * 2 * 64;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    auto sample = [&](int i, int* p, double* a) {
        const int q = i * npixc;
        *p = ((q / NS - npixc / 2) % N + N) % N;
        *a = (double)(q % NS) / NS;
    };
    if (idx < nE) {
        const int lane = idx & 63, ct = (idx >> 6) % NCT, ks = (idx >> 6) / NCT;
        bool imag;
        const int i = col_sample(16 * ct + (lane & 15), &imag);
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float val = 0.f;
            if (i >= 0) {
                const int u = KBL * ks + 8 * (lane >> 4) + e;
                int p;
                double a;
                sample(i, &p, &a);
                const cx<double> w0 = twg[(int)(((long)u * p) % N)];
                const cx<double> w1 = twg[(int)(((long)u * (p + 1)) % N)];
                val = (float)((imag ? (1.0 - a) * w0.y + a * w1.y : (1.0 - a) * w0.x + a * w1.x) *
                              (double)(1 << kTabShift));
            }
            _Float16 h, q;
            split16(val, &h, &q);
            hi[e] = h;
            lo[e] = q;
        }
        h8* dst = E + ((size_t)(l * nks + ks) * NCT + ct) * 2 * 64;
        dst[lane] = hi;
        dst[64 + lane] = lo;
    }
    if (idx < nG) {
        const int lane = idx & 63, xy = (idx >> 6) & 1, jt = (idx >> 7) % NJT, mt = (idx >> 7) / NJT;
        const int j = 16 * jt + (lane & 15);
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = MTL * mt + 4 * (lane >> 4) + e;
            float val = 0.f;
            if (j < NSH && v <= N / 2) {
                int p;
                double a;
                sample(j, &p, &a);
                const cx<double> w0 = twg[(int)(((long)v * p) % N)];
                const cx<double> w1 = twg[(int)(((long)v * (p + 1)) % N)];
                const double wv = (v == 0 || v == N / 2) ? 1.0 : 2.0;
                // conj(W^(v p)): the imaginary part changes sign
                val = (float)((xy ? -wv * ((1.0 - a) * w0.y + a * w1.y) : wv * ((1.0 - a) * w0.x + a * w1.x)) *
                              (double)(1 << kTabShift));
            }
            _Float16 h, q;
            split16(val, &h, &q);
            hi[e] = h;
            lo[e] = q;
        }
        h4* dst = G + (((size_t)(l * nmt + mt) * NJT + jt) * 2 + xy) * 2 * 64;
        dst[lane] = hi;
        dst[64 + lane] = lo;
    }
}

// ------------------------------------------------------------------------------------------
// K_MF_TEL: constant tables of the telescope OTF (once per context).
//   tl2[v][u] = log2 tel[v][u] + kShift  (-inf where tel = 0 and on the padding lines v > N/2):
//               the telescope OTF goes into the exponent, tel 2^(c D) = 2^(c D + log2 tel)
//   tlb[mt][ks] = log2 of the block maximum of tel (block pruning, see K_OTF_MFMA)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_mf_tel(int N, const float* __restrict__ telT, float* __restrict__ tl2, float* __restrict__ tlb) {
    const int mt = blockIdx.x, nks = mf_nks(N);
    __shared__ float bmax[64];
    if (threadIdx.x < 64) bmax[threadIdx.x] = 0.f;
    __syncthreads();
    for (int e = threadIdx.x; e < MTL * N; e += 256) {
        const int v = MTL * mt + e / N, u = e % N;
        const float t = v <= N / 2 ? telT[(size_t)v * N + u] : 0.f;
        tl2[(size_t)v * N + u] = __builtin_amdgcn_logf(t) + kShift;        // log2(0) = -inf
        atomicMax(reinterpret_cast<int*>(&bmax[u / KBL]), __float_as_int(t));   // t >= 0: int order
    }
    __syncthreads();
    if ((int)threadIdx.x < nks) tlb[mt * nks + threadIdx.x] = __builtin_amdgcn_logf(bmax[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// Pruning of K_OTF_MFMA1.  vkeep[task][pair] (K_VKEEP, stage_a.hip) bounds the lines; inside them
// a block of 16 lines x 32 columns is generated only if  2^(c' dminb + tlb) > 2^thr,  dminb the
// block minimum of D over lines, columns and directions (K_DMIN, K_VKEEP) and tlb the block maximum
// of log2 tel: every element of a dropped block is below 2^thr, and the host sets thr so that all
// blocks together weigh less than eps / 2 of the PSF peak (>= OTF[0][0] = 1).
// ------------------------------------------------------------------------------------------
struct MfArgs {
    int N, ntask, ndir, nl;
    const float* D0t;        // [ntask ndir][N/2+1][N]
    const float* tl2;        // [nmt 16][N]
    const LamPar* lp;
    const h8* E;
    const h4* G;
    const int* vkeep;        // [ntask][(nl+1)/2] or nullptr
    const float* dminb;      // [ntask][nmt][nks] or nullptr
    const float* tlb;        // [nmt][nks]
    float thr;               // log2 of the block threshold (the eps rule)
    const float* thrf;       // [ntask] floor of the task (K_PEAK_FLOOR: the precision tier under its budget), or nullptr
    float tq_scale;          // 2^-ceil(log2 N): first-pass sums back into the fp16 range
    float* pre;              // [ntask][nl][40][40]
    const int* order;        // [ntask] dispatch order of the tasks, or nullptr
    unsigned long long* clk; // experiments: per-wave phase time stamps (or nullptr)
};


// ------------------------------------------------------------------------------------------
// K_OTF_MFMA1: the per-wavelength kernel (MULTI: several directions, see the template comment).
//
// What bounds the stage is not the matrix pipe but the vector-memory path of a CU (about 30 B/clk
// from the L2 in gather-shaped loads): a tile step consumes 4 KB of D and log2 tel and shares a
// 6 KB slab of E with the other tile steps of its k-step.  The kernel is therefore blocked like a
// matrix product over (m-tiles of one task) x (wavelengths):
//   * a workgroup is ONE task and a group of up to eight wavelengths, one wavelength per wave;
//   * the D | log2 tel tiles of a k-step (eight m-tiles, 32 KB) are staged in LDS ONCE per
//     workgroup by LDS-DMA (global_load_lds_dwordx4, wave w fetches tile w) and read by every
//     wave, each applying its own wavelength: 4 KB from the L2 feed up to eight tile steps;
//   * a wave keeps the accumulators of eight m-tiles (96 registers), so its E slab of the k-step
//     (private: it depends on the wavelength) feeds up to eight tile steps too;
//   * per k-step that is 32 + 48 KB of loads for 64 tile steps instead of 64 x 7 KB.
// The staging buffer is double buffered: the loads of the next k-step are in flight behind the
// tile steps of this one, and one s_barrier per k-step hands the buffers over (raw s_barrier with
// explicit waits: hipcc does not count asm memory operations).  The wavelengths of a group prune
// differently -- the bound grows with the wavelength -- so the workgroup stages what its longest
// wavelength needs and every wave skips the tile steps its own mask drops.
// A wave owns its stamp from the OTF to the normalised 40 x 40 pixels: no inter-wave reduction,
// results bit-identical for any chunking or lane count.
// ------------------------------------------------------------------------------------------
#ifndef MPSFR_MF_CLOCK
#define MPSFR_MF_CLOCK 0
#endif
#ifndef MPSFR_MF_KNOCK
#define MPSFR_MF_KNOCK 0         // kernel experiments: 1 = no loads, 2 = loads only, 3 = no products, 4 = no OTF arithmetic
#endif
constexpr int kMfTiles = 8;                         // m-tiles per sweep over the k-steps
constexpr int kMfStage = kMfTiles * 4096;           // one staging buffer: 8 x (D 2 KB | log2 tel 2 KB)
constexpr int kMfLds = 2 * kMfStage + 8 * 6 * 1024; // two staging buffers + one E slab per wave
constexpr int kMfStageMulti = 52 * 1024;            // several directions: a tile of 25 directions + log2 tel
constexpr int kMfLdsMulti = 2 * kMfStageMulti + 8 * 6 * 1024;

// x = 2^(c d + t) for two elements, split
__device__ __forceinline__ void otf_pair(f2 cc, f2 d, f2 t, unsigned* hi, unsigned* lo) {
    f2 y;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(cc), "v"(d), "v"(t));
    split_pair(__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1]), hi, lo);
}

// MULTI: several directions.  A staged tile is then ndir x D | log2 tel ((ndir + 1) x 2 KB), a sweep
// takes as many m-tiles as fit the 52 KB staging buffer (tpg: five at four directions, two at nine), and
// the OTF tile costs ndir exponentials per element -- the vector pipe, not the loads, sets the pace.
template <bool MULTI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_otf_mfma1(const MfArgs a, int per, int ngr, int tpg) {
    constexpr int GRP = kMfTiles;
    const int ndir = MULTI ? a.ndir : 1;
    const int tile_bytes = (ndir + 1) * 2048;
    constexpr int STAGE = MULTI ? kMfStageMulti : kMfStage;
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, H1 = N / 2 + 1, nks = mf_nks(N), nmt_all = mf_nmt(N);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    // workgroup -> (task, wavelength group).  Consecutive blockIdx go round the 8 XCDs (each with
    // its own L2): XCD x takes the tasks of rank x, x + 8, ... in dispatch order (heaviest first),
    // all wavelength groups of a task one after the other (they read its D from that L2), longest
    // wavelengths first.
    // The tasks beyond the last full round of eight (the lightest) are dealt group by group, so
    // that no XCD gets a whole task more than another: 500 workgroups on 8 x 32 CUs must be 62 or
    // 63 per XCD (two rounds), not 65 and 60.
    const int xcd = blockIdx.x & 7, ql = blockIdx.x >> 3;
    const int full = (a.ntask >> 3) * ngr;               // workgroups per XCD from the full rounds
    int rank, gi;
    if (ql < full) {
        rank = (ql / ngr) * 8 + xcd;
        gi = ql % ngr;
    } else {
        const int j = (ql - full) * 8 + xcd;
        if (j >= (a.ntask & 7) * ngr) return;            // whole workgroup
        rank = (a.ntask & ~7) + j / ngr;
        gi = j % ngr;
    }
    const int grp = ngr - 1 - gi;
    const int task = a.order != nullptr ? a.order[rank] : rank;
    // experiment clock (scripts/mf_clock.py): compiled in with -DMPSFR_MF_CLOCK=1 only -- even untaken,
    // its lane-0 branches inside the k-loop cost 4 % of the launch
    unsigned long long* const clk = MPSFR_MF_CLOCK && a.clk != nullptr && lane == 0
                                        ? a.clk + ((size_t)blockIdx.x * 8 + wave) * 8 : nullptr;
#define MF_STAMP(i_) if (clk != nullptr) clk[i_] = __builtin_readcyclecounter()
    MF_STAMP(0);
    const int lmax = min(a.nl - 1, grp * per + per - 1);  // the group's longest wavelength
    const bool lv = wave < per && grp * per + wave < a.nl;
    const int l = lv ? grp * per + wave : lmax;
    const float kLog2e = 1.44269504088896340736f;
    const float c2 = (float)a.lp[l].c * kLog2e;
    // the staging mask of the group is that of its LONGEST wavelength (the bound grows with the
    // wavelength), whichever member that is: the caller's wavelengths come in any order
    double cu = a.lp[lmax].c;
    for (int j = grp * per; j < lmax; ++j) cu = fmax(cu, a.lp[j].c);
    const float c2u = (float)cu * kLog2e;
    const f2 cc = {c2, c2};
    const int npair = (a.nl + 1) / 2;
    const int nv = a.vkeep != nullptr ? a.vkeep[(size_t)task * npair + (l >> 1)] : H1;
    const int nvu = a.vkeep != nullptr ? a.vkeep[(size_t)task * npair + (lmax >> 1)] : H1;
    const int nmt = lv ? (nv + MTL - 1) / MTL : 0, nmtu = (nvu + MTL - 1) / MTL;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* slab = smem + 2 * STAGE + wave * 6 * 1024;
    const unsigned slab_lds = lds0 + 2 * STAGE + (unsigned)wave * 6 * 1024;

    f4 P0[NJT], Q0[NJT], R2x[NJT], R2y[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        P0[jt] = f4{0.f, 0.f, 0.f, 0.f};
        Q0[jt] = f4{0.f, 0.f, 0.f, 0.f};
        R2x[jt] = f4{0.f, 0.f, 0.f, 0.f};
        R2y[jt] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // lane part of a tile's addresses (bytes); the rest is wave-uniform.  Lines beyond N/2 of
    // the last m-tile read the zeroed padding behind D (log2 tel = -inf there).
    const unsigned voff = (unsigned)(((size_t)lr * N + 8 * lk) * sizeof(float)), voff16 = voff + 16;
    const unsigned voffb = (unsigned)lane * 16;
    const float thr_t = a.thrf != nullptr ? fmaxf(a.thr, a.thrf[task]) : a.thr;
    const char* dtask = reinterpret_cast<const char*>(a.D0t + (size_t)task * ndir * H1 * N);
    const size_t dstride = (size_t)H1 * N * sizeof(float);       // one direction of D
    float dirshift = 0.f;                      // several directions: their sum <= 2^dirshift
    while ((1 << (int)dirshift) < ndir) dirshift += 1.f;
    const f2 dsh = {dirshift, dirshift};
    const char* ttab = reinterpret_cast<const char*>(a.tl2);
    const char* etab = reinterpret_cast<const char*>(a.E + (size_t)l * nks * NCT * 2 * 64);
    const h4* Gl = a.G + (size_t)l * nmt_all * NJT * 2 * 2 * 64 + lane;

    unsigned long long t_kloop = 0, t_pass2 = 0, t_wload = 0, t_wbar = 0, n_iter = 0, t_issue = 0, t_comp = 0;
    unsigned mytiles = 0;                  // the tiles of a sweep this wave fetches
    for (int g = wave; g < GRP; g += per) mytiles |= 1u << g;
    for (int g0 = 0; g0 < nmtu; g0 += tpg) {
        // which k-steps each m-tile needs: for this wave's wavelength and for the group's longest
        unsigned long long own[GRP], uni[GRP], kown = 0, kuni = 0;
        float dm[GRP], tb[GRP];
        if (a.dminb != nullptr) {          // all loads in flight before the first ballot waits
#pragma unroll
            for (int g = 0; g < GRP; ++g) {
                const int mt = min(g0 + g, nmt_all - 1), kk = min(lane, nks - 1);
                dm[g] = a.dminb[((size_t)task * nmt_all + mt) * nks + kk];
                tb[g] = a.tlb[mt * nks + kk];
            }
        }
#pragma unroll
        for (int g = 0; g < GRP; ++g) {
            const int mt = g0 + g;
            bool ou = g < tpg && mt < nmtu && lane < nks, oo = g < tpg && mt < nmt && lane < nks;
            if (a.dminb != nullptr) {
                ou = ou && fmaf(c2u, dm[g], tb[g]) > thr_t;
                oo = oo && fmaf(c2, dm[g], tb[g]) > thr_t;
            }
            uni[g] = __ballot(ou);
            own[g] = __ballot(oo) & uni[g];
            kuni |= uni[g];
            kown |= own[g];
        }
        f4 acc[GRP][NCT];
#pragma unroll
        for (int g = 0; g < GRP; ++g)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[g][ct] = f4{0.f, 0.f, 0.f, 0.f};

        // stage the tiles of k-step ks (wave w fetches tiles w, w + per, ...) and this wave's E slab
        auto stage = [&](int ks, int buf) {
            if (MPSFR_MF_KNOCK == 1) return;
            if constexpr (!MULTI) {
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    if (!((mytiles >> g) & 1) || !((uni[g] >> ks) & 1)) continue;
                    const size_t off = ((size_t)(MTL * (g0 + g)) * N + (size_t)KBL * ks) * sizeof(float);
                    glds_tile(dtask + off, ttab + off, voff, voff16, lds0 + buf * STAGE + g * 4096);
                }
            } else {
                // pieces (tile, direction or log2 tel) of 2 KB, dealt to the waves in turn
                int turn = 0;                                    // whose piece it is (no division)
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    if (!((uni[g] >> ks) & 1)) continue;          // (uni[g] = 0 for g >= tpg)
                    const size_t off = ((size_t)(MTL * (g0 + g)) * N + (size_t)KBL * ks) * sizeof(float);
                    const unsigned dst = lds0 + buf * STAGE + g * tile_bytes;
                    for (int d = 0; d <= ndir; ++d) {
                        const bool mine = turn == wave;
                        turn = turn + 1 == per ? 0 : turn + 1;
                        if (!mine) continue;
                        const char* src = d < ndir ? dtask + d * dstride + off : ttab + off;
                        glds16s(src, voff, dst + d * 2048);
                        glds16s(src, voff16, dst + d * 2048 + 1024);
                    }
                }
            }
            if ((kown >> ks) & 1) {
                const char* Ek = etab + (size_t)ks * NCT * 2 * 1024;
#pragma unroll
                for (int i = 0; i < 2 * NCT; ++i) glds16s(Ek + i * 1024, voffb, slab_lds + i * 1024);
            }
        };
        unsigned long long rest = kuni;
        int buf = 0;
        const unsigned long long tk0 = clk != nullptr ? __builtin_readcyclecounter() : 0;
        if (g0 == 0) MF_STAMP(1);
        if (rest != 0) stage(__builtin_ctzll(rest), 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g0 == 0) MF_STAMP(2);
        while (rest != 0) {
            const int ks = __builtin_ctzll(rest);
            rest &= rest - 1;
            const bool mine = (kown >> ks) & 1;
            const unsigned long long ti0 = clk != nullptr ? __builtin_readcyclecounter() : 0;
            h8 bh[NCT], bl[NCT];
            if (mine) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    bh[ct] = *reinterpret_cast<const h8*>(slab + (ct * 2) * 1024 + lane * 16);
                    bl[ct] = *reinterpret_cast<const h8*>(slab + (ct * 2 + 1) * 1024 + lane * 16);
                }
                // the slab is in registers before the next one may land on it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (rest != 0) stage(__builtin_ctzll(rest), buf ^ 1);
            const unsigned long long ti1 = clk != nullptr ? __builtin_readcyclecounter() : 0;
            if (mine) {
                // The tile steps of the k-step, software pipelined inside the wave: the operands of
                // the NEXT tile the wave needs are read from LDS and turned into the fp16 halves of
                // its OTF tile in the same basic block as the nine products of the current tile, so
                // the scheduler puts the vector work into the shadow of the MFMAs.  The m-tile of
                // the products is a compile-time index (the accumulators stay where they are); the
                // tile that is prepared is a run-time one.  The last tile prepares itself again.
                const unsigned char* tbuf = smem + buf * STAGE + lane * 16;
                unsigned gbits = 0;
#pragma unroll
                for (int g = 0; g < GRP; ++g) gbits |= (unsigned)((own[g] >> ks) & 1) << g;
                h8 ah, al;
                auto prepare = [&](int g) {
                    const unsigned char* tp = tbuf + g * (MULTI ? tile_bytes : 4096);
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    unsigned hi[4], lo[4];
                    if constexpr (!MULTI) {
                        const f4 d0 = *reinterpret_cast<const f4*>(tp);
                        const f4 d1 = *reinterpret_cast<const f4*>(tp + 1024);
                        const f4 t0 = *reinterpret_cast<const f4*>(tp + 2048);
                        const f4 t1 = *reinterpret_cast<const f4*>(tp + 3072);
                        if (MPSFR_MF_KNOCK == 4) {          // experiment: no OTF arithmetic
                            ah = __builtin_bit_cast(h8, d0 + t0);
                            al = __builtin_bit_cast(h8, d1 + t1);
                            return;
                        }
                        otf_pair(cc, f2{d0[0], d0[1]}, f2{t0[0], t0[1]}, &hi[0], &lo[0]);
                        otf_pair(cc, f2{d0[2], d0[3]}, f2{t0[2], t0[3]}, &hi[1], &lo[1]);
                        otf_pair(cc, f2{d1[0], d1[1]}, f2{t1[0], t1[1]}, &hi[2], &lo[2]);
                        otf_pair(cc, f2{d1[2], d1[3]}, f2{t1[2], t1[3]}, &hi[3], &lo[3]);
                    } else {
                        // x = sum_dir 2^(c D_dir + log2 tel - dirshift): one fma + v_exp_f32 per direction
                        const f4 t0 = *reinterpret_cast<const f4*>(tp + ndir * 2048);
                        const f4 t1 = *reinterpret_cast<const f4*>(tp + ndir * 2048 + 1024);
                        f2 tt[4] = {f2{t0[0], t0[1]} - dsh, f2{t0[2], t0[3]} - dsh, f2{t1[0], t1[1]} - dsh,
                                    f2{t1[2], t1[3]} - dsh};
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = 0.f;
                        // two directions per turn: their LDS reads and exp chains overlap
                        auto add_dir = [&](const f4& d0, const f4& d1) {
                            const f2 dd[4] = {f2{d0[0], d0[1]}, f2{d0[2], d0[3]}, f2{d1[0], d1[1]}, f2{d1[2], d1[3]}};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                f2 y;
                                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(cc), "v"(dd[k]), "v"(tt[k]));
                                x[2 * k] += __builtin_amdgcn_exp2f(y[0]);
                                x[2 * k + 1] += __builtin_amdgcn_exp2f(y[1]);
                            }
                        };
                        int d = 0;
                        for (; d + 1 < ndir; d += 2) {
                            const f4 a0 = *reinterpret_cast<const f4*>(tp + d * 2048);
                            const f4 a1 = *reinterpret_cast<const f4*>(tp + d * 2048 + 1024);
                            const f4 b0 = *reinterpret_cast<const f4*>(tp + d * 2048 + 2048);
                            const f4 b1 = *reinterpret_cast<const f4*>(tp + d * 2048 + 3072);
                            add_dir(a0, a1);
                            add_dir(b0, b1);
                        }
                        if (d < ndir)
                            add_dir(*reinterpret_cast<const f4*>(tp + d * 2048),
                                    *reinterpret_cast<const f4*>(tp + d * 2048 + 1024));
#pragma unroll
                        for (int k = 0; k < 4; ++k) split_pair(x[2 * k], x[2 * k + 1], &hi[k], &lo[k]);
                    }
                    ah = __builtin_bit_cast(h8, u4{hi[0], hi[1], hi[2], hi[3]});
                    al = __builtin_bit_cast(h8, u4{lo[0], lo[1], lo[2], lo[3]});
                };
                prepare(__builtin_ctz(gbits));
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    if (!((gbits >> g) & 1)) continue;
                    const h8 ch = ah, cl = al;
                    const unsigned later = gbits & ~((2u << g) - 1u);
                    prepare(later != 0 ? __builtin_ctz(later) : g);
                    if (MPSFR_MF_KNOCK == 2 || MPSFR_MF_KNOCK == 3) {   // experiments: loads only / no products
                        acc[g][0][0] += (float)ch[0] + (float)cl[0];
                        continue;
                    }
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, bh[ct], acc[g][ct], 0, 0, 0);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, bl[ct], acc[g][ct], 0, 0, 0);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, bh[ct], acc[g][ct], 0, 0, 0);
                }
            }
            // the next k-step's tiles have landed, and nobody reads this k-step's any more
            const unsigned long long tw0 = clk != nullptr ? __builtin_readcyclecounter() : 0;
            if (clk != nullptr) { t_issue += ti1 - ti0; t_comp += tw0 - ti1; }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long tw1 = clk != nullptr ? __builtin_readcyclecounter() : 0;
            __builtin_amdgcn_s_barrier();
            if (clk != nullptr) {
                t_wload += tw1 - tw0;
                t_wbar += __builtin_readcyclecounter() - tw1;
                n_iter += 1;
            }
            buf ^= 1;
        }
        const unsigned long long tk1 = clk != nullptr ? __builtin_readcyclecounter() : 0;
        if (g0 == 0) MF_STAMP(3);
        // second pass: the accumulator tile (rows = lines on registers / lane groups, column on
        // the lane) is the A operand of a 16x16x16 product that sums over its rows
        // (the G fragments of the next tile the wave needs are in flight behind the products of
        // the current one: tile index at run time, accumulator index at compile time, as above)
        unsigned tbits = 0;
#pragma unroll
        for (int g = 0; g < GRP; ++g) tbits |= (unsigned)(own[g] != 0) << g;
        h4 gq[NJT][2][2];
        auto fetch_g = [&](int g) {
            const h4* Gm = Gl + (size_t)(g0 + g) * NJT * 2 * 2 * 64;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) gq[jt][xy][hl] = Gm[((jt * 2 + xy) * 2 + hl) * 64];
        };
        // (several directions: the registers of the prefetch are what the kernel spills without)
        if (!MULTI && tbits != 0) fetch_g(__builtin_ctz(tbits));
#pragma unroll
        for (int g2 = 0; g2 < GRP; ++g2) {
            if (!((tbits >> g2) & 1)) continue;
            if constexpr (MULTI) fetch_g(g2);
            h4 gc[NJT][2][2];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) gc[jt][xy][hl] = gq[jt][xy][hl];
            if constexpr (!MULTI) {
                const unsigned later = tbits & ~((2u << g2) - 1u);
                fetch_g(later != 0 ? __builtin_ctz(later) : g2);
            }
            h4 th[NCT], tw[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 h, w;
                    split16(acc[g2][ct][r] * a.tq_scale, &h, &w);
                    th[ct][r] = h;
                    tw[ct][r] = w;
                }
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                P0[jt] = mm16(P0[jt], th[0], tw[0], gc[jt][0][0], gc[jt][0][1]);
                Q0[jt] = mm16(Q0[jt], th[1], tw[1], gc[jt][1][0], gc[jt][1][1]);
                R2x[jt] = mm16(R2x[jt], th[2], tw[2], gc[jt][0][0], gc[jt][0][1]);
                R2y[jt] = mm16(R2y[jt], th[2], tw[2], gc[jt][1][0], gc[jt][1][1]);
            }
        }
        if (clk != nullptr) {
            t_kloop += tk1 - tk0;
            t_pass2 += __builtin_readcyclecounter() - tk1;
        }
    }
    MF_STAMP(4);
    if (clk != nullptr) { clk[6] = t_kloop; clk[7] = t_pass2; clk[1] = t_wload; clk[2] = t_wbar; clk[3] = n_iter; clk[0] = t_issue; clk[4] = t_comp; }
    if (lv) write_stamp(P0, Q0, R2x, R2y, lr, lk, a.pre + ((size_t)task * a.nl + l) * NS * NS);
    MF_STAMP(5);
#undef MF_STAMP
}

}  // namespace

size_t mf_etab_bytes(int N, int nl) { return (size_t)nl * mf_nks(N) * NCT * 2 * 64 * sizeof(h8); }
size_t mf_gtab_bytes(int N, int nl) { return (size_t)nl * mf_nmt(N) * NJT * 2 * 2 * 64 * sizeof(h4); }
size_t mf_tl2_bytes(int N) { return (size_t)mf_nmt(N) * MTL * N * sizeof(float); }
size_t mf_tlb_bytes(int N) { return (size_t)mf_nmt(N) * mf_nks(N) * sizeof(float); }
size_t mf_dminb_bytes(int N, int ntask) { return (size_t)ntask * mf_nmt(N) * mf_nks(N) * sizeof(float); }
int mf_block_count(int N) { return mf_nmt(N) * mf_nks(N); }

void launch_mf_tables(hipStream_t s, int N, int nl, const LamPar* d_lp, const void* d_tw64,
                      void* d_E, void* d_G) {
    const int nE = mf_nks(N) * NCT * 64, nG = mf_nmt(N) * NJT * 2 * 64;
    const int n = nE > nG ? nE : nG;
    hipLaunchKernelGGL(k_mf_tables, dim3((n + 255) / 256, nl), dim3(256), 0, s, N, nl, d_lp,
                       (const cx<double>*)d_tw64, (h8*)d_E, (h4*)d_G);
}

// K_PEAK_FLOOR: the floor tier of the kernel for several directions under its budget.  Per task:
// S = the OTF of the SHORTEST wavelength (the smallest) summed exactly over the first `nlines` lines of the half
// plane and over the directions -- a lower bound of the PSF peak of every wavelength of the task -- and
//   thrf[task] = min(floor, log2(tier_half S / (2 x 512 x ndir x blocks))):
// every element of a dropped block is below 2^thrf in every direction, so all dropped blocks together weigh less than
// tier_half of S.  (One threshold per task, counted not summed: coarser than K_MF_PREP's per-wavelength masses.)
__global__ void __launch_bounds__(256) k_peak_floor(int N, int ndir, int nlines, const float* __restrict__ D0t,
                                                    const float* __restrict__ tl2, float c2min, float tier_half,
                                                    float floor_nominal, int nblocks, float* __restrict__ thrf) {
    __shared__ float part[4];
    const int task = blockIdx.x, H1 = N / 2 + 1;
    float s = 0.f;
    const int per = nlines * N, tot = per * ndir;          // (direction, element) pairs: independent loads
    const float* dt = D0t + (size_t)task * ndir * H1 * N;
#pragma unroll 4
    for (int i = threadIdx.x; i < tot; i += 256) {
        const int d = i / per, e = i - d * per;
        const float x = __builtin_amdgcn_exp2f(fmaf(c2min, fmaxf(dt[(size_t)d * H1 * N + e], 0.f), tl2[e]));
        s += e < N ? x : 2.f * x;             // line 0 once, the others stand for both half planes
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float S = ((part[0] + part[1]) + (part[2] + part[3])) * __builtin_amdgcn_exp2f(-kShift);   // (tl2 carries 2^kShift)
        const float lim = __builtin_amdgcn_logf(tier_half * S / (1024.f * (float)ndir * (float)nblocks));
        thrf[task] = fminf(floor_nominal, lim);
    }
}

void launch_peak_floor(hipStream_t s, int N, int ntask, int ndir, const void* d_D0t, const float* d_tl2, float c2min,
                       float tier_half, float floor_nominal, float* d_thrf) {
    hipLaunchKernelGGL(k_peak_floor, dim3(ntask), dim3(256), 0, s, N, ndir, N > 512 ? 2 : 4, (const float*)d_D0t, d_tl2,
                       c2min, tier_half, floor_nominal, mf_nmt(N) * mf_nks(N), d_thrf);
}

void launch_mf_tel(hipStream_t s, int N, const void* d_tel, float* d_tl2, float* d_tlb) {
    hipLaunchKernelGGL(k_mf_tel, dim3(mf_nmt(N)), dim3(256), 0, s, N, (const float*)d_tel, d_tl2, d_tlb);
}

void launch_otf_mfma(hipStream_t s, int N, int ntask, int ndir, int nl, const void* d_D0t,
                     const float* d_tl2, const LamPar* d_lp, const void* d_E, const void* d_G,
                     const int* d_vkeep, const float* d_dminb, const float* d_tlb, float thr,
                     void* d_pre, const int* d_order, void* d_clk, const float* d_thrf) {
    MfArgs a;
    a.thrf = d_thrf;
    a.N = N; a.ntask = ntask; a.ndir = ndir; a.nl = nl;
    a.D0t = (const float*)d_D0t; a.tl2 = d_tl2; a.lp = d_lp;
    a.E = (const h8*)d_E; a.G = (const h4*)d_G;
    a.vkeep = d_vkeep; a.dminb = d_dminb; a.tlb = d_tlb; a.thr = thr;
    int lg = 0;
    while ((1 << lg) < N) ++lg;
    a.tq_scale = 1.0f / ((float)(1 << lg) * (float)(1 << kTabShift));    // first-pass sums -> <= 2^kShift ndir
    a.pre = (float*)d_pre;
    a.clk = (unsigned long long*)d_clk;
    a.order = d_order;
    // wavelength groups of at most eight, as even as possible: one wave per wavelength
    const int ngr = (nl + 7) / 8, per = (nl + ngr - 1) / ngr;
    const int per_xcd = (ntask / 8) * ngr + ((ntask % 8) * ngr + 7) / 8;
    // LDS: two staging buffers + one E slab per wave -- no more than the waves there are, so that
    // a workgroup of another kernel (the other pipeline lane) fits beside it on the CU
    if (ndir == 1) {
        const size_t sm = 2 * (size_t)kMfStage + (size_t)per * 6 * 1024;
        allow_smem(k_otf_mfma1<false>, (size_t)kMfLds);
        hipLaunchKernelGGL(k_otf_mfma1<false>, dim3(8 * per_xcd), dim3(64 * per), sm, s, a, per, ngr, kMfTiles);
    } else {
        // a staged tile is ndir x D + log2 tel, 2 KB each: 52 KB at the 25 directions of npsflin = 5
        const int fit = kMfStageMulti / ((ndir + 1) * 2048), tpg = fit < kMfTiles ? fit : kMfTiles;
        const size_t sm = 2 * (size_t)kMfStageMulti + (size_t)per * 6 * 1024;
        allow_smem(k_otf_mfma1<true>, (size_t)kMfLdsMulti);
        hipLaunchKernelGGL(k_otf_mfma1<true>, dim3(8 * per_xcd), dim3(64 * per), sm, s, a, per, ngr, tpg);
    }
}

}  // namespace mpsfr
