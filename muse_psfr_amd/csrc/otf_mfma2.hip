// Per-wavelength stage of the mixed-precision path on the matrix cores, second generation
// (gfx950, wave64).  Reference citations are to /root/reference/muse_psfr/psfrec.py.
//
// Same contraction as otf_mfma.hip:
//     stamp[i][j] = sum_v sum_u  G[v][j] * OTF[v][u] * E[u][i]              (psfrec.py:672-685, 793-801)
// with OTF[v][u] = tel[v][u] 2^(c D[v][u]) generated in registers in the A-operand layout, the E / G
// tables of K_MF_TABLES as B operands, split-fp16 products with fp32 accumulation.  What is new:
//
//  * Precision tiers per block of 16 lines x 32 columns, decided from the same bound as the block
//    pruning (every element of the block is below 2^(c' dmin + log2 telmax)):
//      - below 2^-29 of OTF[0][0] both fp16 halves of every element are subnormal (the OTF is
//        generated times 2^15): the block is dropped;
//      - below 2^-18 the LOW half of every element is subnormal: the block runs two of the three
//        products and no low half ("mid" blocks: 6 MFMA and 16 vector instructions per tile step
//        instead of 9 and 28);
//      - above, the full three products.
//    Round 3 took the first two for bit-neutral ("the matrix cores flush fp16 subnormals").  They are
//    not: the gfx950 matrix cores multiply subnormal fp16 inputs like any other, so the tiers are
//    approximations of the same kind as the block pruning, and as small -- what a tier leaves out is
//    below 2^-29 (2^-18 x 2^-11) of the largest OTF element per element: measured against a run
//    without the tier, no stamp pixel moves by more than 2e-7 of its peak and beta by 1.2e-6
//    (tests/test_gpu_parity.py::test_precision_tiers_of_the_matrix_core_stage holds them to 3e-7 of
//    the peak, the bound of the block pruning).
//  * The masks (which blocks a wavelength keeps, in which tier; which blocks a wavelength group
//    stages) are computed once per (task, wavelength) by K_MF_PREP instead of by every wave, and
//    the group's staging mask is the exact union of its members (no assumption on the order of
//    the wavelengths).
//  * Thin waves: a workgroup still shares the D | log2 tel tiles of a k-step through LDS between
//    the up to seven wavelengths of a group, but every wavelength has TWO waves, each owning four
//    of the sweep's eight m-tiles (48 accumulator registers instead of 96).  12-14 waves per CU
//    instead of 7 hide each other's LDS latency, LDS-DMA issue and barrier waits; the E slab of a
//    wavelength is fetched once for its two waves (double buffered, so no wave waits for its
//    partner before the next slab is requested).  The two partial stamps meet in LDS.
//  * Work items and a queue.  The unit of work is (task, wavelength group, SWEEP of eight m-tiles)
//    instead of (task, group): the sharpest PSFs keep three sweeps and six times the blocks of the
//    broadest, and with one workgroup per (task, group) the launch ended on them at twice the
//    balanced time.  K_MF_PREP files every non-empty item in one of 16 lists by the number of
//    blocks it stages; one persistent workgroup per CU takes the items from an atomic counter,
//    heaviest class first (longest-processing-time-first, to the width of a class).  A (task,
//    group) with one sweep is finished by its workgroup; with several, every sweep leaves its partial
//    tiles in memory and K_MF_FINISH adds them in sweep order (fixed order: results do not depend on
//    who ran what).  (Whole (task, group)s as items, their sweeps run in sequence by one workgroup,
//    need no K_MF_FINISH and balance so much worse that the launch is 20 % longer: DESIGN.md.)
//  * The queue runs a little ahead: the next item's number is drawn during the last k-step, its
//    descriptor and masks are loaded behind the second pass and the epilogue.
#include <hip/hip_ext.h>
#include "mf_common.h"

namespace mpsfr {

namespace {

typedef unsigned long long u64;

constexpr int kTW = 4;                  // m-tiles per wave: tiles 2 i + half of the sweep
constexpr int kStage2 = kT2 * 4096;     // one staging buffer: 8 x (D 2 KB | log2 tel 2 KB)
constexpr int kSlab = 6 * 1024;         // E slab of one wavelength and k-step: 3 column tiles x (hi | lo)

// ------------------------------------------------------------------------------------------
// K_MF_PREP: one workgroup per (task, wavelength group): the block masks of the group's wavelengths
// and its work items, from the block minima of D (K_DMIN, stage_a.hip; a version that computed them
// here, one workgroup reading all of a task's D, took 43 us against 12 + 6: a single CU draws
// 30 GB/s; one workgroup per task walking its groups in rounds took 16 us, all of it latency).
// One wave per wavelength, lane = k-step:
//   own[task][l][mt][2]   bit ks of word 0: block (mt, ks) is a full block of wavelength l;
//                         word 1: a mid block (see the file comment)
//   uni[task][grp][mt]    bit ks: some wavelength of the group keeps the block (what the workgroup
//                         of K_OTF_MFMA2 stages)
//   ksum[task][l][sw], kuni[task][grp][sw]   the k-steps with work, per sweep of eight m-tiles
//   gsw[task][grp]        bit sw: the sweep has work
//   items[cls][]          the work lists of K_OTF_MFMA2: {task, grp, sweep, sweeps of the (task, grp)} of
//                         every sweep with work, filed by work class -- the blocks it stages, in 64 steps up
//                         to the eight m-tiles x nks of a full sweep -- with one atomic add on the class
//                         counter sched[cls].  (Sixteen classes were sixteen hot words: the launch took
//                         13 ns per item, 12 / 30 / 60 us for 900 / 1800 / 4500 items, whatever else it
//                         did; 64 take 11 / 25 / 41 us and order the queue more finely.)  The order inside
//                         a class is left to the hardware and changes no result (items are independent).
// A block is kept if its bound e = c' dmin + log2 telmax is above thr (every element of a dropped
// block is below 2^thr); it is full if e is above thr_mid.  Without pruning (dminb = nullptr) every
// block of the half plane is full.  (The line pruning of the FFT path, K_VKEEP, is not needed here:
// a line it drops consists of blocks this rule drops.)
// ------------------------------------------------------------------------------------------
struct MaskArgs {
    int N, nl, per, ngr;
    const LamPar* lp;
    const float* dminb;      // [ntask][nmt][nks] block minima of D (K_DMIN), or nullptr = no pruning
    const float* dlin;       // instead of dminb: [ntask][N/2+1][nks] minima per LINE and block of 32 columns
                             // (K_DPHI_SERIES); the minimum over a block's 16 lines is taken here
    const float* tlb;        // [nmt][nks]
    float thr, thr_mid;      // thr: the eps rule (blocks below are dropped whatever the tiers do)
    // Precision tiers under a budget: blocks with a bound in (thr, thr_floor] are dropped and
    // blocks in (thr_floor, thr_mid] run without the low fp16 half of the OTF -- as far as the OTF mass each of
    // the two leaves out stays below tier_half of a lower bound of the PSF peak, the OTF summed exactly over
    // the first peak_lines(N) lines of the half plane (D0t, tl2).  Where it would not, the wave lowers the
    // threshold for its (task, wavelength) until it does.  tier_half <= 0 or D0t == nullptr: no budget.
    float thr_floor, tier_half;
    const float* D0t;        // [ntask][N/2+1][N]
    const float* tl2;        // [nmt 16][N] log2 of the telescope OTF
    u64* own;
    u64* uni;
    u64* ksum;
    u64* kuni;
    int* gsw;
    int* sched;              // [0..63] items per list (zeroed by stage A), [64] queue head of K_OTF_MFMA2
    int4* items;             // [kMfLists][cap]
    int cap;                 // ntask ngr nsw
};

constexpr int kMaxNmt = 1280 / 2 / MTL + 1, kMaxNks = 1280 / KBL;      // 41, 40

// lines of the half plane summed for the lower bound of the peak (two on the large grids: as many elements)
__host__ __device__ constexpr int peak_lines(int N) { return N > 512 ? 2 : 4; }

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one workgroup per (wavelength group, task): wave = wavelength slot of the group, lane = k-step
__global__ void __launch_bounds__(64 * 8) k_mf_prep(const MaskArgs a) {
    __shared__ float s_dm[kMaxNmt * kMaxNks], s_tb[kMaxNmt * kMaxNks];
    __shared__ u64 s_any[8][kMaxNmt + 7];
    const int grp = blockIdx.x, task = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int N = a.N, nks = mf_nks(N), nmt = mf_nmt(N), nsw = (nmt + kT2 - 1) / kT2;
    const int nthr = 64 * a.per;
    for (int e = threadIdx.x; e < nmt * nks; e += nthr) {
        float dm = 0.f;
        if (a.dlin != nullptr) {
            const int mt = e / nks, ks = e - mt * nks, H1 = N / 2 + 1;
            const float* src = a.dlin + ((size_t)task * H1 + MTL * mt) * nks + ks;
            const int last = min(MTL, H1 - MTL * mt) - 1;
            float v[MTL];
#pragma unroll
            for (int i = 0; i < MTL; ++i) v[i] = src[(size_t)min(i, last) * nks];      // all sixteen in flight
            dm = v[0];
#pragma unroll
            for (int i = 1; i < MTL; ++i) dm = fminf(dm, v[i]);
        } else if (a.dminb != nullptr) {
            dm = a.dminb[(size_t)task * nmt * nks + e];
        }
        s_dm[e] = dm;
        s_tb[e] = a.tlb[e];
    }
    const int l = grp * a.per + w;
    const bool lv = l < a.nl;
    const float c2 = lv ? (float)a.lp[l].c * 1.44269504088896340736f : 0.f;
    __syncthreads();
    const int kk = min(lane, nks - 1);
    const bool nopr = a.dminb == nullptr && a.dlin == nullptr;
    float thr_keep = fmaxf(a.thr, a.thr_floor), thr_mid = a.thr_mid;
    if (!nopr && lv && a.tier_half > 0.f && a.D0t != nullptr && fmaxf(thr_mid, thr_keep) > a.thr) {
        // mass of the blocks with a bound in (lo, hi]: 2 half planes x 512 elements x the bound each
        auto mass = [&](float lo, float hi) {
            float m = 0.f;
            for (int mt = 0; mt < nmt; ++mt) {
                const float e = fmaf(c2, s_dm[mt * nks + kk], s_tb[mt * nks + kk]);
                m += (lane < nks && e > lo && e <= hi) ? __builtin_amdgcn_exp2f(e) : 0.f;
            }
            return 1024.f * wave_sum_f(m);
        };
        // both tiers at their nominal thresholds in one pass (the usual case needs no other)
        float mf = 0.f, mm = 0.f;
        for (int mt = 0; mt < nmt; ++mt) {
            const float e = fmaf(c2, s_dm[mt * nks + kk], s_tb[mt * nks + kk]);
            const float x = (lane < nks && e > a.thr && e <= fmaxf(thr_mid, thr_keep)) ? __builtin_amdgcn_exp2f(e) : 0.f;
            mf += e <= thr_keep ? x : 0.f;
            mm += e <= thr_keep ? 0.f : x;
        }
        mf = 1024.f * wave_sum_f(mf);
        mm = 1024.f * wave_sum_f(mm);
        if (mf > 0.f || mm > 0.f) {          // (wave-uniform)
            // lower bound of the PSF peak = sum of the OTF: its first lines, exactly (every element is >= 0)
            const int H1 = N / 2 + 1;
            const float4* d4 = reinterpret_cast<const float4*>(a.D0t + (size_t)task * H1 * N);
            const float4* t4 = reinterpret_cast<const float4*>(a.tl2);
            float S = 0.f;
            const int n4 = peak_lines(N) * N / 4;         // a multiple of 128
            for (int i0 = lane; i0 < n4; i0 += 128) {     // four loads in flight
                const float4 da = d4[i0], ta = t4[i0], db = d4[i0 + 64], tb = t4[i0 + 64];
                const float xa = (__builtin_amdgcn_exp2f(fmaf(c2, fmaxf(da.x, 0.f), ta.x)) + __builtin_amdgcn_exp2f(fmaf(c2, fmaxf(da.y, 0.f), ta.y))) +
                                 (__builtin_amdgcn_exp2f(fmaf(c2, fmaxf(da.z, 0.f), ta.z)) + __builtin_amdgcn_exp2f(fmaf(c2, fmaxf(da.w, 0.f), ta.w)));
                const float xb = (__builtin_amdgcn_exp2f(fmaf(c2, fmaxf(db.x, 0.f), tb.x)) + __builtin_amdgcn_exp2f(fmaf(c2, fmaxf(db.y, 0.f), tb.y))) +
                                 (__builtin_amdgcn_exp2f(fmaf(c2, fmaxf(db.z, 0.f), tb.z)) + __builtin_amdgcn_exp2f(fmaf(c2, fmaxf(db.w, 0.f), tb.w)));
                // line 0 once, the others stand for both half planes
                S += (i0 < N / 4 ? xa : 2.f * xa) + (i0 + 64 < N / 4 ? xb : 2.f * xb);
            }
            // (tl2 carries the 2^kShift of the fp16 scaling: mf_common.h)
            const float budget = a.tier_half * wave_sum_f(S) * __builtin_amdgcn_exp2f(-kShift);
            if (mf > budget) {
                do {
                    thr_keep -= 2.f;
                    if (thr_keep <= a.thr) { thr_keep = a.thr; break; }
                } while (mass(a.thr, thr_keep) > budget);
                mm = mass(thr_keep, thr_mid);
            }
            // a mid block loses the low half of every element: 2^-11 of it at most (round to nearest fp16)
            while (thr_mid > thr_keep && mm * (1.f / 2048.f) > budget) {
                thr_mid -= 2.f;
                mm = mass(thr_keep, thr_mid);
            }
        }
    }
    u64 myf = 0, mym = 0;                    // lane mt keeps the two words of m-tile mt
    for (int mt = 0; mt < nmt; ++mt) {
        const float e = fmaf(c2, s_dm[mt * nks + kk], s_tb[mt * nks + kk]);
        const bool keep = lv && lane < nks && (nopr || e > thr_keep);
        const bool full = keep && (nopr || e > thr_mid);
        const u64 bf = __ballot(full), bm = __ballot(keep && !full);
        if (lane == mt) { myf = bf; mym = bm; }
    }
    if (lane < nmt) {
        if (lv) {                            // one coalesced store of the wavelength's words
            ulonglong2* o = reinterpret_cast<ulonglong2*>(a.own + ((size_t)task * a.nl + l) * nmt * 2) + lane;
            *o = make_ulonglong2(myf, mym);
        }
        s_any[w][lane] = myf | mym;
    }
    u64 ks_l = lane < nmt ? (myf | mym) : 0;
#pragma unroll
    for (int o = 1; o < kT2; o <<= 1) ks_l |= __shfl_xor(ks_l, o, 64);
    if (lv && (lane & 7) == 0 && (lane >> 3) < nsw) a.ksum[((size_t)task * a.nl + l) * nsw + (lane >> 3)] = ks_l;
    __syncthreads();
    if (w == 0) {                            // the group's first wave: lane = m-tile
        u64 u = 0;
        if (lane < nmt)
            for (int j = 0; j < a.per; ++j) u |= s_any[j][lane];
        if (lane < nmt) a.uni[((size_t)task * a.ngr + grp) * nmt + lane] = u;
        u64 ku = u;
        int pc = __builtin_popcountll(u);
#pragma unroll
        for (int o = 1; o < kT2; o <<= 1) {      // over the eight m-tiles of a sweep
            ku |= __shfl_xor(ku, o, 64);
            pc += __shfl_xor(pc, o, 64);
        }
        const bool lead = (lane & 7) == 0 && (lane >> 3) < nsw;
        if (lead) a.kuni[((size_t)task * a.ngr + grp) * nsw + (lane >> 3)] = ku;
        const u64 has = __ballot(lead && ku != 0);     // bit 8 sw
        int sweeps = 0;
#pragma unroll
        for (int sw = 0; sw < 8; ++sw) sweeps |= (int)((has >> (8 * sw)) & 1) << sw;
        if (lane == 0) a.gsw[task * a.ngr + grp] = sweeps;
        if (lead && ku != 0) {               // file the item in its work class
            const int cls = min(kMfLists - 1, pc * (kMfLists / kT2) / nks);
            a.items[(size_t)cls * a.cap + atomicAdd(a.sched + cls, 1)] =
                make_int4(task, grp, lane >> 3, __builtin_popcount(sweeps));
        }
    }
}

// the 64-bit word that lane `src` (wave-uniform) holds, into scalar registers
__device__ __forceinline__ u64 lane_word(u64 w, int src) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)w, src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(w >> 32), src);
    return ((u64)hi << 32) | lo;
}

// two LDS-DMA loads of one staging unit: 16 bytes per lane from sbase + voff / + voff16 to
// lds_dst + 16 lane / + 1024 + 16 lane (wait states: mf_common.h)
__device__ __forceinline__ void glds_pair(const void* sbase, unsigned voff, unsigned voff16, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_nop " MF_XSTR(MPSFR_MF_BASE_NOP) "\n\ts_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "v"(voff16), "s"(sbase), "s"(lds_dst)
        : "memory", "scc");
}

struct Mf2Args {
    int N, ntask, nl, per, ngr;
    const float* D0t;        // [ntask][N/2+1][N] (+ 16 finite padding lines)
    const float* tl2;        // [nmt 16][N]
    const LamPar* lp;
    const h8* E;
    const h4* G;
    const u64* own;          // K_MF_PREP
    const u64* uni;
    const u64* ksum;
    const u64* kuni;
    int* sched;              // [0..63] items per list (K_MF_PREP), [64] queue head
    const int4* items;       // [kMfLists][cap]
    int cap;
    f4* part;                // [ntask][nl][nsw][8][64]: partial tiles of (task, group)s with several sweeps
    unsigned long long* clk; // experiments (-DMPSFR_MF_CLOCK=1): per-wave phase times, or nullptr
    float tq_scale;          // 2^-ceil(log2 N) 2^-kTabShift: first-pass sums back into the fp16 range
    float* pre;              // [ntask][nl][40][40]
};

#ifndef MPSFR_MF2_PKFMA
#define MPSFR_MF2_PKFMA 0        // 1: v_pk_fma_f32 (inline asm) for the exponent, 0: two v_fma_f32
#endif

// x = 2^(c d + t) for two elements: the fp16 high halves, and the low halves if `full`
template <bool FULL>
__device__ __forceinline__ void otf_pair2(float c, f2 d, f2 t, unsigned* hi, unsigned* lo) {
#if MPSFR_MF2_PKFMA
    f2 y;
    const f2 cc = {c, c};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(cc), "v"(d), "v"(t));
    const float x0 = __builtin_amdgcn_exp2f(y[0]), x1 = __builtin_amdgcn_exp2f(y[1]);
#else
    const float x0 = __builtin_amdgcn_exp2f(__builtin_fmaf(c, d[0], t[0]));
    const float x1 = __builtin_amdgcn_exp2f(__builtin_fmaf(c, d[1], t[1]));
#endif
    if constexpr (FULL) {
        split_pair(x0, x1, hi, lo);
    } else {
        *hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x0, x1}, h2));
    }
}

#ifndef MPSFR_MF_CLOCK
#define MPSFR_MF_CLOCK 0
#endif

// ------------------------------------------------------------------------------------------
// K_OTF_MFMA2 (one direction).  Persistent workgroups (one per CU) of 2 per waves take items (task,
// wavelength group of `per` <= 8 wavelengths, sweep of eight m-tiles) from the list of K_MF_PREP.
// Wave w < per is the even half (m-tiles 0, 2, 4, 6 of the sweep) of wavelength slot w, wave
// per + w the odd half.  Per k-step (32 columns) the workgroup stages, by LDS-DMA into the buffer
// the previous k-step does not read: the D | log2 tel tiles the group's union mask keeps (16 units
// of 2 KB dealt over the waves) and, per wavelength with work in the k-step, its 6 KB E slab (high
// halves by the even wave, low halves by the odd one).  One raw s_barrier per k-step hands the
// buffers over (hipcc does not count asm memory operations: the waits are explicit).
// WPE = waves per SIMD the register budget is cut for (4: up to 16 waves, 128 registers; 3: up to
// 12 waves, 168).
// ------------------------------------------------------------------------------------------
template <int WPE>
__global__ void __launch_bounds__(WPE * 256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k_otf_mfma2(const Mf2Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, H1 = N / 2 + 1, nks = mf_nks(N), nmt_all = mf_nmt(N), nsw = (nmt_all + kT2 - 1) / kT2;
    const int per = a.per, ngr = a.ngr;
    const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    const int nw = 2 * per;
    const char* ttab = reinterpret_cast<const char*>(a.tl2);
    // The item number travels through a word of LDS behind the staging buffers and slabs.  The queue
    // runs a little ahead: the number of the next item is drawn during the last k-step of the current
    // one, its descriptor is loaded behind the second pass and its masks behind the epilogue, so
    // that an item starts without the three dependent memory round trips (atomic, descriptor,
    // masks: 13 % of a wave's lifetime when they sat at the head of every item).  (Drawn a whole
    // item ahead, the queue balances worse: the launch ended 12 us later on its last workgroups.)
    volatile int* s_next = reinterpret_cast<volatile int*>(smem + 2 * kStage2 + 2 * per * kSlab);
    // the 64 lists as one queue, heaviest class first: item i of the queue is entry i - first[c] of
    // the list c with first[c] <= i < first[c] + count[c]  (lane c keeps the bounds of list c)
    const int ccnt = a.sched[lane0];
    int cfirst = 0, nitems = 0;
#pragma unroll
    for (int c = kMfLists - 1; c >= 0; --c) {
        if (lane0 == c) cfirst = nitems;
        nitems += __builtin_amdgcn_readlane(ccnt, c);
    }
#if MPSFR_MF_CLOCK
    unsigned long long* const clk = a.clk != nullptr && lane0 == 0 ? a.clk + ((size_t)blockIdx.x * 16 + wave0) * 16 : nullptr;
    unsigned long long t_kloop = 0, t_stage = 0, t_tiles = 0, t_wload = 0, t_wbar = 0, t_pass2 = 0, t_masks = 0, t_tail = 0;
    unsigned long long n_ks = 0, n_tiles = 0, n_dma = 0, n_items = 0, t_first = 0;
#define MF2_NOW() __builtin_readcyclecounter()
    const unsigned long long t_begin = MF2_NOW();
#else
#define MF2_NOW() 0ull
#endif
    // descriptor {task, group, sweep, sweeps of the (task, group)} of queue item `item`
    auto load_item = [&](int item) -> int4 {
        const int cls = __builtin_ctzll(__ballot(item >= cfirst && item < cfirst + ccnt));
        const int ipos = item - __builtin_amdgcn_readlane(cfirst, cls);
        return a.items[(size_t)cls * a.cap + ipos];
    };
    // The masks of a sweep (K_MF_PREP) with ONE load instruction: lane 8 k + g holds, for m-tile
    // g0 + g, the group's staging mask (k = 0), this wavelength's full blocks (1) and mid blocks
    // (2); lane 24 the k-steps in which the wavelength has work, lane 25 the group's k-steps.
    auto load_masks = [&](int task, int grp, int sw, float* c2) -> u64 {
        const int half = wave0 >= per ? 1 : 0, slot = wave0 - half * per;
        const int l = grp * per + slot;
        const bool lv = l < a.nl;
        const int lc = lv ? l : a.nl - 1;
        *c2 = (float)a.lp[lc].c * 1.44269504088896340736f;
        const int g = lane0 & 7, mt = sw * kT2 + g, kind = lane0 >> 3;
        const size_t tl_ = (size_t)task * a.nl + lc, tg_ = (size_t)task * ngr + grp;
        u64 mw = 0;
        if (kind == 0) { if (mt < nmt_all) mw = a.uni[tg_ * nmt_all + mt]; }
        else if (kind < 3) { if (mt < nmt_all && lv) mw = a.own[(tl_ * nmt_all + mt) * 2 + kind - 1]; }
        else if (lane0 == 24) { if (lv) mw = a.ksum[tl_ * nsw + sw]; }
        else if (lane0 == 25) mw = a.kuni[tg_ * nsw + sw];
        return mw;
    };
    // The first round is dealt without a draw -- workgroup b starts on item b (the heaviest items, in
    // queue order) -- and the counter numbers the items behind it: 256 workgroups drawing from one word
    // at the same moment waited ~3 us for the last answer.
    int item = blockIdx.x;
    int4 it = make_int4(0, 0, 0, 0);
    u64 mw = 0;
    float c2 = 0.f;
    if (item < nitems) {
        it = load_item(item);
        mw = load_masks(__builtin_amdgcn_readfirstlane(it.x), __builtin_amdgcn_readfirstlane(it.y),
                        __builtin_amdgcn_readfirstlane(it.z), &c2);
    }

    while (item < nitems) {
#if MPSFR_MF_CLOCK
        const unsigned long long ti0 = MF2_NOW();
#endif
        // Everything derived from the lane and wave numbers is recomputed per item: hoisted out of
        // this loop the two dozen addresses and constants live across it and spill (the reloads sat
        // on the serial path of every item's epilogue).
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(wave));
        const int half = wave >= per ? 1 : 0, slot = wave - half * per;
        const int lr = lane & 15, lk = lane >> 4;
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const unsigned slab_off = 2 * kStage2 + (unsigned)slot * kSlab;      // + buf * per * kSlab
        // lane part of a tile's addresses (bytes); the rest is wave-uniform.  Lines beyond N/2 of
        // the last m-tile read the finite padding behind D (log2 tel = -inf there).
        const unsigned voff = (unsigned)((lr * N + 8 * lk) * sizeof(float)), voff16 = voff + 16;
        const unsigned voffb = (unsigned)lane * 16;
        const int task = __builtin_amdgcn_readfirstlane(it.x), grp = __builtin_amdgcn_readfirstlane(it.y);
        const int sw = __builtin_amdgcn_readfirstlane(it.z), nsweeps = __builtin_amdgcn_readfirstlane(it.w);
        const int g0 = sw * kT2;
        const int l = grp * per + slot;
        const bool lv = l < a.nl;
        const int lc = lv ? l : a.nl - 1;
        const char* dtask = reinterpret_cast<const char*>(a.D0t + (size_t)task * H1 * N);
        const char* etab = reinterpret_cast<const char*>(a.E + (size_t)lc * nks * NCT * 2 * 64);
        const h4* Gl = a.G + (size_t)lc * nmt_all * NJT * 2 * 2 * 64 + lane;
        const u64 kuni = lane_word(mw, 25);            // != 0: K_MF_PREP lists no empty item
        const u64 sany = lane_word(mw, 24);
        u64 of[kTW], om[kTW];              // this wave's tiles 2 i + half: full / mid blocks
#pragma unroll
        for (int i = 0; i < kTW; ++i) {
            of[i] = lane_word(mw, 8 + 2 * i + half);
            om[i] = lane_word(mw, 16 + 2 * i + half);
        }
        f4 acc[kTW][NCT];
#pragma unroll
        for (int i = 0; i < kTW; ++i)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = f4{0.f, 0.f, 0.f, 0.f};
#if MPSFR_MF_CLOCK
        const unsigned long long tm1 = MF2_NOW();
        t_masks += tm1 - ti0;
        n_items += 1;
#endif

        // This wave's share of the loads of k-step ks, into staging buffer `buf`.  Staging units of a
        // k-step: unit q = 2 g + kind (kind 0: D of tile g, 1: log2 tel), dealt over the waves from the
        // top (the waves with a second unit change with `per`).
        auto stage = [&](int ks, int buf) {
            const unsigned koff = (unsigned)(KBL * ks) * (unsigned)sizeof(float);
            for (int q = nw - 1 - wave; q < 2 * kT2; q += nw) {
                const int g = q >> 1;
                if (!((lane_word(mw, g) >> ks) & 1)) continue;
                const unsigned off = (unsigned)(MTL * (g0 + g)) * (unsigned)N * (unsigned)sizeof(float) + koff;
                glds_pair(((q & 1) ? ttab : dtask) + off, voff, voff16,
                          lds0 + buf * kStage2 + g * 4096 + (q & 1) * 2048);
#if MPSFR_MF_CLOCK
                n_dma += 2;
#endif
            }
            if ((sany >> ks) & 1) {
                const char* Ek = etab + (unsigned)ks * (NCT * 2 * 1024) + half * 1024;
                const unsigned dst = lds0 + slab_off + buf * per * kSlab + half * 1024;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) glds16s(Ek + ct * 2048, voffb, dst + ct * 2048);
#if MPSFR_MF_CLOCK
                n_dma += 3;
#endif
            }
        };
        u64 rest = kuni;
        int buf = 0;
        stage(__builtin_ctzll(rest), 0);
        // (lgkmcnt: the first thread's store of the next item's number)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if MPSFR_MF_CLOCK
        const unsigned long long tk0 = MF2_NOW();
        t_first += tk0 - tm1;
#endif
        for (;;) {
            const int ks = __builtin_ctzll(rest);
            rest &= rest - 1;
            // the next item's number is drawn during the last k-step (the atomic is in flight behind
            // its tile steps) and read behind the barrier that ends the k-loop
            if (rest == 0 && threadIdx.x == 0) *s_next = (int)gridDim.x + atomicAdd(a.sched + kMfLists, 1);
            unsigned fb = 0, mb = 0;
#pragma unroll
            for (int i = 0; i < kTW; ++i) {
                fb |= (unsigned)((of[i] >> ks) & 1) << i;
                mb |= (unsigned)((om[i] >> ks) & 1) << i;
            }
            const unsigned ab = fb | mb;
            h8 bh[NCT], bl[NCT];
            if (ab != 0) {
                const unsigned char* slab = smem + slab_off + buf * per * kSlab + lane * 16;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    bh[ct] = *reinterpret_cast<const h8*>(slab + (ct * 2) * 1024);
                    bl[ct] = *reinterpret_cast<const h8*>(slab + (ct * 2 + 1) * 1024);
                }
            }
#if MPSFR_MF_CLOCK
            const unsigned long long ts0 = MF2_NOW();
#endif
            if (rest != 0) stage(__builtin_ctzll(rest), buf ^ 1);
#if MPSFR_MF_CLOCK
            const unsigned long long ts1 = MF2_NOW();
            t_stage += ts1 - ts0;
            n_ks += 1;
            n_tiles += __builtin_popcount(ab);
#endif
            if (ab != 0) {
                const unsigned char* tbuf = smem + buf * kStage2 + half * 4096 + lane * 16;
#pragma unroll
                for (int i = 0; i < kTW; ++i) {
                    if (!((ab >> i) & 1)) continue;
                    const unsigned char* tp = tbuf + i * 8192;
                    const f4 d0 = *reinterpret_cast<const f4*>(tp);
                    const f4 d1 = *reinterpret_cast<const f4*>(tp + 1024);
                    const f4 t0 = *reinterpret_cast<const f4*>(tp + 2048);
                    const f4 t1 = *reinterpret_cast<const f4*>(tp + 3072);
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    unsigned hi[4], lo[4];
                    if ((fb >> i) & 1) {
                        otf_pair2<true>(c2, f2{d0[0], d0[1]}, f2{t0[0], t0[1]}, &hi[0], &lo[0]);
                        otf_pair2<true>(c2, f2{d0[2], d0[3]}, f2{t0[2], t0[3]}, &hi[1], &lo[1]);
                        otf_pair2<true>(c2, f2{d1[0], d1[1]}, f2{t1[0], t1[1]}, &hi[2], &lo[2]);
                        otf_pair2<true>(c2, f2{d1[2], d1[3]}, f2{t1[2], t1[3]}, &hi[3], &lo[3]);
                        const h8 ah = __builtin_bit_cast(h8, u4{hi[0], hi[1], hi[2], hi[3]});
                        const h8 al = __builtin_bit_cast(h8, u4{lo[0], lo[1], lo[2], lo[3]});
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ct], acc[i][ct], 0, 0, 0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ct], acc[i][ct], 0, 0, 0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ct], acc[i][ct], 0, 0, 0);
                    } else {
                        otf_pair2<false>(c2, f2{d0[0], d0[1]}, f2{t0[0], t0[1]}, &hi[0], &lo[0]);
                        otf_pair2<false>(c2, f2{d0[2], d0[3]}, f2{t0[2], t0[3]}, &hi[1], &lo[1]);
                        otf_pair2<false>(c2, f2{d1[0], d1[1]}, f2{t1[0], t1[1]}, &hi[2], &lo[2]);
                        otf_pair2<false>(c2, f2{d1[2], d1[3]}, f2{t1[2], t1[3]}, &hi[3], &lo[3]);
                        const h8 ah = __builtin_bit_cast(h8, u4{hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ct], acc[i][ct], 0, 0, 0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ct], acc[i][ct], 0, 0, 0);
                    }
                }
            }
#if MPSFR_MF_CLOCK
            const unsigned long long tw0 = MF2_NOW();
            t_tiles += tw0 - ts1;
#endif
            if (rest == 0) break;       // the last k-step's tail is below
            // the next k-step's tiles and slabs have landed, and nobody reads this k-step's any more
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if MPSFR_MF_CLOCK
            const unsigned long long tw1 = MF2_NOW();
#endif
            __builtin_amdgcn_s_barrier();
#if MPSFR_MF_CLOCK
            t_wload += tw1 - tw0;
            t_wbar += MF2_NOW() - tw1;
#endif
            buf ^= 1;
        }
        // Tail of the last k-step: no load is in flight (the wait returns at once; it is there for the
        // static check of tools/isa_lint.py, which cannot know that), and the wave has finished its LDS
        // reads before the barrier that frees the staging buffers.  The next item's masks and the G
        // fragments of the second pass's first tile are requested BEFORE that barrier: they land while
        // the wave waits for the others.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        unsigned tbits = 0;
#pragma unroll
        for (int i = 0; i < kTW; ++i) tbits |= (unsigned)((of[i] | om[i]) != 0) << i;
        h4 gq[NJT][2][2];
        auto fetch_g = [&](int i) {
            const h4* Gm = Gl + (size_t)(g0 + 2 * i + half) * NJT * 2 * 2 * 64;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) gq[jt][xy][hl] = Gm[((jt * 2 + xy) * 2 + hl) * 64];
        };
        if (tbits != 0) fetch_g(__builtin_ctz(tbits));
#if MPSFR_MF_CLOCK
        const unsigned long long tw1l = MF2_NOW();
#endif
        __builtin_amdgcn_s_barrier();
        const int item_next = __builtin_amdgcn_readfirstlane(*s_next);
        int4 it_next = make_int4(0, 0, 0, 0);
        if (item_next < nitems) it_next = load_item(item_next);     // lands during the second pass
#if MPSFR_MF_CLOCK
        const unsigned long long tk1 = MF2_NOW();
        t_wbar += tk1 - tw1l;
        t_kloop += tk1 - tk0;
#endif
        // second pass: the accumulator tile (rows = lines on registers / lane groups, column on
        // the lane) is the A operand of a 16x16x16 product that sums over its rows; the G fragments
        // of the next tile the wave needs are in flight behind the products of the current one
        f4 P0[NJT], Q0[NJT], R2x[NJT], R2y[NJT];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            P0[jt] = f4{0.f, 0.f, 0.f, 0.f};
            Q0[jt] = f4{0.f, 0.f, 0.f, 0.f};
            R2x[jt] = f4{0.f, 0.f, 0.f, 0.f};
            R2y[jt] = f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < kTW; ++i) {
            if (!((tbits >> i) & 1)) continue;
            h4 gc[NJT][2][2];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) gc[jt][xy][hl] = gq[jt][xy][hl];
            const unsigned later = tbits & ~((2u << i) - 1u);
            fetch_g(later != 0 ? __builtin_ctz(later) : i);
            h4 th[NCT], tw[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 h, w;
                    split16(acc[i][ct][r] * a.tq_scale, &h, &w);
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
        // the next item's masks: in flight behind the epilogue
        u64 mw_next = 0;
        float c2_next = 0.f;
        if (item_next < nitems)
            mw_next = load_masks(__builtin_amdgcn_readfirstlane(it_next.x), __builtin_amdgcn_readfirstlane(it_next.y),
                                 __builtin_amdgcn_readfirstlane(it_next.z), &c2_next);
#if MPSFR_MF_CLOCK
        const unsigned long long t_red = MF2_NOW();
        t_pass2 += t_red - tk1;
#endif
        // The two halves of the sweep meet in LDS (the staging buffers are free: the k-loop ends on a
        // barrier behind its last reads): the odd wave leaves its partial tiles, the even wave adds
        // them to its own -- a fixed order.  A (task, group) with a single sweep is finished here;
        // otherwise the sweep's tiles go to memory for K_MF_FINISH.
        f4* red = reinterpret_cast<f4*>(smem + slot * 8192) + lane;
        if (half == 1) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                red[(0 * NJT + jt) * 64] = P0[jt];
                red[(1 * NJT + jt) * 64] = Q0[jt];
                red[(2 * NJT + jt) * 64] = R2x[jt];
                red[(3 * NJT + jt) * 64] = R2y[jt];
            }
        }
        __syncthreads();
        if (half == 0 && lv) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                P0[jt] += red[(0 * NJT + jt) * 64];
                Q0[jt] += red[(1 * NJT + jt) * 64];
                R2x[jt] += red[(2 * NJT + jt) * 64];
                R2y[jt] += red[(3 * NJT + jt) * 64];
            }
            if (nsweeps == 1) {
                write_stamp(P0, Q0, R2x, R2y, lr, lk, a.pre + ((size_t)task * a.nl + l) * NS * NS);
            } else {
                f4* pt = a.part + (((size_t)task * a.nl + l) * nsw + sw) * (4 * NJT * 64) + lane;
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    if (!part_col(jt, lr)) continue;
                    pt[(0 * NJT + jt) * 64] = P0[jt];
                    pt[(1 * NJT + jt) * 64] = Q0[jt];
                    if (part_r2x(lk)) pt[(2 * NJT + jt) * 64] = R2x[jt];
                    if (part_r2y(lk)) pt[(3 * NJT + jt) * 64] = R2y[jt];
                }
            }
        }
        __syncthreads();           // the tiles in LDS have been read: the next item may stage over them
        item = item_next;
        it = it_next;
        mw = mw_next;
        c2 = c2_next;
#if MPSFR_MF_CLOCK
        t_tail += MF2_NOW() - t_red;
#endif
    }
#if MPSFR_MF_CLOCK
    if (clk != nullptr) {
        const unsigned long long t_end = MF2_NOW();
        clk[0] = t_begin; clk[1] = t_masks; clk[2] = t_kloop; clk[3] = t_stage; clk[4] = t_tiles;
        clk[5] = t_wload; clk[6] = t_wbar; clk[7] = t_pass2; clk[8] = t_end; clk[9] = n_ks;
        clk[10] = n_tiles; clk[11] = t_tail; clk[12] = n_dma; clk[13] = n_items; clk[14] = t_first;
    }
#endif
#undef MF2_NOW
}

// K_MF_FINISH: the stamps of the (task, group)s with several sweeps: their partial tiles added in
// sweep order, then the epilogue of K_OTF_MFMA2.  One wave per (task, wavelength).  (One workgroup
// per (task, group) with a wave per wavelength measured slower: 9.3 us against 7.7.)
__global__ void __launch_bounds__(64) k_mf_finish(int N, int nl, int per, int ngr, const int* __restrict__ gsw,
                                                  const f4* __restrict__ part, float* __restrict__ pre) {
    const int l = blockIdx.x, task = blockIdx.y, lane = threadIdx.x;
    const int m = gsw[task * ngr + l / per];
    if (__builtin_popcount(m) <= 1) return;
    MfFinish f;
    f.gsw = gsw; f.part = part; f.per = per; f.ngr = ngr; f.nsw = (mf_nmt(N) + kT2 - 1) / kT2;
    finish_stamp(f, m, task, nl, l, lane, pre + ((size_t)task * nl + l) * NS * NS);
}

}  // namespace

static size_t mf2_nsw(int N) { return (size_t)(mf_nmt(N) + kT2 - 1) / kT2; }
size_t mf2_own_bytes(int N, int ntask, int nl) { return (size_t)ntask * nl * mf_nmt(N) * 2 * sizeof(u64); }
// uni | ksum | kuni in one buffer (any grouping: at most nl groups)
size_t mf2_uni_bytes(int N, int ntask, int nl) {
    return (size_t)ntask * nl * (mf_nmt(N) + 2 * mf2_nsw(N)) * sizeof(u64);
}
void mf2_groups(int nl, int permax, int* per, int* ngr);
// sched[68] | gsw | items (int4) [kMfLists][ntask ngr nsw]
static size_t mf2_cap(int N, int ntask, int ngr) { return (size_t)ntask * ngr * mf2_nsw(N); }
size_t mf2_sched_bytes(int N, int ntask, int nl, int permax) {
    int per, ngr;
    mf2_groups(nl, permax, &per, &ngr);
    return (68 + (size_t)ntask * nl + 4 + kMfLists * mf2_cap(N, ntask, ngr) * 4) * sizeof(int);
}
size_t mf2_part_bytes(int N, int ntask, int nl) { return (size_t)ntask * nl * mf2_nsw(N) * 4 * NJT * 64 * sizeof(f4); }

// wavelength groups of at most `permax` (7: 14 waves at 128 registers; 6: 12 waves at 168), as even
// as possible.  (Eight would fill the 160 KB of LDS to the last byte and leave no word for the queue.)
void mf2_groups(int nl, int permax, int* per, int* ngr) {
    if (permax > 7) permax = 7;
    *ngr = (nl + permax - 1) / permax;
    *per = (nl + *ngr - 1) / *ngr;
}

namespace {
struct SchedPtrs { int* sched; int* gsw; int4* items; };
SchedPtrs sched_ptrs(void* d_sched, int ntask, int nl) {
    SchedPtrs p;
    p.sched = (int*)d_sched;
    p.gsw = p.sched + 68;
    p.items = (int4*)(p.sched + 68 + (((size_t)ntask * nl + 3) & ~(size_t)3));
    return p;
}
}  // namespace

// masks of every (task, wavelength) and the work lists (d_sched[0..16] must be zero: launch_colfft_dphi
// does that).  d_dminb = nullptr: no pruning.
void launch_mf_prep(hipStream_t s, int N, int ntask, int nl, int permax, const LamPar* d_lp,
                    const float* d_dminb, const float* d_tlb, float thr, float thr_floor, float thr_mid,
                    float tier_half, const void* d_D0t, const float* d_tl2, void* d_own,
                    void* d_uni, void* d_sched, const float* d_dlin) {
    MaskArgs a;
    a.dlin = d_dlin;
    a.thr_floor = thr_floor; a.tier_half = tier_half;
    a.D0t = (const float*)d_D0t; a.tl2 = d_tl2;
    a.N = N; a.nl = nl;
    mf2_groups(nl, permax, &a.per, &a.ngr);
    a.lp = d_lp; a.dminb = d_dminb; a.tlb = d_tlb;
    a.thr = thr; a.thr_mid = thr_mid;
    a.own = (u64*)d_own; a.uni = (u64*)d_uni;
    a.ksum = a.uni + (size_t)ntask * nl * mf_nmt(N);
    a.kuni = a.ksum + (size_t)ntask * nl * mf2_nsw(N);
    const SchedPtrs p = sched_ptrs(d_sched, ntask, nl);
    a.gsw = p.gsw; a.sched = p.sched; a.items = p.items;
    a.cap = (int)mf2_cap(N, ntask, a.ngr);
    hipLaunchKernelGGL(k_mf_prep, dim3(a.ngr, ntask), dim3(64 * a.per), 0, s, a);
}

void launch_otf_mfma2(hipStream_t s, int N, int ntask, int nl, int permax, int ncu, const void* d_D0t,
                      const float* d_tl2, const LamPar* d_lp, const void* d_E, const void* d_G,
                      const void* d_own, const void* d_uni, void* d_sched, void* d_part, void* d_pre,
                      void* d_clk, hipEvent_t ev_start, hipEvent_t ev_stop, bool finish) {
    Mf2Args a;
    a.N = N; a.ntask = ntask; a.nl = nl;
    mf2_groups(nl, permax, &a.per, &a.ngr);
    a.D0t = (const float*)d_D0t; a.tl2 = d_tl2; a.lp = d_lp;
    a.E = (const h8*)d_E; a.G = (const h4*)d_G;
    a.own = (const u64*)d_own; a.uni = (const u64*)d_uni;
    a.ksum = a.uni + (size_t)ntask * nl * mf_nmt(N);
    a.kuni = a.ksum + (size_t)ntask * nl * mf2_nsw(N);
    const SchedPtrs p = sched_ptrs(d_sched, ntask, nl);
    a.sched = p.sched; a.items = p.items;
    a.cap = (int)mf2_cap(N, ntask, a.ngr);
    a.part = (f4*)d_part;
    a.clk = (unsigned long long*)d_clk;
    int lg = 0;
    while ((1 << lg) < N) ++lg;
    a.tq_scale = 1.0f / ((float)(1 << lg) * (float)(1 << kTabShift));
    a.pre = (float*)d_pre;
    // one persistent workgroup per CU (its LDS admits no second one), never more than there can be items
    const int maxitems = ntask * a.ngr * (int)mf2_nsw(N);
    const int nwg = ncu < maxitems ? ncu : maxitems;
    const size_t sm = 2 * (size_t)kStage2 + 2 * (size_t)a.per * kSlab + 64;     // + the queue word
    // (ev_start / ev_stop: the kernel's own start and end time stamps through the dispatch packet -- no marker
    // packets before and behind it, which cost the queue ~7 us each)
    if (permax > 6) {
        allow_smem(k_otf_mfma2<4>, 2 * (size_t)kStage2 + 2 * (size_t)7 * kSlab + 64);
        if (ev_start != nullptr)
            hipExtLaunchKernelGGL(k_otf_mfma2<4>, dim3(nwg), dim3(128 * a.per), (unsigned)sm, s, ev_start, ev_stop, 0, a);
        else
            hipLaunchKernelGGL(k_otf_mfma2<4>, dim3(nwg), dim3(128 * a.per), sm, s, a);
    } else {
        allow_smem(k_otf_mfma2<3>, 2 * (size_t)kStage2 + 2 * (size_t)6 * kSlab + 64);
        if (ev_start != nullptr)
            hipExtLaunchKernelGGL(k_otf_mfma2<3>, dim3(nwg), dim3(128 * a.per), (unsigned)sm, s, ev_start, ev_stop, 0, a);
        else
            hipLaunchKernelGGL(k_otf_mfma2<3>, dim3(nwg), dim3(128 * a.per), sm, s, a);
    }
    if (finish)
        hipLaunchKernelGGL(k_mf_finish, dim3(nl, ntask), dim3(64), 0, s, N, nl, a.per, a.ngr, (const int*)p.gsw,
                           (const f4*)a.part, a.pre);
}

// K_MF_FINISH on its own (launch_otf_mfma2 with finish = false leaves the stamps of the (task, group)s with several
// sweeps as partial tiles: K_CONV_FFT finishes them on its way -- MfFinishArgs -- and a reader of `pre` itself,
// debug fetches and psf_muse, asks for this)
void launch_mf_finish(hipStream_t s, int N, int ntask, int nl, int permax, void* d_sched, const void* d_part, void* d_pre) {
    int per, ngr;
    mf2_groups(nl, permax, &per, &ngr);
    const SchedPtrs p = sched_ptrs(d_sched, ntask, nl);
    hipLaunchKernelGGL(k_mf_finish, dim3(nl, ntask), dim3(64), 0, s, N, nl, per, ngr, (const int*)p.gsw,
                       (const f4*)d_part, (float*)d_pre);
}

MfFinishArgs mf2_finish_args(int N, int ntask, int nl, int permax, void* d_sched, const void* d_part) {
    MfFinishArgs f;
    mf2_groups(nl, permax, &f.per, &f.ngr);
    f.gsw = sched_ptrs(d_sched, ntask, nl).gsw;
    f.part = d_part;
    f.nsw = (int)mf2_nsw(N);
    return f;
}

}  // namespace mpsfr
