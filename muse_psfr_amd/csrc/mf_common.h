// Shared pieces of the matrix-core per-wavelength kernels (otf_mfma.hip, otf_mfma2.hip): operand
// types, tile geometry, the scaling of the split-fp16 operands, LDS-DMA statements, the fp16
// split, the second-pass product and the stamp epilogue.  Internal linkage throughout.
// Reference citations are to /root/reference/muse_psfr/psfrec.py.
#pragma once
#include "device_common.h"

namespace mpsfr {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int MTL = 16;      // lines per m-tile
constexpr int KBL = 32;      // columns (u) per k-step
constexpr int NCT = 3;       // column tiles of the first pass: 48 >= 42 real columns
constexpr int NJT = 2;       // column tiles of the second pass: 32 >= 21
// The OTF (<= 1) is generated times 2^kShift = 32768, so that the low half of an element (<= 2^-12 of
// it) is a NORMAL fp16 number -- full 11 bits -- down to elements of 2^-14 / 2^-12 / 2^15 = 7.6e-6; below
// that the low half is subnormal (fewer bits: an absolute 2e-9 of the largest element at worst), and an
// element below 2^-29 has both halves subnormal.  (The gfx950 matrix cores multiply subnormal fp16
// inputs like any other -- they do not flush them: tests/test_gpu_parity.py::
// test_precision_tiers_of_the_matrix_core_stage.)  The first-pass sums, scaled back by the table factors,
// stay below 2^kShift as well.  The stamp is normalised to sum 1 at the end (psfrec.py:685), so the
// factor drops out.
constexpr float kShift = 15.0f;
// The E and G tables (|E| <= 1, |G| <= 2) are stored times 2^kTabShift for the same reason: the low
// half of an entry (<= 2^-12 of it) would otherwise sit in the fp16 subnormal range.
constexpr int kTabShift = 9;

__host__ __device__ constexpr int mf_nks(int N) { return N / KBL; }
__host__ __device__ constexpr int mf_nmt(int N) { return (N / 2 + 1 + MTL - 1) / MTL; }

[[maybe_unused]] __device__ __forceinline__ void split16(float x, _Float16* hi, _Float16* lo) {
    const _Float16 h = (_Float16)x;
    *hi = h;
    *lo = (_Float16)(x - (float)h);
}

// three fp16 products for one fp32-grade product: c += a b with a = ahi + alo, b = bhi + blo
[[maybe_unused]] __device__ __forceinline__ f4 mm16(f4 c, h4 ahi, h4 alo, h4 bhi, h4 blo) {
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(alo, bhi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ahi, blo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(ahi, bhi, c, 0, 0, 0);
}

// Epilogue of the per-wavelength kernels: the second-pass tiles of one stamp -> the 40 x 40 stamp.
[[maybe_unused]] __device__ __forceinline__ void write_stamp(const f4* P0, const f4* Q0, const f4* R2x,
                                                            const f4* R2y, int lr, int lk,
                                                            float* __restrict__ out) {
    // Result tiles: column j = 16 jt + lr on the lane, row 4 lk + r in register r.
    //   P0 / Q0: rows = samples i = 0..15;  R2x rows 0..4 = P of i = 16..20;  R2y rows 8..12 = Q of
    //   i = 16..20 (32 lanes away).  stamp[i][j] = P + Q, and with Tq[v][40-i] = conj Tq[v][i],
    //   G[v][40-j] = conj G[v][j]:  stamp[40-i][j] = stamp[i][40-j] = P - Q,  stamp[40-i][40-j] = P + Q.
    // Clamp (psfrec.py:680), normalise to sum 1 (:685).
    float va[2 * NJT * 4], vb[2 * NJT * 4];     // P + Q, P - Q (clamped); first the 0..15 rows
    float part = 0.f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int jj = 16 * jt + lr;
        const bool jok = jj < NSH;
        const float mj = (jj >= 1 && jj <= NS / 2 - 1) ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float qhi = __shfl_xor(R2y[jt][r], 32, 64);
#pragma unroll
            for (int top = 0; top < 2; ++top) {
                const int row = 4 * lk + r;
                const int ii = top ? 16 + row : row;
                const bool ok = jok && (top ? row < 5 : true);
                const float Pv = top ? R2x[jt][r] : P0[jt][r];
                const float Qv = top ? qhi : Q0[jt][r];
                const float pa = ok ? fmaxf(Pv + Qv, 0.f) : 0.f;
                const float pb = ok ? fmaxf(Pv - Qv, 0.f) : 0.f;
                const float mi = (ii >= 1 && ii <= NS / 2 - 1) ? 1.f : 0.f;
                part += pa * (1.f + mi * mj) + pb * (mi + mj);
                va[(top * NJT + jt) * 4 + r] = pa;
                vb[(top * NJT + jt) * 4 + r] = pb;
            }
        }
    }
    const double tot = wave_sum((double)part);
    const float inv = (float)(1.0 / tot);
#pragma unroll
    for (int top = 0; top < 2; ++top)
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * lk + r, ii = top ? 16 + row : row, jj = 16 * jt + lr;
                if (jj >= NSH || (top && row >= 5)) continue;
                const float pa = va[(top * NJT + jt) * 4 + r] * inv, pb = vb[(top * NJT + jt) * 4 + r] * inv;
                const bool mi = ii >= 1 && ii <= NS / 2 - 1, mj = jj >= 1 && jj <= NS / 2 - 1;
                out[ii * NS + jj] = pa;
                if (mi) out[(NS - ii) * NS + jj] = pb;
                if (mj) out[ii * NS + NS - jj] = pb;
                if (mi && mj) out[(NS - ii) * NS + NS - jj] = pa;
            }
}

// Which lanes of the second-pass result tiles the epilogue (write_stamp) reads: columns
// 16 jt + lr < 21; of R2x rows 0..4 (lane groups lk = 0, 1), of R2y rows 8..12 (lk = 2, 3).  Only
// those travel through memory between K_OTF_MFMA2 and whoever finishes the stamp (half of the 8 KB per sweep).
[[maybe_unused]] __device__ __forceinline__ bool part_col(int jt, int lr) { return 16 * jt + lr < NSH; }
[[maybe_unused]] __device__ __forceinline__ bool part_r2x(int lk) { return lk < 2; }
[[maybe_unused]] __device__ __forceinline__ bool part_r2y(int lk) { return lk >= 2; }

constexpr int kT2 = 8;                  // m-tiles staged per sweep over the k-steps (K_OTF_MFMA2)

// What finishes a stamp whose (task, wavelength group) K_OTF_MFMA2 ran as several sweeps: where the partial
// result tiles lie and which sweeps wrote one.
struct MfFinish {
    const int* gsw;          // [ntask][ngr] bit mask of the sweeps of a (task, group); nullptr: nothing to finish
    const f4* part;          // [ntask][nl][nsw][4 NJT][64] partial tiles
    int per, ngr, nsw;
};

// One wave: the partial tiles of stamp (task, l) added in sweep order, then the epilogue of K_OTF_MFMA2
// (write_stamp) into `out` (global memory, or LDS for a consumer that takes the stamp from there).  `m`: the
// sweep mask of the stamp's (task, group), more than one bit set.
[[maybe_unused]] __device__ __forceinline__ void finish_stamp(const MfFinish& f, int m, int task, int nl, int l,
                                                             int lane, float* __restrict__ out) {
    f4 P0[NJT], Q0[NJT], R2x[NJT], R2y[NJT];
    bool first = true;
    for (int sw = 0; sw < f.nsw; ++sw) {
        if (!((m >> sw) & 1)) continue;
        const f4* pt = f.part + (((size_t)task * nl + l) * f.nsw + sw) * (4 * NJT * 64) + lane;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const f4 z = {0.f, 0.f, 0.f, 0.f};
            const bool col = part_col(jt, lane & 15);
            const f4 p = col ? pt[(0 * NJT + jt) * 64] : z, q = col ? pt[(1 * NJT + jt) * 64] : z;
            const f4 x = col && part_r2x(lane >> 4) ? pt[(2 * NJT + jt) * 64] : z;
            const f4 y = col && part_r2y(lane >> 4) ? pt[(3 * NJT + jt) * 64] : z;
            P0[jt] = first ? p : P0[jt] + p;
            Q0[jt] = first ? q : Q0[jt] + q;
            R2x[jt] = first ? x : R2x[jt] + x;
            R2y[jt] = first ? y : R2y[jt] + y;
        }
        first = false;
    }
    write_stamp(P0, Q0, R2x, R2y, lane & 15, lane >> 4, out);
}

#ifndef MPSFR_MF_BASE_NOP
#define MPSFR_MF_BASE_NOP 4      // s_nop N opening every LDS-DMA statement (see below)
#endif
#ifndef MF_XSTR
#define MF_STR(x_) #x_
#define MF_XSTR(x_) MF_STR(x_)
#endif

// Wait states of the inline-asm statements below (hipcc pads nothing inside an asm string and
// does not know what the string reads).  The scalar operands of a load statement (its base, the
// LDS address that goes to M0) may have been written by a VECTOR instruction -- v_readfirstlane, or
// a v_cmp / v_readlane feeding the scalar address arithmetic -- and an SGPR written by the vector
// pipe needs five wait states before a vector-memory instruction reads it as its base; the
// compiler pads that for its own loads, not for one inside an asm string.  Every load statement
// therefore opens with s_nop 4, and M0 gets one wait state (s_nop 0) before the LDS-DMA that uses
// it.  (A version without the s_nop 4 passed every test until an unrelated edit moved the address
// arithmetic next to the statement: stamps off by 1e-5, no fault.)  tests/test_isa_lint.py checks
// the emitted code object for both distances.
// one LDS-DMA load: 16 bytes per lane from sbase + voff to lds_dst + 16 lane
[[maybe_unused]] __device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_nop " MF_XSTR(MPSFR_MF_BASE_NOP) "\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
// the four loads of a tile: D (two halves of the lane's 8 columns), log2 tel (likewise)
[[maybe_unused]] __device__ __forceinline__ void glds_tile(const void* dbase, const void* tbase, unsigned voff,
                                                          unsigned voff16, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_nop " MF_XSTR(MPSFR_MF_BASE_NOP) "\n\ts_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "v"(voff16), "s"(dbase), "s"(tbase), "s"(lds_dst)
        : "memory", "scc");
}
// two elements split into fp16 halves: hi = rne(x), lo = rne(x - hi).  The conversions are the
// compiler's (v_cvt_pk_f16_f32; it knows the wait states between a vector write and the MFMA that
// takes it as an operand, and after a transcendental); only the two v_fma_mix_f32, which read the
// fp16 halves of hi directly, are asm -- a vector instruction between vector instructions.
[[maybe_unused]] __device__ __forceinline__ void split_pair(float x0, float x1, unsigned* hi, unsigned* lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x0, x1}, h2));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(x1));
    *lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{l0, l1}, h2));
    *hi = h;
}

}  // namespace
}  // namespace mpsfr
