// Launchers for the HIP kernels of the PSF-reconstruction hot path (implemented in stage_a.hip,
// per_lambda.hip and stamps.hip).
// Host code (mpsfr_api.cpp) sees only these plain functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpsfr {

constexpr int NS = 40;       // dimpsf, psfrec.py:658
constexpr int NSH = NS / 2 + 1;  // samples 0..20 of a line; sample 40-i is the conjugate of i
constexpr int KS = 41;       // Moffat kernel side, psfrec.py:911-916
constexpr int NAO = 80;      // AO-corrected zone, psfrec.py:103, 138
constexpr int NFIT = 16;
constexpr int KHAT = 33 * 64;  // complex entries of one kernel spectrum (k_khat)
constexpr int MAXLGS = 4;

// per-task scalars of the PSD model
struct TaskPar {
    double r0m53;     // r0^(-5/3), r0 at 0.5 um (psfrec.py:108, 183-187)
    double inv_l0sq;  // (1/L0)^2
    double cn2_0;     // normalised Cn2 of the ground layer (psfrec.py:57-58)
    double cn2_1;
    int geom;         // 0 = 4 LGS, 1 = 3 LGS (psfrec.py:86-91)
    int basis;        // 0: a task.  k + 1: "basis task" k of the series form of stage A (stage_a2.hip):
                      // PSD = r0m53 cfit (f^2 + inv_l0sq)^(-11/6 - k) [f >= fc], no corrected zone
};

// per-wavelength scalars
struct LamPar {
    double c;         // -0.5 * (2 pi / lambda_nm)^2   (psfrec.py:717, 793)
    int npixc;        // psfrec.py:663-664
    int pad;
};

struct AoGeom {
    double h[2];            // layer altitudes [m]
    double wind[2][2];      // wind[xy][layer] m/s  (psfrec.py:594)
    int nlgs[2];            // per geometry
    double poslgs[2][2][MAXLGS];  // [geom][xy][lgs] arcmin (psfrec.py:536)
    int ndir;
    double dir[2][25];      // [xy][dir] arcmin
};

// work lists of the thin-wave matrix-core kernel: one per work class (a lane of the consumer's wave keeps the
// bounds of one list), then the queue head
constexpr int kMfLists = 64;
constexpr int kMfSchedInts = kMfLists + 1;

enum KernelId {
    K_AO_TABLES = 0,
    K_TEL_OTF,
    K_PSD_ROWFFT,
    K_COLFFT_DPHI,
    K_GTABLE,
    K_MOFFAT_KERNELS,
    K_OTF_ROWFFT,
    K_COLPASS,
    K_CONV,
    K_FIT,
    K_STAMP_SUM,
    K_VKEEP,        // FFT path: K_DMIN + K_VKEEP
    K_OTF_MFMA,     // matrix-core per-wavelength stage (K_OTF_MFMA2 + K_MF_FINISH, or K_OTF_MFMA1)
    K_MF_PREP,      // its preparation: K_DMIN + K_MF_PREP (or K_DMIN + K_VKEEP + K_TASK_ORDER)
    K_PATCH,        // series form of stage A (stage_a2.hip): K_PATCH_GEN + K_PATCH_ROWS
    K_DPHI_SERIES,  //   and K_DPHI_SERIES
    K_COUNT
};

void launch_ao_tables(hipStream_t s, const AoGeom& g, const uint8_t* d_mask_rec,
                      const uint8_t* d_mask_res, double* d_tab);
void launch_tel_otf(hipStream_t s, int N, const uint64_t* d_rows, int words, double pupsum,
                    void* d_tel, bool f64out);
// d_dcpart: [ntd][psd_rowfft_groups(N)] the workgroups' shares of the PSD sum (bg[0,0], psfrec.py:721),
// added by launch_colfft_dphi; f64: the f64 mode (two Newton steps in x^(-11/6) instead of one)
int psd_rowfft_groups(int N);
void launch_psd_rowfft(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp,
                       const double* d_aotab, double cfit, void* d_C, const void* d_tw64,
                       double* d_dcpart, bool f64);
// d_zero: kMfSchedInts ints the kernel sets to zero (the work-list counters of launch_mf_prep), or nullptr
void launch_colfft_dphi(hipStream_t s, int N, int ntd, const void* d_C, const double* d_dcpart,
                        double scale2, void* d_D0t, bool f64out, const void* d_tw64,
                        int* d_zero = nullptr);
// Pruning of the per-wavelength stage (stage_a.hip, "Line pruning"): minima of D per line
// ([ntd][N/2+1]) and per block of 16 lines x 32 columns ([ntd][nmt][N/32]), then the lines to keep
// per (task, wavelength pair) and the block minima over the directions
// (f64: D / the telescope OTF are double; the minima are rounded down, the maxima up)
void launch_dmin(hipStream_t s, int N, int ntd, const void* d_D0t, float* d_dline, float* d_dblk,
                 bool f64 = false);
// Stage-level entry points (stage_a.hip): the PSD of one task as an image [ndir][N][N] (centred, in the
// reference's units: `unit` = (lambda_ref 1000 / 2 pi)^2), and the structure function of an arbitrary
// PSD image: d_Cm [ndir][N][N/2+1] complex workspace, scale = 2 / L^2 for a PSD in the reference's units.
void launch_psd_image(hipStream_t s, int N, int ndir, const TaskPar& p, const double* d_aotab, double cfit,
                      double unit, double* d_psd);
void launch_dphi_from_psd(hipStream_t s, int N, int ndir, const double* d_psd, void* d_Cm, double scale,
                          void* d_D0t, bool f64out, const void* d_tw64);
// Series form of stage A (stage_a2.hip).  d_coef: [N/2+1][N][series_terms] structure functions of the
// terms of the expansion of the fitting PSD in 1/L0^2 about series_eps0() (launch_series_coef from the
// fp64 planes [terms][N/2+1][N] that launch_colfft_dphi produced for the basis tasks);
// launch_patch: d_P [ntd][80][80], d_T [ntd][N/2+1][80] complex, d_sp [ntd];
// launch_dphi_series: D0t as launch_colfft_dphi writes it.  Valid for 1/L0^2 <= series_eps_max().
// d_twk: the per-lane twiddle table of launch_series_twiddles (series_twiddle_bytes), which takes the
// place of d_tw64 in launch_patch / launch_dphi_series
size_t series_twiddle_bytes(int N);
void launch_series_twiddles(hipStream_t s, int N, const void* d_tw64, void* d_twk);
int series_terms(bool f64);
double series_eps0();
inline double series_eps_max() { return 1.0 / 49.0; }
void launch_series_coef(hipStream_t s, int N, const double* d_planes, void* d_coef, bool f64);
// Round 6: the queue-fed form of launch_dphi_series (K_DPHI_SERIES_Q) and the lines it may skip.  queue: one int the
// launch in front (launch_patch: PatchExtras::queue_zero) sets to zero; perm [ntd]: the order in which the lines of
// a y are dealt (nullptr: as they come); tlmax [N/2+1] (launch_tel_linemax) + c2max (log2(e) c of the longest
// wavelength) + thr_elem / thr_mass (log2): see SeriesSkip in stage_a2.hip; tlmax = nullptr: nothing is skipped.
struct SeriesQueue {
    int* queue = nullptr;
    const int* perm = nullptr;
    const float* tlmax = nullptr;
    float c2max = 0.f, thr_elem = -3.0e38f, thr_mass = -3.0e38f;
};
// What else rides in the two launches of launch_patch (round 6; all optional):
//  * the call's parameter blob: blob_src (pinned host) -> blob_dst (device) as extra workgroups of K_PATCH_GEN, which
//    then reads its tasks from `tp_host` (the TaskPar array INSIDE the pinned blob); `flag`: the pinned word the
//    first workgroup of K_PATCH_ROWS sets to `seq` -- the blob may be refilled (may be NULL);
//  * the spectra of `khat_n` tip-tilt Moffat kernels (launch_khat) as extra workgroups of K_PATCH_ROWS.
struct PatchExtras {
    const void* blob_src = nullptr;
    void* blob_dst = nullptr;
    size_t blob_bytes = 0;
    const TaskPar* tp_host = nullptr;
    unsigned long long* flag = nullptr;
    unsigned long long seq = 0;
    int* queue_zero = nullptr;
    int khat_n = 0;
    const double* khat_gam = nullptr;
    const double* khat_alp = nullptr;
    void* khat_out = nullptr;
    bool khat_f64 = false;
};
void launch_patch(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp, const double* d_aotab,
                  double cfit, const void* d_twk, double* d_P, void* d_T, double* d_sp, bool f64,
                  const PatchExtras& x = PatchExtras());
void launch_dphi_series(hipStream_t s, int N, int ntd, int ndir, const TaskPar* d_tp, const void* d_T,
                        const double* d_sp, const void* d_coef, const void* d_twk, double scale2,
                        void* d_D0t, float* d_dlin, bool f64out, int* d_zero, int ncu,
                        const unsigned* d_support = nullptr, const SeriesQueue& qx = SeriesQueue());
// d_support [N/2+1]: per line, the pieces of series_lanes(N) columns inside the support of the telescope OTF
// (launch_series_support from d_tel, once per context); launch_dphi_series neither evaluates nor stores the others
void launch_series_support(hipStream_t s, int N, const void* d_tel, bool f64, unsigned* d_support);
// d_dlin: [ntd][N/2+1][N/32] minima of max(D, 0) per line and block of 32 columns, written by
// launch_dphi_series when not nullptr; launch_dmin16 turns them into what launch_dmin computes from D
void launch_dmin16(hipStream_t s, int N, int ntd, const float* d_dlin, float* d_dline, float* d_dblk);
void launch_tel_linemax(hipStream_t s, int N, const void* d_tel, float* d_tlmax, bool f64 = false);
void launch_vkeep(hipStream_t s, int N, int ntask, int ndir, int nl, const LamPar* d_lp,
                  const float* d_dline, const float* d_dblk, const float* d_tlmax, float thr_sum,
                  int* d_vkeep, int fixed, float* d_dminb);
// d_order: [ntask] the tasks by descending work (dispatch order of launch_otf_mfma)
void launch_task_order(hipStream_t s, int ntask, int nl, const int* d_vkeep, int* d_order);
// Per-wavelength stage of the mixed-precision path on the matrix cores (otf_mfma.hip): operand
// tables per wavelength set, constant telescope tables, and the fused kernel D -> stamps.
size_t mf_etab_bytes(int N, int nl);
size_t mf_gtab_bytes(int N, int nl);
size_t mf_tl2_bytes(int N);
size_t mf_tlb_bytes(int N);
size_t mf_dminb_bytes(int N, int ntask);
int mf_block_count(int N);
void launch_mf_tables(hipStream_t s, int N, int nl, const LamPar* d_lp, const void* d_tw64,
                      void* d_E, void* d_G);
void launch_mf_tel(hipStream_t s, int N, const void* d_tel, float* d_tl2, float* d_tlb);
// d_vkeep / d_dminb: nullptr = no line / block pruning; thr: log2 of the block threshold;
// d_order: dispatch order of the tasks (or nullptr)
// K_PEAK_FLOOR: per-task floor of launch_otf_mfma's block rule under the tier budget (d_thrf [ntask])
void launch_peak_floor(hipStream_t s, int N, int ntask, int ndir, const void* d_D0t, const float* d_tl2, float c2min,
                       float tier_half, float floor_nominal, float* d_thrf);
void launch_otf_mfma(hipStream_t s, int N, int ntask, int ndir, int nl, const void* d_D0t,
                     const float* d_tl2, const LamPar* d_lp, const void* d_E, const void* d_G,
                     const int* d_vkeep, const float* d_dminb, const float* d_tlb, float thr,
                     void* d_pre, const int* d_order = nullptr, void* d_clk = nullptr, const float* d_thrf = nullptr);
// Second generation of the matrix-core stage (otf_mfma2.hip, one direction): block masks per
// (task, wavelength) in two precision tiers, then the thin-wave kernel.  permax: wavelengths per
// workgroup (7: 14 waves of 128 registers; 6: 12 waves of 168).
size_t mf2_own_bytes(int N, int ntask, int nl);
size_t mf2_uni_bytes(int N, int ntask, int nl);
size_t mf2_sched_bytes(int N, int ntask, int nl, int permax);
size_t mf2_part_bytes(int N, int ntask, int nl);
void mf2_groups(int nl, int permax, int* per, int* ngr);
// K_MF_PREP: block masks and the work lists of launch_otf_mfma2 from the block minima of launch_dmin
// (d_dminb = nullptr: no pruning); d_sched[0..16] must be zero (launch_colfft_dphi does that)
// thr: the eps rule; thr_floor / thr_mid: the precision tiers, applied per (task, wavelength) as far as the OTF
// mass each leaves out stays below tier_half of a lower bound of the PSF peak (d_D0t, d_tl2; tier_half <= 0: no budget)
void launch_mf_prep(hipStream_t s, int N, int ntask, int nl, int permax, const LamPar* d_lp,
                    const float* d_dminb, const float* d_tlb, float thr, float thr_floor, float thr_mid,
                    float tier_half, const void* d_D0t, const float* d_tl2, void* d_own,
                    void* d_uni, void* d_sched, const float* d_dlin = nullptr);
// (d_dlin: instead of d_dminb, the per-line block minima of launch_dphi_series -- the minimum over a
// block's 16 lines is then taken by the kernel itself and launch_dmin16 is not needed)
// K_OTF_MFMA2 (persistent, ncu workgroups) + K_MF_FINISH
void launch_otf_mfma2(hipStream_t s, int N, int ntask, int nl, int permax, int ncu, const void* d_D0t,
                      const float* d_tl2, const LamPar* d_lp, const void* d_E, const void* d_G,
                      const void* d_own, const void* d_uni, void* d_sched, void* d_part, void* d_pre,
                      void* d_clk = nullptr, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                      bool finish = true);
// finish = false: K_MF_FINISH is not launched -- the stamps of the (task, wavelength group)s that ran as several sweeps
// stay partial tiles, which K_CONV_FFT adds up on its way (round 6: one launch less between the per-wavelength stage
// and the convolutions).  MfFinishArgs tells it where they lie; launch_mf_finish completes `pre` for other readers.
struct MfFinishArgs {
    const int* gsw = nullptr;      // [ntask][ngr] sweep masks; nullptr: `pre` is complete
    const void* part = nullptr;    // partial tiles
    int per = 1, ngr = 1, nsw = 1;
};
MfFinishArgs mf2_finish_args(int N, int ntask, int nl, int permax, void* d_sched, const void* d_part);
void launch_mf_finish(hipStream_t s, int N, int ntask, int nl, int permax, void* d_sched, const void* d_part, void* d_pre);
void launch_gtable(hipStream_t s, int N, int nl, const LamPar* d_lp, const void* d_tw64,
                   int* d_samp_p, void* d_samp_a, void* d_G, bool f64);
void launch_moffat_kernels(hipStream_t s, int nker, const double* d_gamma, const double* d_alpha,
                           void* d_out, bool f64);
// d_xtab: extraction table of the radix-16 plans (launch_xtab), used when otf_uses_r16()
void launch_otf_rowfft(hipStream_t s, int N, int ntask, int ndir, int nl, const void* d_D0t,
                       const void* d_tel, const LamPar* d_lp, const int* d_samp_p,
                       const void* d_samp_a, const void* d_xtab, void* d_Tq, const void* d_tw64,
                       bool f64, bool fast_exp, const int* d_vkeep);
bool otf_uses_r16(int N, bool f64, int nl, int ndir);
size_t xtab_bytes(int nl);
void launch_xtab(hipStream_t s, int N, int nl, const int* d_samp_p, const void* d_samp_a,
                 const void* d_tw64, void* d_xtab);
// d_pre: stamps before the convolutions, float (mixed) or double (f64)
// d_vkeep: [ntask][(nl+1)/2] lines to read per (task, wavelength pair), or nullptr for all
void launch_colpass(hipStream_t s, int N, int ntask, int nl, const void* d_Tq, const void* d_G,
                    void* d_pre, bool f64, const int* d_vkeep);
void launch_conv(hipStream_t s, int ntask, int nl, const void* d_pre, const void* d_ktt,
                 const void* d_kmuse, double* d_fin, bool f64);
void launch_khat(hipStream_t s, int nker, const double* d_gamma, const double* d_alpha,
                 void* d_khat, bool f64 = false);
// fin_f32 / stamps_f32: the final stamps are float (inside the pipeline) instead of double
// f64: double stamps in and out, fp64 transforms, khat tables of complex double
void launch_conv_fft(hipStream_t s, int ntask, int nl, const void* d_pre, const void* d_khat_tt,
                     const void* d_khat_muse, void* d_fin, bool fin_f32, bool f64 = false,
                     const MfFinishArgs& finish = MfFinishArgs());
// (sum_*: the deterministic sum of the stamps over the sum_ntask tasks of the chunk -- [sum_nl][40][40] into d_sum,
// added to it if sum_accumulate -- as the first workgroups of the same launch; d_sum = nullptr: none)
void launch_fit(hipStream_t s, int nstamp, const void* d_stamps, bool stamps_f32, double* d_fit,
                bool f64, int sum_ntask = 0, int sum_nl = 0, double* d_sum = nullptr, int sum_accumulate = 0);
void launch_stamp_sum(hipStream_t s, int ntask, int nl, const void* d_fin, bool fin_f32, double* d_sum,
                      int accumulate);
// the call's parameter blob from pinned host memory into device memory, as a kernel of the call's own queue
// (bytes: a multiple of 16); h_flag_pinned: a pinned host word that receives `seq` once the blob has been read
void launch_param_copy(hipStream_t s, void* d_dst, const void* h_src_pinned, size_t bytes,
                       unsigned long long* h_flag_pinned = nullptr, unsigned long long seq = 0);

}  // namespace mpsfr
