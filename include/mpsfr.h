/* libmpsfr -- C ABI of the MI355X-native PSF-reconstruction hot path of muse-psfr.
 *
 * The reference (musevlt/muse-psfr) has no FFI: its boundary for this path is two Python
 * functions.  This header is what a ctypes binding of those functions binds to; every entry
 * point cites the reference interface it replaces (file:line in /root/reference/muse_psfr/).
 *
 * Conventions: plain pointers and sizes only; all arrays C-contiguous; every function returns 0
 * on success or a negative error code (MPSFR_E_*), with a thread-local message available from
 * mpsfr_last_error(); nothing is allocated across the ABI; no exceptions cross it.  A context
 * owns one GPU (one process per GPU, one context per process is the intended use), its HIP
 * streams and all device workspaces; it is not re-entrant.
 *
 * Asynchronous calls (on_device = 1) are pipelined inside the context: consecutive calls run on
 * alternating internal streams ("lanes") with their own workspaces, so the transforms of one
 * call overlap the convolutions and fits of the one before.  What a caller may rely on:
 *   - mpsfr_stream() is ordered after every call made so far (wait on it, or mpsfr_sync, before
 *     reading results);
 *   - two calls that write the same output buffer run in call order;
 *   - calls with distinct output buffers may run concurrently and finish in any order;
 *   - to make the NEXT call wait for the caller's own GPU work, register an event with
 *     mpsfr_wait_event (work queued on mpsfr_stream() does not hold back later calls).
 * Option "pipeline_calls" = 0 restores strictly serial calls.
 */
#ifndef MPSFR_H
#define MPSFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpsfr_ctx mpsfr_ctx;

#define MPSFR_OK            0
#define MPSFR_E_INVALID    -1   /* bad argument (message says which) */
#define MPSFR_E_HIP        -2   /* a HIP runtime call failed */
#define MPSFR_E_GRID       -3   /* wavelength too short for the grid: npixc(lambda) > dim.  The
                                   reference raises ValueError from interpn here
                                   (psfrec.py:663-664, 672-683) */
#define MPSFR_E_NOMEM      -4

/* precision modes */
#define MPSFR_PREC_MIXED    0   /* fp64 PSD -> structure function, fp32 per-wavelength stage,
                                   fp64 Moffat fit */
#define MPSFR_PREC_F64      1   /* fp64 everywhere (reference arithmetic type) */

/* number of doubles per fitted stamp in fit_out */
#define MPSFR_NFIT         16
/* fit_out[k]: 0 peak  1 p0 (row centre, px)  2 q0 (col centre, px)  3 alpha (px)  4 n (beta)
 *             5 fwhm (px) = 2 alpha sqrt(2^(1/n) - 1)   6 chi2   7 iterations
 *             8 err_peak  9 err_p0  10 err_q0  11 err_alpha  12 err_n  13 err_fwhm (px)
 *             14 status (0 converged, 1 iteration cap, 2 singular; + MPSFR_FIT_ILL_CONDITIONED: see below)
 *             15 flux = peak pi alpha^2/(n-1)
 * Status bit MPSFR_FIT_ILL_CONDITIONED (4): the least-squares minimum was found, but the stamp does not pin
 * (fwhm, n) to the parity tolerance: n^2 sqrt((J^T J)^-1[eta, eta]) * peak >= 100, i.e. iid pixel noise of 1e-6 of
 * the peak moves n by 1e-4 or more (the covariance is the one err_n comes from).  That is the case where the stamp
 * is narrower than the PSF core (the 128^2 / 256^2 grids with the rescaled pixel scale, seeing > 2 arcsec at 512^2);
 * on the 512^2 ... 1280^2 grids with the SPARTA range of inputs the number stays below 30.
 */
#define MPSFR_FIT_ILL_CONDITIONED 4

/* Side of the AO-corrected zone grid (psfrec.py:103, 138: Dimpup * 2). */
#define MPSFR_DIM_AO       80

/* Create a context on HIP device `device_id` for an N x N spatial-frequency grid.
 * dim      : N, the `dim` argument of simul_psd_wfm (psfrec.py:37; compute_psf hard-codes 1280,
 *            psfrec.py:954-955).  Supported: 128, 256, 512, 1024, 1280.
 * dimpsf   : side of the output stamps (psfrec.py:658); only 40 is supported.
 * pixscale : arcsec per output pixel (psfrec.py:659, :899, :868).
 * Builds the wavelength- and row-independent telescope OTF (psfrec.py:784-790) once. */
int mpsfr_create(mpsfr_ctx** out, int device_id, int dim, int dimpsf, double pixscale,
                 int precision);
void mpsfr_destroy(mpsfr_ctx* ctx);
const char* mpsfr_last_error(void);

/* Tunables: "chunk_tasks" (tasks per pipeline pass, 0 = automatic: one pass up to 512 tasks /
 * 65536 stamps / 4 GiB of workspace, balanced passes beyond; a synchronous call -- host outputs --
 * of 8192 stamps or more: one pass per lane); "fast_exp" (mixed mode only,
 * default 1: hardware exp2 for the OTF); "fft_conv" (mixed mode only, default 1: the two 41x41
 * convolutions through 64-point FFTs instead of the direct form); "cu_partition" (default 0; 1: every lane's stream is created with a CU mask and owns
 * 1/lanes of the compute units -- measured slower, profiles/r05_experiments.md); "streams" (0 = automatic = 2,
 * or 1..4 pipeline lanes: consecutive chunks -- of one call and of consecutive asynchronous
 * calls -- go to successive HIP streams with their own workspaces so that one chunk's tail
 * overlaps the next one's body; results are independent of it except for the summation order of
 * psf_sum_out in multi-chunk calls); "pipeline_calls" (default 1: asynchronous calls rotate over
 * the lanes; 0: every call starts on the first lane); "prune_eps" (mixed mode only, default 1e-9:
 * the parts of the OTF half plane -- trailing lines, and 16 x 32 blocks inside the lines kept --
 * whose elements together weigh less than eps of OTF[0][0] (<= eps of the PSF peak, the sum of the OTF, whose
 * elements are all >= 0) are neither generated nor summed; 0 = everything); "tier_eps" (mixed mode, matrix-core
 * stage only, default 4e-6: the two precision tiers of that stage -- blocks whose largest element is below
 * 2^-29 of OTF[0][0] are dropped, blocks below 2^-18 run without the low fp16 half of the OTF -- are applied,
 * per task and wavelength, only as far as the OTF mass each of them leaves out stays below tier_eps / 2 of a
 * lower bound of the PSF peak (the OTF summed exactly over its first four lines; two on grids above 512^2); where it would not, the
 * thresholds of that (task, wavelength) are lowered until it does; the kernel for several directions has the floor
 * tier only and takes one floor per task, from the OTF of the task's shortest wavelength and the number of blocks.
 * 0 = no tiers; inf = tiers without a budget).
 * WHAT THE TWO TOGETHER GUARANTEE, for every input: with eps = prune_eps + tier_eps no pixel of a stamp before
 * the convolutions (psf_muse, psfrec.py:644-686) moves by more than eps of that stamp's peak -- up to the
 * stamp's normalisation to unit sum (psfrec.py:685), which in the worst case (all 1600 pixels moved the same
 * way) rescales the stamp by 1600 eps peak / sum; the Moffat fwhm, n and centre do not depend on the
 * normalisation, and the two convolutions (non-negative kernels of unit sum) do not increase a difference.
 * What the approximations really do on the workloads measured (tests/test_gpu_parity.py::
 * test_precision_tiers_of_the_matrix_core_stage): <= 3e-7 of the peak, |d beta| <= 2e-6, and the budget
 * is not reached (it is a guarantee for inputs nobody measured, e.g. an OTF whose coherent plateau sits just
 * below the floor); "otf_mfma" (mixed mode only,
 * default 1: the per-wavelength stage as split-fp16 contractions on the matrix cores; 0: LDS
 * FFTs on the vector pipe); "profile" (0/1: bracket every kernel launch
 * with HIP events on the stream it is launched on -- the event packets cost ~8 % of a step);
 * "profile_only" (-1 = all kernels, else the kernel id of mpsfr_profile_name to time alone);
 * "prune_eps_f64" (f64 mode only, default 1e-13, at most 1e-6: the same bound for the line pruning
 * of the reference-precision mode; 0 = everything).
 * "stage_a" (default 1 = automatic: from 512^2 on (at 256^2 with several directions) the structure function of a task is the sum of a
 * per-pixel polynomial in 1/L0^2 -- the fitting term, from tables built once per context -- and a pruned
 * transform of the 80 x 80 corrected zone (stage_a2.hip); below, and for any call with L0 < 7 m, the
 * full-size fp64 transforms of the PSD (stage_a.hip); 0 = always the full-size transforms, 2 = the
 * series form on every grid; the two forms agree to the rounding of the stored structure function).
 * Round 6, where a call's work is queued (results do not depend on any of these, bit for bit):
 * "persist_reserve" / "persist_reserve_mf" / "persist_reserve_a" (default -1 = automatic; 0..128): the two persistent
 * kernels (the matrix-core per-wavelength kernel / the column kernel of stage A's series form) launch ncu - R
 * workgroups instead of one per CU, so that R CUs -- R / 8 per XCD -- stay free for the latency-bound kernels of the
 * context's other lane while the big kernel runs; automatic: R = ncu / 8 when the call has chunks on several lanes
 * or the other lane still has work in flight, 0 for a call that runs alone (one lane loses 5 % to a reserve of 32,
 * two lanes gain 5 %: profiles/r06_experiments.md);
 * "head_fusion" (default 1: in the series form of stage A the spectra of a chunk's tip-tilt Moffat kernels are
 * computed by trailing workgroups of the patch's row kernel; 0: by a kernel of their own at the head of the call);
 * "copy_fusion" (default 0; 1: the parameter blob is fetched by workgroups of the call's first kernel, which reads
 * its tasks straight from pinned host memory -- measured, no gain); "finish_fusion" (default 0; 1: the FFT
 * convolution kernel adds up the partial tiles of the stamps the matrix-core kernel split into sweeps instead of
 * a kernel between the two -- measured, -1 %); "support_skip" (default 1: the series form of stage A neither
 * evaluates nor stores the structure function on the pieces of a line where the telescope OTF is identically zero
 * -- a fifth of the half plane; the buffer keeps the zero it was allocated with there).
 * "stage_a_queue" (default 0; 1: the lines of stage A's series form dealt in blocks from a queue instead of equal
 * contiguous shares -- bit-identical, 13 % slower; 2: and the lines of a task on which a lower bound of the
 * structure function from the patch's row transforms puts the whole OTF line below the eps rule of "prune_eps", or
 * its mass below tier_eps / (8 (N/2+1)), at the longest wavelength, are skipped: 28 % of the lines on the bench
 * rows, stamps within 1e-7 -- and 2 % of the kernel's time: measured, lost, profiles/r06_experiments.md).
 * "cold_stagger" (default 0 = off; 1 / 2: after the GPU has drained, the second lane's first chunk
 * waits once for the first lane's column transforms / per-wavelength preparation, so that the two
 * lanes do not start in step: +3 % in a sustained run of 100-row calls, -1 % on a burst of 20;
 * results do not depend on it).
 * Experiment switches of the matrix-core stage (results depend on them within the bound of "tier_eps"): "mf_kernel" (2 = thin-wave kernel with precision tiers
 * for one direction, 1 = the blocked kernel that several directions always use), "mf_permax"
 * (1..7 wavelengths per workgroup, default 6), "mf_floor" (default 1: blocks below the fp16
 * representation floor are skipped, within "tier_eps"), "mf_mid_log2" (default -18.01: blocks below 2^this of
 * OTF[0][0] run without the low half of the OTF, within "tier_eps"), "mf_clock" (phase time stamps; builds with
 * -DMPSFR_MF_CLOCK=1 only), "prune_fixed" (a fixed number of lines for the FFT form). */
int mpsfr_set_option(mpsfr_ctx* ctx, const char* key, double value);

/* Batched replacement of  Parallel(n_jobs)(delayed(compute_psf)(*args) ...)  (psfrec.py:1082-1083)
 * i.e. of compute_psf (psfrec.py:933-978) = simul_psd_wfm (:36-151) -> psf_muse (:644-686) ->
 * convolve_final_psf (:874-930) -> fit_psf_cube (:861-871), for ntask (seeing, GL, L0) triples.
 *
 * seeing, gl, l0 : [ntask] arcsec @500 nm, ground-layer fraction, outer scale [m]
 * three_lgs      : [ntask] 0/1, three_lgs_mode of simul_psd_wfm (psfrec.py:86-91)
 * h              : layer altitudes [m] (psfrec.py:60); exactly two layers (psfrec.py:66 fixes two
 *                  wind directions)
 * wind_speed     : m/s; the reference uses np.full_like(h, 12.5) = 12 for integer h, 12.5 for
 *                  float h (psfrec.py:61) -- the caller decides
 * npsflin        : linear number of evaluation directions (psfrec.py:154-158), 1..5
 * lbda_nm        : [nl] wavelengths in nm
 * mask_rec/res   : [80*80] 0/1 cut-off masks of psfrec.py:257 (>=) and :435 (>) indexed
 *                  [i_fx][j_fy] like the reference's arrays, or NULL for the exact rule
 *                  |k| >= 24 / |k| > 24 on the integer frequency grid.  (In the reference these
 *                  masks depend on last-bit libm rounding; see DESIGN.md "cut-off masks".)
 *                  NOTE: NULL is NOT what the Python API passes by default.  muse_psfr_amd.compute_psf /
 *                  compute_psf_from_sparta default to cutoff_masks='host' -- the masks as the caller's NumPy
 *                  evaluates psfrec.py:257/:435, i.e. what the reference itself would compute on that
 *                  machine -- which differs from the exact rule on 44-52 of the 160 boundary pixels and
 *                  moves beta by up to 3e-3 (30x the parity tolerance).  A C caller that wants the results
 *                  of the Python default (or of a given reference installation) passes that installation's
 *                  masks; cutoff_masks='exact' in Python is this NULL.
 * psf_out        : [ntask][nl][dimpsf][dimpsf] final stamps (after both convolutions), or NULL
 * psf_sum_out    : [nl][dimpsf][dimpsf] sum over the ntask stamps (the caller divides by the
 *                  global task count to get PSF_MEAN, psfrec.py:1104), or NULL
 * fit_out        : [ntask][nl][MPSFR_NFIT] Moffat fit of every stamp, or NULL
 * on_device      : 0 = the three outputs are host pointers, the call returns with the results;
 *                  1 = device pointers on this context's device (results complete after mpsfr_sync);
 *                  2 = host pointers, asynchronous: the call returns once its work is queued (like
 *                  on_device = 1 it takes its turn on the pipeline lanes, so consecutive calls
 *                  overlap), the results travel to a pinned staging set of the library and reach the
 *                  caller's arrays in mpsfr_wait(ctx, mpsfr_last_ticket(ctx)) -- or in mpsfr_sync, or
 *                  when the fourth asynchronous call after this one is made (the ring of staging
 *                  sets has four); the arrays must stay allocated until then
 * All outputs are float64.  The call is asynchronous when on_device != 0. */
int mpsfr_reconstruct(mpsfr_ctx* ctx, int ntask, const double* seeing, const double* gl,
                      const double* l0, const uint8_t* three_lgs, const double h[2],
                      double wind_speed, int npsflin, int nl, const double* lbda_nm,
                      const uint8_t* mask_rec, const uint8_t* mask_res, double* psf_out,
                      double* psf_sum_out, double* fit_out, int on_device);

/* The same over several devices: the reference's  Parallel(n_jobs=...)  fans the rows out over
 * worker processes (psfrec.py:1082-1083); here the rows go in contiguous, balanced shards (the
 * first ntask % nctx contexts take one row more) to `nctx` contexts -- normally one per device,
 * created by the caller with the same dim, dimpsf, pixscale and precision -- one host thread per
 * context.  Host buffers only, synchronous.  Per-row outputs are those of the single-context
 * call; psf_sum_out adds the shards' sums in context order.  ntask < nctx: the first context
 * takes everything. */
int mpsfr_reconstruct_multi(mpsfr_ctx* const* ctxs, int nctx, int ntask, const double* seeing,
                            const double* gl, const double* l0, const uint8_t* three_lgs,
                            const double h[2], double wind_speed, int npsflin, int nl,
                            const double* lbda_nm, const uint8_t* mask_rec,
                            const uint8_t* mask_res, double* psf_out, double* psf_sum_out,
                            double* fit_out);

/* The same without waiting for the GPUs: every context takes its shard as an asynchronous host-output call
 * (on_device = 2 of mpsfr_reconstruct: no host thread, the call returns once the shards are queued), and
 * mpsfr_wait_multi(ctxs, nctx) -- same contexts, same order -- blocks until all of them have finished, hands
 * the per-row outputs to the caller's arrays and adds the shards' stamp sums in context order into the
 * psf_sum_out of the call.  One such call may be pending per ctxs[0]; the output arrays must stay allocated
 * until mpsfr_wait_multi returns (or mpsfr_abandon on every context gives them up).  Several tables -- or the
 * parts of one -- can so be kept in flight on several devices, like on_device = 2 does on one.
 * A shard that fails (the error names its context) ABANDONS every pending asynchronous call of EVERY context of
 * the call -- the shards already queued and any single on_device = 2 call still in flight on them: their arrays
 * are never written, mpsfr_wait on their tickets fails.  The refusals made before anything is queued (NULL
 * context, a multi-context call already pending on ctxs[0], contexts of different shape) touch nothing. */
int mpsfr_reconstruct_multi_async(mpsfr_ctx* const* ctxs, int nctx, int ntask, const double* seeing,
                            const double* gl, const double* l0, const uint8_t* three_lgs,
                            const double h[2], double wind_speed, int npsflin, int nl,
                            const double* lbda_nm, const uint8_t* mask_rec,
                            const uint8_t* mask_res, double* psf_out, double* psf_sum_out,
                            double* fit_out);
int mpsfr_wait_multi(mpsfr_ctx* const* ctxs, int nctx);

/* Replacement of fit_psf_cube (psfrec.py:861-871) on caller-provided stamps, e.g. the mean PSF
 * (psfrec.py:1105).  stamps: [nstamp][dimpsf][dimpsf] float64; fit_out: [nstamp][MPSFR_NFIT]. */
int mpsfr_fit_stamps(mpsfr_ctx* ctx, int nstamp, const double* stamps, double* fit_out,
                     int on_device);

/* The stages of the path as the reference exports them (muse_psfr/__init__.py:16: `from .psfrec import *`),
 * for a caller that holds its own PSD or its own stamps.  Host buffers, synchronous, float64; the same
 * kernels as mpsfr_reconstruct entered or left at another stage.
 *
 * mpsfr_simul_psd      simul_psd_wfm (psfrec.py:36-151) for one (seeing, GL, L0): psd_out [npsflin^2][dim][dim],
 *                      centred (DC at [dim/2][dim/2]) and in the reference's units (times (0.5 1000 / 2 pi)^2,
 *                      psfrec.py:151).  Arguments as for mpsfr_reconstruct.
 * mpsfr_psf_from_psd   psf_muse (psfrec.py:644-686): psd [ndir][dim][dim] as above -- ANY real image, not only
 *                      the model's: every row is transformed -- to psf_out [nl][dimpsf][dimpsf], the stamps
 *                      BEFORE the convolutions (mean over the ndir directions, normalised to sum 1).
 * mpsfr_convolve_stamps  convolve_final_psf (psfrec.py:874-930): psf_in [ntask][nl][dimpsf][dimpsf] convolved
 *                      with each task's tip-tilt Moffat kernel and the instrument's, into psf_out. */
int mpsfr_simul_psd(mpsfr_ctx* ctx, double seeing, double gl, double l0, int three_lgs, const double h[2],
                    double wind_speed, int npsflin, const uint8_t* mask_rec, const uint8_t* mask_res,
                    double* psd_out);
int mpsfr_psf_from_psd(mpsfr_ctx* ctx, int ndir, const double* psd, int nl, const double* lbda_nm,
                       double* psf_out);
int mpsfr_convolve_stamps(mpsfr_ctx* ctx, int ntask, const double* seeing, const double* gl, const double* l0,
                          int nl, const double* lbda_nm, const double* psf_in, double* psf_out);

/* FIT_ROWS assembly on the host (pure C, no GPU): the columns fit_psf_cube keeps from the fit object
 * (psfrec.py:866-870) -- center[2], flux, fwhm[2] (arcsec), n, peak, err_center[2], err_flux, err_fwhm[2] (arcsec),
 * err_n, err_peak: 14 doubles -- of `n` fit rows ([n][MPSFR_NFIT], as mpsfr_reconstruct writes them) into
 * out[r * stride + 0..13].  `stride` (in doubles, >= 14) lets the caller write straight into the records of a
 * table that carries further columns (compute_psf_from_sparta's FIT_ROWS: lbda in front, SEEING, GL, L0, row_idx,
 * lgs_idx behind, psfrec.py:1086-1101).  err_flux: the relative errors of peak, alpha^2 and (n - 1) in quadrature. */
int mpsfr_fit_rows(const double* fit, long n, double pixscale, double* out, long stride);

/* Block until every call made so far has finished (and hand over the results of every
 * asynchronous host-output call). */
int mpsfr_sync(mpsfr_ctx* ctx);

/* Asynchronous host outputs (on_device = 2).  mpsfr_last_ticket: the ticket (0, 1, 2, ...) of the most
 * recent such call of the context, -1 before the first.  mpsfr_wait: block until the call with that
 * ticket has finished and copy its results (and those of every earlier ticket not yet handed over)
 * into the arrays it was given.  This is how the rows of a table larger than one call are kept in
 * flight (the reference: Parallel(...)(delayed(compute_psf) ...), psfrec.py:1082-1083, results
 * collected at :1086-1113). */
long mpsfr_last_ticket(mpsfr_ctx* ctx);
int mpsfr_wait(mpsfr_ctx* ctx, long ticket);

/* Give up every asynchronous host-output call that has not been handed over yet: wait for the GPU to
 * drain, then forget the callers' arrays WITHOUT writing to them.  The error path of a caller whose output
 * arrays are about to go away (an exception between the call and its mpsfr_wait): after this, no later
 * mpsfr_wait / mpsfr_sync / mpsfr_reconstruct of the context touches those arrays.  The abandoned tickets
 * count as completed; the context stays usable. */
int mpsfr_abandon(mpsfr_ctx* ctx);

/* The context's hipStream_t (as void*): it is ordered after every asynchronous (on_device = 1)
 * call made so far, so a caller can queue its own GPU work behind the results without a host
 * sync, e.g. torch.cuda.current_stream().wait_stream(torch.cuda.ExternalStream(...)). */
void* mpsfr_stream(mpsfr_ctx* ctx);

/* Make `caller_stream` (a hipStream_t of the caller, as void*) wait, on the GPU, for every asynchronous call made so far
 * on this context: the cheap form of "wait on mpsfr_stream()" -- the call's work stays on its lane and one event at the
 * lane's end is all that is queued (asking for mpsfr_stream() makes every later call join its lanes into a further
 * queue).  E.g. before a collective that reads the call's device outputs:
 *   mpsfr_stream_wait(ctx, (void*)torch.cuda.current_stream().cuda_stream). */
int mpsfr_stream_wait(mpsfr_ctx* ctx, void* caller_stream);

/* Make the next mpsfr_reconstruct wait (on the GPU, no host sync) for `hip_event`, a recorded
 * hipEvent_t of the caller, e.g. the end of a collective that still reads the buffers the call
 * will overwrite.  One-shot: consumed by the next call. */
int mpsfr_wait_event(mpsfr_ctx* ctx, void* hip_event);

/* Host wall time spent inside mpsfr_reconstruct since the last mpsfr_profile_reset, and the
 * number of calls: what queueing a call costs the host thread.  Time spent blocked because the
 * host ran four calls ahead of the GPU (the ring of parameter blobs) is not counted. */
int mpsfr_host_time(mpsfr_ctx* ctx, double* seconds, long* calls);

/* Copy an intermediate of the most recent mpsfr_reconstruct pipeline pass (last chunk) to the
 * host as float64 (parity tests, tests/test_gpu_parity.py).  `what`:
 *   "ao_tables"  [2 geometries][ndir][3 (T0,T1,noise)][80][80]   (psfrec.py:531-613)
 *   "tel"        [dim/2+1][dim]  telescope OTF, transposed half plane (psfrec.py:784-790)
 *   "dphi0"      [chunk tasks][ndir][dim/2+1][dim] structure function / lambda-factor,
 *                transposed half plane (psfrec.py:717-722)
 *   "pre"        [chunk tasks][nl][dimpsf][dimpsf] stamps before the convolutions (psfrec.py:685)
 *   "vkeep"      [chunk tasks][(nl+1)/2] lines of the half plane transformed per wavelength pair
 *                (option "prune_eps")
 *   "mf_work"    [7] (the first 3 / 5 if capacity < 5 / 7) matrix-core stage, last chunk: tile steps executed,
 *                m-tiles with a second pass, tile steps without pruning, tile steps with all three
 *                products, tile steps without the low half of the OTF, blocks kept by at least one
 *                wavelength summed over the tasks, blocks inside the support of the telescope OTF x tasks
 * Returns the number of doubles written (<= capacity) or a negative error. */
long mpsfr_debug_fetch(mpsfr_ctx* ctx, const char* what, double* out, size_t capacity);

/* Per-kernel timing with HIP events recorded on the context's stream (option "profile" = 1). */
int mpsfr_profile_count(void);
const char* mpsfr_profile_name(int kernel_id);
int mpsfr_profile_get(mpsfr_ctx* ctx, int kernel_id, double* total_ms, long* launches);
int mpsfr_profile_reset(mpsfr_ctx* ctx);

/* Number of HIP devices visible to the process (0 if there is none or the runtime fails): what the
 * drop-in Python layer fans a large SPARTA table out over, one context and one host thread per
 * device -- the reference's joblib fan-out over worker processes (psfrec.py:1082-1083). */
int mpsfr_device_count(void);

/* Library/ABI version (major*100 + minor). */
int mpsfr_version(void);

/* Hash of the sources, headers and compiler flags this binary was built from (written by
 * muse_psfr_amd/_build.py; "unstamped" for a hand build).  bench.py records it. */
const char* mpsfr_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* MPSFR_H */
