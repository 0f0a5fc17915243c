/*
 * tspws_hip.h -- the thin C-ABI HIP layer under tspws_main().
 *
 * Plain C types only (pointers, sizes, scalars); every `d_` pointer is a HIP
 * device pointer on the plan's device, `stream` is a hipStream_t passed as
 * void* (NULL = the default stream).  All functions return 0 on success or a
 * TSPWS_E_* code; tspws_hip_last_error() gives the text.  Nothing here exists
 * in the reference (it has no device code): each entry point names the
 * reference routine whose work it takes over, paths relative to
 * /root/reference/src.
 *
 * Coefficient containers are the reference's ragged [S][N_s] layout
 * (FWTa/wavelet_mem_v7.c:78-106) flattened: scale s starts at coef_off[s],
 * N_s = ceil(N / D_s); complex values are interleaved (re, im) doubles.
 */
#ifndef TSPWS_HIP_H
#define TSPWS_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "ts_pws1f_lib.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
	TSPWS_OK        = 0,
	TSPWS_E_ARG     = -1, /* NULL / inconsistent argument (tspws_main returns -1 for NULL too)   */
	TSPWS_E_NOMEM   = 4,  /* host or device allocation failed (reference code, ts_pws1f_lib.c:199) */
	TSPWS_E_NODEV   = 5,  /* no usable HIP device or kernel image                               */
	TSPWS_E_HIP     = 6,  /* a HIP runtime call failed                                          */
	TSPWS_E_FRAME   = 7   /* frame cannot be built (bad family id, wavelet longer than trace)   */
};

typedef struct tspws_hip_plan tspws_hip_plan;

typedef struct {
	int      type;          /* -1 / -2 / -3                                     */
	unsigned S, V, J, N;    /* scales, voices, octaves, trace length            */
	double   s0, b0, w0;    /* as passed to plan_create                         */
	double   Cpsi;          /* admissibility constant                           */
	size_t   ncoef;         /* sum of N_s  (complex coefficients per trace)     */
	size_t   ntaps;         /* sum of L_s  (complex taps; the dual has as many) */
	int      device;
} tspws_hip_frame_info;

/* ---- runtime helpers so that the C host needs no HIP headers ------------------ */
int         tspws_hip_device_count(void);
const char *tspws_hip_last_error(void);
int  tspws_hip_alloc(void **d_ptr, size_t bytes, int device);
int  tspws_hip_free(void *d_ptr);
int  tspws_hip_upload(void *d_dst, const void *h_src, size_t bytes, void *stream);
int  tspws_hip_download(void *h_dst, const void *d_src, size_t bytes, void *stream);
int  tspws_hip_zero(void *d_ptr, size_t bytes, void *stream);
int  tspws_hip_sync(void *stream);

/* ---- parameters and plan ------------------------------------------------------ */

/* Resolve w0 / V / b0 / s0 / J from the user's knobs, rewriting *p in place.
 * Takes over ts_pws1f_lib.c:91-124 (host arithmetic, bit-for-bit the same rules). */
void tspws_resolve_params(t_tsPWS *p, unsigned nsamp, float dt);

/* Build the frame on `device`: geometry tables on the host, taps / dual taps generated
 * by a device kernel, container offsets.  Takes over CreateWaveletFamily
 * (FWTa/wavelet_def_v7.c:204-295) and the container set-up (wavelet_mem_v7.c:78-106).
 * Argument meaning and order follow CreateWaveletFamily(type, J, V, N, s0, b0, -, w0, uni). */
int  tspws_hip_plan_create(tspws_hip_plan **plan, int type, unsigned J, unsigned V, unsigned N,
                           double s0, double b0, double w0, int uni, int device);
void tspws_hip_plan_destroy(tspws_hip_plan *plan);   /* DestroyWaveletFamily, wavelet_def_v7.c:341 */
int  tspws_hip_plan_info(const tspws_hip_plan *plan, tspws_hip_frame_info *info);
/* host copies of the per-scale tables; any pointer may be NULL */
int  tspws_hip_plan_tables(const tspws_hip_plan *plan, double *scale, unsigned *L, int *c, int *cd,
                           unsigned *D, unsigned *Ns);
/* device -> host copy of the (re,im) tap tables, 2*ntaps doubles each (tests) */
int  tspws_hip_plan_taps(const tspws_hip_plan *plan, double *h_w, double *h_wd);

/* ---- trace prologue (in place, float) ----------------------------------------- */
/* fold: x[n] = x[max-1-n] = 0.5f*(x[n]+x[max-1-n]).   ts_pws1f_lib.c:76-86 */
int  tspws_hip_fold(float *d_sigall, size_t mtr, size_t max, size_t ld, void *stream);
/* remove each trace's mean (FP64 sum, float subtraction).   ts_pws1f_lib.c:159-169 */
int  tspws_hip_remove_mean(float *d_sigall, size_t mtr, size_t max, size_t ld, void *stream);

/* ---- stage 1 of the two-stage stack -------------------------------------------- */
/* P[g][n] (+)= sum of this shard's traces whose GLOBAL index i = first + local falls in group
 * g = floor(i*Kmax/mtr_global); P is [Kmax][ldP] doubles and is overwritten (rows of groups the
 * shard does not touch become 0).  Takes over partial_linear_stacks, ts_pws1f_lib.c:866-881.
 * This is the HBM-streaming kernel: every input sample is read exactly once. */
int  tspws_hip_partial_stacks(tspws_hip_plan *plan, const float *d_sigall, size_t ld,
                              size_t mtr_local, size_t first, size_t mtr_global, unsigned Kmax,
                              double *d_P, size_t ldP, void *stream);

/* Same, restricted to the groups [g_begin, g_end): only those rows of P are written.  Lets a multi-GPU caller start the
 * all-reduce of the first groups while the later ones are still being streamed. */
int  tspws_hip_partial_stacks_range(tspws_hip_plan *plan, const float *d_sigall, size_t ld,
                                    size_t mtr_local, size_t first, size_t mtr_global, unsigned Kmax,
                                    unsigned g_begin, unsigned g_end, double *d_P, size_t ldP, void *stream);

/* ---- frame transforms ------------------------------------------------------------ */
/* Forward frame CWT of ntr real traces (row stride ld elements) into d_Y[ntr][ncoef] complex.
 * Takes over complex_1D_wavelet_dec (wavelet_v7.c:43-64) -> cdotx_dc (cdotx.c:35-72). */
int  tspws_hip_forward_f64(tspws_hip_plan *plan, const double *d_x, size_t ntr, size_t ld, double *d_Y, void *stream);
int  tspws_hip_forward_f32(tspws_hip_plan *plan, const float  *d_x, size_t ntr, size_t ld, double *d_Y, void *stream);
/* The same coefficients through the SPECTRAL engine (csrc/spectral.hip): only the scales of the frame's spectral set for octaves of at
 * most nsmax outputs -- [tspws_hip_spectral_first_scale(plan, nsmax), tspws_hip_spectral_end_scale(plan)) -- are written, the rest of
 * d_Y[ntr][ncoef] is left alone.  Exact for those scales because the reference's FIR is a circular correlation (cdotx.c:35-72): for N a
 * power of two it is the transform's own (D divides N); for any other N >= 1024 it is evaluated as a linear correlation over a window of
 * the trace's periodic extension, tspws_hip_spectral_transform_length(plan) >= N + L - 1 samples long -- scales whose filters do not fit
 * that window lie at and behind tspws_hip_spectral_end_scale (S for most frames).  Returns TSPWS_E_ARG when the frame has no such set
 * (N < 1024, decimations that are not powers of two >= 8, ...). */
unsigned tspws_hip_spectral_first_scale(const tspws_hip_plan *plan, unsigned nsmax);
unsigned tspws_hip_spectral_end_scale(const tspws_hip_plan *plan);
unsigned tspws_hip_spectral_transform_length(const tspws_hip_plan *plan);
/* First scale of the spectral set a single-stage batch of ntr traces gets by the library's own rule (S: FIR kernels only): batches of
 * >= 64 traces and >= 1 M samples (or >= 256 traces) send the octaves with D >= 32 (two-voice frames: D >= 16) through the spectrum;
 * TSPWS_ENGINE=fir / spectral pins the choice. */
unsigned tspws_hip_spectral_choice(const tspws_hip_plan *plan, size_t ntr);
int  tspws_hip_forward_spectral_f64(tspws_hip_plan *plan, const double *d_x, size_t ntr, size_t ld, double *d_Y, unsigned nsmax, void *stream);
int  tspws_hip_forward_spectral_f32(tspws_hip_plan *plan, const float  *d_x, size_t ntr, size_t ld, double *d_Y, unsigned nsmax, void *stream);
/* Real part of the inverse frame transform of nrec coefficient sets d_Y[nrec][ncoef] into
 * d_x[nrec][N] doubles.  Takes over Re_complex_1D_wavelet_rec (wavelet_v7.c:124-150) ->
 * re_cdotx_upsampling_cc (cdotx.c:305-340) / re_cdotx_cc (cdotx.c:176-211). */
int  tspws_hip_inverse(tspws_hip_plan *plan, const double *d_Y, size_t nrec, double *d_x, void *stream);

/* ---- stacks in the time-scale domain ----------------------------------------------- */
/* ST += sum_b Y_b ; PS += sum_b Y_b/|Y_b| (non-unit quotients skipped); zero_first clears
 * ST/PS before.  The loop body of ts_pws1f_lib.c:486-494 / :897-904. */
int  tspws_hip_accumulate(tspws_hip_plan *plan, const double *d_Y, size_t ntr, double *d_ST, double *d_PS,
                          int zero_first, void *stream);
/* forward + accumulate of K double-precision partial stacks.  tspws_stacks_double, :885-906 */
int  tspws_hip_stacks_double(tspws_hip_plan *plan, const double *d_P, unsigned K, size_t ldP,
                             double *d_ST, double *d_PS, void *stream);
/* forward + accumulate of mtr float traces (ST/PS are cleared first).  tspws_stacks_float, :466-499 */
int  tspws_hip_stacks_float(tspws_hip_plan *plan, const float *d_sigall, size_t mtr, size_t ld,
                            double *d_ST, double *d_PS, void *stream);
/* OUT = ST * weight(PS).  wu==2 && unbiased -> tspws_unbiased (:965-984) else tspws_biased (:909-943) */
int  tspws_hip_weight(tspws_hip_plan *plan, double *d_OUT, const double *d_ST, const double *d_PS,
                      unsigned K, unsigned M, double wu, int unbiased, void *stream);
/* ls[n] = (float)x_st[n] / mtr (float division), tsPWS[n] = (float)x_out[n].   :233-241 */
int  tspws_hip_epilogue(float *d_ls, float *d_tsPWS, const double *d_x_st, const double *d_x_out,
                        size_t N, unsigned mtr, void *stream);

/* ---- whole call on HBM-resident traces ------------------------------------------------ */
/* What tspws_main does between reading `in` and writing `out`, for one shard of traces that
 * already sits in device memory.  `p` must be resolved (tspws_resolve_params) and the plan
 * built from it.  Split in two so that a multi-GPU caller can sum the shard results between
 * the halves (one all-reduce of tspws_hip_reduce_buffer):
 *   _local : two-stage  -> partial stacks of the shard        (reduce buffer = P[Kmax][N])
 *            single     -> ST and PS of the shard's traces    (reduce buffer = ST||PS)
 *   _finish: (two-stage: forward+accumulate of the K partials,) weight, two inverses, epilogue.
 * d_ls / d_tsPWS receive `max` floats each. */
int  tspws_hip_stack_local(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld,
                           size_t mtr_local, size_t first, size_t mtr_global, void *stream);
int  tspws_hip_reduce_buffer(tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global,
                             double **d_buf, size_t *ndoubles);
int  tspws_hip_stack_finish(tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global,
                            float *d_ls, float *d_tsPWS, void *stream);

/* The finish stage in pieces, for callers that overlap it with the reduction of the second half of the groups (_range:
 * two-stage calls only).  _range transforms the already reduced partial stacks g_begin <= g < g_end of the reduce buffer
 * and adds them to the linear / phase stacks (ts_pws1f_lib.c:885-906); g_begin == 0 starts from zero, and ranges must
 * be given in increasing order so that the sums keep the reference's trace order.  _tail applies the weight, both inverse
 * transforms and the epilogue (ts_pws1f_lib.c:226-241).  tspws_hip_stack_finish == _range(0, Kmax) + _tail. */
int  tspws_hip_stack_finish_range(tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global,
                                  unsigned g_begin, unsigned g_end, void *stream);
int  tspws_hip_stack_finish_tail(tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global,
                                 float *d_ls, float *d_tsPWS, void *stream);

/* Single-GPU convenience: _local + _finish in one call; on return all work is ordered on `stream`. */
int  tspws_hip_stack(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld, size_t mtr,
                     float *d_ls, float *d_tsPWS, void *stream);
/* Optional timing inside tspws_hip_stack: HIP events on `stream` at the start of the call, after its streaming stage (the
 * partial-stack launches) and at its end, for up to max_calls calls.  _read synchronises the device and returns the per-call
 * durations in ms (either array may be NULL; *ncalls = calls recorded); _end returns the mean of the streaming stage and
 * resets the recorder. */
int  tspws_hip_profile_begin(tspws_hip_plan *plan, size_t max_calls);
int  tspws_hip_profile_read(tspws_hip_plan *plan, double *stage_ms, double *call_ms, size_t cap, size_t *ncalls);
int  tspws_hip_profile_end(tspws_hip_plan *plan, double *mean_ms, size_t *ncalls);
/* Number of streaming-kernel (k_partial) launches the last partial-stack / stack_local call issued: the traces are walked a
 * few groups at a time so that every launch puts one workgroup on every CU. */
int  tspws_hip_stream_launches(const tspws_hip_plan *plan);

/* ---- scale-sharded finish stage (multi-GPU; no counterpart in the reference) ---------------------------------------------
 * After the all-reduce every rank holds the same Kmax partial stacks.  The transforms, the weighting and the inverse are
 * separable by scale and the reconstruction is a SUM over scales: rank r finishes only its share of the scales and the
 * ranks add their partial reconstructions (2 * max doubles) before tspws_hip_epilogue.
 *   _finish_shard : work-balanced, contiguous share [*s_begin, *s_end) of the scales (whole decimation octaves, possibly
 *                   empty) of `rank` in `world`; returns 1 (not an error) when the plan / parameters have no sharded finish
 *                   (single-stage calls, debug kernels, >= 64 groups): finish with tspws_hip_stack_finish then.
 *   _finish_scales: transforms + stacks + weights + inverse of those scales from the reduced partial stacks
 *                   (tspws_hip_reduce_buffer); d_x2 receives [ICWT(OUT) | ICWT(ST)] restricted to them, 2 * max doubles. */
int  tspws_hip_finish_shard(const tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global, unsigned rank, unsigned world,
                            unsigned *s_begin, unsigned *s_end);
int  tspws_hip_stack_finish_scales(tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global, unsigned s_begin, unsigned s_end,
                                   double *d_x2, void *stream);

/* ---- jackknife (two-stage only, like the reference) ------------------------------------- */
/* Host: deletion masks sel[C][mtr] (1 = kept) from start times.  JackknifePlans, :385-430.
 * Returns 0, 1 for NULL arguments, -2 when time[0]==0 (no start times). */
int  tspws_jackknife_plan(char *sel, const time_t *time, size_t mtr, unsigned d, unsigned n, unsigned C);
/* Device: all C replicas from ONE pass over the traces.  TwoStage_jackknife_float, :719-831.
 * d_ls_out / d_ts_out are [C][max] floats; h_mtr_out receives the C replica sizes. */
int  tspws_hip_jackknife(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld,
                         size_t mtr, const char *h_sel, unsigned C,
                         float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *stream);

/* The two-stage stack AND its C jackknife replicas from ONE pass over the device-resident traces (the reference walks them
 * 1 + C times, :216 and :758-772): same outputs as tspws_hip_stack followed by tspws_hip_jackknife, except that the stack's
 * groups are summed class by class (last-bit differences in the FP64 partial stacks).  Falls back to tspws_hip_stack for
 * single-stage parameters or C == 0.  h_mtr_out is filled on return; the device outputs are ordered on `stream` like tspws_hip_stack's
 * (the call waits for the stream itself only when it had to upload the tables of a selection it has not seen in its last call). */
int  tspws_hip_stack_jackknife(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld, size_t mtr,
                               float *d_ls, float *d_tsPWS, const char *h_sel, unsigned C,
                               float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *stream);

/* Trace-sharded jackknife (one shard per GPU; the reference has no counterpart -- its replicas, :735-813, walk one host
 * array).  h_sel is the selection over the WHOLE ensemble, [C][mtr_global]; a trace's group in a replica is its rank among
 * all selected traces (:766), so every shard evaluates the signatures globally and sums only its own traces.
 *   _buffer : device rows [C * Kmax][max] doubles (row c * Kmax + g = group g of replica c), *nd = their count.
 *   _local  : ONE pass over the shard fills those rows AND the rows of the plain two-stage groups
 *             (tspws_hip_reduce_buffer) with the shard's sums; an empty shard gives zeros.  The caller adds the shards
 *             (all-reduce, or a reduce of each replica's rows to the rank that owns it).
 *   _finish : replicas [c_begin, c_end) from the (reduced) rows; outputs go to rows c_begin.. of the [C][max] arrays,
 *             replica sizes to h_mtr_out[c_begin..]. */
int  tspws_hip_jackknife_buffer(tspws_hip_plan *plan, const t_tsPWS *p, unsigned C, double **d_buf, size_t *nd);
int  tspws_hip_jackknife_local(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld, size_t mtr_local,
                               size_t first, size_t mtr_global, const char *h_sel, unsigned C, void *stream);
int  tspws_hip_jackknife_finish(tspws_hip_plan *plan, const t_tsPWS *p, size_t mtr_global, const char *h_sel, unsigned C,
                                unsigned c_begin, unsigned c_end, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out,
                                void *stream);

/* ---- random subsampling ---------------------------------------------------------------------- */
/* Host: keep K of J traces at random with libc rand(), flipping whichever symbol is rarer.
 * SubsamplingPlan, ts_pws1f_lib.c:355-383 (same rand() call order, so the same masks from the same state). */
int  tspws_subsampling_plan(char *sel, size_t J, size_t K);
/* M random subsamples of K = ceil(mtr * p->subsmpl_p) traces; d_ls_out / d_ts_out are [M][max] floats.
 * Single-stage: tspws_subsmpl_float (:501-610); two-stage: TwoStage_subsmpl_float (:612-709). */
int  tspws_hip_subsample(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld, size_t mtr, unsigned M,
                         float *d_ls_out, float *d_ts_out, void *stream);
/* The same with the M masks given (h_sel[M][mtr], 1 = kept: M calls of tspws_subsampling_plan with K = ceil(mtr * subsmpl_p)).
 * tspws_main draws them BEFORE its first HIP call: the initialisation of the HIP runtime inside a process's first call consumes
 * libc rand() values, and the masks are to come from the state the caller seeded, as in the reference. */
int  tspws_hip_subsample_sel(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld, size_t mtr, unsigned M,
                             const char *h_sel, float *d_ls_out, float *d_ts_out, void *stream);

/* ---- convergence curves ------------------------------------------------------------------------ */
/* Similarity / misfit of the stack of the first i+1 traces against a reference, for i = 0..mtr-1
 * (ts_pws1f_lib.c:247-314, similarity :433-449, misfit :452-462).  d_ref_ts / d_ref_ls are [max] floats on the
 * device (the caller passes in->reference for both, or the final tsPWS / ls); the four h_ arrays receive mtr doubles;
 * d_*_steps, when not NULL, receive the [mtr][max] float stacks of every step. */
int  tspws_hip_convergence(tspws_hip_plan *plan, const t_tsPWS *p, const float *d_sigall, size_t ld, size_t mtr,
                           const float *d_ref_ts, const float *d_ref_ls, double *h_ts_sim, double *h_ts_misfit,
                           double *h_ls_sim, double *h_ls_misfit, float *d_ts_steps, float *d_ls_steps, void *stream);

/* ---- several devices of one process (single process, ncclCommInitAll, one stream per device) ---------------
 * Traces shard contiguously by global index; what shards is the sum of partial_linear_stacks (ts_pws1f_lib.c:866-881) --
 * single-stage: of ST || PS (:486-494).  The devices' buffers are added by ONE RCCL all-reduce (fp64, sum) over xGMI.
 * RCCL is bound at run time (librccl.so.1), so single-device users never load it.
 *   TSPWS_COMM=local  own reduction kernel instead of RCCL; also chosen when the device list names a device twice (RCCL
 *                     refuses that) -- lets the N-way bookkeeping run on a one-GPU box; not a fallback: a list with
 *                     DISTINCT devices is refused under it (several physical devices always reduce through RCCL)
 *   TSPWS_COMM=rccl   go through RCCL even for a single device */
typedef struct tspws_hip_comm tspws_hip_comm;
/* communicator over `ndev` devices of this process; devices == NULL: 0 .. ndev-1 */
int   tspws_hip_comm_create(tspws_hip_comm **comm, int ndev, const int *devices);
void  tspws_hip_comm_destroy(tspws_hip_comm *comm);
int   tspws_hip_comm_size(const tspws_hip_comm *comm);
int   tspws_hip_comm_device(const tspws_hip_comm *comm, int i);
void *tspws_hip_comm_stream(const tspws_hip_comm *comm, int i);      /* the communicator's own stream on device i */
const char *tspws_hip_comm_backend(const tspws_hip_comm *comm);      /* "rccl 2.x.y" or "local" */
/* In-place sum: d_bufs[i] = `count` doubles on device i; ordered on streams[i] (NULL array / entry: the communicator's
 * stream of that device).  One ncclAllReduce per device inside ncclGroupStart / ncclGroupEnd. */
int   tspws_hip_allreduce_f64(tspws_hip_comm *comm, double *const *d_bufs, size_t count, void *const *streams);
/* Sum into device `root` only (one ncclReduce per device): the replica rows of a sharded jackknife go to the device that finishes them. */
int   tspws_hip_reduce_f64(tspws_hip_comm *comm, double *const *d_bufs, size_t count, int root, void *const *streams);
/* contiguous shard of `r` of `n`: traces [*first, *first + *count) */
void  tspws_shard_range(size_t mtr, unsigned r, unsigned n, size_t *first, size_t *count);

/* The whole call over trace shards: one plan per device + the communicator.  tspws_main uses it when TSPWS_DEVICES names
 * several devices ("0,1,2,3" or "all").  d_shards[r] = the shard of device r (tspws_shard_range), row stride ld; d_ls /
 * d_tsPWS live on the first device.  The finish stage is split by scales over the devices (tspws_hip_finish_shard) and a
 * second small all-reduce adds the partial reconstructions.  Returns after synchronising the devices. */
typedef struct tspws_hip_multi tspws_hip_multi;
int   tspws_hip_multi_create(tspws_hip_multi **m, int ndev, const int *devices, int type, unsigned J, unsigned V, unsigned N,
                             double s0, double b0, double w0, int uni);
void  tspws_hip_multi_destroy(tspws_hip_multi *m);
tspws_hip_comm *tspws_hip_multi_comm(tspws_hip_multi *m);
tspws_hip_plan *tspws_hip_multi_plan(tspws_hip_multi *m, int i);
/* host traces -> shards (the host array is pinned once, every device pulls its shard on its own stream: all PCIe links at
 * once); *d_shards = per-device buffers owned by m, *d_ls / *d_tsPWS = output buffers on the first device */
int   tspws_hip_multi_upload(tspws_hip_multi *m, const float *h_sigall, size_t ld, size_t mtr, const float *const **d_shards,
                             float **d_ls, float **d_tsPWS);
/* fold / mean removal on the uploaded shards, mirrored back into h_sigall (ts_pws1f_lib.c:71-88, :159-169) */
int   tspws_hip_multi_prologue(tspws_hip_multi *m, float *h_sigall, size_t max, size_t ld, size_t mtr, int fold, int rm);
int   tspws_hip_multi_stack(tspws_hip_multi *m, const t_tsPWS *p, const float *const *d_shards, size_t ld, size_t mtr,
                            float *d_ls, float *d_tsPWS);
/* ... and its C jackknife replicas (two-stage only): one pass per shard, the rows all-reduced, device r finishes replicas
 * [r C / n, (r + 1) C / n) and writes its rows of the HOST arrays h_ls_out / h_ts_out ([C][max] floats). */
int   tspws_hip_multi_stack_jackknife(tspws_hip_multi *m, const t_tsPWS *p, const float *const *d_shards, size_t ld, size_t mtr,
                                      float *d_ls, float *d_tsPWS, const char *h_sel, unsigned C,
                                      float *h_ls_out, float *h_ts_out, unsigned *h_mtr_out);

/* ---- the drop-in on a named device, and its cache -----------------------------------------------------
 * tspws_main (ts_pws1f_lib.h; reference: ts_pws1f_lib.h:104) runs on the device TSPWS_DEVICE names (default 0; a one-entry
 * TSPWS_DEVICES means the same).  tspws_main_on is the same call on an explicit device -- no environment variable is
 * consulted -- for hosts that run one stacking thread per GPU (several station pairs at once): calls on different devices
 * run concurrently, calls on the same device are serialised (per-device lock).
 * Each device keeps the frame and the device trace buffer of its last call for the next one (same parameters: nothing to
 * rebuild; the memory stays allocated between calls).  tspws_main_release frees all of them; TSPWS_PLAN_CACHE=0 in the
 * environment disables the cache altogether.  tspws_main_cached_devices: bit i set = device i holds a cached frame. */
int  tspws_main_on(int device, t_tsPWS *tspws, t_tsPWS_out *out, t_data *in);
void tspws_main_release(void);
unsigned long long tspws_main_cached_devices(void);

/* ---- synthetic ensembles for bench / tests ------------------------------------ */
/* trace i = first+local, sample n: 0.2 sin(2pi(n-N/2)/200) exp(-((n-N/2)/(0.05N))^2/2) + U(-.5,.5) */
int  tspws_hip_synth(float *d_sigall, size_t mtr, size_t N, size_t ld, uint64_t seed, size_t first, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSPWS_HIP_H */
