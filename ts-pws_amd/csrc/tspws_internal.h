// tspws_internal.h -- what the translation units of libtspws_hip.so share: the plan object, the per-scale descriptor,
// error plumbing, a few device helpers and the host functions that cross unit boundaries.  Not installed; the public
// surface is include/tspws_hip.h.
//
//   plan.hip      parameters, frame geometry, tap generation, runtime helpers, synthetic traces
//   stream.hip    trace prologue + stage 1: the HBM-streaming partial-stack pass
//   forward.hip   forward frame CWT (fwd_lds.h, fwd_poly.h, fwd_tl.h) + linear / phase stacks
//   inverse.hip   phase weighting, inverse frame CWT (inv_poly.h), epilogue
//   stack.hip     the whole call on device-resident traces: local half, finish stage (whole / in pieces / by scales)
//   resample.hip  jackknife, random subsampling, convergence curves
//   comm.hip      trace shards on several devices of one process: RCCL all-reduce, sharded tspws_main driver
//
// Layout in HBM
//   traces       float   [mtr][ld]     row-major, one trace per row (the reference's sigall)
//   partials     double  [Kmax][ldP]   stage-1 group sums of the two-stage stack
//   taps         double2 [ntaps]       ragged per scale, tap_off[s] .. ; dual taps likewise
//   coefficients double2 [ncoef]       ragged [S][N_s], coef_off[s] ..  (N_s = ceil(N/D_s))
// All arithmetic on the path is FP64 (the reference is double / double complex throughout); MFMA is deliberately unused:
// the per-scale FIRs are skinny 1-D correlations.  Reference citations are relative to /root/reference/src.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tspws_hip.h"

#ifndef TSPWS_PI
#define TSPWS_PI 3.14159265358979328
#endif

// ------------------------------------------------------------------------------------------
// error plumbing (plan.hip owns the thread-local text)
// ------------------------------------------------------------------------------------------
int tspws_fail(int code, const char *what, hipError_t e = hipSuccess);
#define fail tspws_fail

#define HIP_TRY(expr)                                                              \
	do {                                                                           \
		hipError_t e_ = (expr);                                                    \
		if (e_ != hipSuccess) return tspws_fail(e_ == hipErrorOutOfMemory ? TSPWS_E_NOMEM : TSPWS_E_HIP, #expr, e_); \
	} while (0)

static inline hipStream_t S_(void *s) { return (hipStream_t)s; }

// Tuning / A-B / test switches are read from the environment only in builds with -DTSPWS_SWEEPS (`make sweeps`:
// lib/libtspws_hip_sweeps.so, what tools/build_variant.sh and the engine-agreement tests load through TSPWS_LIB_PATH).  The shipped
// library's environment is TSPWS_DEVICE / TSPWS_DEVICES / TSPWS_PLAN_CACHE (tspws_main.c), TSPWS_COMM (comm.hip), TSPWS_PART_MB,
// TSPWS_ENGINE (forward.hip) and, in the Python binding, TSPWS_SCHEDULE / TSPWS_SHARD_FINISH -- nothing else a caller's
// environment could reach.  (TSPWS_SCHEDULE is also read by comm.hip's device-list call.)
#ifdef TSPWS_SWEEPS
static inline const char *sweep_env(const char *name) { return getenv(name); }
#else
static inline const char *sweep_env(const char *) { return nullptr; }
#endif

// ------------------------------------------------------------------------------------------
// per-scale descriptor (host table sc / sc_tl, device copies d_sc / d_sc_tl)
// ------------------------------------------------------------------------------------------
struct ScaleDesc {
	unsigned L, D, Ns;
	int c, cd;
	unsigned Q;                 // ceil(L / D): taps per phase
	unsigned long long tap_off, coef_off;
	double scale, gain;         // gain = ln2 / (2 Cpsi V scale), wavelet_v7.c:145
	// work decomposition of the polyphase forward kernel (fwd_poly.h)
	unsigned DL, logDL;         // phase lanes per output group (power of two <= 64)
	unsigned MC, cps, nsplit;   // 64-phase chunks, chunks per wave, waves sharing one output group
	unsigned ngw;               // group-blocks (waves) per split
	unsigned wave_off;          // first wave of this scale in the launch
	unsigned inv_fast;          // 1: D divides N, handled by the polyphase inverse (inv_poly.h)
	unsigned acc_off;           // first 256-coefficient block of this scale in k_accumulate_masked
	unsigned use_lds;           // 1: forward transform by k_fwd_lds (fwd_lds.h), 0: k_fwd_poly
	unsigned lds_off, lds_bps;  // first workgroup of this scale in k_fwd_lds, workgroups per split
	unsigned acc2_off;          // first block of this scale in k_accumulate_parts (4 coefficients per block when split)
	unsigned fuse_ok;           // 1: k_fwd_lds<FUSE> keeps this scale's linear / phase stacks in registers (no partials)
	unsigned r16;               // 1: the direct kernel gives a thread 16 outputs of this scale (ngw counts 16-output groups)
	unsigned long long part_off; // offset of this scale's [nsplit][Ns] partial block
};

struct Chunk { // one streaming work item of the partial-stack kernel
	unsigned long long t0; // first local trace
	unsigned count;        // traces
	unsigned row;          // destination row (group / class)
};

struct RunDesc { // one run of consecutive traces with one signature, for k_rows_walk (stream.hip)
	unsigned long long t0;   // first trace
	unsigned count;          // traces
	unsigned member, flush;  // bit c: the run belongs to column c / column c's group ends with this run
	unsigned frow;           // first entry of the run's flush destinations in flush_rows (ascending column order)
	unsigned pad[2];
};

// A contiguous run of scales [s0, s1) a finish-stage launch is restricted to (scale-sharded finish); s1 == 0: all scales.
struct ScaleRange {
	unsigned s0 = 0, s1 = 0;
	bool on() const { return s1 != 0; }
};

// Output of the fused forward + phase-stack kernel: slice j of the launch (traces [j*tps, (j+1)*tps)) leaves the linear and
// phase stacks of every fuse_ok scale in accST / accPS + j*stride ([ncoef] planes).
// Optional last step of the fused kernel (the staged masked replicas, resample.hip): the slice's stacks are COMPLETED there -- the plane
// pairs of the `nprev` earlier stages of the same column are added in front (stage order), the weighted coefficient goes straight to
// OUT (with the column's trace count Mv[slice]) and no pass of k_accumulate_parts over the fused scales (98 % of the coefficients) is
// needed; the slice `keep_slice` also keeps its linear stack (the plain stack's ST is reconstructed for ls).
struct FuseFinal {
	const double2 *pST = nullptr, *pPS = nullptr; // plane pairs of stage 0: slice j at + j * slice_stride, stage q at + q * pair_stride
	size_t pair_stride = 0, slice_stride = 0;
	unsigned nprev = 0;
	double2 *OUT = nullptr;                       // nullptr: off
	size_t out_stride = 0;
	const double *Mv = nullptr;                   // trace count per slice (nullptr: M for all)
	double M = 0;
	int mode = 0, keep_slice = -1;
	double2 *keepST = nullptr;                    // [ncoef] set that receives the linear stack of slice keep_slice
	double K = 0, wu = 0;
};

struct FuseOut {
	double2 *accST = nullptr, *accPS = nullptr;
	size_t stride = 0;
	unsigned tps = 1;
	bool applied = false; // set by the forward launch when the fused kernel ran
	FuseFinal fin;        // (default: off)
	// few rows in columns (slices of tps rows) whose far-decimated scales may go through the spectral engine (spectral.hip): the caller
	// allows it; the forward launch reports the first scale it took that way (S: none) -- those scales are COMPLETE afterwards (weighted
	// sets / plane pairs of every slice written), whatever their fuse_ok / nsplit say
	bool allow_spec = false;
	unsigned spec_first = ~0u;
	// recorded behind the transposition of the spectral chain (side by side with the FIR kernels: the default) or behind the whole chain (one after
	// the other): work that only must not run beside that transposition can start here (resample.hip: the replicas' linear stacks)
	hipEvent_t ev_mid = nullptr;
	bool mid_recorded = false;
	// the caller joins the FIR kernels' stream itself (it has more work for that stream: the early half of the inverses, resample.hip): when the
	// FIR kernels ran on a stream of their own beside the spectral chain, the forward launch leaves it un-joined and names it here
	bool defer_fir_join = false;
	hipStream_t fir_stream = nullptr;
};

// Phase weighting of one coefficient (tspws_biased :909-943, tspws_unbiased :965-984).
// mode 0: wu == 2 biased, 1: wu == 1, 2: general power, 3: unbiased (K > 1); K = stacked units, M = traces.
struct WeightArgs {
	double2 *OUT = nullptr;   // nullptr: no weighting
	int mode = 0;
	double K = 0, M = 0, wu = 0;
	// prefix outputs (convergence curves, ts_pws1f_lib.c:247-314): after trace b of the batch the weighted coefficients of the
	// first k0 + b + 1 traces go to OUTP[b * outp_stride + i] (K = M = that count, mode1 when it is 1); the traces are added one by one
	double2 *OUTP = nullptr;
	size_t outp_stride = 0;
	unsigned k0 = 0;
	int mode1 = 0;
	// stacks side by side in one launch (grid.y: jackknife replicas) with their own trace counts: M = Mv[blockIdx.y]
	const double *Mv = nullptr;
	size_t out_stride = 0;    // ... and their own weighted-coefficient sets: OUT + blockIdx.y * out_stride
	int planes_batch = -1;    // -1: every stack's ST / PS planes are written; -2: none (only OUT is wanted); b >= 0: those of stack b only
};

// optional geometry of tspws_launch_accumulate for stacks whose transformed traces / slice planes are interleaved in memory
struct AccExtra {
	size_t trace_stride = 0;          // distance of consecutive transformed traces in `part` (0: npart)
	size_t y_fz = 0;                  // slice planes of stack b start b * y_fz further
	const unsigned *rowmap = nullptr; // device table [nbatch][nb]: transformed trace t of stack b is trace rowmap[b nb + t] of `part`
	bool fused_done = false;          // the fused forward kernel completed (and weighted) the stacks of its scales itself: only the others are left
};

enum { SCR_Y = 0, SCR_PART, SCR_XT, SCR_OBUF, SCR_SEL, SCR_SUBST, SCR_CONV, SCR_CHUNK, SCR_P, SCR_STPS, SCR_OUT, SCR_X2, SCR_CLS, SCR_JKP, SCR_JKOUT, SCR_TAB, SCR_FZ, SCR_JKTAB, SCR_SPA, SCR_SPB, SCR_SPG, SCR_SPH, SCR_SPM, SCR_GEMM, SCR_N };

struct OctDesc; // inverse work items (inv_poly.h)
struct TLItem;  // many-trace forward work items (fwd_tl.h)

// One work decomposition of the many-trace forward path: its own scale table (partial layout, fused flags, accumulate
// geometry), the trace-lane work items of k_fwd_tl and the waves of the direct kernel for the scales it leaves out.
struct GemmCol; // columns of the matrix-pipe kernel for the coarsest scales (fwd_gemm.h)
struct TlTable {
	unsigned minns = 0;              // octaves with fewer outputs than this stay on the direct kernel
	std::vector<ScaleDesc> sc;
	ScaleDesc *d_sc = nullptr;
	TLItem *d_items = nullptr;
	unsigned n = 0, wgs = 0, waves = 0, acc2_blocks = 0; // items, workgroups per trace block, direct-kernel waves, accumulate blocks
	size_t npart = 0, lds = 0;
	// spectral decompositions of frames whose coarsest filters do not fit the transform window: those scales as one dense contraction (k_fwd_gemm)
	GemmCol *d_gcols = nullptr;
	unsigned gcoltiles = 0, gKS = 0, gKC = 0; // tiles of 16 columns, runs of samples per (trace block, tile), samples per run
};

struct SpecDecomp; // spectral engine (spectral.h)

struct tspws_hip_plan {
	int device = 0, type = -1;
	unsigned S = 0, V = 0, J = 0, N = 0;
	double s0 = 0, b0 = 0, w0 = 0, Cpsi = 0;
	size_t ncoef = 0, ntaps = 0;
	size_t npart = 0;          // complex partial coefficients per trace (sum of nsplit*Ns)
	unsigned fwd_waves = 0;    // waves per trace batch of k_fwd_poly
	unsigned acc_blocks = 0;   // blocks of k_accumulate_masked (256 coefficients each)
	unsigned acc2_blocks = 0;  // blocks of k_accumulate_parts
	unsigned lds_blocks = 0;   // workgroups per trace slice of k_fwd_lds
	unsigned lds_qt = 24;      // tap rows resident in LDS: the k_fwd_lds instantiation of this frame (24 or 32, tspws_build_forward)
	unsigned n_fusable = 0;    // scales whose stacks the fused forward kernel keeps in registers
	// many-trace decompositions (fwd_tl.h): tl[0] for batches of many 64-trace blocks, tl[1] for few (more scales on the
	// direct kernel, whose parallelism is in the taps): see TlTable
	TlTable tl[2];
	std::vector<SpecDecomp *> spec; // spectral sets built so far, with their many-trace decompositions (spectral.hip)
	std::vector<unsigned> oc_s0, oc_nv, oc_wave_off, oc_nwaves, oc_gen; // host copy of the inverse's octave items (launch order)
	std::vector<unsigned> og_s0, og_nv; // the decimation octaves (first scale, voices) in scale order, whatever the items are
	unsigned inv_waves_lds = 0; // ... of which the first inv_waves_lds (octaves with D < 64) run the LDS-staged instantiation
	unsigned inv_waves = 0, inv_waves_fast = 0, inv_noct = 0, inv_ngeneric = 0; // polyphase inverse: waves (of the octaves whose D divides N first), octave items, scales left to the generic kernel
	OctDesc *d_oc = nullptr;
	std::vector<ScaleDesc> sc;
	ScaleDesc *d_sc = nullptr;
	double2 *d_w = nullptr, *d_wd = nullptr;
	// lazily grown device scratch
	void *scr[SCR_N] = {nullptr};
	size_t scr_bytes[SCR_N] = {0};
	// forward transform: the direct kernel (coarse scales, latency-bound) runs beside the LDS kernel (FP64-bound) on a
	// side stream, forked from and joined back into the caller's stream
	hipStream_t side = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	hipStream_t xs = nullptr;          // ... and the spectral chain of a few-row launch beside both (forward.hip)
	hipEvent_t ev_xs0 = nullptr, ev_xs1 = nullptr, ev_xs2 = nullptr;
	hipEvent_t ev_mid = nullptr;         // behind the spectral chain of the masked call's last stage (FuseOut::ev_mid)
	hipEvent_t ev_lin = nullptr;         // behind the replicas' linear stacks when they run on the second stream
	// optional timing inside tspws_hip_stack (bench.py): three events per call -- start, end of the streaming stage, end
	// Events that ride on kernel launches instead of being recorded as packets of their own (hipExtLaunchKernelGGL: the launch's
	// start / completion signal IS the event; tools/probes/xstream_probe.hip: the next kernel of the stream follows 2.5 us after
	// the kernel instead of 4.0).  Armed by tspws_hip_stack for the duration of that ONE call and consumed by the launches they
	// name (arguments of the call in all but form; every field is NULL between API calls):
	//   first_start / last_stop: first / last launch of the streaming pass (tspws_run_chunks); last_stop then becomes `ready`,
	//   ready: the producer of the forward transforms' input has signalled it -- the side stream waits for it instead of a fork record,
	//   call_end: the launch that writes the float outputs (tspws_inverse_pair_out).
	struct LaunchEvents { hipEvent_t first_start = nullptr, last_stop = nullptr, ready = nullptr, call_end = nullptr; } le;
	std::vector<hipEvent_t> prof_ev;
	size_t prof_used = 0;
	// cached chunk table: the host copy is keyed on (mtr_local, first, mtr_global, K); the device copy becomes valid only
	// once its upload has been enqueued (ck_dev), other streams order themselves behind it through ck_ev
	std::vector<Chunk> chunks;
	std::vector<unsigned> row_first; // per destination row: first chunk, rows+1 entries
	size_t ck_mtr = 0, ck_first = 0, ck_glob = 0;
	unsigned ck_K = 0;
	bool ck_valid = false, ck_dev = false;
	hipEvent_t ck_ev = nullptr;
	hipStream_t ck_stream = nullptr;
	unsigned last_stream_launches = 0; // k_partial launches of the last streaming pass (bench.py: per-launch roofline figures)
	// masked replicas (resample.hip): generation of the class / combine tables the table blocks (SCR_TAB, SCR_JKTAB) hold right
	// now (0: none -- any other upload into those blocks resets it); a call with the same selection skips the uploads
	unsigned long long jk_gen = 0;
	hipStream_t xf = nullptr;          // second stream of the pipelined masked-replica call: transforms of finished groups
	std::vector<hipEvent_t> stage_ev;  // ... one event per hand-over
	// blocks of exported slots (tspws_hip_reduce_buffer hands out SCR_P / SCR_STPS, tspws_hip_jackknife_buffer SCR_JKP) that
	// were outgrown: a caller may still hold the old pointer (e.g. as the buffer of an in-flight collective), so they live
	// until plan_destroy
	std::vector<void *> retired;
};

int tspws_scratch(tspws_hip_plan *p, int slot, size_t bytes, void **out);
#define scratch tspws_scratch

// ------------------------------------------------------------------------------------------
// device helpers shared by the kernels of several units
// ------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ unsigned find_scale(const ScaleDesc *sc, unsigned S, unsigned long long idx, bool taps)
{
	unsigned lo = 0, hi = S; // last s with off[s] <= idx
	while (hi - lo > 1) {
		unsigned mid = (lo + hi) >> 1;
		unsigned long long off = taps ? sc[mid].tap_off : sc[mid].coef_off;
		if (off <= idx) lo = mid; else hi = mid;
	}
	return lo;
}

__device__ __forceinline__ double wave_sum(double v) // wave = 64 lanes
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

// a double of lane `src` (compile-time or wave-uniform) as a wave-uniform value
__device__ __forceinline__ double readlane_f64(const double x, const int src)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
	return __hiloint2double(hi, lo);
}

// PS += Y/|Y| unless the quotient is not a unit phasor (Y == 0 gives NaN and is skipped), ts_pws1f_lib.c:491-492.
// Fast path: one rsqrt instead of hypot + two divisions whenever |Y|^2 is comfortably inside the double range;
// the literal form handles the rest (zeros, subnormals, huge values).
__device__ __forceinline__ void add_unit_phasor(double2 &ps, const double2 v)
{
	const double r2 = fma(v.x, v.x, v.y * v.y);
	if (r2 > 1e-280 && r2 < 1e280) {
		const double inv = rsqrt(r2);
		ps.x = fma(v.x, inv, ps.x);
		ps.y = fma(v.y, inv, ps.y);
	} else {
		const double r = hypot(v.x, v.y);
		const double ux = v.x / r, uy = v.y / r;
		if (ux * ux + uy * uy <= 1.001) { ps.x += ux; ps.y += uy; }
	}
}

__device__ __forceinline__ double2 weight_value(const double2 st, const double2 ps, const int mode, const double K, const double M, const double wu)
{
	double a;
	if (mode == 0) {
		const double g = 1. / (K * K * M);
		a = (ps.x * ps.x + ps.y * ps.y) * g;
		return make_double2(a * st.x, a * st.y);
	} else if (mode == 1) {
		const double g = 1. / (K * M);
		const double r = hypot(ps.x, ps.y);
		return make_double2(st.x * r * g, st.y * r * g);
	} else if (mode == 2) {
		a = hypot(ps.x, ps.y) / K;
		a = pow(a, wu);
		return make_double2(st.x * a / M, st.y * a / M);
	}
	const double iK = 1. / K, iK1 = 1. / (K - 1), iM = 1. / M;
	const double px = ps.x * iK, py = ps.y * iK;
	a = px * px + py * py;
	a = (K * a - 1) * iK1;
	return make_double2(st.x * a * iM, st.y * a * iM);
}
#endif

// ------------------------------------------------------------------------------------------
// host functions that cross unit boundaries
// ------------------------------------------------------------------------------------------
// forward.hip
int  tspws_build_forward(tspws_hip_plan *p);              // work decomposition of the forward kernels (few-trace and many-trace tables)
int  tspws_forward_parts_f32(tspws_hip_plan *p, const float *d_x, size_t ntr, size_t ld, double2 *d_part, hipStream_t st, FuseOut *fz, ScaleRange rg);
int  tspws_forward_parts_f64(tspws_hip_plan *p, const double *d_x, size_t ntr, size_t ld, double2 *d_part, hipStream_t st, FuseOut *fz, ScaleRange rg);
int  tspws_join_fir_stream(tspws_hip_plan *p, hipStream_t fir, hipStream_t st); // FuseOut::defer_fir_join: st waits for what is enqueued on fir
// ST / PS of ntr traces (keep: add to the stacks already there; wa: weighting applied by the launch that completes the
// stacks, *weighted tells whether that happened; rg: only these scales)
int  tspws_stacks_f32(tspws_hip_plan *p, const float *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                      const WeightArgs *wa, bool *weighted, ScaleRange rg);
unsigned tspws_first_unfused_scale(const tspws_hip_plan *p); // scales [it, S) hold every scale the fused forward kernel does not stack itself
int  tspws_stacks_f64(tspws_hip_plan *p, const double *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                      const WeightArgs *wa, bool *weighted, ScaleRange rg);
// k_accumulate_parts for nb transformed traces (nbatch independent stacks side by side: y_part / y_stack apart)
void tspws_launch_accumulate(tspws_hip_plan *p, const double2 *part, unsigned nb, double2 *ST, double2 *PS, int zero_first, const FuseOut *fz,
                             unsigned nslices, hipStream_t st, unsigned nbatch, size_t y_part, size_t y_stack, const TlTable *tl, const WeightArgs *wa,
                             ScaleRange rg, const AccExtra *ex = nullptr);
bool tspws_fused_forward(const tspws_hip_plan *p);        // the few-trace forward kernel stacks some scales in registers
bool tspws_many_trace_path(const tspws_hip_plan *p, size_t ntr); // a batch this size goes to the trace-lane kernel
unsigned tspws_spectral_choice(const tspws_hip_plan *p, size_t ntr); // first scale of the batch's spectral set (S: none)
size_t tspws_part_budget_bytes();
int  tspws_engine_pin();   // TSPWS_ENGINE: 0 auto, 1 fir, 2 spectral, -1 not a value (refused by plan_create)
bool tspws_generic_forward();
// spectral.hip: the far-decimated octaves of a many-trace batch through the traces' spectra
unsigned tspws_spectral_first_scale(const tspws_hip_plan *p, unsigned nsmax); // first scale of the spectral set [first, tspws_spectral_end_scale) for octaves of <= nsmax outputs (S: none)
int  tspws_spectral_decomp(tspws_hip_plan *p, unsigned s_first, unsigned nblk_hint, SpecDecomp **out, bool few = false);
int  tspws_spectral_rows_f64(tspws_hip_plan *p, SpecDecomp *dc, const double *d_x, size_t ld, unsigned ntr, unsigned tps, const FuseOut &fz, hipStream_t st,
                             hipEvent_t after_transposition = nullptr); // few rows in columns
int  tspws_build_tl_spectral(tspws_hip_plan *p, unsigned s_first, unsigned s_end, unsigned nblk_hint, TlTable &T); // (forward.hip) scale table + trace-lane items of that decomposition
void tspws_spectral_geometry(const tspws_hip_plan *p, unsigned *NT, unsigned *s_end, unsigned *cneg); // transform length, end of the scales that fit its window, samples in front of the trace
unsigned tspws_spectral_end_scale(const tspws_hip_plan *p); // end of the spectral set (S unless the coarsest filters do not fit the transform window)
int  tspws_spectral_run_f32(tspws_hip_plan *p, SpecDecomp *dc, const float *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st);
int  tspws_spectral_run_f64(tspws_hip_plan *p, SpecDecomp *dc, const double *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st);
// the batch transposed (xT[n][t], TP = padded trace count) + the traces' largest |sample| where the spectral chain looks for them
int  tspws_spectral_transpose_f32(tspws_hip_plan *p, const float *d_x, size_t ld, unsigned ntr, float *xT, unsigned TP, hipStream_t st);
int  tspws_spectral_transpose_f64(tspws_hip_plan *p, const double *d_x, size_t ld, unsigned ntr, double *xT, unsigned TP, hipStream_t st);
void tspws_spectral_destroy(tspws_hip_plan *p);
// inverse.hip
int  tspws_build_inverse(tspws_hip_plan *p);
int  tspws_weight_mode(double wu, int unbiased, unsigned K);
// OUT[j] = ST[j] * weight(PS[j]) for nb stacks side by side (y_out / y_stack doubles2 apart), trace counts d_Mv[j]
void tspws_weight_batched(tspws_hip_plan *p, double2 *OUT, const double2 *ST, const double2 *PS, int mode, double K, double wu, const double *d_Mv,
                          unsigned nb, size_t y_out, size_t y_stack, hipStream_t st, double M = 0.0); // d_Mv: per-stack trace counts (NULL: M for all)
// the stack's pair of reconstructions (set 0 = OUT, set 1 = ST of Y) straight to the float outputs
int  tspws_inverse_pair_out(tspws_hip_plan *p, const double2 *Y, float *d_ts, float *d_ls, float mtr, hipStream_t st);
int  tspws_inverse_scales(tspws_hip_plan *p, const double2 *Y, double *x2, hipStream_t st, ScaleRange rg);
// nb pairs of reconstructions in two halves: the octaves of the scales [0, s_split) early on the stream where their sets are complete, the rest + the
// combining kernel late on the caller's stream (which the caller has made wait for `early` in between); *done = false: no such split, nothing launched
int  tspws_inverse_pairs_early(tspws_hip_plan *p, const double2 *Y, unsigned nb, unsigned s_split, hipStream_t early, bool *done);
int  tspws_inverse_pairs_late(tspws_hip_plan *p, const double2 *Y, double *x, unsigned nb, unsigned s_split, hipStream_t st);
bool tspws_generic_inverse();
// ts rows: (float) x[j][n] for nb rows (replica outputs)
void tspws_epilogue_rows(float *d_ts, const double *d_x, size_t N, unsigned nb, hipStream_t st);
// stream.hip
int  tspws_run_chunks(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const std::vector<Chunk> &chunks,
                      const std::vector<unsigned> &row_first, unsigned rows, double *d_P, size_t ldP, hipStream_t st, bool cached,
                      unsigned row_begin = 0, unsigned row_end = ~0u);
int  tspws_chunks_upload(tspws_hip_plan *p, const std::vector<Chunk> &chunks, const std::vector<unsigned> &row_first, unsigned rows, hipStream_t st,
                         bool cached);
int  tspws_chunks_launch(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const std::vector<Chunk> &chunks,
                         const std::vector<unsigned> &row_first, unsigned rows, double *d_P, size_t ldP, hipStream_t st,
                         unsigned row_begin = 0, unsigned row_end = ~0u);
// running sums with snapshots after every run of traces + rows as signed sums of snapshots (masked replicas): see k_prefix_walk
int  tspws_prefix_launch(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const Chunk *d_runs, const unsigned *d_seg_first, unsigned nseg,
                         size_t nruns_total, const unsigned *d_carry, unsigned ncarry, double **d_snap, size_t *ldpc, hipStream_t st);
void tspws_combine_terms_launch(const double *d_snap, size_t ldpc, const unsigned *d_row_ptr, const unsigned *d_idx, const float *d_coef, unsigned nrows,
                                double *d_P, size_t N, hipStream_t st);
// few columns: the rows themselves from one walk with a running sum per column, a stage in two segments (k_rows_walk, k_seg_fix)
int  tspws_rows_walk_launch(const float *d_x, size_t ld, size_t N, const RunDesc *d_runs, unsigned q0, unsigned qm, unsigned q1, unsigned W,
                            const unsigned *d_flush_rows, const unsigned *d_fix_row, double *d_rows, double *d_blk, int carry_in, int carry_out, hipStream_t st);
unsigned tspws_rows_walk_wmax();
unsigned tspws_chunk_len_for(size_t N, size_t mtr);
// stack.hip
bool tspws_is_two_stage(const t_tsPWS *p, size_t mtr_global);
