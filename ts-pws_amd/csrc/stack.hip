// stack.hip -- the whole call on device-resident traces: what tspws_main does between reading `in` and writing `out`
// (ts_pws1f_lib.c:194-242), split into the shard-local half, the reduction buffer a multi-GPU caller sums, and the finish
// stage (as a whole, in pieces of groups, or restricted to a run of scales).
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"

bool tspws_is_two_stage(const t_tsPWS *p, size_t mtr_global) { return !(!p->Kmax || p->Kmax > mtr_global); }
#define is_two_stage tspws_is_two_stage

extern "C" int tspws_hip_reduce_buffer(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, double **d_buf, size_t *nd)
{
	if (!pl || !p || !d_buf || !nd) return fail(TSPWS_E_ARG, "reduce_buffer: NULL");
	HIP_TRY(hipSetDevice(pl->device));
	void *b = nullptr;
	int rc;
	if (is_two_stage(p, mtr_global)) {
		*nd = (size_t)p->Kmax * pl->N;
		if ((rc = scratch(pl, SCR_P, *nd * sizeof(double), &b))) return rc;
	} else {
		*nd = 4 * pl->ncoef;
		if ((rc = scratch(pl, SCR_STPS, *nd * sizeof(double), &b))) return rc;
	}
	*d_buf = (double *)b;
	return 0;
}

extern "C" int tspws_hip_stack_local(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr_local, size_t first,
                                     size_t mtr_global, void *s)
{
	if (!pl || !p || (!d_x && mtr_local)) return fail(TSPWS_E_ARG, "stack_local: NULL");
	double *buf; size_t nd; int rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &buf, &nd))) return rc;
	// an empty shard is legal: its rows are zeros, so that every rank of a trace-sharded call reaches the collective
	if (!mtr_local) { HIP_TRY(hipMemsetAsync(buf, 0, nd * sizeof(double), S_(s))); return 0; }
	if (is_two_stage(p, mtr_global)) return tspws_hip_partial_stacks(pl, d_x, ld, mtr_local, first, mtr_global, p->Kmax, buf, pl->N, s);
	return tspws_hip_stacks_float(pl, d_x, mtr_local, ld, buf, buf + 2 * pl->ncoef, s);
}

// coefficient block of the finish stage: [OUT | ST | PS] so that the two inverses read rows 0 and 1
static int finish_block(tspws_hip_plan *pl, double **OUT, double **ST, double **PS)
{
	void *v;
	const size_t nc = pl->ncoef;
	int rc = scratch(pl, SCR_OUT, 6 * nc * sizeof(double), &v);
	if (rc) return rc;
	*OUT = (double *)v; *ST = *OUT + 2 * nc; *PS = *ST + 2 * nc;
	return 0;
}

static WeightArgs weight_args(const t_tsPWS *p, double *OUT, unsigned K, size_t mtr_global)
{
	WeightArgs wa;
	wa.OUT = (double2 *)OUT; wa.mode = tspws_weight_mode(p->wu, p->unbiased, K); wa.K = (double)K; wa.M = (double)(unsigned)mtr_global; wa.wu = p->wu;
	return wa;
}

extern "C" int tspws_hip_stack_finish_range(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, unsigned g_begin, unsigned g_end, void *s)
{
	if (!pl || !p) return fail(TSPWS_E_ARG, "stack_finish_range: NULL");
	if (!is_two_stage(p, mtr_global) || g_begin > g_end || g_end > p->Kmax) return fail(TSPWS_E_ARG, "stack_finish_range: two-stage calls, 0 <= g_begin <= g_end <= Kmax");
	HIP_TRY(hipSetDevice(pl->device));
	double *OUT, *ST, *PS, *P;
	size_t nd;
	int rc;
	if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &P, &nd))) return rc;
	return tspws_stacks_f64(pl, P + (size_t)g_begin * pl->N, g_end - g_begin, pl->N, ST, PS, S_(s), g_begin != 0, nullptr, nullptr, ScaleRange());
}

// weight (unless the accumulation already wrote OUT), both inverse transforms, epilogue (ts_pws1f_lib.c:226-241)
static int finish_tail(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, float *d_ls, float *d_ts, void *s, bool weighted)
{
	HIP_TRY(hipSetDevice(pl->device));
	double *OUT, *ST, *PS;
	int rc;
	if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
	const unsigned K = is_two_stage(p, mtr_global) ? p->Kmax : (unsigned)mtr_global;
	if (!weighted && (rc = tspws_hip_weight(pl, OUT, ST, PS, K, (unsigned)mtr_global, p->wu, p->unbiased, s))) return rc;
	return tspws_inverse_pair_out(pl, (const double2 *)OUT, d_ts, d_ls, (float)(unsigned)mtr_global, S_(s));
}

extern "C" int tspws_hip_stack_finish_tail(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, float *d_ls, float *d_ts, void *s)
{
	if (!pl || !p) return fail(TSPWS_E_ARG, "stack_finish_tail: NULL");
	return finish_tail(pl, p, mtr_global, d_ls, d_ts, s, false);
}

extern "C" int tspws_hip_stack_finish(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, float *d_ls, float *d_ts, void *s)
{
	if (!pl || !p) return fail(TSPWS_E_ARG, "stack_finish: NULL");
	HIP_TRY(hipSetDevice(pl->device));
	int rc;
	double *OUT, *ST, *PS, *B;
	size_t nd;
	if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &B, &nd))) return rc;
	if (is_two_stage(p, mtr_global)) {
		// all Kmax partial stacks at once: the launch that completes ST / PS also writes the weighted coefficients
		const WeightArgs wa = weight_args(p, OUT, p->Kmax, mtr_global);
		bool weighted = false;
		if ((rc = tspws_stacks_f64(pl, B, p->Kmax, pl->N, ST, PS, S_(s), false, &wa, &weighted, ScaleRange()))) return rc;
		return finish_tail(pl, p, mtr_global, d_ls, d_ts, s, weighted);
	}
	HIP_TRY(hipMemcpyAsync(ST, B, 4 * pl->ncoef * sizeof(double), hipMemcpyDeviceToDevice, S_(s)));
	return finish_tail(pl, p, mtr_global, d_ls, d_ts, s, false);
}

// ------------------------------------------------------------------------------------------
// Scale-sharded finish stage (multi-GPU).  After the all-reduce every rank holds the same K partial stacks; instead of
// finishing redundantly, rank r transforms, weights and reconstructs only ITS share of the scales -- the reconstruction
// is a sum over scales -- and the ranks add their partial reconstructions (2 N doubles) before the epilogue.
// ------------------------------------------------------------------------------------------
static bool finish_shardable(const tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global)
{
	return is_two_stage(p, mtr_global) && !tspws_generic_forward() && !tspws_generic_inverse() && pl->inv_noct && !pl->inv_ngeneric &&
	       !tspws_many_trace_path(pl, p->Kmax);
}

// Contiguous, work-balanced share of the scales for `rank` of `world`: whole decimation octaves (the inverse sums the voices
// of an octave in registers), cost = weighted forward MACs.  Returns 0 with [*s_begin, *s_end) (possibly empty), 1 when this plan /
// parameter set has no sharded finish (the caller then finishes as a whole), an error code for bad arguments.
extern "C" int tspws_hip_finish_shard(const tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, unsigned rank, unsigned world, unsigned *s_begin,
                                      unsigned *s_end)
{
	if (!pl || !p || !s_begin || !s_end || !world || rank >= world) return fail(TSPWS_E_ARG, "finish_shard: bad argument");
	if (!finish_shardable(pl, p, mtr_global)) return 1;
	const std::vector<unsigned> &og_s0 = pl->og_s0, &og_nv = pl->og_nv; // the decimation octaves in scale order
	std::vector<unsigned> items(og_s0.size());
	for (unsigned i = 0; i < items.size(); i++) items[i] = i;
	std::vector<double> cost(items.size());
	double total = 0;
	for (size_t k = 0; k < items.size(); k++) {
		double c = 0;
		// forward MACs.  Shares of one or two octaves (world >= 4) are latency-bound -- a launch lasts as long as its
		// longest workgroup -- and the far-decimated octaves then cost about twice as much (tools/shard_finish_timing.py, per
		// octave on the north-star frame: 13 us above the launch-chain floor for D <= 128, ~30 us for D >= 256); shares of
		// many octaves (world 2: 0.148 | 0.157 ms for an even split of the MACs) follow the MAC count
		const double far_w = world >= 4 ? 2.0 : 1.0;
		for (unsigned s = og_s0[items[k]]; s < og_s0[items[k]] + og_nv[items[k]]; s++)
			c += (double)pl->sc[s].L * (double)pl->sc[s].Ns * (pl->sc[s].D >= 256 ? far_w : 1.0);
		cost[k] = c; total += c;
	}
	// item k goes to rank floor(world * (cum_before + cost / 2) / total): contiguous, balanced, deterministic on every rank
	unsigned lo = ~0u, hi = 0;
	double cum = 0;
	for (size_t k = 0; k < items.size(); k++) {
		unsigned owner = total > 0 ? (unsigned)((double)world * (cum + 0.5 * cost[k]) / total) : 0u;
		if (owner >= world) owner = world - 1;
		cum += cost[k];
		if (owner != rank) continue;
		if (lo == ~0u) lo = og_s0[items[k]];
		hi = og_s0[items[k]] + og_nv[items[k]];
	}
	*s_begin = lo == ~0u ? 0u : lo;
	*s_end = lo == ~0u ? 0u : hi;
	return 0;
}

extern "C" int tspws_hip_stack_finish_scales(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, unsigned s_begin, unsigned s_end, double *d_x2,
                                             void *s)
{
	if (!pl || !p || !d_x2) return fail(TSPWS_E_ARG, "stack_finish_scales: NULL");
	if (!finish_shardable(pl, p, mtr_global)) return fail(TSPWS_E_ARG, "stack_finish_scales: no sharded finish for this plan / parameter set (tspws_hip_finish_shard tells)");
	if (s_begin > s_end || s_end > pl->S) return fail(TSPWS_E_ARG, "stack_finish_scales: 0 <= s_begin <= s_end <= S");
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	if (s_begin == s_end) { HIP_TRY(hipMemsetAsync(d_x2, 0, 2 * (size_t)pl->N * sizeof(double), st)); return 0; }
	bool lo_ok = false, hi_ok = false; // whole octaves only
	for (size_t i = 0; i < pl->og_s0.size(); i++) { lo_ok |= pl->og_s0[i] == s_begin; hi_ok |= pl->og_s0[i] + pl->og_nv[i] == s_end; }
	if (!lo_ok || !hi_ok) return fail(TSPWS_E_ARG, "stack_finish_scales: the range must consist of whole decimation octaves");
	double *OUT, *ST, *PS, *P;
	size_t nd;
	int rc;
	if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &P, &nd))) return rc;
	const WeightArgs wa = weight_args(p, OUT, p->Kmax, mtr_global);
	bool weighted = false;
	ScaleRange rg; rg.s0 = s_begin; rg.s1 = s_end; // the finish-stage launches below cover these scales only
	if ((rc = tspws_stacks_f64(pl, P, p->Kmax, pl->N, ST, PS, st, false, &wa, &weighted, rg))) return rc;
	// (several forward batches: weight afterwards -- over all coefficients; those of other scales are never read)
	if (!weighted && (rc = tspws_hip_weight(pl, OUT, ST, PS, p->Kmax, (unsigned)mtr_global, p->wu, p->unbiased, s))) return rc;
	return tspws_inverse_scales(pl, (const double2 *)OUT, d_x2, st, rg);
}

// ------------------------------------------------------------------------------------------
// Whole call on one GPU = stack_local + stack_finish, with optional HIP events on the caller's stream around the call and
// around its streaming stage (bench.py: roofline of the streaming kernel, per-call durations).
//
// Measured and not kept (docs/history/round-2-3.md, profiles/r03_overlap_experiments.txt): transforming finished groups on a
// second stream while the next ones are streamed -- with one launch per hand-over or with ONE persistent streaming kernel
// that releases the groups through hipStreamWaitValue32 -- shortens the tail by 0.1 ms and slows the streaming pass by as much.
// ------------------------------------------------------------------------------------------
extern "C" int tspws_hip_profile_begin(tspws_hip_plan *pl, size_t max_calls)
{
	if (!pl) return fail(TSPWS_E_ARG, "profile_begin: NULL");
	HIP_TRY(hipSetDevice(pl->device));
	const size_t need = max_calls * 3;
	while (pl->prof_ev.size() < need) {
		hipEvent_t e;
		HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence)); // timing on, device-scope ordering only: a system-scope release would write the fresh partial stacks back out of L2 (8 us per call)
		pl->prof_ev.push_back(e);
	}
	pl->prof_used = 0;
	return 0;
}

extern "C" int tspws_hip_stream_launches(const tspws_hip_plan *pl) { return pl ? (int)pl->last_stream_launches : 0; }

extern "C" int tspws_hip_profile_read(tspws_hip_plan *pl, double *stage_ms, double *call_ms, size_t cap, size_t *ncalls)
{
	if (!pl || !ncalls) return fail(TSPWS_E_ARG, "profile_read: NULL");
	HIP_TRY(hipSetDevice(pl->device));
	HIP_TRY(hipDeviceSynchronize());
	const size_t n = pl->prof_used / 3;
	for (size_t i = 0; i < n && i < cap; i++) {
		float a = 0, b = 0;
		HIP_TRY(hipEventElapsedTime(&a, pl->prof_ev[3 * i], pl->prof_ev[3 * i + 1]));
		HIP_TRY(hipEventElapsedTime(&b, pl->prof_ev[3 * i], pl->prof_ev[3 * i + 2]));
		if (stage_ms) stage_ms[i] = a;
		if (call_ms) call_ms[i] = b;
	}
	*ncalls = n;
	return 0;
}

extern "C" int tspws_hip_profile_end(tspws_hip_plan *pl, double *mean_ms, size_t *ncalls)
{
	if (!pl || !mean_ms || !ncalls) return fail(TSPWS_E_ARG, "profile_end: NULL");
	size_t n = 0;
	int rc = tspws_hip_profile_read(pl, nullptr, nullptr, 0, &n);
	if (rc) return rc;
	std::vector<double> st(n);
	if ((rc = tspws_hip_profile_read(pl, st.data(), nullptr, n, &n))) return rc;
	double tot = 0;
	for (double v : st) tot += v;
	*mean_ms = n ? tot / (double)n : 0.0;
	*ncalls = n;
	pl->prof_used = 0;
	return 0;
}

static int hip_rc(hipError_t e, const char *what) { return e == hipSuccess ? 0 : tspws_fail(e == hipErrorOutOfMemory ? TSPWS_E_NOMEM : TSPWS_E_HIP, what, e); }

extern "C" int tspws_hip_stack(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, float *d_ls, float *d_ts,
                               void *s)
{
	if (!pl || !p || !d_x || !mtr) return fail(TSPWS_E_ARG, "stack: bad argument");
	HIP_TRY(hipSetDevice(pl->device));
	int rc;
	// Events of the call ride on its launches where they can (plan->le): start of the first and end of the last streaming launch
	// -- the latter is also what the side stream of the forward transforms waits for --, end of the launch that writes the outputs.
	// Whatever a path does not consume (single-stage calls, chunked passes, generic kernels) is recorded the ordinary way.
	const bool prof = pl->prof_used + 3 <= pl->prof_ev.size();
	hipEvent_t *pe = prof ? &pl->prof_ev[pl->prof_used] : nullptr;
	if (!pl->ev_fork) HIP_TRY(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence));
	pl->le = tspws_hip_plan::LaunchEvents();
	const bool ride = is_two_stage(p, mtr); // the streaming pass of a two-stage call takes both of its events (tspws_run_chunks)
	if (ride) { pl->le.first_start = prof ? pe[0] : nullptr; pl->le.last_stop = prof ? pe[1] : pl->ev_fork; }
	else if (prof) HIP_TRY(hipEventRecord(pe[0], S_(s)));
	if (!ride) {
		// single-stage on one device: the launch that completes ST / PS also writes the weighted coefficients -- no copy of the stacks
		// into the finish block, no weighting pass (the halves stack_local / stack_finish keep them for the all-reduce in between)
		double *OUT, *ST, *PS;
		bool weighted = false;
		if (!(rc = finish_block(pl, &OUT, &ST, &PS))) {
			const WeightArgs wa = weight_args(p, OUT, (unsigned)mtr, mtr);
			rc = tspws_stacks_f32(pl, d_x, mtr, ld, ST, PS, S_(s), false, &wa, &weighted, ScaleRange());
		}
		if (!rc && prof) rc = hip_rc(hipEventRecord(pe[1], S_(s)), "stack: event");
		pl->le.call_end = (!rc && prof) ? pe[2] : nullptr;
		if (!rc) rc = finish_tail(pl, p, mtr, d_ls, d_ts, s, weighted);
		if (!rc && prof) {
			if (pl->le.call_end) rc = hip_rc(hipEventRecord(pe[2], S_(s)), "stack: event");
			pl->prof_used += 3;
		}
		pl->le = tspws_hip_plan::LaunchEvents();
		return rc;
	}
	rc = tspws_hip_stack_local(pl, p, d_x, ld, mtr, 0, mtr, s);
	if (!rc) {
		if (prof && (!ride || pl->le.last_stop)) rc = hip_rc(hipEventRecord(pe[1], S_(s)), "stack: event");
		if (!rc && prof && pl->le.first_start) rc = hip_rc(hipEventRecord(pe[0], S_(s)), "stack: event"); // (never: both are consumed together)
		pl->le.first_start = pl->le.last_stop = nullptr;
		pl->le.call_end = prof ? pe[2] : nullptr;
	}
	if (!rc) rc = tspws_hip_stack_finish(pl, p, mtr, d_ls, d_ts, s);
	if (!rc && prof) {
		if (pl->le.call_end) rc = hip_rc(hipEventRecord(pe[2], S_(s)), "stack: event");
		pl->prof_used += 3;
	}
	pl->le = tspws_hip_plan::LaunchEvents();
	return rc;
}
