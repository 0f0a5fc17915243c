// inverse.hip -- phase weighting, inverse frame CWT (real part), epilogue.
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"
#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------
// phase weighting (tspws_biased :909-943, tspws_unbiased :965-984)
// mode 0: wu==2 biased, 1: wu==1, 2: general power, 3: unbiased (K>1)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_weight(double2 *__restrict__ OUT, const double2 *__restrict__ ST, const double2 *__restrict__ PS,
                                                size_t ncoef, int mode, double K, double M, double wu, const double *__restrict__ Mv,
                                                size_t y_out, size_t y_stack)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= ncoef) return;
	// blockIdx.y = independent stack (jackknife replica) with its own trace count M
	OUT += (size_t)blockIdx.y * y_out; ST += (size_t)blockIdx.y * y_stack; PS += (size_t)blockIdx.y * y_stack;
	if (Mv) M = Mv[blockIdx.y];
	OUT[i] = weight_value(ST[i], PS[i], mode, K, M, wu);
}

extern "C" int tspws_hip_weight(tspws_hip_plan *p, double *d_OUT, const double *d_ST, const double *d_PS, unsigned K, unsigned M,
                                double wu, int unbiased, void *s)
{
	if (!p || !d_OUT || !d_ST || !d_PS) return fail(TSPWS_E_ARG, "weight: NULL");
	HIP_TRY(hipSetDevice(p->device));
	const int mode = tspws_weight_mode(wu, unbiased, K);
	hipLaunchKernelGGL(k_weight, dim3((unsigned)((p->ncoef + 255) / 256)), dim3(256), 0, S_(s), (double2 *)d_OUT, (const double2 *)d_ST,
	                   (const double2 *)d_PS, p->ncoef, mode, (double)K, (double)M, wu, (const double *)nullptr, (size_t)0, (size_t)0);
	HIP_TRY(hipGetLastError());
	return 0;
}

int tspws_weight_mode(double wu, int unbiased, unsigned K)
{
	if (wu == 2 && unbiased && K != 1) return 3; // selection rule ts_pws1f_lib.c:226-228, K==1 falls back :972
	if (wu == 2) return 0;
	if (wu == 1) return 1;
	return 2;
}

void tspws_weight_batched(tspws_hip_plan *p, double2 *OUT, const double2 *ST, const double2 *PS, int mode, double K, double wu, const double *d_Mv,
                          unsigned nb, size_t y_out, size_t y_stack, hipStream_t st, double M)
{
	hipLaunchKernelGGL(k_weight, dim3((unsigned)((p->ncoef + 255) / 256), nb), dim3(256), 0, st, OUT, ST, PS, p->ncoef, mode, K, M, wu, d_Mv, y_out, y_stack);
}

// ------------------------------------------------------------------------------------------
// inverse frame transform, real part (gather form), generic:
//   x^[n] = sum_s gain_s * D_s * sum_{l: (n - cd + l) on the decimation grid} Re(conj(wd_s[l]) Y_s[.])
// with the grid restarting at the circular seam (cdotx.c:313-337); D==1 -> cdotx.c:176-211.
// One thread per output sample; NREC coefficient sets share the tap reads.
// ------------------------------------------------------------------------------------------
template <int NREC>
__global__ void __launch_bounds__(256) k_inverse_generic(const double2 *__restrict__ Y, size_t ncoef, unsigned N, const ScaleDesc *__restrict__ sc,
                                                         unsigned S, const double2 *__restrict__ wd, double *__restrict__ xout, int only_slow,
                                                         size_t y_coef, size_t y_out)
{
	const unsigned n = blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	Y += (size_t)blockIdx.y * y_coef; xout += (size_t)blockIdx.y * y_out; // blockIdx.y = independent reconstruction set
	double tot[NREC];
#pragma unroll
	for (int r = 0; r < NREC; r++) tot[r] = 0;
	for (unsigned s = 0; s < S; s++) {
		const ScaleDesc d = sc[s];
		if (only_slow && d.inv_fast) continue;
		const double2 *ws = wd + d.tap_off;
		const double2 *ys = Y + d.coef_off;
		long long n0 = (long long)n - d.cd;
		if (n0 < 0) n0 += N;
		const unsigned l0 = N - (unsigned)n0; // taps before the seam
		const unsigned lim = d.L < l0 ? d.L : l0;
		double acc[NREC];
#pragma unroll
		for (int r = 0; r < NREC; r++) acc[r] = 0;
		if (d.D > 1) {
			const unsigned D = d.D;
			unsigned l = (D - (unsigned)(n0 % D)) % D;
			unsigned q = (unsigned)((n0 + D - 1) / D);
			for (; l < lim; l += D, q++) {
				const double2 t = ws[l];
#pragma unroll
				for (int r = 0; r < NREC; r++) {
					const double2 y = ys[(size_t)r * ncoef + q];
					acc[r] = fma(t.x, y.x, fma(t.y, y.y, acc[r]));
				}
			}
			q = 0;
			for (l = l0; l < d.L; l += D, q++) {
				const double2 t = ws[l];
#pragma unroll
				for (int r = 0; r < NREC; r++) {
					const double2 y = ys[(size_t)r * ncoef + q];
					acc[r] = fma(t.x, y.x, fma(t.y, y.y, acc[r]));
				}
			}
#pragma unroll
			for (int r = 0; r < NREC; r++) tot[r] += d.gain * ((double)D * acc[r]);
		} else {
			for (unsigned l = 0; l < lim; l++) {
				const double2 t = ws[l];
#pragma unroll
				for (int r = 0; r < NREC; r++) {
					const double2 y = ys[(size_t)r * ncoef + (unsigned)n0 + l];
					acc[r] = fma(t.x, y.x, fma(t.y, y.y, acc[r]));
				}
			}
			for (unsigned l = lim; l < d.L; l++) {
				const double2 t = ws[l];
#pragma unroll
				for (int r = 0; r < NREC; r++) {
					const double2 y = ys[(size_t)r * ncoef + (l - l0)];
					acc[r] = fma(t.x, y.x, fma(t.y, y.y, acc[r]));
				}
			}
#pragma unroll
			for (int r = 0; r < NREC; r++) tot[r] += d.gain * acc[r];
		}
	}
#pragma unroll
	for (int r = 0; r < NREC; r++) xout[(size_t)r * N + n] = tot[r];
}

#include "inv_poly.h"

// octaves up to this decimation take the LDS-staged instantiation (TSPWS_INV_LDS_MAXD, default 1: the full-rate scales of the
// Mexican hat / `uni` frames, whose per-lane windows touch 64 cache lines per wave-load; 0: none)
static unsigned inv_lds_maxd()
{
	static int v = -1;
	if (v < 0) { const char *e = sweep_env("TSPWS_INV_LDS_MAXD"); v = e ? std::max(0, atoi(e)) : 1; }
	return (unsigned)v;
}

// work list of the polyphase inverse: one item group per run of consecutive scales with the same D (an octave: its voices are
// summed in registers).  SHORT frames -- all octave items together fewer than 768 waves, i.e. N < ~28 000 for the default frames --
// get one item per SCALE instead: the kernel is a chain of dependent round trips per voice, and with less than two waves per SIMD
// nothing hides them (N = 16501: 105 workgroups on 256 CUs); four times the waves, each a quarter as long, and the combining
// kernel adds a row per scale (TSPWS_INV_SPLIT=0 / 1 forces either form).  Long frames fill the chip with octave items (round 1:
// the voices on separate waves 49 vs 50 us at N = 131072).
int tspws_build_inverse(tspws_hip_plan *p)
{
	std::vector<OctDesc> oc;
	unsigned woff = 0;
	p->inv_ngeneric = 0;
	p->og_s0.clear(); p->og_nv.clear();
	bool split = false;
	{
		unsigned waves = 0;
		for (unsigned s = 0; s < p->S;) {
			unsigned e = s + 1;
			while (e < p->S && p->sc[e].D == p->sc[s].D) e++;
			p->og_s0.push_back(s); p->og_nv.push_back(e - s); // the octaves themselves: what a share of the sharded finish is made of
			const unsigned D = p->sc[s].D, dl = D >= 64 ? 64u : std::max(1u, 1u << (unsigned)ceil(log2((double)std::max(1u, D))));
			const unsigned NG = (p->sc[s].Ns + INV_R - 1) / INV_R, GW = 64 / dl;
			waves += (D > 64 ? (D + 63) / 64 : 1) * ((NG + GW - 1) / GW);
			s = e;
		}
		split = waves < 768; // (tools/experiments/inv_split.sh: 499 x 16501 two-stage 0.157 -> 0.134 ms, 64 x 8192 0.165 -> 0.158; N = 32768 / 65536 / 131072 unchanged or worse)
		if (const char *e = sweep_env("TSPWS_INV_SPLIT")) split = atoi(e) != 0;
	}
	for (unsigned s = 0; s < p->S;) {
		unsigned e = s + 1;
		while (!split && e < p->S && p->sc[e].D == p->sc[s].D) e++;
		OctDesc o;
		memset(&o, 0, sizeof o);
		o.gen = p->sc[s].inv_fast ? 0u : 1u;
		o.s0 = s; o.nv = e - s; o.D = p->sc[s].D; o.Ns = p->sc[s].Ns;
		unsigned dl = 1, lg = 0;
		while (dl < o.D && dl < 64) { dl <<= 1; lg++; }
		o.DL = dl; o.logDL = lg;
		o.MC = o.D > 64 ? (o.D + 63) / 64 : 1;
		const unsigned NG = (o.Ns + INV_R - 1) / INV_R, GW = 64 / o.DL;
		o.ngw = (NG + GW - 1) / GW;
		oc.push_back(o);
		s = e;
	}
	// octaves whose decimation divides N first: the two classes are launched separately (k_inv_poly<., GEN>)
	std::stable_sort(oc.begin(), oc.end(), [](const OctDesc &x, const OctDesc &y) { return x.gen < y.gen; });
	p->inv_waves_fast = 0; p->inv_waves_lds = 0;
	for (size_t i = 0; i < oc.size(); i++) {
		oc[i].wave_off = woff; oc[i].slot = (unsigned)i;
		woff += oc[i].MC * oc[i].ngw;
		if (!oc[i].gen) p->inv_waves_fast = woff;
		// the finely decimated octaves (D < 64) come first in scale order: their waves take the LDS-staged instantiation
		if (!oc[i].gen && oc[i].D <= inv_lds_maxd() && p->inv_waves_lds == oc[i].wave_off) p->inv_waves_lds = woff;
	}
	if (const char *e = sweep_env("TSPWS_INV_LDS")) if (*e == '0') p->inv_waves_lds = 0; // (A/B: the per-lane form for every octave)
	p->inv_waves = woff; p->inv_noct = (unsigned)oc.size();
	p->oc_s0.clear(); p->oc_nv.clear(); p->oc_wave_off.clear(); p->oc_nwaves.clear(); p->oc_gen.clear();
	for (const OctDesc &o : oc) {
		p->oc_s0.push_back(o.s0); p->oc_nv.push_back(o.nv); p->oc_wave_off.push_back(o.wave_off); p->oc_nwaves.push_back(o.MC * o.ngw);
		p->oc_gen.push_back(o.gen);
	}
	if (!oc.empty()) {
		HIP_TRY(hipMalloc(&p->d_oc, oc.size() * sizeof(OctDesc)));
		HIP_TRY(hipMemcpy(p->d_oc, oc.data(), oc.size() * sizeof(OctDesc), hipMemcpyHostToDevice));
	}
	return 0;
}

#if FL_ABLATE
// debug build only: only the octave classes of k_inv_poly named by the hex mask TSPWS_INV_CLASSES run (results wrong: timing ablation)
extern "C" int tspws_hip_inv_ablate(void)
{
	if (const char *e = sweep_env("TSPWS_INV_CLASSES")) {
		const unsigned m = (unsigned)strtoul(e, nullptr, 16);
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(inv_class_mask), &m, sizeof m));
	}
	return 0;
}
#endif

bool tspws_generic_inverse()
{
	static int v = -1;
	if (v < 0) { const char *e = sweep_env("TSPWS_INV_GENERIC"); v = (e && *e == '1') ? 1 : 0; }
	return v == 1;
}

// nb independent NREC-set reconstructions in one launch (grid.y): set j reads Y + j NREC ncoef, writes x + j NREC N
// f_ts / f_ls (NREC == 2, nb == 1 only): the stack's float outputs are written by the combining kernel itself (no FP64
// reconstructions in memory, no epilogue launch); x may then be NULL.
template <int NREC>
static int inverse_launch(tspws_hip_plan *p, const double2 *Y, double *x, hipStream_t st, unsigned nb = 1, float *f_ts = nullptr, float *f_ls = nullptr,
                          float f_mtr = 1.0f)
{
	const unsigned nbx = (p->N + 255) / 256;
	const size_t slot = (size_t)NREC * p->N;
	if (tspws_generic_inverse() || p->inv_noct == 0) {
		hipLaunchKernelGGL(k_inverse_generic<NREC>, dim3(nbx, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->S, p->d_wd, x, 0,
		                   (size_t)NREC * p->ncoef, slot);
		return 0;
	}
	const unsigned nslots = p->inv_noct + (p->inv_ngeneric ? 1 : 0);
	void *v;
	int rc = scratch(p, SCR_OBUF, (size_t)nb * nslots * slot * sizeof(double), &v);
	if (rc) return rc;
	double *obuf = (double *)v;
	// the LDS-staged octaves (D = 1: two 80-KB workgroups per CU) run BESIDE the others (latency-bound, eight waves per SIMD) on the
	// plan's side stream: one after the other they took 64 + 153 us for cfg4's twelve reconstructions, the single per-lane launch 262
	// (batched reconstructions only: a single pair is 206 vs 197 us that way -- the fork / join and two waves per SIMD cost more than
	// the one octave's coalescing gains)
	const unsigned lds_w = nb >= 2 ? p->inv_waves_lds : 0u;
	hipStream_t sl = st;
	const bool beside = lds_w && p->inv_waves_fast > lds_w;
	if (beside) {
		const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
		if (!p->side) HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
		if (!p->ev_fork) HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, evf));
		if (!p->ev_join) HIP_TRY(hipEventCreateWithFlags(&p->ev_join, evf));
		HIP_TRY(hipEventRecord(p->ev_fork, st));
		HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
		sl = p->side;
	}
	if (lds_w)
		hipLaunchKernelGGL((k_inv_poly<NREC, false, true>), dim3((lds_w + 3) / 4, nb), dim3(256), 0, sl, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, lds_w, (size_t)NREC * p->ncoef, (size_t)nslots * slot, 0u);
	if (beside) HIP_TRY(hipEventRecord(p->ev_join, sl));
	if (p->inv_waves_fast > lds_w)
		hipLaunchKernelGGL((k_inv_poly<NREC, false>), dim3((p->inv_waves_fast - lds_w + 3) / 4, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, p->inv_waves_fast, (size_t)NREC * p->ncoef, (size_t)nslots * slot, lds_w);
	if (p->inv_waves > p->inv_waves_fast)
		hipLaunchKernelGGL((k_inv_poly<NREC, true>), dim3((p->inv_waves - p->inv_waves_fast + 3) / 4, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc,
		                   p->d_oc, p->inv_noct, p->d_wd, obuf, slot, p->inv_waves, (size_t)NREC * p->ncoef, (size_t)nslots * slot, p->inv_waves_fast);
	if (p->inv_ngeneric)
		hipLaunchKernelGGL(k_inverse_generic<NREC>, dim3(nbx, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->S, p->d_wd,
		                   obuf + (size_t)p->inv_noct * slot, 1, (size_t)NREC * p->ncoef, (size_t)nslots * slot);
	if (beside) HIP_TRY(hipStreamWaitEvent(st, p->ev_join, 0));
	if (NREC == 2 && nb == 1 && (f_ts || f_ls))
		if (hipEvent_t e1 = p->le.call_end) { // the call's end event rides on its last launch (tspws_hip_stack)
			p->le.call_end = nullptr;
			hipExtLaunchKernelGGL(k_inv_combine_out, dim3(nbx, 2), dim3(256), 0, st, nullptr, e1, 0, (const double *)obuf, slot, nslots, (size_t)p->N, f_ts, f_ls, f_mtr);
		} else
			hipLaunchKernelGGL(k_inv_combine_out, dim3(nbx, 2), dim3(256), 0, st, obuf, slot, nslots, (size_t)p->N, f_ts, f_ls, f_mtr);
	else
		hipLaunchKernelGGL(k_inv_combine, dim3((unsigned)((slot + 255) / 256), nb), dim3(256), 0, st, obuf, slot, nslots, slot, x,
		                   (size_t)nslots * slot, slot);
	return 0;
}

extern "C" int tspws_hip_inverse(tspws_hip_plan *p, const double *d_Y, size_t nrec, double *d_x, void *s)
{
	if (!p || !d_Y || !d_x) return fail(TSPWS_E_ARG, "inverse: NULL");
	HIP_TRY(hipSetDevice(p->device));
	int rc;
	const size_t pairs = nrec / 2;
	for (size_t r = 0; r < pairs; r += 32768) { // pairs of coefficient sets share the tap reads; all pairs in one launch
		const unsigned nb = (unsigned)std::min<size_t>(pairs - r, 32768);
		if ((rc = inverse_launch<2>(p, (const double2 *)d_Y + 2 * r * p->ncoef, d_x + 2 * r * p->N, S_(s), nb))) return rc;
	}
	if (nrec & 1)
		if ((rc = inverse_launch<1>(p, (const double2 *)d_Y + (nrec - 1) * p->ncoef, d_x + (nrec - 1) * p->N, S_(s)))) return rc;
	HIP_TRY(hipGetLastError());
	return 0;
}

// Batched pairs of reconstructions in TWO halves (the one-pass stack + jackknife call, resample.hip): the octaves of the scales [0, s_split) --
// those the FIR forward kernels completed -- EARLY, on the stream `early` on which their weighted coefficient sets are complete, beside whatever
// the caller's stream still runs (the spectral chain of the other scales: its tail is a run of short, latency-bound kernels); the other octaves
// and the combining kernel LATE, on the caller's stream once it has been made to wait for `early`.  The octave items are independent (one row
// of the octave buffer each), so the halves are two wave ranges of the same launch list.  *done = false: this plan / call has no such split
// (generic octaves, a single reconstruction, octave items not in scale order) -- the caller takes tspws_hip_inverse.
static bool inv_split_point(const tspws_hip_plan *p, unsigned s_split, unsigned *w_split)
{
	if (tspws_generic_inverse() || !p->inv_noct || p->inv_ngeneric || p->inv_waves_fast != p->inv_waves) return false;
	unsigned w = p->inv_waves_fast;
	for (unsigned i = 0; i < p->inv_noct; i++) {
		if (i && p->oc_s0[i] < p->oc_s0[i - 1]) return false;
		if (p->oc_s0[i] < s_split && p->oc_s0[i] + p->oc_nv[i] > s_split) return false; // (an octave item across the split)
		if (p->oc_s0[i] >= s_split) { w = p->oc_wave_off[i]; break; }
	}
	*w_split = w;
	return w > 0 && w < p->inv_waves_fast && p->inv_waves_lds <= w;
}

int tspws_inverse_pairs_early(tspws_hip_plan *p, const double2 *Y, unsigned nb, unsigned s_split, hipStream_t early, bool *done)
{
	*done = false;
	unsigned w_split = 0;
	if (nb < 2 || !inv_split_point(p, s_split, &w_split)) return 0;
	const size_t slot = (size_t)2 * p->N;
	const unsigned nslots = p->inv_noct;
	void *v;
	int rc = scratch(p, SCR_OBUF, (size_t)nb * nslots * slot * sizeof(double), &v);
	if (rc) return rc;
	double *obuf = (double *)v;
	const unsigned lds_w = p->inv_waves_lds;
	if (lds_w && w_split > lds_w) { // the LDS-staged octaves (D = 1) beside the per-lane ones, as in inverse_launch
		const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
		if (!p->side) HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
		if (!p->ev_fork) HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, evf));
		if (!p->ev_join) HIP_TRY(hipEventCreateWithFlags(&p->ev_join, evf));
		HIP_TRY(hipEventRecord(p->ev_fork, early));
		HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
		hipLaunchKernelGGL((k_inv_poly<2, false, true>), dim3((lds_w + 3) / 4, nb), dim3(256), 0, p->side, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, lds_w, (size_t)2 * p->ncoef, (size_t)nslots * slot, 0u);
		HIP_TRY(hipEventRecord(p->ev_join, p->side));
		hipLaunchKernelGGL((k_inv_poly<2, false>), dim3((w_split - lds_w + 3) / 4, nb), dim3(256), 0, early, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, w_split, (size_t)2 * p->ncoef, (size_t)nslots * slot, lds_w);
		HIP_TRY(hipStreamWaitEvent(early, p->ev_join, 0));
	} else if (lds_w) {
		hipLaunchKernelGGL((k_inv_poly<2, false, true>), dim3((lds_w + 3) / 4, nb), dim3(256), 0, early, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, lds_w, (size_t)2 * p->ncoef, (size_t)nslots * slot, 0u);
	} else {
		hipLaunchKernelGGL((k_inv_poly<2, false>), dim3((w_split + 3) / 4, nb), dim3(256), 0, early, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, w_split, (size_t)2 * p->ncoef, (size_t)nslots * slot, 0u);
	}
	HIP_TRY(hipGetLastError());
	*done = true;
	return 0;
}

int tspws_inverse_pairs_late(tspws_hip_plan *p, const double2 *Y, double *x, unsigned nb, unsigned s_split, hipStream_t st)
{
	unsigned w_split = 0;
	if (nb < 2 || !inv_split_point(p, s_split, &w_split)) return fail(TSPWS_E_ARG, "inverse_pairs_late: no early half was launched");
	const size_t slot = (size_t)2 * p->N;
	const unsigned nslots = p->inv_noct;
	void *v;
	int rc = scratch(p, SCR_OBUF, (size_t)nb * nslots * slot * sizeof(double), &v);
	if (rc) return rc;
	double *obuf = (double *)v;
	hipLaunchKernelGGL((k_inv_poly<2, false>), dim3((p->inv_waves_fast - w_split + 3) / 4, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
	                   p->d_wd, obuf, slot, p->inv_waves_fast, (size_t)2 * p->ncoef, (size_t)nslots * slot, w_split);
	hipLaunchKernelGGL(k_inv_combine, dim3((unsigned)((slot + 255) / 256), nb), dim3(256), 0, st, obuf, slot, nslots, slot, x, (size_t)nslots * slot, slot);
	HIP_TRY(hipGetLastError());
	return 0;
}

// Two reconstructions (sets 0 and 1 of Y) from the scales [s_lo, s_hi) ONLY -- whole decimation octaves -- as FP64 partial
// sums x2[2][N]: the reconstruction is a sum over scales, so the shares of disjoint scale ranges add up to the whole.
int tspws_inverse_scales(tspws_hip_plan *p, const double2 *Y, double *x2, hipStream_t st, ScaleRange rg)
{
	const unsigned s_lo = rg.s0, s_hi = rg.s1;
	const size_t slot = 2 * (size_t)p->N;
	void *v;
	int rc = scratch(p, SCR_OBUF, (size_t)p->inv_noct * slot * sizeof(double), &v);
	if (rc) return rc;
	double *obuf = (double *)v;
	// the octave items are stored class by class (decimation divides N first), in scale order inside a class: the items
	// of a scale range are one contiguous run per class
	unsigned first[2] = {~0u, ~0u}, last[2] = {0, 0};
	for (unsigned i = 0; i < p->inv_noct; i++) {
		if (p->oc_s0[i] < s_lo || p->oc_s0[i] >= s_hi) continue;
		const unsigned c = p->oc_gen[i] ? 1u : 0u;
		if (first[c] == ~0u) first[c] = i;
		last[c] = i;
	}
	for (unsigned c = 0; c < 2; c++) {
		if (first[c] == ~0u) continue;
		const unsigned w0 = p->oc_wave_off[first[c]], w1 = p->oc_wave_off[last[c]] + p->oc_nwaves[last[c]];
		if (c == 0)
			hipLaunchKernelGGL((k_inv_poly<2, false>), dim3((w1 - w0 + 3) / 4, 1), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct, p->d_wd,
			                   obuf, slot, w1, (size_t)2 * p->ncoef, (size_t)p->inv_noct * slot, w0);
		else
			hipLaunchKernelGGL((k_inv_poly<2, true>), dim3((w1 - w0 + 3) / 4, 1), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct, p->d_wd,
			                   obuf, slot, w1, (size_t)2 * p->ncoef, (size_t)p->inv_noct * slot, w0);
	}
	const unsigned a0 = first[0] == ~0u ? 0u : first[0], na = first[0] == ~0u ? 0u : last[0] - first[0] + 1;
	const unsigned b0 = first[1] == ~0u ? 0u : first[1], nb = first[1] == ~0u ? 0u : last[1] - first[1] + 1;
	hipLaunchKernelGGL(k_inv_combine_ranges, dim3((unsigned)((slot + 255) / 256)), dim3(256), 0, st, (const double *)obuf, slot, a0, na, b0, nb, slot, x2);
	HIP_TRY(hipGetLastError());
	return 0;
}

// epilogue, ts_pws1f_lib.c:233-241 (ls is a FLOAT division by the converted trace count)
__global__ void __launch_bounds__(256) k_epilogue(float *__restrict__ ls, float *__restrict__ ts, const double *__restrict__ xst,
                                                  const double *__restrict__ xout, size_t N, float mtr)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const size_t o = (size_t)blockIdx.y * N; // blockIdx.y = row of a batch of outputs
	if (ls) ls[o + n] = (float)xst[o + n] / mtr;
	if (ts) ts[o + n] = (float)xout[o + n];
}

extern "C" int tspws_hip_epilogue(float *d_ls, float *d_ts, const double *d_xst, const double *d_xout, size_t N, unsigned mtr, void *s)
{
	if ((d_ls && !d_xst) || (d_ts && !d_xout)) return fail(TSPWS_E_ARG, "epilogue: NULL");
	hipLaunchKernelGGL(k_epilogue, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, S_(s), d_ls, d_ts, d_xst, d_xout, N, (float)mtr);
	HIP_TRY(hipGetLastError());
	return 0;
}

void tspws_epilogue_rows(float *d_ts, const double *d_x, size_t N, unsigned nb, hipStream_t st)
{
	hipLaunchKernelGGL(k_epilogue, dim3((unsigned)((N + 255) / 256), nb), dim3(256), 0, st, (float *)nullptr, d_ts, (const double *)nullptr, d_x, N, 1.0f);
}

// The stack's two reconstructions: set 0 = ICWT(OUT) -> tsPWS, set 1 = ICWT(ST) -> ls (float division by the trace count); the
// combining kernel of the polyphase inverse writes the floats itself.
int tspws_inverse_pair_out(tspws_hip_plan *p, const double2 *Y, float *d_ts, float *d_ls, float mtr, hipStream_t st)
{
	int rc;
	if (!tspws_generic_inverse() && p->inv_noct) {
		if ((rc = inverse_launch<2>(p, Y, nullptr, st, 1, d_ts, d_ls, mtr))) return rc;
		HIP_TRY(hipGetLastError());
		return 0;
	}
	void *v;
	if ((rc = scratch(p, SCR_X2, 2 * (size_t)p->N * sizeof(double), &v))) return rc;
	double *x2 = (double *)v;
	if ((rc = tspws_hip_inverse(p, (const double *)Y, 2, x2, (void *)st))) return rc; // row 0 = ICWT(OUT), row 1 = ICWT(ST)
	hipLaunchKernelGGL(k_epilogue, dim3((unsigned)((p->N + 255) / 256)), dim3(256), 0, st, d_ls, d_ts, (const double *)(x2 + p->N), (const double *)x2, (size_t)p->N, mtr);
	HIP_TRY(hipGetLastError());
	return 0;
}
