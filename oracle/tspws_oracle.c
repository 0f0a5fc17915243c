/*
 * tspws_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C (C11, 64-bit indices) restatement of the reference ts-PWS stacking
 * path, written from the algorithm, not from the reference text.  Nothing in
 * the product (ts-pws_amd/, the CLI, libtspws_hip.so) may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Pinning: the reference ships no golden vectors, so the
 * oracle is pinned against outputs of the reference itself built in place from
 * /root/reference/src by oracle/Makefile into oracle/_ref/ (see
 * tests/golden/make_golden.py and tests/test_oracle_vs_golden.py) and against
 * the known-answer values of SURVEY.md 8(c).
 *
 * Every routine cites the reference lines whose behaviour it reproduces
 * (paths relative to /root/reference/src).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/ts_pws1f_lib.h"
#include "tspws_oracle.h"

typedef double complex cplx;

struct orc_frame {
	int       type;
	unsigned  S, V, N;
	double    Cpsi;
	double   *scale;
	unsigned *L, *D, *Ns;
	int      *c, *cd;
	size_t   *tap_off;  /* S+1 */
	size_t   *coef_off; /* S+1 */
	cplx     *w, *wd;
};

/* --------------------------------------------------------------------------
 * Frame geometry.  FWTa/wavelet_def_v7.c:299-309 (scales by repeated product),
 * :312-322 (support = +-5 sigma, clipped to N), :325-339 (dyadic decimation
 * table, truncation towards zero, floor of 1), wavelet_v7.c:61 (N_s).
 * -------------------------------------------------------------------------- */
static void frame_geometry(orc_frame *f, unsigned J, double s0, double b0, int uni)
{
	const unsigned S = f->S, V = f->V, N = f->N;
	double sc = s0;
	const double ratio = pow(2.0, 1.0 / (double)V);
	for (unsigned s = 0; s < S; s++) { f->scale[s] = sc; sc *= ratio; }

	for (unsigned s = 0; s < S; s++) {
		unsigned len = 2u * (unsigned)ceil(5.0 * f->scale[s]) + 1u;
		f->L[s]  = len > N ? N : len;
		f->c[s]  = (int)(f->L[s] / 2u);
		f->cd[s] = (int)f->L[s] - 1 - f->c[s];
	}
	if (uni) {
		for (unsigned s = 0; s < S; s++) f->D[s] = 1;
	} else {
		double step = s0 * b0;
		unsigned s = 0;
		for (unsigned j = 0; j < J; j++) {
			unsigned d = step < 1.0 ? 1u : (unsigned)step;
			for (unsigned v = 0; v < V; v++) f->D[s++] = d;
			step *= 2.0;
		}
	}
	f->tap_off[0] = f->coef_off[0] = 0;
	for (unsigned s = 0; s < S; s++) {
		f->Ns[s] = (N + f->D[s] - 1u) / f->D[s];
		f->tap_off[s + 1]  = f->tap_off[s] + f->L[s];
		f->coef_off[s + 1] = f->coef_off[s] + f->Ns[s];
	}
}

/* 500-term Maclaurin series of erfi.  wavelet_def_v7.c:104-117 */
static double erfi_series(double z)
{
	const double zz = z * z;
	double term = z, sum = z;
	for (unsigned n = 1; n < 500; n++) {
		term *= zz / n;
		sum += term / (2 * n + 1);
	}
	return sum * (2 / sqrt(PI));
}

/* Tap tables.  Morlet :38-52, exact Morlet :71-87, complex Mexican hat
 * :119-131.  Note the argument is u*(1/scale), not u/scale (Appendix A.6). */
static void frame_taps(orc_frame *f, double w0)
{
	for (unsigned s = 0; s < f->S; s++) {
		cplx *w = f->w + f->tap_off[s];
		const unsigned L = f->L[s];
		const int u0 = (int)(L / 2u);
		const double sc = f->scale[s], inv = 1 / sc;
		if (f->type == -1) {
			const double k = 1 / sqrt(sqrt(PI) * sc);
			for (int u = -u0; u < (int)L - u0; u++) {
				double t = inv * u, ph = w0 * t;
				t *= t;
				*w++ = k * cexp(I * ph) * exp(-0.5 * t);
			}
		} else if (f->type == -2) {
			const double ze = exp((-w0 * w0) / 2);
			const double k = 1 / sqrt(sqrt(PI) * sc);
			for (int u = -u0; u < (int)L - u0; u++) {
				double t = inv * u, ph = w0 * t;
				t *= t;
				t = k * exp(-0.5 * t);
				*w++ = t * (cexp(I * ph) - ze);
			}
		} else { /* -3 */
			const double k = 2 / sqrt(3 * sqrt(PI) * sc);
			for (int u = -u0; u < (int)L - u0; u++) {
				double t = inv * u;
				*w++ = k * ((t * t - 1) * exp(-t * t / 2) * (1 + I * erfi_series(t / sqrt(2))) - I * sqrt(2 / PI) * t);
			}
		}
	}
	/* dual frame = conjugate, time reversed.  wavelet_def_v7.c:152-188 */
	for (unsigned s = 0; s < f->S; s++) {
		const cplx *w = f->w + f->tap_off[s];
		cplx *wd = f->wd + f->tap_off[s];
		const unsigned L = f->L[s];
		for (unsigned l = 0; l < L; l++) wd[l] = conj(w[L - 1 - l]);
	}
}

/* Admissibility constant.  Morlet: wavelet_def_v7.c:133-144 (Riemann sum on an
 * accumulated double, reproduced literally); Mexican hat :146-148. */
static double frame_cpsi(int type, double w0)
{
	if (type == -3) return (4. / 3.) * sqrt(PI);
	double acc = 0;
	for (double om = 0.01; om < 100; om += 0.01) {
		double d = om - w0;
		d *= d;
		acc += exp(-d) / om;
	}
	return acc * (0.01 * sqrt(PI) / 2);
}

/* CreateWaveletFamily, wavelet_def_v7.c:204-295 (complex families only). */
orc_frame *orc_frame_create(int type, unsigned J, unsigned V, unsigned N, double s0, double b0, double w0, int uni)
{
	if (type > -1 || type < -3 || V == 0) return NULL;
	orc_frame *f = calloc(1, sizeof *f);
	if (!f) return NULL;
	const unsigned S = J * V;
	f->type = type; f->S = S; f->V = V; f->N = N;
	f->scale = malloc((S + 1) * sizeof(double));
	f->L = malloc((S + 1) * sizeof(unsigned)); f->D = malloc((S + 1) * sizeof(unsigned));
	f->Ns = malloc((S + 1) * sizeof(unsigned));
	f->c = malloc((S + 1) * sizeof(int)); f->cd = malloc((S + 1) * sizeof(int));
	f->tap_off = malloc((S + 1) * sizeof(size_t)); f->coef_off = malloc((S + 1) * sizeof(size_t));
	frame_geometry(f, J, s0, uni ? 1.0 : b0, uni);
	f->w  = malloc((f->tap_off[S] + 1) * sizeof(cplx));
	f->wd = malloc((f->tap_off[S] + 1) * sizeof(cplx));
	frame_taps(f, w0);
	f->Cpsi = frame_cpsi(type, w0);
	return f;
}

void orc_frame_destroy(orc_frame *f)
{
	if (!f) return;
	free(f->scale); free(f->L); free(f->D); free(f->Ns); free(f->c); free(f->cd);
	free(f->tap_off); free(f->coef_off); free(f->w); free(f->wd); free(f);
}

unsigned orc_frame_S(const orc_frame *f) { return f->S; }
size_t orc_frame_ncoef(const orc_frame *f) { return f->coef_off[f->S]; }
size_t orc_frame_ntaps(const orc_frame *f) { return f->tap_off[f->S]; }
double orc_frame_cpsi(const orc_frame *f) { return f->Cpsi; }

void orc_frame_tables(const orc_frame *f, double *scale, unsigned *L, int *c, int *cd, unsigned *D, unsigned *Ns)
{
	for (unsigned s = 0; s < f->S; s++) {
		scale[s] = f->scale[s]; L[s] = f->L[s]; c[s] = f->c[s]; cd[s] = f->cd[s]; D[s] = f->D[s]; Ns[s] = f->Ns[s];
	}
}

void orc_frame_taps(const orc_frame *f, double *w, double *wd)
{
	memcpy(w, f->w, f->tap_off[f->S] * sizeof(cplx));
	memcpy(wd, f->wd, f->tap_off[f->S] * sizeof(cplx));
}

/* --------------------------------------------------------------------------
 * Forward frame transform of one real trace.
 * wavelet_v7.c:43-64 -> cdotx.c:35-72:  Y_s[k] = conj( sum_l x[(kD - c + l) mod N] w_s[l] )
 * The window wraps at most once because L <= N.
 * -------------------------------------------------------------------------- */
void orc_forward(const orc_frame *f, const double *x, double *Yout)
{
	cplx *Y = (cplx *)Yout;
	const size_t N = f->N;
	for (unsigned s = 0; s < f->S; s++) {
		const cplx *w = f->w + f->tap_off[s];
		cplx *y = Y + f->coef_off[s];
		const size_t L = f->L[s], D = f->D[s], c = (size_t)f->c[s];
		for (size_t k = 0; k < f->Ns[s]; k++) {
			const size_t n0 = (N + k * D - c) % N;
			const size_t l0 = (N - n0 < L) ? N - n0 : L;
			cplx acc = 0;
			const double *xp = x + n0;
			for (size_t l = 0; l < l0; l++) acc += xp[l] * w[l];
			for (size_t l = l0; l < L; l++) acc += x[l - l0] * w[l];
			y[k] = conj(acc);
		}
	}
}

/* --------------------------------------------------------------------------
 * Real part of the inverse frame transform.
 * wavelet_v7.c:124-150; per scale cdotx.c:305-340 (D>1, zero-stuffed grid that
 * restarts at the circular seam) or cdotx.c:176-211 (D==1); gain
 * ln2/(2 Cpsi V scale), wavelet_v7.c:145.
 * -------------------------------------------------------------------------- */
void orc_inverse(const orc_frame *f, const double *Yin, double *xrec)
{
	const cplx *Y = (const cplx *)Yin;
	const size_t N = f->N;
	double *bf = malloc(N * sizeof(double));
	for (size_t n = 0; n < N; n++) xrec[n] = 0;
	for (unsigned s = 0; s < f->S; s++) {
		const cplx *wd = f->wd + f->tap_off[s];
		const cplx *y = Y + f->coef_off[s];
		const size_t L = f->L[s], D = f->D[s], cd = (size_t)f->cd[s];
		for (size_t n = 0; n < N; n++) {
			const size_t n0 = (N + n - cd) % N;   /* position of tap 0 */
			const size_t l0 = N - n0;             /* taps before the seam */
			double acc = 0;
			if (D > 1) {
				/* taps landing on grid points p = n0 + l, D | p, p < N */
				size_t l = (D - n0 % D) % D;
				size_t q = (n0 + D - 1) / D;
				const size_t lim = L < l0 ? L : l0;
				for (; l < lim; l += D, q++)
					acc += creal(wd[l]) * creal(y[q]) + cimag(wd[l]) * cimag(y[q]);
				/* after the seam the grid restarts at sample 0 */
				q = 0;
				for (l = l0; l < L; l += D, q++)
					acc += creal(wd[l]) * creal(y[q]) + cimag(wd[l]) * cimag(y[q]);
				bf[n] = (double)D * acc;
			} else {
				const size_t lim = L < l0 ? L : l0;
				for (size_t l = 0; l < lim; l++)
					acc += creal(wd[l]) * creal(y[n0 + l]) + cimag(wd[l]) * cimag(y[n0 + l]);
				for (size_t l = lim; l < L; l++)
					acc += creal(wd[l]) * creal(y[l - l0]) + cimag(wd[l]) * cimag(y[l - l0]);
				bf[n] = acc;
			}
		}
		const double gain = log(2.0) / (2 * f->Cpsi * f->V * f->scale[s]);
		for (size_t n = 0; n < N; n++) xrec[n] += gain * bf[n];
	}
	free(bf);
}

/* --------------------------------------------------------------------------
 * Stack accumulation: ST += Y ; PS += Y/|Y| unless the quotient is not a unit
 * phasor (NaN for Y == 0).   ts_pws1f_lib.c:486-494, :897-904
 * -------------------------------------------------------------------------- */
static void accumulate(cplx *ST, cplx *PS, const cplx *Y, size_t ncoef)
{
#pragma omp parallel for schedule(static)
	for (size_t i = 0; i < ncoef; i++) {
		cplx v = Y[i];
		ST[i] += v;
		v /= cabs(v);
		if (creal(v * conj(v)) <= 1.001) PS[i] += v;
	}
}

/* Phase weighting.  biased: ts_pws1f_lib.c:909-943; unbiased: :965-984 */
static void weight_biased(cplx *OUT, const cplx *ST, const cplx *PS, size_t ncoef, unsigned K, unsigned M, double wu)
{
	if (wu == 2) {
		double g = (double)K;
		g = 1. / (g * g * (double)M);
		for (size_t i = 0; i < ncoef; i++) {
			double a = (creal(PS[i]) * creal(PS[i]) + cimag(PS[i]) * cimag(PS[i])) * g;
			OUT[i] = a * ST[i];
		}
	} else if (wu == 1) {
		const double g = 1. / ((double)K * (double)M);
		for (size_t i = 0; i < ncoef; i++) OUT[i] = ST[i] * cabs(PS[i]) * g;
	} else {
		for (size_t i = 0; i < ncoef; i++) {
			double a = cabs(PS[i]) / K;
			a = pow(a, wu);
			OUT[i] = ST[i] * a / M;
		}
	}
}

static void weight_unbiased(cplx *OUT, const cplx *ST, const cplx *PS, size_t ncoef, unsigned K, unsigned M)
{
	if (K == 1) { weight_biased(OUT, ST, PS, ncoef, K, M, 2); return; }
	const double iK = 1. / (double)K, iK1 = 1. / (double)(K - 1), iM = 1. / (double)M;
	for (size_t i = 0; i < ncoef; i++) {
		cplx p = PS[i] * iK;
		double a = creal(p) * creal(p) + cimag(p) * cimag(p);
		a = (K * a - 1) * iK1;
		OUT[i] = ST[i] * a * iM;
	}
}

static void weight(cplx *OUT, const cplx *ST, const cplx *PS, size_t ncoef, unsigned K, unsigned M, double wu, int unbiased)
{
	/* selection rule ts_pws1f_lib.c:226-228 */
	if (wu == 2 && unbiased) weight_unbiased(OUT, ST, PS, ncoef, K, M);
	else weight_biased(OUT, ST, PS, ncoef, K, M, wu);
}

/* Stage 1 of the two-stage stack: contiguous groups in trace order.
 * ts_pws1f_lib.c:866-881 */
void orc_partial_stacks(double *P, const float *sigall, size_t max, size_t mtr, unsigned Kmax)
{
	memset(P, 0, (size_t)Kmax * max * sizeof(double));
	for (size_t i = 0; i < mtr; i++) {
		const size_t g = (size_t)floor((double)(i * Kmax) / (double)mtr);
		double *p = P + g * max;
		const float *x = sigall + i * max;
		for (size_t n = 0; n < max; n++) p[n] += (double)x[n];
	}
}

/* Stage 2: transform each partial and accumulate.  ts_pws1f_lib.c:885-906 */
static void stacks_of_doubles(const orc_frame *f, cplx *ST, cplx *PS, cplx *Y, const double *P, size_t max, unsigned K)
{
	const size_t nc = orc_frame_ncoef(f);
	memset(ST, 0, nc * sizeof(cplx));
	memset(PS, 0, nc * sizeof(cplx));
	for (unsigned g = 0; g < K; g++) {
		orc_forward(f, P + (size_t)g * max, (double *)Y);
		accumulate(ST, PS, Y, nc);
	}
}

/* Parameter resolution, ts_pws1f_lib.c:91-124 (mutates *p exactly as there). */
void orc_resolve(t_tsPWS *p, unsigned nsamp, float dt)
{
	if (p->fmin != 0 && p->fmin < 1 / (dt * nsamp)) {
		printf("Warning: fmin is too low. Replaced by the default value.\n");
		p->fmin = 0;
	}
	if (p->w0set == 1) p->w0 = 2 * sqrt(log(2)) * p->Q;
	else if (p->w0set == 2) p->w0 = PI / sqrt(log(2)) * p->cycle;
	if (p->type == -1 || p->type == -2) {
		const double rel = p->w0 / (PI * sqrt(2 / log(2)));
		if (!p->lVfix)  p->V  = (unsigned)ceil(4. * rel);
		if (!p->lb0fix) p->b0 = (unsigned)pow(2, round(log2(rel)));
		if (!p->ls0fix) p->s0 = 2.;
	}
	if (p->type == -3) {
		p->w0 = sqrt(2);
		if (!p->lVfix)  p->V  = 2;
		if (!p->lb0fix) p->b0 = 0.5;
		if (!p->ls0fix) p->s0 = 1.;
	}
	if (p->fmin) {
		double a = p->w0 / (2 * PI * dt * p->fmin);
		if (p->J) {
			a /= pow(2, p->J - 1 / (double)p->V);
			while (a < p->s0 * 0.9) { a *= 2; p->J--; }
			p->s0 = a;
		} else p->J = (unsigned)floor(log2(a / p->s0) + 1 / (double)p->V);
	} else if (!p->J) {
		const double a = nsamp * p->w0 / (2 * PI * 4. * p->s0);
		p->J = (unsigned)floor(log2(a) + 1 / (double)p->V);
	}
}

/* Jackknife deletion masks.  ts_pws1f_lib.c:385-430: bin = floor(yday*n/365),
 * all C(n,d) deletions in lexicographic order, sel = 0 when the trace's bin is
 * deleted.  Returns 0, or -2 when no start times are available. */
int orc_jackknife_plan(char *sel, const time_t *tm, size_t mtr, unsigned d, unsigned n, unsigned C)
{
	if (!sel || !tm) return 1;
	if (tm[0] == 0) return -2;
	unsigned *bin = malloc(mtr * sizeof(unsigned));
	unsigned *comb = malloc((size_t)d * sizeof(unsigned));
	for (size_t i = 0; i < mtr; i++) {
		struct tm g;
		gmtime_r(tm + i, &g);
		bin[i] = (unsigned)floor((double)(g.tm_yday * (int)n) / 365.);
	}
	for (unsigned i = 0; i < d; i++) comb[i] = i;
	for (unsigned c = 0; c < C; c++) {
		if (c) { /* next combination */
			int i = (int)d - 1;
			while (i >= 0 && comb[i] >= n - d + (unsigned)i) i--;
			if (i < 0) break;
			comb[i]++;
			for (unsigned j = (unsigned)i + 1; j < d; j++) comb[j] = comb[j - 1] + 1;
		}
		char *row = sel + (size_t)c * mtr;
		for (size_t t = 0; t < mtr; t++) {
			row[t] = 1;
			for (unsigned i = 0; i < d; i++) if (bin[t] == comb[i]) row[t] = 0;
		}
	}
	free(comb); free(bin);
	return 0;
}


/* Convergence metrics.  similarity = normalised cross-correlation at lag 0, misfit = sum of squared
 * differences.  ts_pws1f_lib.c:433-462 */
static double similarity_of(const double *a, const float *b, size_t N)
{
	double y = 0, s = 0;
	for (size_t n = 0; n < N; n++) y += a[n] * b[n];
	for (size_t n = 0; n < N; n++) s += a[n] * a[n];
	y /= sqrt(s);
	s = 0;
	for (size_t n = 0; n < N; n++) s += b[n] * b[n];
	return y / sqrt(s);
}

static double misfit_of(const double *a, const float *b, size_t N)
{
	double y = 0;
	for (size_t n = 0; n < N; n++) { const double e = a[n] - (double)b[n]; y += e * e; }
	return y;
}

/* Random K-of-J selection with libc rand(), flipping the rarer symbol.  ts_pws1f_lib.c:355-383 */
int orc_subsampling_plan(char *sel, size_t J, size_t K)
{
	if (!sel) return 1;
	if (K > J) return 2;
	size_t k = 0;
	if (2 * K < J) {
		memset(sel, 0, J);
		while (k < K) { const size_t j = (size_t)rand() % J; if (!sel[j]) { k++; sel[j] = 1; } }
	} else {
		memset(sel, 1, J);
		K = J - K;
		while (k < K) { const size_t j = (size_t)rand() % J; if (sel[j]) { k++; sel[j] = 0; } }
	}
	return 0;
}

/* One masked two-stage replica: partial stacks of the selected traces (group by rank among the selected),
 * stacks, weight(Kmax, K), inverse; time-domain linear stack.  Shared by the jackknife (:758-811) and the
 * two-stage random subsampling (:642-691). */
static size_t masked_two_stage(const orc_frame *f, const t_tsPWS *p, const float *sigall, size_t max, size_t mtr, const char *row,
                               float *ts_out, float *ls_out)
{
	const size_t nc = orc_frame_ncoef(f);
	const unsigned KM = p->Kmax;
	cplx *y = malloc(nc * sizeof(cplx)), *st = malloc(nc * sizeof(cplx)), *ps = malloc(nc * sizeof(cplx)), *o = malloc(nc * sizeof(cplx));
	double *P = calloc((size_t)KM * max, sizeof(double)), *xr = malloc(max * sizeof(double));
	size_t K = 0, k = 0;
	for (size_t i = 0; i < mtr; i++) if (row[i] == 1) K++;
	for (size_t i = 0; i < mtr; i++) {
		if (row[i] != 1) continue;
		const size_t g = (size_t)floor((double)(k * KM) / (double)K);
		double *pg = P + g * max;
		const float *x = sigall + i * max;
		for (size_t n = 0; n < max; n++) pg[n] += (double)x[n];
		k++;
	}
	stacks_of_doubles(f, st, ps, y, P, max, KM);
	weight(o, st, ps, nc, KM, (unsigned)K, p->wu, p->unbiased);
	orc_inverse(f, (double *)o, xr);
	for (size_t n = 0; n < max; n++) ts_out[n] = (float)(1. * xr[n]);
	memcpy(xr, P, max * sizeof(double));
	for (unsigned g = 1; g < KM; g++) for (size_t n = 0; n < max; n++) xr[n] += P[(size_t)g * max + n];
	const double sc = 1. / K;
	for (size_t n = 0; n < max; n++) ls_out[n] = (float)(xr[n] * sc);
	free(y); free(st); free(ps); free(o); free(P); free(xr);
	return K;
}

/* Whole call.  ts_pws1f_lib.c:48-352 (main path, two-stage, unbiased, convergence, random subsampling,
 * jackknife). */
int orc_tspws_main(t_tsPWS *p, t_tsPWS_out *out, t_data *in)
{
	if (!p || !out || !in) { printf("tspws_main: NULL input\n"); return -1; }
	float *sigall = in->sigall;
	const int max = in->hdr.max;
	const size_t mtr = p->Nmax ? p->Nmax : in->hdr.mtr;
	const float beg = in->hdr.beg, dt = in->hdr.dt;
	const unsigned nsamp = (unsigned)max;

	/* fold, :71-88 */
	if (p->fold) {
		if (2 * beg + (max - 1) * dt > 0.5 * dt) {
			printf("Warning: Folding ignored. B = %f, E = %f, nsamp = %u\n", beg, beg + (max - 1) * dt, nsamp);
			p->fold = 0;
		} else {
			const size_t half = nsamp / 2;
			for (size_t i = 0; i < mtr; i++) {
				float *x = sigall + i * (size_t)max;
				for (size_t n = 0; n < half; n++) {
					float v = x[n];
					v += x[max - 1 - n];
					v *= 0.5;
					x[max - 1 - n] = v;
					x[n] = v;
				}
			}
		}
	}
	orc_resolve(p, nsamp, dt);
	orc_frame *f = orc_frame_create(p->type, p->J, p->V, nsamp, p->s0, p->b0, p->w0, (int)p->uni);

	/* mean removal, :159-169 */
	if (p->lrm) {
#pragma omp parallel for schedule(static)
		for (size_t i = 0; i < mtr; i++) {
			float *x = sigall + i * (size_t)max;
			double m = 0.;
			for (size_t n = 0; n < nsamp; n++) m += (double)x[n];
			const float mf = (float)(m / nsamp);
			for (size_t n = 0; n < nsamp; n++) x[n] -= mf;
		}
	}
	if (!mtr) { orc_frame_destroy(f); return 0; }
	if (!f) return 4;
	/* a family without scales (J resolved to 0: fmin above the first scale) fails CheckWaveletFamily
	 * (FWTa/wavelet_mem_v7.c:40-45: Ns <= 0), the container is NULL and the call returns 4 (:199-204) -- after fold / rm */
	if (f->S == 0) { printf("Error: The waveletFamily variable is not nice.\n"); orc_frame_destroy(f); return 4; }

	const size_t nc = orc_frame_ncoef(f);
	cplx *Y = malloc(nc * sizeof(cplx)), *ST = calloc(nc, sizeof(cplx)), *PS = calloc(nc, sizeof(cplx));
	cplx *OUT = malloc(nc * sizeof(cplx));
	double *xa = malloc((size_t)max * sizeof(double)), *xb = malloc((size_t)max * sizeof(double));
	unsigned Kmax = p->Kmax;
	const int two_stage = !(!p->Kmax || p->Kmax > mtr);

	if (!two_stage) { /* :466-499 */
		Kmax = (unsigned)mtr;
		for (size_t i = 0; i < mtr; i++) {
			const float *x = sigall + i * (size_t)max;
			for (size_t n = 0; n < nsamp; n++) xa[n] = (double)x[n];
			orc_forward(f, xa, (double *)Y);
			accumulate(ST, PS, Y, nc);
		}
	} else {
		double *P = malloc((size_t)Kmax * max * sizeof(double));
		orc_partial_stacks(P, sigall, (size_t)max, mtr, Kmax);
		stacks_of_doubles(f, ST, PS, Y, P, (size_t)max, Kmax);
		free(P);
	}
	weight(OUT, ST, PS, nc, Kmax, (unsigned)mtr, p->wu, p->unbiased);
	orc_inverse(f, (double *)OUT, xa);
	orc_inverse(f, (double *)ST, xb);
	/* epilogue :233-241 -- ls is a FLOAT division by the converted count */
	for (size_t n = 0; n < nsamp; n++) out->ls[n] = (float)xb[n] / (unsigned)mtr;
	for (size_t n = 0; n < nsamp; n++) out->tsPWS[n] = (float)xa[n];

	/* convergence curves, :247-314 */
	if (p->convergence) {
		const float *ref_ts = in->reference ? in->reference : out->tsPWS;
		const float *ref_ls = in->reference ? in->reference : out->ls;
		double *xr = malloc((size_t)max * sizeof(double)), *P = malloc((size_t)(p->Kmax ? p->Kmax : 1) * max * sizeof(double));
		memset(ST, 0, nc * sizeof(cplx));
		memset(PS, 0, nc * sizeof(cplx));
		float *steps = out->tsPWS_steps;
		for (size_t i = 0; i < mtr; i++) {
			const size_t Tr = i + 1;
			unsigned K;
			if (!p->Kmax || p->Kmax >= Tr) { /* incremental single-stage step, :835-863 */
				K = (unsigned)Tr;
				const float *x = sigall + i * (size_t)max;
				for (size_t n = 0; n < nsamp; n++) xr[n] = (double)x[n];
				orc_forward(f, xr, (double *)Y);
				accumulate(ST, PS, Y, nc);
			} else {
				K = p->Kmax;
				orc_partial_stacks(P, sigall, (size_t)max, Tr, K);
				stacks_of_doubles(f, ST, PS, Y, P, (size_t)max, K);
			}
			weight(OUT, ST, PS, nc, K, (unsigned)Tr, p->wu, p->unbiased);
			orc_inverse(f, (double *)OUT, xr);
			out->tsPWS_sim[i] = similarity_of(xr, ref_ts, nsamp);
			out->tsPWS_misfit[i] = misfit_of(xr, ref_ts, nsamp);
			if (steps) { for (size_t n = 0; n < nsamp; n++) steps[n] = (float)xr[n]; steps += max; }
		}
		/* linear stack: running sum scaled by a FLOAT reciprocal and scaled back, :288-308 */
		steps = out->ls_steps;
		for (size_t n = 0; n < nsamp; n++) xr[n] = 0;
		for (size_t i = 0; i < mtr; i++) {
			const float *x = sigall + i * (size_t)max;
			for (size_t n = 0; n < nsamp; n++) xr[n] += x[n];
			const float inv = 1.0 / (i + 1);
			for (size_t n = 0; n < nsamp; n++) xr[n] *= inv;
			out->ls_sim[i] = similarity_of(xr, ref_ls, nsamp);
			out->ls_misfit[i] = misfit_of(xr, ref_ls, nsamp);
			if (steps) { for (size_t n = 0; n < nsamp; n++) steps[n] = (float)xr[n]; steps += max; }
			for (size_t n = 0; n < nsamp; n++) xr[n] *= (i + 1);
		}
		free(xr); free(P);
	}

	/* random subsampling, :324-333 */
	if (p->subsmpl_N > 0 && p->subsmpl_p > 0) {
		const size_t K = (size_t)ceil((double)mtr * p->subsmpl_p);
		const unsigned M = p->subsmpl_N;
		char *sel = malloc((size_t)M * mtr);
		if (!two_stage) { /* :501-610 */
			for (unsigned m = 0; m < M; m++) orc_subsampling_plan(sel + (size_t)m * mtr, mtr, K);
			cplx *st = calloc((size_t)M * nc, sizeof(cplx)), *ps = calloc((size_t)M * nc, sizeof(cplx));
			double *xr = malloc((size_t)max * sizeof(double));
			for (unsigned m = 0; m < M; m++) { memset(out->ls_subsmpl[m], 0, (size_t)max * sizeof(float)); memset(out->tsPWS_subsmpl[m], 0, (size_t)max * sizeof(float)); }
			for (size_t i = 0; i < mtr; i++) {
				const float *x = sigall + i * (size_t)max;
				for (size_t n = 0; n < nsamp; n++) xr[n] = (double)x[n];
				orc_forward(f, xr, (double *)Y);
				for (unsigned m = 0; m < M; m++) {
					if (sel[(size_t)m * mtr + i] != 1) continue;
					float *l = out->ls_subsmpl[m];
					for (size_t n = 0; n < nsamp; n++) l[n] += 1. * xr[n];   /* float accumulator */
					accumulate(st + (size_t)m * nc, ps + (size_t)m * nc, Y, nc);
				}
			}
			for (unsigned m = 0; m < M; m++) {
				const float sc = 1. / K;
				float *l = out->ls_subsmpl[m];
				for (size_t n = 0; n < (size_t)max; n++) l[n] *= sc;
				weight(OUT, st + (size_t)m * nc, ps + (size_t)m * nc, nc, (unsigned)K, (unsigned)K, p->wu, p->unbiased);
				orc_inverse(f, (double *)OUT, xr);
				for (size_t n = 0; n < nsamp; n++) out->tsPWS_subsmpl[m][n] = (float)(1. * xr[n]);
			}
			free(st); free(ps); free(xr);
		} else { /* :612-709 (sequential here: the reference draws the masks inside an OpenMP loop) */
			for (unsigned m = 0; m < M; m++) {
				orc_subsampling_plan(sel, mtr, K);
				masked_two_stage(f, p, sigall, (size_t)max, mtr, sel, out->tsPWS_subsmpl[m], out->ls_subsmpl[m]);
			}
		}
		free(sel);
	}

	/* jackknife, :335-345 -> :719-831 (two-stage only; the single-stage variant
	 * is an empty stub at :711-716) */
	if (p->jackknife_n > 0 && p->jackknife_d > 0 && two_stage) {
		const unsigned C = out->M;
		char *sel = malloc((size_t)C * mtr);
		if (orc_jackknife_plan(sel, in->time, mtr, p->jackknife_d, p->jackknife_n, C) == 0) {
			const unsigned KM = p->Kmax;
#pragma omp parallel
			{
				cplx *y = malloc(nc * sizeof(cplx)), *st = malloc(nc * sizeof(cplx)), *ps = malloc(nc * sizeof(cplx));
				cplx *o = malloc(nc * sizeof(cplx));
				double *P = malloc((size_t)KM * max * sizeof(double)), *xr = malloc((size_t)max * sizeof(double));
#pragma omp for schedule(static)
				for (unsigned c = 0; c < C; c++) {
					const char *row = sel + (size_t)c * mtr;
					size_t K = 0, k = 0;
					memset(P, 0, (size_t)KM * max * sizeof(double));
					for (size_t i = 0; i < mtr; i++) if (row[i] == 1) K++;
					for (size_t i = 0; i < mtr; i++) {
						if (row[i] != 1) continue;
						const size_t g = (size_t)floor((double)(k * KM) / (double)K);
						double *pg = P + g * (size_t)max;
						const float *x = sigall + i * (size_t)max;
						for (size_t n = 0; n < nsamp; n++) pg[n] += (double)x[n];
						k++;
					}
					stacks_of_doubles(f, st, ps, y, P, (size_t)max, KM);
					weight(o, st, ps, nc, KM, (unsigned)K, p->wu, p->unbiased);
					orc_inverse(f, (double *)o, xr);
					out->mtr_subsmpl[c] = (unsigned)K;
					float *dst = out->tsPWS_subsmpl[c];
					for (size_t n = 0; n < nsamp; n++) dst[n] = (float)(1. * xr[n]);
					/* time-domain linear stack of the replica, :799-811 */
					memcpy(xr, P, (size_t)max * sizeof(double));
					for (unsigned g = 1; g < KM; g++)
						for (size_t n = 0; n < nsamp; n++) xr[n] += P[(size_t)g * max + n];
					const double sc = 1. / K;
					dst = out->ls_subsmpl[c];
					for (size_t n = 0; n < nsamp; n++) dst[n] = (float)(xr[n] * sc);
				}
				free(y); free(st); free(ps); free(o); free(P); free(xr);
			}
		}
		free(sel);
	}
	free(Y); free(ST); free(PS); free(OUT); free(xa); free(xb);
	orc_frame_destroy(f);
	return 0;
}

/* --------------------------------------------------------------------------
 * Trace-/scale-parallel OpenMP variant of the main stack (single- or two-stage, no resampling): the "honest
 * stronger" host baseline of BASELINE.md section 3.  Same arithmetic and the same per-element summation order as
 * orc_tspws_main (results are bit-identical); only the loops are distributed: partial stacks by column blocks,
 * forward transforms by (trace, scale), accumulation / weighting by coefficient, inverse by sample.
 * -------------------------------------------------------------------------- */
static void forward_one_scale(const orc_frame *f, unsigned s, const double *x, cplx *Y)
{
	const size_t N = f->N;
	const cplx *w = f->w + f->tap_off[s];
	cplx *y = Y + f->coef_off[s];
	const size_t L = f->L[s], D = f->D[s], c = (size_t)f->c[s];
	for (size_t k = 0; k < f->Ns[s]; k++) {
		const size_t n0 = (N + k * D - c) % N;
		const size_t l0 = (N - n0 < L) ? N - n0 : L;
		cplx acc = 0;
		const double *xp = x + n0;
		for (size_t l = 0; l < l0; l++) acc += xp[l] * w[l];
		for (size_t l = l0; l < L; l++) acc += x[l - l0] * w[l];
		y[k] = conj(acc);
	}
}

static void inverse_mt(const orc_frame *f, const cplx *Y, double *xrec)
{
	const size_t N = f->N;
#pragma omp parallel for schedule(static)
	for (size_t n = 0; n < N; n++) {
		double tot = 0;
		for (unsigned s = 0; s < f->S; s++) {
			const cplx *wd = f->wd + f->tap_off[s];
			const cplx *y = Y + f->coef_off[s];
			const size_t L = f->L[s], D = f->D[s], cd = (size_t)f->cd[s];
			const size_t n0 = (N + n - cd) % N, l0 = N - n0, lim = L < l0 ? L : l0;
			double acc = 0, bf;
			if (D > 1) {
				size_t l = (D - n0 % D) % D, q = (n0 + D - 1) / D;
				for (; l < lim; l += D, q++) acc += creal(wd[l]) * creal(y[q]) + cimag(wd[l]) * cimag(y[q]);
				q = 0;
				for (l = l0; l < L; l += D, q++) acc += creal(wd[l]) * creal(y[q]) + cimag(wd[l]) * cimag(y[q]);
				bf = (double)D * acc;
			} else {
				for (size_t l = 0; l < lim; l++) acc += creal(wd[l]) * creal(y[n0 + l]) + cimag(wd[l]) * cimag(y[n0 + l]);
				for (size_t l = lim; l < L; l++) acc += creal(wd[l]) * creal(y[l - l0]) + cimag(wd[l]) * cimag(y[l - l0]);
				bf = acc;
			}
			tot += (log(2.0) / (2 * f->Cpsi * f->V * f->scale[s])) * bf;
		}
		xrec[n] = tot;
	}
}

int orc_tspws_main_mt(t_tsPWS *p, t_tsPWS_out *out, t_data *in)
{
	if (!p || !out || !in) return -1;
	if (p->fold || p->lrm || p->convergence || p->subsmpl_N || p->jackknife_n) return orc_tspws_main(p, out, in);
	const float *sigall = in->sigall;
	const size_t max = (size_t)in->hdr.max, mtr = p->Nmax ? p->Nmax : in->hdr.mtr;
	orc_resolve(p, (unsigned)max, in->hdr.dt);
	orc_frame *f = orc_frame_create(p->type, p->J, p->V, (unsigned)max, p->s0, p->b0, p->w0, (int)p->uni);
	if (!f) return 4;
	if (!mtr) { orc_frame_destroy(f); return 0; }
	if (f->S == 0) { orc_frame_destroy(f); return 4; } /* no scales: see orc_tspws_main */
	const size_t nc = orc_frame_ncoef(f);
	const int two_stage = !(!p->Kmax || p->Kmax > mtr);
	const unsigned K = two_stage ? p->Kmax : (unsigned)mtr;
	cplx *ST = calloc(nc, sizeof(cplx)), *PS = calloc(nc, sizeof(cplx)), *OUT = malloc(nc * sizeof(cplx));
	double *xa = malloc(max * sizeof(double)), *xb = malloc(max * sizeof(double));
	const size_t batch = two_stage ? K : (mtr < 64 ? mtr : 64);
	cplx *Y = malloc(batch * nc * sizeof(cplx));
	double *X = malloc(batch * max * sizeof(double));
	for (size_t t0 = 0; t0 < (two_stage ? (size_t)K : mtr); t0 += batch) {
		const size_t nb = two_stage ? K : (mtr - t0 < batch ? mtr - t0 : batch);
		if (two_stage) { /* partial stacks by column blocks: every element still sums its traces in order */
#pragma omp parallel for schedule(static)
			for (size_t n0 = 0; n0 < max; n0 += 2048) {
				const size_t n1 = n0 + 2048 < max ? n0 + 2048 : max;
				for (unsigned g = 0; g < K; g++) for (size_t n = n0; n < n1; n++) X[g * max + n] = 0;
				for (size_t i = 0; i < mtr; i++) {
					const size_t g = (size_t)floor((double)(i * K) / (double)mtr);
					const float *x = sigall + i * max;
					double *pg = X + g * max;
					for (size_t n = n0; n < n1; n++) pg[n] += (double)x[n];
				}
			}
		} else {
#pragma omp parallel for schedule(static)
			for (size_t b = 0; b < nb; b++)
				for (size_t n = 0; n < max; n++) X[b * max + n] = (double)sigall[(t0 + b) * max + n];
		}
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
		for (size_t b = 0; b < nb; b++)
			for (unsigned s = 0; s < f->S; s++) forward_one_scale(f, f->S - 1 - s, X + b * max, Y + b * nc);
#pragma omp parallel for schedule(static)
		for (size_t i = 0; i < nc; i++)
			for (size_t b = 0; b < nb; b++) {
				cplx v = Y[b * nc + i];
				ST[i] += v;
				v /= cabs(v);
				if (creal(v * conj(v)) <= 1.001) PS[i] += v;
			}
	}
	weight(OUT, ST, PS, nc, K, (unsigned)mtr, p->wu, p->unbiased);
	inverse_mt(f, OUT, xa);
	inverse_mt(f, ST, xb);
	for (size_t n = 0; n < max; n++) out->ls[n] = (float)xb[n] / (unsigned)mtr;
	for (size_t n = 0; n < max; n++) out->tsPWS[n] = (float)xa[n];
	free(Y); free(X); free(ST); free(PS); free(OUT); free(xa); free(xb);
	orc_frame_destroy(f);
	return 0;
}

/* Stand-alone pieces exposed to the tests ------------------------------------ */
void orc_accumulate(double *ST, double *PS, const double *Y, size_t ncoef)
{
	accumulate((cplx *)ST, (cplx *)PS, (const cplx *)Y, ncoef);
}

void orc_weight(double *OUT, const double *ST, const double *PS, size_t ncoef, unsigned K, unsigned M, double wu, int unbiased)
{
	weight((cplx *)OUT, (const cplx *)ST, (const cplx *)PS, ncoef, K, M, wu, unbiased);
}
