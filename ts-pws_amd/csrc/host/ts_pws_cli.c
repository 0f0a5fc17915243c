/*
 * ts_pws -- SAC-file command-line front-end of the MI355X ts-PWS engine.
 *
 * Same command line, input conventions and output files as the reference front-end
 * (/root/reference/src/ts_pws1f.c:139-431): first argument = text file listing SAC files (or an msacs
 * binary with `bin`), then key=value / bare-word options matched by prefix in the reference's order
 * (:163-215); outputs tl[_X].sac and ts_pws[_X].sac (+ jackknife replicas), fold trimming as :322-328.
 * The stacking itself is tspws_main() from libtspws_hip.so.  SAC I/O is this repo's own minimal
 * implementation (sacio_min.c) because SAC's sacio.a is not available.
 *
 * Extensions over the reference:
 *  - when reading a SAC list, each trace's start time is taken from the SAC reference time (nzyear/nzjday/...), so the
 *    jackknife also works without an msacs file (the reference leaves `time` zero there and runs on uninitialised masks);
 *  - `ts_pws @batch.txt [options]`: SEVERAL ensembles in one process.  Every line of batch.txt is `<filelist> [tag]`; ensemble
 *    i is stacked with the same options and its outputs are named as a single run with `osac=<tag>` names them (:335-349;
 *    tag defaults to the list's base name without extension).  One HIP start-up (~0.25 s, most of a one-call process), one
 *    frame per (N, options) kept by tspws_main, and the files of ensemble i + 1 are read by a second thread while ensemble i
 *    is stacked -- the paper's workload is tens of station pairs per run.
 * A malformed input (missing / truncated SAC file, npts <= 0, a `bin` file whose header is short or promises more traces
 * than the file could hold) ends the ensemble with the reference's message and a non-zero status, every buffer freed.
 */
#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sacio_min.h"
#include "ts_pws1f_lib.h"

static int is_uint(const char *s)
{
	while (*s == ' ') s++;
	if (*s == '+') s++;
	const char *d = s;
	while (isdigit((unsigned char)*s)) s++;
	return *s == '\0' && s > d;
}

static int is_double(const char *s)
{
	while (*s == ' ') s++;
	if (*s == '+' || *s == '-') s++;
	while (isdigit((unsigned char)*s)) s++;
	if (*s == '.') { s++; while (isdigit((unsigned char)*s)) s++; }
	return *s == '\0';
}

/* invalid numbers are ignored, as in the reference (the parse status is never checked, :163-215) */
static void rd_uint(unsigned *x, const char *s) { if (is_uint(s)) *x = (unsigned)atoi(s); else fprintf(stderr, "ts_pws: ignoring bad unsigned value '%s'\n", s); }
static void rd_double(double *x, const char *s) { if (is_double(s)) *x = atof(s); else fprintf(stderr, "ts_pws: ignoring bad value '%s'\n", s); }

static unsigned binomial(unsigned n, unsigned d)
{
	if (d > n - d) d = n - d;
	unsigned out = 1;
	for (unsigned i = n - d + 1; i <= n; i++) out *= i;
	for (unsigned i = 2; i <= d; i++) out /= i;
	return out;
}

static void usage(void)
{
	puts("\nTime-scale phase-weighted stack (ts-PWS) on AMD MI355X -- drop-in for ts_pws (Ventosa et al., GJI 2017).\n"
	     "USAGE: ts_pws filelist [options]      |      ts_pws @batch [options]\n"
	     "  filelist           text file with one SAC file per line (or an msacs binary with `bin`)\n"
	     "  @batch             text file with one `filelist [tag]` per line: every ensemble with the same options, outputs as osac=tag\n"
	     "  wu=2               phase-weight power            J= V= s0= b0= fmin=   frame sampling\n"
	     "  Q= | cycles= | cyc= | w0=   Morlet shape         MexHat   complex Mexican-hat frame\n"
	     "  rm  fold  uni  verbose  unbiased                 TwoStage[=10]   two-stage stack\n"
	     "  jackknife_n= jackknife_d=   (TwoStage only)      obin   replicas to one msacs file\n"
	     "  Nmax=              use the first Nmax traces     osac=X kinst=S  output naming / header\n"
	     "  convergence[=ref.sac] AllSteps   convergence curves   subsmpl_N= subsmpl_prob=   random subsampling\n"
	     "OUTPUT: tl[_X].sac (linear stack), ts_pws[_X].sac (ts-PWS), *_subsmpl_<m>.sac replicas.\n"
	     "Device: environment variable TSPWS_DEVICE (default 0).\n");
}

static int starts(const char *a, const char *p) { return !strncmp(a, p, strlen(p)); }

/* ---- input --------------------------------------------------------------------------------- */
static char **read_list(const char *path, unsigned *n_out)
{
	FILE *f = fopen(path, "r");
	if (!f) return NULL;
	size_t cap = 64, n = 0;
	char **v = (char **)malloc(cap * sizeof *v), line[1024];
	while (v && fgets(line, sizeof line, f)) {
		char *nl = strchr(line, '\n');
		if (nl) *nl = '\0';
		if (n == cap) { cap *= 2; v = (char **)realloc(v, cap * sizeof *v); if (!v) break; }
		v[n++] = strdup(line);
	}
	fclose(f);
	*n_out = (unsigned)n;
	return v;
}

static void free_list(char **v, unsigned n)
{
	if (!v) return;
	for (unsigned i = 0; i < n; i++) free(v[i]);
	free(v);
}

static void free_data(t_data *in)
{
	free(in->sigall); free(in->time); free(in->lag0); free(in->reference);
	in->sigall = NULL; in->time = NULL; in->lag0 = NULL; in->reference = NULL;
}

static void trim_copy(char dst[9], const char *src8)
{
	memcpy(dst, src8, 8);
	dst[8] = '\0';
}

/* reference: ReadData, ts_pws1f.c:570-723.  On a non-zero return nothing is left allocated in *in. */
static int read_data(t_data *in, const char *filein, int bin, int verbose)
{
	t_hdr *hdr = &in->hdr;
	if (bin) {
		FILE *f = fopen(filein, "rb");
		if (!f) { printf("tspws_main: cannot open the %s file\n", filein); return -2; }
		msacs_header b;
		/* (the reference prints this and carries on with an uninitialised header, ts_pws1f.c:587-589: nothing useful can follow) */
		if (fread(&b, sizeof b, 1, f) != 1) { printf("tspws_main: %s is shorter than predicted (header).\n", filein); fclose(f); return -2; }
		long fsize = -1;
		if (!fseek(f, 0, SEEK_END)) fsize = ftell(f);
		if (fsize < 0 || fseek(f, (long)sizeof b, SEEK_SET)) { printf("tspws_main: cannot size the %s file\n", filein); fclose(f); return -2; }
		/* header sanity before anything is sized from it (the reference's check_header only looks at the text fields, :606): at least two
		 * lags (dt divides by nlags - 1, :597), sample count within int (hdr->max is one), and the file must at least hold the time and
		 * lag0 tables and ONE whole trace -- a file cut inside the data block still reads like the reference's (warning, zeros behind) */
		const unsigned long long nl = b.nlags, ns = b.nseq;
		const unsigned long long tables = ns * (unsigned long long)(sizeof(time_t) + sizeof(float));
		if (nl < 2 || nl > (unsigned long long)INT_MAX || (ns && (ns > (1ull << 31) || (unsigned long long)fsize < sizeof b + tables + nl * sizeof(float)))) {
			printf("\a ReadData: Found the header corrupted when reading the %s file (nlags = %u, nseq = %u, %ld bytes)\n", filein, b.nlags, b.nseq, fsize);
			fclose(f);
			return 2;
		}
		hdr->max = (int)b.nlags; hdr->mtr = b.nseq;
		hdr->evla = b.stlat1; hdr->evlo = b.stlon1; hdr->stla = b.stlat2; hdr->stlo = b.stlon2; hdr->stel = b.stel2;
		hdr->dt = (b.lag2 - b.lag1) / (float)(b.nlags - 1);
		hdr->beg = b.lag1;
		trim_copy(hdr->net1, b.net1); trim_copy(hdr->sta1, b.sta1); trim_copy(hdr->loc1, b.loc1); trim_copy(hdr->chn1, b.chn1);
		trim_copy(hdr->net2, b.net2); trim_copy(hdr->sta2, b.sta2); trim_copy(hdr->loc2, b.loc2); trim_copy(hdr->chn2, b.chn2);
		const size_t mtr = hdr->mtr, max = (size_t)hdr->max;
		in->sigall = (float *)calloc((mtr * max) > 0 ? mtr * max : 1, sizeof(float));
		in->time = (time_t *)calloc(mtr ? mtr : 1, sizeof(time_t));
		in->lag0 = (float *)calloc(mtr ? mtr : 1, sizeof(float));
		if (!in->sigall || !in->time || !in->lag0) { fclose(f); free_data(in); return 4; }
		if (fread(in->time, sizeof(time_t), mtr, f) != mtr) printf("tspws_main: %s is shorter than predicted (time).\n", filein);
		if (fread(in->lag0, sizeof(float), mtr, f) != mtr) printf("tspws_main: %s is shorter than predicted (lag0).\n", filein);
		if (fread(in->sigall, sizeof(float), mtr * max, f) != mtr * max) printf("tspws_main: %s is shorter than predicted (data).\n", filein);
		fclose(f);
		return 0;
	}
	unsigned nfiles = 0;
	char **files = read_list(filein, &nfiles);
	if (!files) { printf("tspws_main: cannot read the %s file\n", filein); return -2; }
	if (!nfiles) { printf("tspws_main: nothing to do, %s is empty!\n", filein); free(files); hdr->mtr = 0; return 0; }
	sac_header h;
	int npts = 0, rc = sac_read(files[0], &h, NULL, 0, &npts);
	if (rc) { printf("\a tspws_main: Error reading %s header (nerr=%d).\n", files[0], rc); free_list(files, nfiles); return 2; }
	if (h.f[SAC_F_DELTA] == SAC_UNDEF_F || h.f[SAC_F_B] == SAC_UNDEF_F || npts <= 0) {
		printf("\a tspws_main: Error reading %s header, npts/delta/b is not defined!\n", files[0]);
		free_list(files, nfiles);
		return 2;
	}
	hdr->max = npts; hdr->mtr = nfiles;
	hdr->dt = h.f[SAC_F_DELTA]; hdr->beg = h.f[SAC_F_B];
	hdr->stla = h.f[SAC_F_STLA]; hdr->stlo = h.f[SAC_F_STLO]; hdr->stel = h.f[SAC_F_STEL];
	hdr->evla = h.f[SAC_F_EVLA]; hdr->evlo = h.f[SAC_F_EVLO];
	if (verbose && (hdr->stla == SAC_UNDEF_F || hdr->evla == SAC_UNDEF_F)) printf("tspws_main: station/event coordinates are not defined in %s.\n", files[0]);
	sac_get_k(&h, SAC_K_KNETWK, 8, hdr->net2); sac_get_k(&h, SAC_K_KSTNM, 8, hdr->sta2);
	sac_get_k(&h, SAC_K_KHOLE, 8, hdr->loc2); sac_get_k(&h, SAC_K_KCMPNM, 8, hdr->chn2);
	sac_get_k(&h, SAC_K_KUSER0, 8, hdr->net1); sac_get_k(&h, SAC_K_KEVNM, 8, hdr->sta1);
	sac_get_k(&h, SAC_K_KUSER1, 8, hdr->loc1); sac_get_k(&h, SAC_K_KUSER2, 8, hdr->chn1);

	const size_t max = (size_t)npts;
	in->sigall = (float *)calloc((size_t)nfiles * max, sizeof(float));
	in->time = (time_t *)calloc(nfiles, sizeof(time_t));
	in->lag0 = (float *)calloc(nfiles, sizeof(float));
	if (!in->sigall || !in->time || !in->lag0) {
		printf("tspws_main: Out of memory when reading %s (mtr = %u, npts = %d)\n", filein, nfiles, npts);
		free_list(files, nfiles); free_data(in);
		return 4;
	}
	/* The reference's reader, literally (ts_pws1f.c:680-708): trace i is read into slot i - nskip; a trace whose dt differs by more
	 * than 1 % or whose b differs by more than dt is "skipped" by letting the next accepted trace overwrite its slot -- the rows are
	 * not cleared in between (a shorter trace keeps the tail of what its slot held), a skipped LAST trace stays in its slot, and the
	 * trace count handed to tspws_main stays the FILE count (:708 updates a local only), so every skipped trace in the middle
	 * leaves one trailing all-zero row in the stack.  Reproduced as is: the outputs are the reference's. */
	const float dt1 = hdr->dt, beg1 = hdr->beg;
	unsigned nskip = 0;
	for (unsigned i = 0; i < nfiles; i++) {
		float *dst = in->sigall + (size_t)(i - nskip) * max;
		int n = 0;
		rc = sac_read(files[i], &h, dst, npts, &n);
		if (rc) { printf("tspws_main: Error reading %s file (nerr=%d)\n", files[i], rc); free_list(files, nfiles); free_data(in); return -2; }
		if (n > npts) printf("tspws_main: WARNING: using only %d samples on %u trace\n", npts, i);
		else if (n < npts) printf("tspws_main: WARNING: trace %u has only %d samples\n", i, n);
		if (fabs(h.f[SAC_F_DELTA] - dt1) > dt1 * 0.01) { /* :695-700 */
			printf("tspws_main: WARNING: trace %u has a different dt !\ntspws_main: WARNING: skipping trace %u\n", i, i);
			nskip++;
			continue;
		}
		if (fabs(beg1 - h.f[SAC_F_B]) > dt1) { /* :701-706 */
			printf("tspws_main: WARNING: trace %u has a different beg !\ntspws_main: WARNING: skipping trace %u\n", i, i);
			nskip++;
			continue;
		}
		in->time[i - nskip] = sac_reference_time(&h); /* (addition: the reference leaves time[] zero for SAC lists) */
	}
	const unsigned kept = nfiles - nskip;
	/* one line about what the reference's way of skipping leaves behind (the outputs are the reference's either way) */
	if (nskip)
		printf("ts_pws: %u of %u traces skipped (dt / b mismatch): the stack keeps %u rows like the reference's reader (ts_pws1f.c:680-708) -- %u accepted "
		       "traces in front, the rest as the skipping left them (zero rows, or a mismatched LAST trace still in its slot); ls is divided by %u\n",
		       nskip, nfiles, nfiles, kept, nfiles);
	free_list(files, nfiles);
	for (unsigned i = 0; i < kept; i++) {
		const float *x = in->sigall + (size_t)i * max;
		size_t n = 0;
		while (n < max && x[n] == 0.f) n++;
		if (n == max) printf("tspws_main: %s, trace %u of %u is ZERO\n", filein, i, kept); /* (the reference checks the kept traces only, :713-718) */
	}
	return 0;
}

/* ---- output -------------------------------------------------------------------------------- */
/* reference: wrsac, ts_pws1f.c:725-763 */
static int write_sac(const char *name, const float *y, const t_hdr *hdr, const char *kinst, float user0)
{
	sac_header h;
	sac_new_header(&h);
	h.i[SAC_I_IZTYPE] = 11; /* IO */
	h.i[SAC_I_NPTS] = hdr->max;
	h.f[SAC_F_DELTA] = hdr->dt; h.f[SAC_F_B] = hdr->beg; h.f[SAC_F_USER0] = user0;
	h.f[SAC_F_STLA] = hdr->stla; h.f[SAC_F_STLO] = hdr->stlo; h.f[SAC_F_STEL] = hdr->stel;
	h.f[SAC_F_EVLA] = hdr->evla; h.f[SAC_F_EVLO] = hdr->evlo;
	sac_set_k(&h, SAC_K_KINST, 8, kinst);
	sac_set_k(&h, SAC_K_KNETWK, 8, hdr->net2); sac_set_k(&h, SAC_K_KSTNM, 8, hdr->sta2);
	sac_set_k(&h, SAC_K_KHOLE, 8, hdr->loc2); sac_set_k(&h, SAC_K_KCMPNM, 8, hdr->chn2);
	sac_set_k(&h, SAC_K_KUSER0, 8, hdr->net1); sac_set_k(&h, SAC_K_KEVNM, 16, hdr->sta1);
	sac_set_k(&h, SAC_K_KUSER1, 8, hdr->loc1); sac_set_k(&h, SAC_K_KUSER2, 8, hdr->chn1);
	h.i[SAC_I_LCALDA] = 1;
	const int rc = sac_write(name, &h, y);
	if (rc) printf("\a wrsac: Error writing the %s file\n", name);
	return rc;
}

static void put8(char dst[8], const char *src) /* NUL-padded 8-character field, no terminator needed */
{
	size_t n = strlen(src);
	memset(dst, 0, 8);
	memcpy(dst, src, n < 8 ? n : 8);
}

/* reference: wrbin, ts_pws1f.c:765-826 (header, then the replica sizes as time_t, then the rows; no lag0 block) */
static int write_bin(const char *name, float **y, unsigned M, unsigned first, const t_hdr *hdr, const char *kinst, const unsigned *mtr)
{
	msacs_header b;
	memset(&b, 0, sizeof b);
	b.nlags = (uint32_t)hdr->max; b.nseq = M;
	b.stlat1 = hdr->evla; b.stlon1 = hdr->evlo; b.stlat2 = hdr->stla; b.stlon2 = hdr->stlo; b.stel2 = hdr->stel;
	b.lag1 = hdr->beg; b.tlength = hdr->dt * (float)(b.nlags - 1); b.lag2 = b.lag1 + b.tlength;
	put8(b.method, kinst);
	put8(b.net1, hdr->net1); put8(b.sta1, hdr->sta1); put8(b.loc1, hdr->loc1); put8(b.chn1, hdr->chn1);
	put8(b.net2, hdr->net2); put8(b.sta2, hdr->sta2); put8(b.loc2, hdr->loc2); put8(b.chn2, hdr->chn2);
	FILE *f = fopen(name, "wb");
	if (!f) { printf("tspws_main: cannot open the %s file\n", name); return -2; }
	fwrite(&b, sizeof b, 1, f);
	for (unsigned m = 0; m < M; m++) { time_t t = (time_t)mtr[m]; fwrite(&t, sizeof t, 1, f); }
	for (unsigned m = 0; m < M; m++) fwrite(y[m] + first, sizeof(float), (size_t)hdr->max, f);
	fclose(f);
	return 0;
}

/* TSPWS_CLI_TIMES=1: wall time of the phases (tools/cli_timing.py) */
static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void free_out(t_tsPWS_out *out)
{
	if (out->ls_subsmpl) free(out->ls_subsmpl[0]);
	if (out->tsPWS_subsmpl) free(out->tsPWS_subsmpl[0]);
	free(out->ls_subsmpl); free(out->tsPWS_subsmpl); free(out->mtr_subsmpl);
	free(out->ls); free(out->tsPWS);
	free(out->ls_sim); free(out->tsPWS_sim); free(out->ls_misfit); free(out->tsPWS_misfit); free(out->ls_steps); free(out->tsPWS_steps);
	memset(out, 0, sizeof *out);
}

/* One ensemble, already read: set-up of the outputs (ts_pws1f.c:229-280), tspws_main, output files (:313-428).  p is this ensemble's own
 * copy of the parsed options (tspws_main rewrites fold / fmin / w0 / V / b0 / s0 / J); frees *in. */
static int run_ensemble(t_tsPWS p, t_data *in, double t_start, double t_read)
{
	t_tsPWS_out out;
	memset(&out, 0, sizeof out);
	int er = 0;
	const size_t max = (size_t)in->hdr.max;
	out.N = (unsigned)max;
	/* the reference takes Nmax unchecked (:65); tspws_main clamps it to the traces read, so every size derived from it
	   (convergence dumps, the user0 header field) uses the clamped count too */
	out.mtr = (p.Nmax && p.Nmax < in->hdr.mtr) ? p.Nmax : in->hdr.mtr;
	out.ls = (float *)calloc(max, sizeof(float));
	out.tsPWS = (float *)calloc(max, sizeof(float));
	if (!out.ls || !out.tsPWS) { printf("main: Out of memory\n"); er = 4; goto done; }
	if (p.convergence && p.fileconv) { /* alternative reference trace, :287-305 */
		sac_header rh;
		int rn = 0;
		in->reference = (float *)calloc(max, sizeof(float));
		if (!in->reference || sac_read(p.fileconv, &rh, in->reference, (int)max, &rn) || rn != (int)max) {
			if (in->reference && rn != (int)max) printf("The reference for convergence has a different length (%d:%d)\n", (int)max, rn);
			free(in->reference); in->reference = NULL; p.fileconv = NULL;
		}
	}
	if (p.convergence) { /* :233-243 */
		out.ls_sim = (double *)calloc(out.mtr, sizeof(double)); out.tsPWS_sim = (double *)calloc(out.mtr, sizeof(double));
		out.ls_misfit = (double *)calloc(out.mtr, sizeof(double)); out.tsPWS_misfit = (double *)calloc(out.mtr, sizeof(double));
		if (!out.ls_sim || !out.tsPWS_sim || !out.ls_misfit || !out.tsPWS_misfit) { printf("main: Out of memory\n"); er = 4; goto done; }
		if (p.AllSteps) {
			out.ls_steps = (float *)calloc((size_t)out.mtr * max, sizeof(float));
			out.tsPWS_steps = (float *)calloc((size_t)out.mtr * max, sizeof(float));
			if (!out.ls_steps || !out.tsPWS_steps) { printf("main: Out of memory\n"); er = 4; goto done; }
		}
	}
	if (p.jackknife_n) { /* :247-252 */
		p.subsmpl_N = 0; p.subsmpl_p = 0;
		if (p.jackknife_d == 0 || p.jackknife_d >= p.jackknife_n) { p.jackknife_d = 0; p.jackknife_n = 0; }
		else out.M = binomial(p.jackknife_n, p.jackknife_d);
	}
	if (p.subsmpl_N) { /* :254-258 */
		if (p.subsmpl_p < 0 || p.subsmpl_p > 1) { p.subsmpl_N = 0; p.subsmpl_p = 0; }
		else out.M = p.subsmpl_N;
	}
	if (p.jackknife_n || p.subsmpl_N) {
		out.mtr_subsmpl = (unsigned *)calloc(out.M, sizeof(unsigned));
		out.ls_subsmpl = (float **)calloc(out.M, sizeof(float *));
		out.tsPWS_subsmpl = (float **)calloc(out.M, sizeof(float *));
		float *a = (float *)calloc((size_t)out.M * max, sizeof(float)), *b = (float *)calloc((size_t)out.M * max, sizeof(float));
		if (!out.mtr_subsmpl || !out.ls_subsmpl || !out.tsPWS_subsmpl || !a || !b) { printf("main: Out of memory\n"); free(a); free(b); er = 4; goto done; }
		for (unsigned m = 0; m < out.M; m++) { out.ls_subsmpl[m] = a + (size_t)m * max; out.tsPWS_subsmpl[m] = b + (size_t)m * max; }
	}

	if (p.subsmpl_N) for (unsigned i = 1; i < out.M; i++) out.mtr_subsmpl[i] = (unsigned)(out.mtr * p.subsmpl_p); /* :277 (entry 0 stays 0) */

	const double t_call0 = now_s();
	er = tspws_main(&p, &out, in);
	const double t_call1 = now_s();
	(void)t_start; (void)t_read; (void)t_call0; (void)t_call1;
#ifdef TSPWS_SWEEPS
	if (getenv("TSPWS_CLI_TIMES")) printf("cli: read %.1f ms, set-up %.1f ms, tspws_main %.1f ms\n", 1e3 * (t_read - t_start), 1e3 * (t_call0 - t_read), 1e3 * (t_call1 - t_call0));
#endif

	if (!er) {
		t_hdr hdr = in->hdr;
		unsigned first = 0;
		if (p.fold) { /* :322-328 */
			if (2 * hdr.beg + (hdr.max - 1) * hdr.dt < 0.5 * hdr.dt) {
				first = (unsigned)(hdr.max / 2);
				hdr.max = (hdr.max + 1) / 2;
				hdr.beg += hdr.dt * first;
			} else printf("Warning: Folding ignored. B = %f, E = %f\n", hdr.beg, hdr.beg + (hdr.max - 1) * hdr.dt);
		}
		char name[1200];
		const char *tag = p.fileout;
		if (tag) snprintf(name, sizeof name, "tl_%s.sac", tag); else strcpy(name, "tl.sac");
		write_sac(name, out.ls + first, &hdr, p.lkinst ? p.kinst : "t-lin", (float)out.mtr);
		if (p.verbose) printf("Output files:\n  Linear stack: %s\n", name);
		if (tag) snprintf(name, sizeof name, "ts_pws_%s.sac", tag); else strcpy(name, "ts_pws.sac");
		write_sac(name, out.tsPWS + first, &hdr, p.lkinst ? p.kinst : "ts_pws", (float)out.mtr);
		if (p.verbose) printf("  ts-PWS:       %s\n", name);
		if (p.convergence) { /* raw double / float dumps, :354-396 */
			const char *t = tag ? tag : "";
			const struct { const char *pre, *suf; const void *buf; size_t bytes; } dumps[] = {
				{"ts_pws_", "_convergence", out.tsPWS_sim, out.mtr * sizeof(double)}, {"tl_", "_convergence", out.ls_sim, out.mtr * sizeof(double)},
				{"ts_pws_", "_misfit", out.tsPWS_misfit, out.mtr * sizeof(double)}, {"tl_", "_misfit", out.ls_misfit, out.mtr * sizeof(double)},
				{"ts_pws_", "_steps", out.tsPWS_steps, (size_t)out.mtr * out.N * sizeof(float)}, {"tl_", "_steps", out.ls_steps, (size_t)out.mtr * out.N * sizeof(float)},
			};
			for (size_t j = 0; j < sizeof dumps / sizeof dumps[0]; j++) {
				if (!dumps[j].buf) continue;
				snprintf(name, sizeof name, "%s%s%s", dumps[j].pre, t, dumps[j].suf);
				FILE *f = fopen(name, "wb");
				if (f) { fwrite(dumps[j].buf, 1, dumps[j].bytes, f); fclose(f); }
			}
		}
		if (p.jackknife_n || p.subsmpl_N) {
			char sub[1100];
			if (tag) snprintf(sub, sizeof sub, "_%s_subsmpl", tag); else strcpy(sub, "_subsmpl");
			if (p.obin) {
				snprintf(name, sizeof name, "tl%s.bin", sub);
				write_bin(name, out.ls_subsmpl, out.M, first, &hdr, p.lkinst ? p.kinst : "t-lin", out.mtr_subsmpl);
				snprintf(name, sizeof name, "ts_pws%s.bin", sub);
				write_bin(name, out.tsPWS_subsmpl, out.M, first, &hdr, p.lkinst ? p.kinst : "ts_pws", out.mtr_subsmpl);
			} else
				for (unsigned m = 0; m < out.M; m++) {
					snprintf(name, sizeof name, "tl%s_%u.sac", sub, m);
					write_sac(name, out.ls_subsmpl[m] + first, &hdr, p.lkinst ? p.kinst : "t-lin", (float)out.mtr_subsmpl[m]);
					snprintf(name, sizeof name, "ts_pws%s_%u.sac", sub, m);
					write_sac(name, out.tsPWS_subsmpl[m] + first, &hdr, p.lkinst ? p.kinst : "ts_pws", (float)out.mtr_subsmpl[m]);
				}
		}
	}
done:
	free_out(&out);
	free_data(in);
	return er; /* the reference returns 0 even when tspws_main failed (:430); a non-zero status is more useful */
}

/* ---- several ensembles in one process: `ts_pws @batch.txt [options]` ----------------------------------------------- */
typedef struct { char *list, *tag; } batch_job;

/* lines `<filelist> [tag]`; tag defaults to the list's base name without its extension */
static batch_job *read_batch(const char *path, unsigned *n_out)
{
	unsigned nl = 0;
	char **lines = read_list(path, &nl);
	if (!lines) return NULL;
	batch_job *jobs = (batch_job *)calloc(nl ? nl : 1, sizeof *jobs);
	unsigned n = 0;
	for (unsigned i = 0; jobs && i < nl; i++) {
		char *l = lines[i];
		while (*l == ' ' || *l == '\t') l++;
		if (!*l || *l == '#') continue;
		char *e = l;
		while (*e && *e != ' ' && *e != '\t') e++;
		char *t = e;
		while (*t == ' ' || *t == '\t') t++;
		if (*e) *e = '\0';
		char *te = t;
		while (*te && *te != ' ' && *te != '\t') te++;
		*te = '\0';
		jobs[n].list = strdup(l);
		if (*t) jobs[n].tag = strdup(t);
		else {
			const char *b = strrchr(l, '/');
			b = b ? b + 1 : l;
			jobs[n].tag = strdup(b);
			char *dot = jobs[n].tag ? strrchr(jobs[n].tag, '.') : NULL;
			if (dot && dot != jobs[n].tag) *dot = '\0';
		}
		if (!jobs[n].list || !jobs[n].tag) { free(jobs[n].list); free(jobs[n].tag); break; }
		n++;
	}
	free_list(lines, nl);
	*n_out = n;
	return jobs;
}

typedef struct { t_data in; const char *list; int bin, verbose, rc; double t0, t1; } read_task;
static void *read_thread(void *arg)
{
	read_task *t = (read_task *)arg;
	t->t0 = now_s();
	t->rc = read_data(&t->in, t->list, t->bin, t->verbose);
	t->t1 = now_s();
	return NULL;
}

int main(int argc, char *argv[])
{
	const double t_start = now_s();
	/* defaults of the reference, ts_pws1f.c:140-142 */
	t_tsPWS p;
	memset(&p, 0, sizeof p);
	p.type = -1; p.V = 4; p.s0 = 2.; p.b0 = 1.0; p.w0 = PI * sqrt(2 / log(2)); p.wu = 2.; p.cycle = 2.;
	if (argc == 1) { usage(); return 0; }
	p.filein = argv[1];
	if (starts(p.filein, "info")) { usage(); return 0; }

	for (int i = 2; i < argc; i++) {
		const char *a = argv[i];
		if (starts(a, "wu=")) rd_double(&p.wu, a + 3);
		else if (starts(a, "J=")) rd_uint(&p.J, a + 2);
		else if (starts(a, "Nmax=")) rd_uint(&p.Nmax, a + 5);
		else if (starts(a, "fmin=")) rd_double(&p.fmin, a + 5);
		else if (starts(a, "V=")) { rd_uint(&p.V, a + 2); p.lVfix = 1; }
		else if (starts(a, "s0=")) { rd_double(&p.s0, a + 3); p.ls0fix = 1; }
		else if (starts(a, "b0=")) { rd_double(&p.b0, a + 3); p.lb0fix = 1; }
		else if (starts(a, "uni")) p.uni = 1;
		else if (starts(a, "rm")) p.lrm = 1;
		else if (starts(a, "bin")) p.bin = 1;
		else if (starts(a, "verbose")) p.verbose = 1;
		else if (starts(a, "fold")) p.fold = 1;
		else if (starts(a, "unbiased")) p.unbiased = 1;
		else if (starts(a, "MexHat")) p.type = -3;
		else if (starts(a, "AllSteps")) p.AllSteps = 1;
		else if (starts(a, "subsmpl_N=")) rd_uint(&p.subsmpl_N, a + 10);
		else if (starts(a, "subsmpl_prob=")) rd_double(&p.subsmpl_p, a + 13);
		else if (starts(a, "jackknife_n=")) rd_uint(&p.jackknife_n, a + 12);
		else if (starts(a, "jackknife_d=")) rd_uint(&p.jackknife_d, a + 12);
		else if (starts(a, "obin")) p.obin = 1;
		else if (starts(a, "TwoStage")) { p.Kmax = 10; if (starts(a, "TwoStage=")) rd_uint(&p.Kmax, a + 9); }
		else if (starts(a, "convergence")) { p.convergence = 1; if (starts(a, "convergence=")) p.fileconv = argv[i] + 12; }
		else if (starts(a, "Q=")) { rd_double(&p.Q, a + 2); if (p.w0set < 1) p.w0set = 1; }
		else if (starts(a, "cycles=")) { rd_double(&p.cycle, a + 7); if (p.w0set < 2) p.w0set = 2; }
		else if (starts(a, "cyc=")) { if (p.w0set < 3) { p.w0set = 3; rd_double(&p.w0, a + 4); p.w0 *= PI; } }
		else if (starts(a, "w0=")) { rd_double(&p.w0, a + 3); if (p.w0set < 4) p.w0set = 4; }
		else if (starts(a, "osac=")) { p.fileout = argv[i] + 5; if (!strlen(p.fileout)) p.fileout = NULL; }
		else if (starts(a, "kinst=")) { p.kinst = argv[i] + 6; p.lkinst = 1; }
		else if (starts(a, "info")) { usage(); return 0; }
	}

	if (p.filein[0] != '@') { /* the reference's command line: one ensemble */
		t_data in;
		memset(&in, 0, sizeof in);
		int er = read_data(&in, p.filein, p.bin, p.verbose);
		if (er) return er;
		const double t_read = now_s();
		if (!in.hdr.mtr) { free_data(&in); return 0; }
		return run_ensemble(p, &in, t_start, t_read);
	}

	/* several ensembles: the files of ensemble i + 1 are read while ensemble i is stacked */
	unsigned njobs = 0;
	batch_job *jobs = read_batch(p.filein + 1, &njobs);
	if (!jobs) { printf("tspws_main: cannot read the %s file\n", p.filein + 1); return -2; }
	int worst = 0;
	read_task cur, nxt;
	memset(&cur, 0, sizeof cur);
	if (njobs) { cur.list = jobs[0].list; cur.bin = p.bin; cur.verbose = p.verbose; read_thread(&cur); }
	for (unsigned j = 0; j < njobs; j++) {
		pthread_t th;
		int have_next = 0;
		if (j + 1 < njobs) {
			memset(&nxt, 0, sizeof nxt);
			nxt.list = jobs[j + 1].list; nxt.bin = p.bin; nxt.verbose = p.verbose;
			have_next = !pthread_create(&th, NULL, read_thread, &nxt);
			if (!have_next) read_thread(&nxt); /* (no thread: read it here) */
		}
		int er = cur.rc;
		if (!er && cur.in.hdr.mtr) {
			t_tsPWS q = p;
			q.filein = jobs[j].list;
			q.fileout = jobs[j].tag;
			er = run_ensemble(q, &cur.in, cur.t0, cur.t1);
		} else free_data(&cur.in);
		if (er) { printf("ts_pws: ensemble %u (%s) ended with status %d\n", j, jobs[j].list, er); if (!worst) worst = er; }
		if (j + 1 < njobs) {
			if (have_next) pthread_join(th, NULL);
			cur = nxt;
		}
	}
	for (unsigned j = 0; j < njobs; j++) { free(jobs[j].list); free(jobs[j].tag); }
	free(jobs);
	return worst;
}
