/*
 * tspws_main.c -- the drop-in entry point, host side, plain C.
 *
 * Same signature, struct layout, return codes and in/out side effects as the
 * reference's tspws_main (/root/reference/src/ts_pws1f_lib.c:48-352); all
 * arithmetic on the traces runs on the MI355X through the C-ABI HIP layer of
 * include/tspws_hip.h.  There is no CPU fallback: without a usable device the
 * call fails with TSPWS_E_NODEV (5).
 *
 * Covered: fold (:71-88), parameter resolution (:91-124), mean removal
 * (:159-169), single- and two-stage stacks (:194-242), biased / unbiased
 * weighting (:226-228), convergence curves (:247-314), random subsampling
 * (:324-333), two-stage jackknife (:335-345).
 *
 * Several devices: TSPWS_DEVICES="0,1,2,3" (or "all") shards in->sigall by
 * traces over those devices of this process -- every device pulls its shard
 * over its own PCIe link, sums it (ts_pws1f_lib.c:866-881), ONE RCCL
 * all-reduce over xGMI adds the shards, the finish stage is split by scales
 * (include/tspws_hip.h, "several devices").  Convergence curves and random
 * subsampling need the whole ensemble on one device and take the single-
 * device path.
 *
 * The frame (plan) and the device trace buffer of the last call ON EACH DEVICE
 * are kept: a caller that stacks many ensembles of the same length with the
 * same wavelet parameters (the CLI in a loop, the MATLAB gateway) pays for the
 * frame once.  The cache is a small table with one slot and one lock per
 * device: tspws_main() uses the device TSPWS_DEVICE names (default 0), the
 * additive entry tspws_main_on(device, ...) any device, so a host that runs
 * one stacking thread per GPU is not serialised (calls on the SAME device
 * are: the slot's frame, scratch and trace buffer belong to one call at a
 * time).  tspws_main_release() frees everything; TSPWS_PLAN_CACHE=0 disables
 * the cache (each call then frees its device memory before it returns).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tspws_hip.h"

static int device_from_env(void)
{
	const char *e = getenv("TSPWS_DEVICE");
	return e ? atoi(e) : 0;
}

#define TRY(call) do { rc = (call); if (rc) goto done; } while (0)

/* ---- frame / trace-buffer cache: one slot per device, the last call's ---------------------------------------------------- */
#define TSPWS_MAX_DEVICES 64
typedef struct {
	pthread_mutex_t lock;  /* one call at a time per device: the slot's frame, its scratch buffers and the trace buffer belong to the holder */
	tspws_hip_plan *plan;
	int type, uni;
	unsigned J, V, N;
	double s0, b0, w0;
	float *d_sig;          /* device trace buffer ON THIS SLOT'S DEVICE, grown on demand */
	size_t sig_bytes;
} dev_slot;
static dev_slot g_slot[TSPWS_MAX_DEVICES];

/* several devices in one call (TSPWS_DEVICES): plans + communicator + shard buffers, keyed on the frame parameters and the device list */
static struct {
	tspws_hip_multi *multi;
	int mtype, muni, mndev, mdevs[TSPWS_MAX_DEVICES];
	unsigned mJ, mV, mN;
	double ms0, mb0, mw0;
} g_cache;
static pthread_mutex_t g_multi_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t g_rand_lock = PTHREAD_MUTEX_INITIALIZER;  /* one call's libc rand() draws (subsampling masks) stay contiguous */

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_slots(void)
{
	for (int i = 0; i < TSPWS_MAX_DEVICES; i++) pthread_mutex_init(&g_slot[i].lock, NULL);
}

static int cache_enabled(void)
{
	const char *e = getenv("TSPWS_PLAN_CACHE");
	return !(e && *e == '0');
}

static void slot_release_locked(dev_slot *sl)
{
	tspws_hip_plan_destroy(sl->plan);
	tspws_hip_free(sl->d_sig);
	sl->plan = NULL; sl->d_sig = NULL; sl->sig_bytes = 0;
}

static void multi_release_locked(void)
{
	tspws_hip_multi_destroy(g_cache.multi);
	memset(&g_cache, 0, sizeof g_cache);
}

void tspws_main_release(void)
{
	pthread_once(&g_once, init_slots);
	for (int i = 0; i < TSPWS_MAX_DEVICES; i++) {
		pthread_mutex_lock(&g_slot[i].lock);
		slot_release_locked(&g_slot[i]);
		pthread_mutex_unlock(&g_slot[i].lock);
	}
	pthread_mutex_lock(&g_multi_lock);
	multi_release_locked();
	pthread_mutex_unlock(&g_multi_lock);
}

/* test hook: which devices hold a cached frame right now (bit i = device i < 64) */
unsigned long long tspws_main_cached_devices(void)
{
	unsigned long long m = 0;
	pthread_once(&g_once, init_slots);
	for (int i = 0; i < TSPWS_MAX_DEVICES; i++) {
		pthread_mutex_lock(&g_slot[i].lock);
		if (g_slot[i].plan) m |= 1ull << i;
		pthread_mutex_unlock(&g_slot[i].lock);
	}
	return m;
}

/* TSPWS_DEVICES: "all" or a comma-separated list of HIP device ids; returns the number of entries (0: not set) */
static int devices_from_env(int *devs, int cap)
{
	const char *e = getenv("TSPWS_DEVICES");
	int n = 0;
	if (!e || !*e) return 0;
	if (!strcmp(e, "all")) {
		const int have = tspws_hip_device_count();
		for (; n < have && n < cap; n++) devs[n] = n;
		return n;
	}
	while (*e && n < cap) {
		char *end;
		const long v = strtol(e, &end, 10);
		if (end == e) break;
		devs[n++] = (int)v;
		e = (*end == ',') ? end + 1 : end;
	}
	return n;
}

static int get_multi(tspws_hip_multi **m, const t_tsPWS *p, unsigned N, const int *devs, int ndev)
{
	if (g_cache.multi && g_cache.mtype == p->type && g_cache.mJ == p->J && g_cache.mV == p->V && g_cache.mN == N && g_cache.ms0 == p->s0 &&
	    g_cache.mb0 == p->b0 && g_cache.mw0 == p->w0 && g_cache.muni == (int)p->uni && g_cache.mndev == ndev &&
	    !memcmp(g_cache.mdevs, devs, (size_t)ndev * sizeof(int))) {
		*m = g_cache.multi;
		return 0;
	}
	tspws_hip_multi *fresh = NULL;
	int rc = tspws_hip_multi_create(&fresh, ndev, devs, p->type, p->J, p->V, N, p->s0, p->b0, p->w0, (int)p->uni);
	if (rc) return rc;
	tspws_hip_multi_destroy(g_cache.multi);
	g_cache.multi = fresh;
	g_cache.mtype = p->type; g_cache.mJ = p->J; g_cache.mV = p->V; g_cache.mN = N; g_cache.ms0 = p->s0; g_cache.mb0 = p->b0; g_cache.mw0 = p->w0;
	g_cache.muni = (int)p->uni; g_cache.mndev = ndev;
	memcpy(g_cache.mdevs, devs, (size_t)ndev * sizeof(int));
	*m = fresh;
	return 0;
}

/* fold / mean removal of the host traces on `dev`, mirrored back (what is left of a call whose frame cannot be built: the
 * reference rewrites sigall, :71-88 and :159-169, before it fails with 4 at :199-204) */
static int prologue_only(t_data *in, size_t mtr, int do_fold, int lrm, int dev)
{
	const size_t ld = (size_t)in->hdr.max, bytes = mtr * ld * sizeof(float);
	float *d = NULL;
	int rc = 0;
	if ((!do_fold && !lrm) || !bytes) return 0;
	TRY(tspws_hip_alloc((void **)&d, bytes, dev));
	TRY(tspws_hip_upload(d, in->sigall, bytes, NULL));
	if (do_fold) TRY(tspws_hip_fold(d, mtr, ld, ld, NULL));
	if (lrm) TRY(tspws_hip_remove_mean(d, mtr, ld, ld, NULL));
	TRY(tspws_hip_download(in->sigall, d, bytes, NULL));
done:
	tspws_hip_free(d);
	return rc;
}

/* The call over trace shards on several devices (no convergence curves, no random subsampling: the caller checked). */
static int main_multi(t_tsPWS *tspws, t_tsPWS_out *out, t_data *in, size_t mtr, int do_fold, const int *devs, int ndev)
{
	const size_t ld = (size_t)in->hdr.max;
	const unsigned nsamp = (unsigned)in->hdr.max;
	tspws_hip_multi *m = NULL;
	const float *const *d_shards = NULL;
	float *d_ls = NULL, *d_ts = NULL, *jk = NULL;
	char *sel = NULL;
	int rc = get_multi(&m, tspws, nsamp, devs, ndev);
	if (rc) {
		printf("tspws_main: cannot set up the devices / the wavelet frame (%s)\n", tspws_hip_last_error());
		if (rc == TSPWS_E_NODEV) return rc;
		/* like the single-device path and the reference: a frame that cannot be built fails AFTER the prologue has rewritten sigall */
		(void)prologue_only(in, mtr, do_fold, tspws->lrm, devs[0]);
		return TSPWS_E_NOMEM;
	}
	TRY(tspws_hip_multi_upload(m, in->sigall, ld, mtr, &d_shards, &d_ls, &d_ts));
	TRY(tspws_hip_multi_prologue(m, in->sigall, ld, ld, mtr, do_fold, tspws->lrm));
	const int want_jk = tspws->jackknife_n > 0 && tspws->jackknife_d > 0 && tspws->Kmax && tspws->Kmax <= mtr && out->M && out->ls_subsmpl &&
	                    out->tsPWS_subsmpl && out->mtr_subsmpl;
	int jk_ready = 0;
	if (want_jk) {
		const unsigned C = out->M;
		sel = (char *)malloc((size_t)C * mtr);
		jk = (float *)malloc(2 * (size_t)C * ld * sizeof(float));
		if (!sel || !jk) { rc = TSPWS_E_NOMEM; goto done; }
		if (tspws_jackknife_plan(sel, in->time, mtr, tspws->jackknife_d, tspws->jackknife_n, C) == 0) {
			jk_ready = 1;
			TRY(tspws_hip_multi_stack_jackknife(m, tspws, d_shards, ld, mtr, d_ls, d_ts, sel, C, jk, jk + (size_t)C * ld, out->mtr_subsmpl));
			for (unsigned c = 0; c < C; c++) {
				memcpy(out->ls_subsmpl[c], jk + (size_t)c * ld, ld * sizeof(float));
				memcpy(out->tsPWS_subsmpl[c], jk + ((size_t)C + c) * ld, ld * sizeof(float));
			}
		} else
			printf("tspws_main: jackknife needs trace start times (binary input); replicas left untouched.\n");
	}
	if (!jk_ready) TRY(tspws_hip_multi_stack(m, tspws, d_shards, ld, mtr, d_ls, d_ts));
	TRY(tspws_hip_download(out->ls, d_ls, ld * sizeof(float), NULL));
	TRY(tspws_hip_download(out->tsPWS, d_ts, ld * sizeof(float), NULL));
done:
	if (rc) printf("tspws_main: HIP path failed (%d: %s)\n", rc, tspws_hip_last_error());
	free(sel);
	free(jk);
	if (!cache_enabled()) multi_release_locked();
	return rc;
}

/* the frame of (type, J, V, N, s0, b0, w0, uni) on the slot's device: the cached one when every parameter matches bit for bit */
static int get_plan(dev_slot *sl, tspws_hip_plan **plan, const t_tsPWS *p, unsigned N, int dev)
{
	if (sl->plan && sl->type == p->type && sl->J == p->J && sl->V == p->V && sl->N == N && sl->s0 == p->s0 &&
	    sl->b0 == p->b0 && sl->w0 == p->w0 && sl->uni == (int)p->uni) {
		*plan = sl->plan;
		return 0;
	}
	tspws_hip_plan *fresh = NULL;
	int rc = tspws_hip_plan_create(&fresh, p->type, p->J, p->V, N, p->s0, p->b0, p->w0, (int)p->uni, dev);
	if (rc) return rc;
	tspws_hip_plan_destroy(sl->plan);
	sl->plan = fresh;
	sl->type = p->type; sl->J = p->J; sl->V = p->V; sl->N = N; sl->s0 = p->s0; sl->b0 = p->b0; sl->w0 = p->w0;
	sl->uni = (int)p->uni;
	*plan = fresh;
	return 0;
}

/* the slot's trace buffer lives on the slot's device by construction (slot index == device) */
static int get_trace_buffer(dev_slot *sl, float **d_sig, size_t bytes, int dev)
{
	if (sl->d_sig && sl->sig_bytes < bytes) { tspws_hip_free(sl->d_sig); sl->d_sig = NULL; sl->sig_bytes = 0; }
	if (!sl->d_sig) {
		int rc = tspws_hip_alloc((void **)&sl->d_sig, bytes, dev);
		if (rc) return rc;
		sl->sig_bytes = bytes;
	}
	*d_sig = sl->d_sig;
	return 0;
}

static int run_call(t_tsPWS *tspws, t_tsPWS_out *out, t_data *in, int dev, int allow_multi);
static int main_single(dev_slot *sl, t_tsPWS *tspws, t_tsPWS_out *out, t_data *in, size_t mtr, int do_fold, int dev, char *sub_sel);

int tspws_main(t_tsPWS *tspws, t_tsPWS_out *out, t_data *in)
{
	if (tspws == NULL || out == NULL || in == NULL) { printf("tspws_main: NULL input\n"); return -1; }
	return run_call(tspws, out, in, device_from_env(), 1);
}

/* tspws_main on an explicit device (additive: include/tspws_hip.h).  TSPWS_DEVICE / TSPWS_DEVICES are not consulted: a host
 * that runs one stacking thread per GPU names the device itself, and calls on different devices run concurrently. */
int tspws_main_on(int device, t_tsPWS *tspws, t_tsPWS_out *out, t_data *in)
{
	if (tspws == NULL || out == NULL || in == NULL) { printf("tspws_main: NULL input\n"); return -1; }
	return run_call(tspws, out, in, device, 0);
}

static int run_call(t_tsPWS *tspws, t_tsPWS_out *out, t_data *in, int dev, int allow_multi)
{
	const int    max   = in->hdr.max;
	size_t mtr = tspws->Nmax ? tspws->Nmax : in->hdr.mtr;
	const float  beg   = in->hdr.beg, dt = in->hdr.dt;
	const unsigned nsamp = (unsigned)max;
	int do_fold = 0;

	/* The reference reads past the end of sigall when Nmax exceeds the trace count (:65, unchecked); here the request
	 * is clamped instead of faulting. */
	if (tspws->Nmax > in->hdr.mtr) {
		printf("tspws_main: Nmax = %u exceeds the %u traces given; using them all.\n", tspws->Nmax, in->hdr.mtr);
		mtr = in->hdr.mtr;
	}

	/* fold test, same float arithmetic as :72 */
	if (tspws->fold) {
		if (2 * beg + (max - 1) * dt > 0.5 * dt) {
			printf("Warning: Folding ignored. B = %f, E = %f, nsamp = %u\n", beg, beg + (max - 1) * dt, nsamp);
			tspws->fold = 0;
		} else do_fold = 1;
	}
	tspws_resolve_params(tspws, nsamp, dt);

	if (tspws->verbose) {
		printf("Sequence length = %d, dt = %f\n", nsamp, dt);
		printf("  PWS power = %f\n", tspws->wu);
		if (tspws->Kmax) printf("  Two-stage on, %d groups\n", tspws->Kmax); else printf("  Two-stage off\n");
		printf("  unbiased %s\n  rm %s\n  fold %s\n", tspws->unbiased ? "on" : "off", tspws->lrm ? "on" : "off", tspws->fold ? "on" : "off");
		printf("Mother wavelet: %s\n", tspws->type == -1 ? "complex Morlet" : tspws->type == -2 ? "exact complex Morlet" : "complex Mexican hat");
		printf("Sampling of the time-frequency domain:\n");
		printf("         V = %d, J = %d, b0 = %f, s0 = %f, w0 = %f\n", tspws->V, tspws->J, tspws->b0, tspws->s0, tspws->w0);
		printf("This is: fmax = %f Hz, fmin = %f Hz\n", tspws->w0 / (2 * PI * tspws->s0 * dt),
		       tspws->w0 / (2 * PI * dt * tspws->s0 * pow(2, tspws->J - 1 / (double)tspws->V)));
		printf("         Q = %f (equivalently cycles = %f)\n", tspws->w0 / (2 * sqrt(log(2))), tspws->w0 * sqrt(log(2)) / PI);
		if (tspws->jackknife_n > 0 && tspws->jackknife_d > 0)
			printf("Jackknife Resampling:\n  n = %d, d = %d\n", tspws->jackknife_n, tspws->jackknife_d);
		printf("Engine: MI355X HIP path, device %d\n\n", dev);
	}

	if (!mtr) return 0; /* the reference builds the family and returns 0 without touching out (:194) -- with or without a device */

	/* Random subsampling (:324-333): the masks are drawn NOW, before the first HIP call of the process -- the initialisation of the
	 * HIP runtime consumes libc rand() values, and the masks must come from the state the caller seeded (srand), as the reference's
	 * do.  Nothing else on this path calls rand(), so the order of the draws is the reference's (SubsamplingPlan, :355-383). */
	char *sub_sel = NULL;
	const int want_sub = tspws->subsmpl_N > 0 && tspws->subsmpl_p > 0 && out->ls_subsmpl && out->tsPWS_subsmpl && tspws->J && tspws->V;
	if (want_sub) {
		const unsigned M = tspws->subsmpl_N;
		const size_t K = (size_t)ceil((double)mtr * tspws->subsmpl_p);
		sub_sel = (char *)malloc((size_t)M * mtr);
		if (!sub_sel) return 4;
		/* one call's draws stay contiguous in libc's rand() stream when callers on several devices arrive together (each call used to
		 * draw under the one global lock of the library) */
		pthread_mutex_lock(&g_rand_lock);
		for (unsigned m = 0; m < M; m++) tspws_subsampling_plan(sub_sel + (size_t)m * mtr, mtr, K);
		pthread_mutex_unlock(&g_rand_lock);
	}

	int devs[TSPWS_MAX_DEVICES];
	int ndev = allow_multi ? devices_from_env(devs, TSPWS_MAX_DEVICES) : 0;
	/* a one-entry TSPWS_DEVICES names THE device of the call (the sharded path on one device only under TSPWS_COMM: tests) */
	if (ndev == 1 && !getenv("TSPWS_COMM")) { dev = devs[0]; ndev = 0; }

	/* several devices (or TSPWS_COMM set: the sharded path even on one device -- tests); convergence curves and random subsampling
	 * need the whole ensemble in one place: such a call runs on the FIRST device of the list, not on TSPWS_DEVICE's default */
	const int needs_one = (tspws->convergence && out->ls_sim && out->tsPWS_sim && out->ls_misfit && out->tsPWS_misfit) ||
	                      (tspws->subsmpl_N > 0 && tspws->subsmpl_p > 0 && out->ls_subsmpl && out->tsPWS_subsmpl);
	if (ndev >= 1) dev = devs[0];
	if (dev < 0 || dev >= TSPWS_MAX_DEVICES || tspws_hip_device_count() <= dev) {
		printf("tspws_main: no usable HIP device %d (%s)\n", dev, tspws_hip_last_error());
		free(sub_sel);
		return TSPWS_E_NODEV;
	}
	pthread_once(&g_once, init_slots);

	if (ndev >= 1) {
		if (!needs_one && tspws->J && tspws->V) {
			free(sub_sel);
			pthread_mutex_lock(&g_multi_lock);
			const int rc = main_multi(tspws, out, in, mtr, do_fold, devs, ndev);
			pthread_mutex_unlock(&g_multi_lock);
			return rc;
		}
	}

	dev_slot *sl = &g_slot[dev];
	pthread_mutex_lock(&sl->lock);
	const int rc = main_single(sl, tspws, out, in, mtr, do_fold, dev, sub_sel);
	pthread_mutex_unlock(&sl->lock);
	return rc;
}

static int main_single(dev_slot *sl, t_tsPWS *tspws, t_tsPWS_out *out, t_data *in, size_t mtr, int do_fold, int dev, char *sub_sel)
{
	const int max = in->hdr.max;
	const unsigned nsamp = (unsigned)max;
	int rc = 0;
	tspws_hip_plan *plan = NULL;
	float *d_sig = NULL, *d_out = NULL, *d_jk = NULL, *d_ref = NULL, *d_steps_ts = NULL, *d_steps_ls = NULL;
	char *sel = NULL;
	unsigned *jk_mtr = NULL;
	float *stage = NULL;
	const size_t ld = (size_t)max;

	int frame_rc = get_plan(sl, &plan, tspws, nsamp, dev);
	if (frame_rc) {
		printf("tspws_main: cannot build the wavelet frame (%s)\n", tspws_hip_last_error());
		if (frame_rc == TSPWS_E_NODEV) { free(sub_sel); return frame_rc; }
		/* A frame that cannot be built (e.g. J resolved to 0: fmin above the first scale) surfaces as 4 in the reference,
		 * when the coefficient containers are created (:199-204) -- i.e. AFTER fold and mean removal have rewritten
		 * sigall (:71-88, :159-169): the prologue below still runs, then the call returns 4. */
		if (!do_fold && !tspws->lrm) { free(sub_sel); return TSPWS_E_NOMEM; }
	}

	/* the two output rows live behind the traces in the slot's buffer: no allocation per call (a small daily ensemble's call is
	 * ~0.1 ms of kernels -- an allocation, a free and a second blocking copy were as much again) */
	const size_t sig_bytes = (mtr * ld * sizeof(float) + 255) & ~(size_t)255;
	TRY(get_trace_buffer(sl, &d_sig, sig_bytes + 2 * ld * sizeof(float), dev));
	d_out = (float *)((char *)d_sig + sig_bytes);
	TRY(tspws_hip_upload(d_sig, in->sigall, mtr * ld * sizeof(float), NULL));

	/* in-place prologue on the device, then mirrored back: the caller sees the same mutated
	 * sigall the reference leaves behind (:83-84, :167) */
	if (do_fold) TRY(tspws_hip_fold(d_sig, mtr, (size_t)max, ld, NULL));
	if (tspws->lrm) TRY(tspws_hip_remove_mean(d_sig, mtr, (size_t)max, ld, NULL));
	if (do_fold || tspws->lrm) TRY(tspws_hip_download(in->sigall, d_sig, mtr * ld * sizeof(float), NULL));
	if (frame_rc) { rc = TSPWS_E_NOMEM; goto done_quiet; }

	/* jackknife masks first (host, :385-430): the stack below then streams the traces ONCE for its own groups and for every
	 * replica; the replicas stay on the device until the point where the reference computes them (:335-345) */
	int jk_ready = 0;
	if (tspws->jackknife_n > 0 && tspws->jackknife_d > 0 && tspws->Kmax && tspws->Kmax <= mtr && out->M && out->ls_subsmpl &&
	    out->tsPWS_subsmpl && out->mtr_subsmpl) {
		const unsigned C = out->M;
		sel = (char *)malloc((size_t)C * mtr);
		jk_mtr = (unsigned *)malloc((size_t)C * sizeof(unsigned));
		if (!sel || !jk_mtr) { rc = TSPWS_E_NOMEM; goto done; }
		if (tspws_jackknife_plan(sel, in->time, mtr, tspws->jackknife_d, tspws->jackknife_n, C) == 0) {
			jk_ready = 1;
			rc = tspws_hip_alloc((void **)&d_jk, 2 * (size_t)C * ld * sizeof(float), dev);
			if (!rc) rc = tspws_hip_stack_jackknife(plan, tspws, d_sig, ld, mtr, d_out, d_out + ld, sel, C, d_jk, d_jk + (size_t)C * ld, jk_mtr, NULL);
			if (rc) goto done;
		}
	}
	if (!jk_ready) TRY(tspws_hip_stack(plan, tspws, d_sig, ld, mtr, d_out, d_out + ld, NULL));
	if (out->ls + ld == out->tsPWS) TRY(tspws_hip_download(out->ls, d_out, 2 * ld * sizeof(float), NULL)); /* (adjacent rows: one copy) */
	else {
		stage = (float *)malloc(2 * ld * sizeof(float)); /* one blocking copy for both rows */
		if (!stage) { rc = TSPWS_E_NOMEM; goto done; }
		TRY(tspws_hip_download(stage, d_out, 2 * ld * sizeof(float), NULL));
		memcpy(out->ls, stage, ld * sizeof(float));
		memcpy(out->tsPWS, stage + ld, ld * sizeof(float));
		free(stage); stage = NULL;
	}

	/* convergence curves, :247-314 */
	if (tspws->convergence && out->ls_sim && out->tsPWS_sim && out->ls_misfit && out->tsPWS_misfit) {
		const float *d_ref_ts = d_out + ld, *d_ref_ls = d_out;
		if (in->reference) { /* alternative reference trace for both curves (:277, :298) */
			TRY(tspws_hip_alloc((void **)&d_ref, ld * sizeof(float), dev));
			TRY(tspws_hip_upload(d_ref, in->reference, ld * sizeof(float), NULL));
			d_ref_ts = d_ref_ls = d_ref;
		}
		if (out->tsPWS_steps) TRY(tspws_hip_alloc((void **)&d_steps_ts, mtr * ld * sizeof(float), dev));
		if (out->ls_steps) TRY(tspws_hip_alloc((void **)&d_steps_ls, mtr * ld * sizeof(float), dev));
		TRY(tspws_hip_convergence(plan, tspws, d_sig, ld, mtr, d_ref_ts, d_ref_ls, out->tsPWS_sim, out->tsPWS_misfit, out->ls_sim,
		                          out->ls_misfit, d_steps_ts, d_steps_ls, NULL));
		if (d_steps_ts) TRY(tspws_hip_download(out->tsPWS_steps, d_steps_ts, mtr * ld * sizeof(float), NULL));
		if (d_steps_ls) TRY(tspws_hip_download(out->ls_steps, d_steps_ls, mtr * ld * sizeof(float), NULL));
	}

	/* random subsampling, :324-333 */
	if (tspws->subsmpl_N > 0 && tspws->subsmpl_p > 0 && out->ls_subsmpl && out->tsPWS_subsmpl) {
		const unsigned M = tspws->subsmpl_N;
		float *d_sub = NULL;
		rc = tspws_hip_alloc((void **)&d_sub, 2 * (size_t)M * ld * sizeof(float), dev);
		if (!rc) rc = sub_sel ? tspws_hip_subsample_sel(plan, tspws, d_sig, ld, mtr, M, sub_sel, d_sub, d_sub + (size_t)M * ld, NULL)
		                      : tspws_hip_subsample(plan, tspws, d_sig, ld, mtr, M, d_sub, d_sub + (size_t)M * ld, NULL);
		if (!rc) { stage = (float *)malloc(2 * (size_t)M * ld * sizeof(float)); if (!stage) rc = TSPWS_E_NOMEM; }
		if (!rc) rc = tspws_hip_download(stage, d_sub, 2 * (size_t)M * ld * sizeof(float), NULL);
		tspws_hip_free(d_sub);
		if (rc) goto done;
		for (unsigned m = 0; m < M; m++) {
			memcpy(out->ls_subsmpl[m], stage + (size_t)m * ld, ld * sizeof(float));
			memcpy(out->tsPWS_subsmpl[m], stage + ((size_t)M + m) * ld, ld * sizeof(float));
		}
		free(stage); stage = NULL;
	}

	if (tspws->jackknife_n > 0 && tspws->jackknife_d > 0 && tspws->Kmax && tspws->Kmax <= mtr) {
		const unsigned C = out->M;
		if (C && out->ls_subsmpl && out->tsPWS_subsmpl && out->mtr_subsmpl) {
			if (jk_ready) { /* computed with the stack (one pass over the traces); handed over here, where the reference does */
				stage = (float *)malloc(2 * (size_t)C * ld * sizeof(float));
				if (!stage) { rc = TSPWS_E_NOMEM; goto done; }
				TRY(tspws_hip_download(stage, d_jk, 2 * (size_t)C * ld * sizeof(float), NULL));
				for (unsigned c = 0; c < C; c++) {
					memcpy(out->ls_subsmpl[c], stage + (size_t)c * ld, ld * sizeof(float));
					memcpy(out->tsPWS_subsmpl[c], stage + ((size_t)C + c) * ld, ld * sizeof(float));
					out->mtr_subsmpl[c] = jk_mtr[c];
				}
			} else
				printf("tspws_main: jackknife needs trace start times (binary input); replicas left untouched.\n");
		}
	}

done:
	if (rc) printf("tspws_main: HIP path failed (%d: %s)\n", rc, tspws_hip_last_error());
done_quiet:
	free(sub_sel);
	free(stage);
	free(jk_mtr);
	free(sel);
	tspws_hip_free(d_steps_ls);
	tspws_hip_free(d_steps_ts);
	tspws_hip_free(d_ref);
	tspws_hip_free(d_jk);
	if (!cache_enabled()) slot_release_locked(sl); /* else: the frame and the trace buffer serve the next call on this device */
	return rc;
}
