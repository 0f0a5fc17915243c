// plan.hip -- parameters, frame geometry, tap generation, plan object, runtime helpers, synthetic traces.
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

extern "C" const char *tspws_hip_last_error(void) { return g_err.c_str(); }

int tspws_fail(int code, const char *what, hipError_t e)
{
	g_err = what;
	if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
	return code;
}

// Lazily grown device scratch, one block per slot.  Blocks of slots whose pointers are handed to callers are retired, not
// freed, when they are outgrown (see tspws_hip_plan::retired).
int tspws_scratch(tspws_hip_plan *p, int slot, size_t bytes, void **out)
{
	if (p->scr_bytes[slot] < bytes) {
		void *fresh = nullptr;
		HIP_TRY(hipMalloc(&fresh, bytes));
		if (p->scr[slot]) {
			if (slot == SCR_P || slot == SCR_STPS || slot == SCR_JKP) p->retired.push_back(p->scr[slot]);
			else (void)hipFree(p->scr[slot]);
		}
		if (slot == SCR_TAB) p->ck_dev = false; // the device chunk table lived in the old block
		if (slot == SCR_TAB || slot == SCR_JKTAB) p->jk_gen = 0; // ... and so did the masked replicas' tables
		p->scr[slot] = fresh;
		p->scr_bytes[slot] = bytes;
	}
	*out = p->scr[slot];
	return 0;
}

// ------------------------------------------------------------------------------------------
// host-side parameter resolution (ts_pws1f_lib.c:91-124)
// ------------------------------------------------------------------------------------------
extern "C" void tspws_resolve_params(t_tsPWS *p, unsigned nsamp, float dt)
{
	if (p->fmin != 0 && p->fmin < 1 / (dt * nsamp)) {
		printf("Warning: fmin is too low. Replaced by the default value.\n");
		p->fmin = 0;
	}
	switch (p->w0set) {
	case 1: p->w0 = 2 * sqrt(log(2)) * p->Q; break;
	case 2: p->w0 = TSPWS_PI / sqrt(log(2)) * p->cycle; break;
	}
	if (p->type == -1 || p->type == -2) {
		const double rel = p->w0 / (TSPWS_PI * sqrt(2 / log(2)));
		if (!p->lVfix)  p->V  = (unsigned)ceil(4. * rel);
		if (!p->lb0fix) p->b0 = (unsigned)pow(2, round(log2(rel)));
		if (!p->ls0fix) p->s0 = 2.;
	} else if (p->type == -3) {
		p->w0 = sqrt(2);
		if (!p->lVfix)  p->V  = 2;
		if (!p->lb0fix) p->b0 = 0.5;
		if (!p->ls0fix) p->s0 = 1.;
	}
	if (p->fmin) {
		double top = p->w0 / (2 * TSPWS_PI * dt * p->fmin); // coarsest scale wanted
		if (p->J) {
			top /= pow(2, p->J - 1 / (double)p->V);           // -> finest scale
			while (top < p->s0 * 0.9) { top *= 2; p->J--; }
			p->s0 = top;
		} else p->J = (unsigned)floor(log2(top / p->s0) + 1 / (double)p->V);
	} else if (!p->J) {
		const double a = nsamp * p->w0 / (2 * TSPWS_PI * 4. * p->s0);
		p->J = (unsigned)floor(log2(a) + 1 / (double)p->V);
	}
}

// ------------------------------------------------------------------------------------------
// tap generation on the device (MorletFun :38-52, Complete_MorletFun :71-87,
// MexicanHatFun :119-131 + erfi :104-117, FillDualFrame :152-188 of FWTa/wavelet_def_v7.c)
// ------------------------------------------------------------------------------------------
__device__ double erfi_series(double z)
{
	const double zz = z * z;
	double term = z, sum = z;
	for (unsigned n = 1; n < 500; n++) {
		term *= zz / n;
		sum += term / (2 * n + 1);
	}
	return sum * (2 / sqrt(TSPWS_PI));
}

__global__ void __launch_bounds__(256) k_gen_taps(const ScaleDesc *__restrict__ sc, unsigned S, int type, double w0,
                                                  unsigned long long ntaps, double2 *__restrict__ w, double2 *__restrict__ wd)
{
	const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= ntaps) return;
	const unsigned s = find_scale(sc, S, t, true);
	const ScaleDesc d = sc[s];
	const unsigned l = (unsigned)(t - d.tap_off);
	const int u = (int)l - (int)(d.L / 2);
	const double inv = 1 / d.scale;
	double re, im;
	if (type == -3) {
		const double k = 2 / sqrt(3 * sqrt(TSPWS_PI) * d.scale);
		const double x = inv * u;
		const double a = (x * x - 1) * exp(-x * x / 2);
		re = k * a;
		im = k * (a * erfi_series(x / sqrt(2.0)) - sqrt(2 / TSPWS_PI) * x);
	} else {
		const double k = 1 / sqrt(sqrt(TSPWS_PI) * d.scale);
		double x = inv * u;
		const double ph = w0 * x;
		x *= x;
		double sn, cs;
		sincos(ph, &sn, &cs);
		if (type == -1) {
			const double e = exp(-0.5 * x);
			re = (k * cs) * e;
			im = (k * sn) * e;
		} else {
			const double ze = exp((-w0 * w0) / 2);
			const double e = k * exp(-0.5 * x);
			re = e * (cs - ze);
			im = e * sn;
		}
	}
	w[t] = make_double2(re, im);
	wd[d.tap_off + (d.L - 1 - l)] = make_double2(re, -im); // dual = conjugate, time reversed
}

static double cpsi_host(int type, double w0)
{
	if (type == -3) return (4. / 3.) * sqrt(TSPWS_PI);     // MexicanHat_Cpsi, wavelet_def_v7.c:146
	double acc = 0;                                        // Morlet_Cpsi :133-144, literal loop
	for (double om = 0.01; om < 100; om += 0.01) {
		double d = om - w0;
		d *= d;
		acc += exp(-d) / om;
	}
	return acc * (0.01 * sqrt(TSPWS_PI) / 2);
}

extern "C" int tspws_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" int tspws_hip_plan_create(tspws_hip_plan **out, int type, unsigned J, unsigned V, unsigned N, double s0,
                                     double b0, double w0, int uni, int device)
{
	if (!out) return fail(TSPWS_E_ARG, "plan_create: NULL plan pointer");
	*out = nullptr;
	if (type > -1 || type < -3) return fail(TSPWS_E_FRAME, "plan_create: only complex families -1/-2/-3");
	if (V == 0 || J == 0 || N == 0) return fail(TSPWS_E_FRAME, "plan_create: empty frame (J, V and N must be > 0)");
	if (tspws_engine_pin() < 0) return fail(TSPWS_E_ARG, "plan_create: TSPWS_ENGINE must be fir, spectral or auto");
	if (tspws_hip_device_count() <= device) return fail(TSPWS_E_NODEV, "plan_create: no such HIP device");
	HIP_TRY(hipSetDevice(device));

	tspws_hip_plan *p = new (std::nothrow) tspws_hip_plan;
	if (!p) return fail(TSPWS_E_NOMEM, "plan_create: host allocation");
	p->device = device; p->type = type; p->V = V; p->J = J; p->N = N; p->S = J * V;
	p->s0 = s0; p->b0 = uni ? 1.0 : b0; p->w0 = w0;
	p->Cpsi = cpsi_host(type, w0);
	const unsigned S = p->S;
	p->sc.resize(S);
	// geometry: setscales0 :299-309, setwaveletlength0 :312-322, setsampling0 :325-339
	double scv = s0;
	const double ratio = pow(2.0, 1.0 / (double)V);
	for (unsigned s = 0; s < S; s++) { p->sc[s].scale = scv; scv *= ratio; }
	double step = s0 * p->b0;
	for (unsigned j = 0, s = 0; j < J; j++) {
		const unsigned d = uni ? 1u : (step < 1.0 ? 1u : (unsigned)step);
		for (unsigned v = 0; v < V; v++) p->sc[s++].D = d;
		step *= 2.0;
	}
	unsigned long long toff = 0, coff = 0;
	for (unsigned s = 0; s < S; s++) {
		ScaleDesc &d = p->sc[s];
		const unsigned len = 2u * (unsigned)ceil(5.0 * d.scale) + 1u; // NSIGMAS = 5
		d.L = len > N ? N : len;
		d.c = (int)(d.L / 2u);
		d.cd = (int)d.L - 1 - d.c;
		d.Ns = (N + d.D - 1u) / d.D;
		d.tap_off = toff; d.coef_off = coff;
		d.gain = log(2.0) / (2 * p->Cpsi * V * d.scale);
		toff += d.L; coff += d.Ns;
	}
	p->ntaps = toff; p->ncoef = coff;
	if (int rc = tspws_build_forward(p)) { tspws_hip_plan_destroy(p); return rc; }
	for (unsigned s = 0; s < S; s++) {
		p->sc[s].inv_fast = (N % p->sc[s].D == 0) ? 1u : 0u;
		p->sc[s].acc_off = p->acc_blocks;
		p->acc_blocks += (p->sc[s].Ns + 255) / 256;
		p->sc[s].acc2_off = p->acc2_blocks;
		p->acc2_blocks += p->sc[s].nsplit > 1 ? (p->sc[s].Ns + 3) / 4 : (p->sc[s].Ns + 255) / 256;
	}

	hipError_t e;
	if ((e = hipMalloc(&p->d_sc, S * sizeof(ScaleDesc))) != hipSuccess ||
	    (e = hipMalloc(&p->d_w, p->ntaps * sizeof(double2))) != hipSuccess ||
	    (e = hipMalloc(&p->d_wd, p->ntaps * sizeof(double2))) != hipSuccess ||
	    (e = hipMemcpy(p->d_sc, p->sc.data(), S * sizeof(ScaleDesc), hipMemcpyHostToDevice)) != hipSuccess) {
		tspws_hip_plan_destroy(p);
		return fail(e == hipErrorOutOfMemory ? TSPWS_E_NOMEM : TSPWS_E_HIP, "plan_create: device tables", e);
	}
	if (int rc = tspws_build_inverse(p)) { tspws_hip_plan_destroy(p); return rc; }
	const unsigned nb = (unsigned)((p->ntaps + 255) / 256);
	hipLaunchKernelGGL(k_gen_taps, dim3(nb), dim3(256), 0, 0, p->d_sc, S, type, w0, (unsigned long long)p->ntaps, p->d_w, p->d_wd);
	if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
		tspws_hip_plan_destroy(p);
		return fail(e == hipErrorNoBinaryForGpu ? TSPWS_E_NODEV : TSPWS_E_HIP, "plan_create: tap kernel", e);
	}
	*out = p;
	return 0;
}

extern "C" void tspws_hip_plan_destroy(tspws_hip_plan *p)
{
	if (!p) return;
	(void)hipSetDevice(p->device);
	for (int i = 0; i < SCR_N; i++) if (p->scr[i]) (void)hipFree(p->scr[i]);
	for (void *b : p->retired) (void)hipFree(b);
	if (p->ck_ev) (void)hipEventDestroy(p->ck_ev);
	for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
	if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
	if (p->ev_join) (void)hipEventDestroy(p->ev_join);
	if (p->side) (void)hipStreamDestroy(p->side);
	if (p->xs) (void)hipStreamDestroy(p->xs);
	if (p->ev_xs0) (void)hipEventDestroy(p->ev_xs0);
	if (p->ev_xs1) (void)hipEventDestroy(p->ev_xs1);
	if (p->ev_xs2) (void)hipEventDestroy(p->ev_xs2);
	if (p->ev_mid) (void)hipEventDestroy(p->ev_mid);
	if (p->ev_lin) (void)hipEventDestroy(p->ev_lin);
	if (p->xf) (void)hipStreamDestroy(p->xf);
	for (hipEvent_t e : p->stage_ev) (void)hipEventDestroy(e);
	if (p->d_oc) (void)hipFree(p->d_oc);
	tspws_spectral_destroy(p);
	for (TlTable &T : p->tl) { if (T.d_sc) (void)hipFree(T.d_sc); if (T.d_items) (void)hipFree(T.d_items); }
	if (p->d_sc) (void)hipFree(p->d_sc);
	if (p->d_w) (void)hipFree(p->d_w);
	if (p->d_wd) (void)hipFree(p->d_wd);
	delete p;
}

extern "C" int tspws_hip_plan_info(const tspws_hip_plan *p, tspws_hip_frame_info *i)
{
	if (!p || !i) return fail(TSPWS_E_ARG, "plan_info: NULL");
	i->type = p->type; i->S = p->S; i->V = p->V; i->J = p->J; i->N = p->N;
	i->s0 = p->s0; i->b0 = p->b0; i->w0 = p->w0; i->Cpsi = p->Cpsi;
	i->ncoef = p->ncoef; i->ntaps = p->ntaps; i->device = p->device;
	return 0;
}

extern "C" int tspws_hip_plan_tables(const tspws_hip_plan *p, double *scale, unsigned *L, int *c, int *cd, unsigned *D, unsigned *Ns)
{
	if (!p) return fail(TSPWS_E_ARG, "plan_tables: NULL");
	for (unsigned s = 0; s < p->S; s++) {
		if (scale) scale[s] = p->sc[s].scale;
		if (L) L[s] = p->sc[s].L;
		if (c) c[s] = p->sc[s].c;
		if (cd) cd[s] = p->sc[s].cd;
		if (D) D[s] = p->sc[s].D;
		if (Ns) Ns[s] = p->sc[s].Ns;
	}
	return 0;
}

extern "C" int tspws_hip_plan_taps(const tspws_hip_plan *p, double *h_w, double *h_wd)
{
	if (!p) return fail(TSPWS_E_ARG, "plan_taps: NULL");
	HIP_TRY(hipSetDevice(p->device));
	if (h_w) HIP_TRY(hipMemcpy(h_w, p->d_w, p->ntaps * sizeof(double2), hipMemcpyDeviceToHost));
	if (h_wd) HIP_TRY(hipMemcpy(h_wd, p->d_wd, p->ntaps * sizeof(double2), hipMemcpyDeviceToHost));
	return 0;
}

// ------------------------------------------------------------------------------------------
// runtime helpers
// ------------------------------------------------------------------------------------------
extern "C" int tspws_hip_alloc(void **d, size_t bytes, int device)
{
	if (!d) return fail(TSPWS_E_ARG, "alloc: NULL");
	*d = nullptr;
	if (tspws_hip_device_count() <= device) return fail(TSPWS_E_NODEV, "alloc: no such HIP device");
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(hipMalloc(d, bytes ? bytes : 1));
	return 0;
}
extern "C" int tspws_hip_free(void *d) { if (d) HIP_TRY(hipFree(d)); return 0; }
// Large host buffers are pinned in place for the duration of the copy, PIECE BY PIECE: piece i + 1 is registered while piece i travels,
// so pinning fresh pages hides behind the link (tools/probes/upload_probe.py, 5.24 GB of fresh pages on the MI355X box: registered as
// a whole + copied 105 ms, plain pageable hipMemcpy 104 ms, 64-MB ... 1-GB pieces 93-97 ms; pages that were pinned before 91 ms = 57.6 GB/s
// either way -- the link).
static bool pin_for_copy(const void *h, size_t bytes)
{
	return bytes >= ((size_t)32 << 20) && hipHostRegister(const_cast<void *>(h), bytes, hipHostRegisterDefault) == hipSuccess;
}

extern "C" int tspws_hip_upload(void *d, const void *h, size_t bytes, void *s)
{
	const size_t piece = (size_t)128 << 20;
	if (bytes < 2 * piece) {
		const bool pinned = pin_for_copy(h, bytes);
		if (!pinned) (void)hipGetLastError(); // a failed registration is not an error: fall back to the pageable path
		hipError_t e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, S_(s));
		if (e == hipSuccess) e = hipStreamSynchronize(S_(s));
		if (pinned) (void)hipHostUnregister(const_cast<void *>(h));
		HIP_TRY(e);
		return 0;
	}
	// pieces end on page boundaries of the HOST address (two registrations must not share a page)
	const uintptr_t h0 = (uintptr_t)h, h1 = h0 + bytes;
	auto cut = [&](size_t k) -> uintptr_t { const uintptr_t a = (h0 + k * piece + 4095) & ~(uintptr_t)4095; return k == 0 ? h0 : std::min(a, h1); };
	hipError_t e = hipSuccess;
	size_t npinned = 0; // leading pieces that are registered (a piece that cannot be pinned travels pageable, like the rest after it)
	bool pin = true;
	for (size_t k = 0; cut(k) < h1 && e == hipSuccess; k++) {
		const uintptr_t a = cut(k), b = cut(k + 1);
		if (pin) {
			if (hipHostRegister((void *)a, b - a, hipHostRegisterDefault) == hipSuccess) npinned++;
			else { (void)hipGetLastError(); pin = false; }
		}
		e = hipMemcpyAsync((char *)d + (a - h0), (const void *)a, b - a, hipMemcpyHostToDevice, S_(s));
	}
	const hipError_t es = hipStreamSynchronize(S_(s));
	if (e == hipSuccess) e = es;
	for (size_t k = 0; k < npinned; k++) (void)hipHostUnregister((void *)cut(k));
	HIP_TRY(e);
	return 0;
}
extern "C" int tspws_hip_download(void *h, const void *d, size_t bytes, void *s)
{
	const bool pinned = pin_for_copy(h, bytes);
	if (!pinned) (void)hipGetLastError();
	hipError_t e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, S_(s));
	if (e == hipSuccess) e = hipStreamSynchronize(S_(s));
	if (pinned) (void)hipHostUnregister(h);
	HIP_TRY(e);
	return 0;
}
extern "C" int tspws_hip_zero(void *d, size_t bytes, void *s) { HIP_TRY(hipMemsetAsync(d, 0, bytes, S_(s))); return 0; }
extern "C" int tspws_hip_sync(void *s) { HIP_TRY(hipStreamSynchronize(S_(s))); return 0; }

// ------------------------------------------------------------------------------------------
// synthetic ensemble: counter-based noise so shards generate independently
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_synth(float *__restrict__ x, size_t N, size_t ld, unsigned long long seed, unsigned long long first)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const double c = (double)n - (double)N / 2;
	const double g = c / (0.05 * (double)N);
	const double sig = 0.2 * sin(2 * TSPWS_PI * c / 200.0) * exp(-0.5 * g * g);
	const unsigned long long i = (first + blockIdx.y) * (unsigned long long)N + n;
	const unsigned long long z = mix64(i + seed * 0xD1342543DE82EF95ull + 0x9E3779B97F4A7C15ull);
	const double u = (double)(z >> 40) * (1.0 / 16777216.0) - 0.5;
	x[(size_t)blockIdx.y * ld + n] = (float)(sig + u);
}

extern "C" int tspws_hip_synth(float *d_x, size_t mtr, size_t N, size_t ld, uint64_t seed, size_t first, void *s)
{
	if (!d_x) return fail(TSPWS_E_ARG, "synth: NULL");
	for (size_t t0 = 0; t0 < mtr; t0 += 65535) {
		const unsigned ny = (unsigned)std::min<size_t>(mtr - t0, 65535);
		hipLaunchKernelGGL(k_synth, dim3((unsigned)((N + 255) / 256), ny), dim3(256), 0, S_(s), d_x + t0 * ld, N, ld, seed, first + t0);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}
