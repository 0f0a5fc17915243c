// tspws_hip.hip -- gfx950 (MI355X) kernels and the thin C-ABI declared in include/tspws_hip.h.
//
// Layout in HBM
//   traces      float  [mtr][ld]          row-major, one trace per row (the reference's sigall)
//   partials    double [Kmax][ldP]        stage-1 group sums of the two-stage stack
//   taps        double2 [ntaps]           ragged per scale, tap_off[s] .. ; dual taps likewise
//   coefficients double2 [ncoef]          ragged [S][N_s], coef_off[s] ..  (N_s = ceil(N/D_s))
// All arithmetic on the path is FP64 (the reference is double / double complex throughout);
// MFMA is deliberately unused: the per-scale FIRs are skinny 1-D correlations.
//
// Reference citations are relative to /root/reference/src.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "tspws_hip.h"

#ifndef TSPWS_PI
#define TSPWS_PI 3.14159265358979328
#endif

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

extern "C" const char *tspws_hip_last_error(void) { return g_err.c_str(); }

static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
	g_err = what;
	if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
	return code;
}

#define HIP_TRY(expr)                                                              \
	do {                                                                           \
		hipError_t e_ = (expr);                                                    \
		if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? TSPWS_E_NOMEM : TSPWS_E_HIP, #expr, e_); \
	} while (0)

static inline hipStream_t S_(void *s) { return (hipStream_t)s; }

// ------------------------------------------------------------------------------------------
// plan
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
	unsigned acc_off;           // first 256-coefficient block of this scale in k_accumulate_parts
	unsigned use_lds;           // 1: forward transform by k_fwd_lds (fwd_lds.h), 0: k_fwd_poly
	unsigned lds_off, lds_bps;  // first workgroup of this scale in k_fwd_lds, workgroups per split
	unsigned acc2_off;          // first block of this scale in k_accumulate_parts (32 coefficients per block when split)
	unsigned fuse_ok;           // 1: k_fwd_lds<FUSE> keeps this scale's linear / phase stacks in registers (no partials)
	unsigned use_oct;           // 1: forward transform by k_fwd_oct (fwd_oct.h) together with the other voices of its octave
	unsigned r16;               // 1: the direct kernel gives a thread 16 outputs of this scale (ngw counts 16-output groups)
	unsigned long long part_off; // offset of this scale's [nsplit][Ns] partial block
};

#include "fwd_mfma_types.h"
#define FM_KQCAP_HOST FM_KQCAP
#define FM_TAMAX_HOST FM_TAMAX
struct OctDesc;
struct OctFwd;
static int build_oct_forward(tspws_hip_plan *p, std::vector<char> &is_oct); // defined next to the forward kernels
static int upload_oct_forward(tspws_hip_plan *p);
static int build_tl_forward(tspws_hip_plan *p, unsigned FWD_STEPS);

#ifndef FL_PASSES
#define FL_PASSES 2
#endif
#ifndef FL_WAVES
#define FL_WAVES 4
#endif
#define FL_PASSES_HOST FL_PASSES
#ifndef FL_PASSES_FINE
#define FL_PASSES_FINE 1
#endif

struct Chunk { // one streaming work item of the partial-stack kernel
	unsigned long long t0; // first local trace
	unsigned count;        // traces
	unsigned row;          // destination row (group / class)
};

enum { SCR_Y = 0, SCR_PART, SCR_XT, SCR_OBUF, SCR_SEL, SCR_SUBST, SCR_CONV, SCR_CHUNK, SCR_P, SCR_STPS, SCR_OUT, SCR_X2, SCR_CLS, SCR_JKP, SCR_JKOUT, SCR_TAB, SCR_FZ, SCR_XCM, SCR_JKTAB, SCR_N };

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
	unsigned n_fusable = 0;    // scales whose stacks the fused forward kernel keeps in registers
	unsigned oct_wgs = 0, oct_n = 0; // k_fwd_oct / k_fwd_oct2: workgroups per trace slice, table entries
	int oct_mode = 2;                // 1: k_fwd_oct (256 threads, voice subsets, two workgroups per CU), 2: k_fwd_oct2 (two-team pipeline)
	size_t oct_lds = 0;
	struct OctFwd *d_ofw = nullptr;
	// many-trace decomposition (fwd_tl.h): second scale table, trace-lane work items
	std::vector<ScaleDesc> sc_tl;
	ScaleDesc *d_sc_tl = nullptr;
	struct TLItem *d_tl = nullptr;
	unsigned tl_n = 0, tl_wgs = 0, tl_waves = 0, tl_acc2_blocks = 0; // items, workgroups per trace block, direct-kernel waves, accumulate blocks
	size_t tl_npart = 0, tl_lds = 0;
	std::vector<unsigned char> ofw_host; // the OctFwd table (bytes; the struct is defined next to the kernel)
	unsigned cm_n[2] = {0, 0}, cm_D[2][8] = {{0}}, cm_MC[2][8] = {{0}}; // chunk-major copies by input type [float, double]
	size_t cm_per_trace[2] = {0, 0};                                    // elements per trace of all copies
	int fwd_kind = 1;          // 0: k_fwd_poly only, 1: k_fwd_lds (+poly), 3: k_fwd_mfma (opt-in)
	std::vector<FwdGroup> pairs; // fwd_kind 3: work groups, their B tables and work items
	FwdGroup *d_pairs = nullptr;
	double *d_bt = nullptr;
	size_t mfma_lds = 0;
	std::vector<unsigned> oc_s0, oc_nv, oc_wave_off, oc_nwaves, oc_gen; // host copy of the inverse's octave items (launch order)
	unsigned inv_waves = 0, inv_waves_fast = 0, inv_noct = 0, inv_ngeneric = 0; // polyphase inverse: waves (of the octaves whose D divides N first), octave items, scales left to the generic kernel
	struct OctDesc *d_oc = nullptr;
	std::vector<ScaleDesc> sc;
	ScaleDesc *d_sc = nullptr;
	double2 *d_w = nullptr, *d_wd = nullptr;
	// work tables of the forward kernels
	unsigned n_short = 0, n_long = 0;           // scales handled wave-per-output / block-per-output
	std::vector<unsigned> long_scales;
	// lazily grown device scratch
	void *scr[SCR_N] = {nullptr};
	size_t scr_bytes[SCR_N] = {0};
	// pipelined single-GPU call: second stream, per-group events
	hipStream_t aux = nullptr;
	std::vector<hipEvent_t> ev_grp;
	hipEvent_t ev_done = nullptr;
	// forward transform: the direct kernel (coarse scales, latency-bound) runs beside the LDS kernel (FP64-bound) on a
	// side stream, forked from and joined back into the caller's stream
	hipStream_t side = nullptr, side2 = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
	// optional timing of the streaming stage inside tspws_hip_stack (bench.py roofline leg)
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
	// class sums of the last masked pass over the traces (jackknife / two-stage subsampling) and, when a jackknife was
	// announced with tspws_hip_jackknife_prepare, of the plain two-stage groups as well: ONE streaming pass then serves
	// the stack and all its replicas
	struct ClassSums {
		bool valid = false, has_main = false;
		const float *d_x = nullptr;
		size_t ld = 0, mtr = 0;                         // mtr: traces of the WHOLE ensemble the selection refers to
		size_t first = 0, mtr_local = 0;                // the shard the sums were taken over (first = 0, mtr_local = mtr: all)
		unsigned C = 0, KM = 0, ncls = 0;
		std::vector<char> sel;                          // the selection the classes were built from
		std::vector<std::vector<unsigned short>> sig;   // per class: group in each replica (0xFFFF = deleted) [+ plain group]
		std::vector<size_t> Kc;                         // traces per replica
		std::vector<Chunk> chunks;                      // host tables stay alive while copies may be in flight
		std::vector<unsigned> row_first, rp, cols;
	} cs;
	unsigned last_stream_launches = 0; // k_partial launches of the last streaming pass (bench.py: per-launch roofline figures)
	bool jk_prepared = false;
	std::vector<char> jk_sel;
	unsigned jk_C = 0, jk_KM = 0;
	size_t jk_mtr = 0;
	// blocks of exported slots (tspws_hip_reduce_buffer hands out SCR_P / SCR_STPS) that were outgrown: a caller may
	// still hold the old pointer (e.g. as the buffer of an in-flight collective), so they live until plan_destroy
	std::vector<void *> retired;
	// Scale sub-range of the finish-stage launches (sharded finish, tspws_hip_stack_finish_scales): scales [rs0, rs1);
	// rs1 == 0 means all.  The launches take their workgroup / wave / block ranges from the per-scale offset tables.
	unsigned rs0 = 0, rs1 = 0;
	bool ranged() const { return rs1 != 0; }
	unsigned lds_begin() const { return ranged() ? sc[rs0].lds_off : 0u; }
	unsigned lds_end() const { return ranged() && rs1 < S ? sc[rs1].lds_off : lds_blocks; }
	unsigned wav_begin() const { return ranged() ? sc[rs0].wave_off : 0u; }
	unsigned wav_end() const { return ranged() && rs1 < S ? sc[rs1].wave_off : fwd_waves; }
	unsigned acc_begin() const { return ranged() ? sc[rs0].acc2_off : 0u; }
	unsigned acc_end() const { return ranged() && rs1 < S ? sc[rs1].acc2_off : acc2_blocks; }
};

static int scratch(tspws_hip_plan *p, int slot, size_t bytes, void **out)
{
	if (p->scr_bytes[slot] < bytes) {
		void *fresh = nullptr;
		HIP_TRY(hipMalloc(&fresh, bytes));
		if (p->scr[slot]) {
			if (slot == SCR_P || slot == SCR_STPS) p->retired.push_back(p->scr[slot]); // exported: see `retired`
			else (void)hipFree(p->scr[slot]);
		}
		if (slot == SCR_TAB) p->ck_dev = false; // the device chunk table lived in the old block
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

static int build_inverse_items(tspws_hip_plan *p); // defined next to the inverse kernels
static int upload_mfma_tables(tspws_hip_plan *p);  // defined next to the forward kernels

extern "C" int tspws_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// Work decomposition of k_fwd_mfma (fwd_mfma.h): the voices of an octave (same D) are paired, up to two pairs share a
// work group (one x image), every group gets its tile geometry, its split of the phases and the partial-block layout of
// its voices (which replaces the one of the VALU kernels).  false = a filter is too long / too many groups.
static bool build_mfma_pairs(tspws_hip_plan *p)
{
	unsigned TARGET = 1024; // matrix instructions per wave (unit run) the decomposition aims at
	if (const char *e = getenv("TSPWS_MFMA_TARGET")) TARGET = (unsigned)std::max(16, atoi(e));
	unsigned TAMAX = FM_TAMAX_HOST;
	if (const char *e = getenv("TSPWS_MFMA_TA")) TAMAX = (unsigned)std::max(1, std::min((int)FM_TAMAX_HOST, atoi(e)));
	p->pairs.clear();
	unsigned long long btoff = 0, poff = 0;
	size_t lds = 0;
	std::vector<unsigned> new_nsplit(p->S);           // committed to the scale table only when every group fits
	std::vector<unsigned long long> new_poff(p->S);
	for (unsigned s = 0; s < p->S;) {
		// voices of this octave that share the decimation
		unsigned e = s + 1;
		while (e < p->S && e / p->V == s / p->V && p->sc[e].D == p->sc[s].D && p->sc[e].Ns == p->sc[s].Ns) e++;
		const unsigned D = p->sc[s].D;
		for (unsigned g0 = s; g0 < e; g0 += 4) {
			const unsigned g1 = std::min(e, g0 + 4);
			FwdGroup d;
			memset(&d, 0, sizeof d);
			d.s0 = g0; d.D = D; d.Ns = p->sc[g0].Ns;
			d.NP = (g1 - g0 + 1) / 2;
			long long cg = 0;
			for (unsigned v = g0; v < g1; v++) cg = std::max<long long>(cg, p->sc[v].c);
			d.cp = cg;
			unsigned rows_needed = 0; // max over the pairs of 4 Kq + row offset
			for (unsigned pi = 0; pi < d.NP; pi++) {
				const unsigned v0 = g0 + 2 * pi, v1 = std::min(v0 + 1, g1 - 1);
				d.nv[pi] = v1 > v0 ? 2 : 1;
				const long long cmax = std::max(p->sc[v0].c, p->sc[v1].c);
				const unsigned r = (unsigned)((cg - cmax) / (long long)D); // whole rows between the group origin and the pair origin
				const long long cpp = cg - (long long)r * D;
				d.rofs[pi] = r;
				unsigned Lp = 0;
				for (unsigned v = v0; v <= v1; v++) Lp = std::max(Lp, p->sc[v].L + (unsigned)(cpp - p->sc[v].c));
				const unsigned Qp = (Lp + D - 1) / D;
				d.Kq[pi] = (Qp + 3) / 4;
				if (d.Kq[pi] > FM_KQCAP_HOST) { if (getenv("TSPWS_DEBUG")) fprintf(stderr, "mfma: Kq %u at scale %u\n", d.Kq[pi], v0); return false; }
				rows_needed = std::max(rows_needed, 4 * d.Kq[pi] + r);
			}
			unsigned mc = 1, lg = 0;
			while (mc < D && mc < 4) { mc <<= 1; lg++; }
			d.Mc = mc; d.logMc = lg;
			d.MC = (D + mc - 1) / mc;
			const unsigned kqsum = d.Kq[0] + d.Kq[1];
			unsigned tq = 1;
			while (tq < TAMAX && 4 * tq < d.Ns) tq <<= 1;
			d.TQ = tq;
			d.nob = (d.Ns + 4 * tq - 1) / (4 * tq);
			d.RT = 4 * tq + rows_needed - 1;
			const unsigned rpi = 64 / mc;
			unsigned P = d.RT + rpi;            // slack: staging rows past RT; nothing is read past RT
			while (P % 32 != 16) P++;
			d.P = P;
			unsigned Pu = mc * P;
			while (Pu % 32 != 8) Pu++;          // the four units of an operand read hit disjoint banks
			d.Pu = Pu;
			const unsigned per_chunk = mc * tq * kqsum;            // matrix instructions per chunk and quad of units
			const unsigned bper = mc * kqsum * 16;                 // B doubles per chunk
			d.cps = std::max(1u, std::min(d.MC, TARGET / std::max(1u, per_chunk)));
			d.nsplit = (d.MC + d.cps - 1) / d.cps;
			// the accumulate kernel walks the splits of every coefficient: at most 64 of them, more (up to 512) only for the
			// coarse scales whose few outputs would otherwise leave most of the GPU without work
			const unsigned smax = std::max(64u, std::min(512u, 4096u / std::max(1u, d.Ns)));
			if (d.nsplit > smax) { d.cps = (d.MC + smax - 1) / smax; d.nsplit = (d.MC + d.cps - 1) / d.cps; }
			d.css = std::max(1u, std::min(d.cps, (8u * 256u) / bper)); // <= 16 KB of tiles per staged sub-split
			if (bper > 8u * 256u) { if (getenv("TSPWS_DEBUG")) fprintf(stderr, "mfma: bper %u\n", bper); return false; }
			d.bl_doubles = (d.css * bper + 63) & ~63u;
			const unsigned nss = (d.cps + d.css - 1) / d.css;
			unsigned nuq = nss > 1 ? 1u : std::max(1u, TARGET / std::max(1u, per_chunk * d.cps)); // quads of units per wave
			nuq = std::min(nuq, 16u);                                                            // one descriptor lane per unit
			d.upi = 16 * nuq;
			d.bper = bper;
			d.bt_off[0] = btoff;                                     // [chunk][pair][quad][kappa][64]
			d.bt_off[1] = btoff + (unsigned long long)mc * d.Kq[0] * 16;
			btoff += (unsigned long long)d.MC * bper;
			lds = std::max(lds, ((size_t)d.bl_doubles + 16 * (size_t)d.Pu + 64) * sizeof(double));
			for (unsigned v = g0; v < g1; v++) { // partial blocks [nsplit][Ns] of the voices
				new_nsplit[v] = d.nsplit; new_poff[v] = poff; d.po[v - g0] = poff;
				poff += (unsigned long long)d.nsplit * d.Ns;
			}
			p->pairs.push_back(d);
		}
		s = e;
	}
	if (p->pairs.size() > FM_MAXGROUPS) { if (getenv("TSPWS_DEBUG")) fprintf(stderr, "mfma: %zu groups\n", p->pairs.size()); return false; }
	if (getenv("TSPWS_DEBUG"))
		for (const FwdGroup &d : p->pairs)
			fprintf(stderr, "mfma group s0=%u D=%u Ns=%u NP=%u Kq=%u,%u rofs=%u,%u Mc=%u MC=%u cps=%u nsplit=%u css=%u TQ=%u nob=%u upi=%u RT=%u P=%u bper=%u\n", d.s0, d.D, d.Ns, d.NP, d.Kq[0], d.Kq[1], d.rofs[0], d.rofs[1], d.Mc, d.MC, d.cps, d.nsplit, d.css, d.TQ, d.nob, d.upi, d.RT, d.P, d.bper);
	for (unsigned v = 0; v < p->S; v++) { p->sc[v].nsplit = new_nsplit[v]; p->sc[v].part_off = new_poff[v]; }
	p->npart = poff;
	p->mfma_lds = lds;
	p->fwd_waves = 0; p->lds_blocks = 0;
	return true;
}

extern "C" int tspws_hip_plan_create(tspws_hip_plan **out, int type, unsigned J, unsigned V, unsigned N, double s0,
                                     double b0, double w0, int uni, int device)
{
	if (!out) return fail(TSPWS_E_ARG, "plan_create: NULL plan pointer");
	*out = nullptr;
	if (type > -1 || type < -3) return fail(TSPWS_E_FRAME, "plan_create: only complex families -1/-2/-3");
	if (V == 0 || J == 0 || N == 0) return fail(TSPWS_E_FRAME, "plan_create: empty frame (J, V and N must be > 0)");
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
	{ // forward decomposition.  Scales with >= 8 output groups and (D >= 64 or D a power of two) run on the
	  // LDS-staged kernel; the rest (very coarse scales, odd small decimations) on the direct kernel, which
	  // aims at ~FWD_STEPS tap steps per wave.
		unsigned FWD_STEPS = 96; // tap steps per wave of the direct kernel (more splits = shorter dependent load chains)
		if (const char *e = getenv("TSPWS_FWD_STEPS")) FWD_STEPS = (unsigned)std::max(8, atoi(e));
		const unsigned R = 8, FL_SLOTS_HOST = FL_WAVES * FL_PASSES_HOST;
		int kind = 1;
		if (const char *e = getenv("TSPWS_FWD_KERNEL")) kind = !strcmp(e, "poly") ? 0 : !strcmp(e, "mfma") ? 3 : 1;
		if (getenv("TSPWS_FWD_NOLDS") && *getenv("TSPWS_FWD_NOLDS") == '1') kind = 0;
		p->fwd_kind = kind;
		// pass 1: per-scale geometry and the kernel that would take the scale on its own
		for (unsigned s = 0; s < S; s++) {
			ScaleDesc &d = p->sc[s];
			d.Q = (d.L + d.D - 1) / d.D;
			unsigned dl = 1, lg = 0;
			while (dl < d.D && dl < 64) { dl <<= 1; lg++; }
			d.DL = dl; d.logDL = lg;
			d.MC = d.D > 64 ? (d.D + 63) / 64 : 1;
			const unsigned NG = (d.Ns + R - 1) / R;
			const bool pow2 = (d.D & (d.D - 1)) == 0;
			d.use_lds = (kind != 0 && NG >= 8 && (d.D >= 64 || pow2)) ? 1u : 0u;
			d.use_oct = 0; d.r16 = 0;
		}
		// pass 2: octaves with D >= 64 whose voices all qualify go to the octave-fused kernel (one x window for all voices)
		std::vector<char> is_oct(S, 0);
		if (kind == 1) { if (int rc = build_oct_forward(p, is_oct)) { tspws_hip_plan_destroy(p); return rc; } }
		// pass 3: work decomposition and offsets
		unsigned woff = 0, boff = 0;
		unsigned long long poff = 0;
		for (unsigned s = 0; s < S; s++) {
			ScaleDesc &d = p->sc[s];
			const unsigned NG = (d.Ns + R - 1) / R, GW = 64 / d.DL;
			d.ngw = (NG + GW - 1) / GW;
			if (is_oct[s]) { d.use_oct = 1; d.use_lds = 0; }
			// direct kernel, 64 phase lanes, at least 16 outputs: 16 outputs per thread (half the operand bytes per FMA)
			static int r16_on = -1;
			if (r16_on < 0) { const char *e = getenv("TSPWS_POLY_R16"); r16_on = (e && *e == '0') ? 0 : 1; }
			if (r16_on && kind != 3 && !d.use_lds && !d.use_oct && d.DL == 64 && d.Ns >= 16) { d.r16 = 1; d.ngw = (d.Ns + 15) / 16; }
			unsigned cps;
			if (d.use_lds || d.use_oct) cps = 1; // one 64-phase chunk per workgroup: its taps stay resident in LDS
			else cps = std::max(1u, (FWD_STEPS + d.Q / 2) / std::max(1u, d.Q));
			d.cps = std::min(cps, d.MC);
			d.nsplit = (d.MC + d.cps - 1) / d.cps;
			d.wave_off = woff; d.lds_off = boff; d.part_off = poff;
			const unsigned slots = (d.D <= 4) ? FL_WAVES * FL_PASSES_FINE : FL_SLOTS_HOST; // as dispatched in k_fwd_lds
			d.lds_bps = (NG + slots * GW - 1) / (slots * GW);
			if (d.use_lds) boff += d.lds_bps * d.nsplit; else if (!d.use_oct) woff += d.ngw * d.nsplit;
			poff += (unsigned long long)d.nsplit * d.Ns;
			d.fuse_ok = (kind == 1 && d.use_lds && d.nsplit == 1) ? 1u : 0u;
		}
		p->fwd_waves = woff; p->lds_blocks = boff; p->npart = poff;
		if (int rc = upload_oct_forward(p)) { tspws_hip_plan_destroy(p); return rc; }
		if (kind == 1) { if (int rc = build_tl_forward(p, FWD_STEPS)) { tspws_hip_plan_destroy(p); return rc; } }
		for (unsigned s = 0; s < S; s++) p->n_fusable += p->sc[s].fuse_ok;
		if (kind == 3 && !build_mfma_pairs(p)) { // a filter too long for the matrix kernel: VALU kernels
			p->fwd_kind = 1; p->pairs.clear();
			for (unsigned s = 0; s < S; s++) { p->sc[s].fuse_ok = (p->sc[s].use_lds && p->sc[s].nsplit == 1) ? 1u : 0u; p->n_fusable += p->sc[s].fuse_ok; }
		}
	}
	for (unsigned s = 0; s < S; s++) {
		p->sc[s].inv_fast = (N % p->sc[s].D == 0) ? 1u : 0u;
		p->sc[s].acc_off = p->acc_blocks;
		p->acc_blocks += (p->sc[s].Ns + 255) / 256;
		p->sc[s].acc2_off = p->acc2_blocks;
		p->acc2_blocks += p->sc[s].nsplit > 1 ? (p->sc[s].Ns + 31) / 32 : (p->sc[s].Ns + 255) / 256;
	}

	hipError_t e;
	if ((e = hipMalloc(&p->d_sc, S * sizeof(ScaleDesc))) != hipSuccess ||
	    (e = hipMalloc(&p->d_w, p->ntaps * sizeof(double2))) != hipSuccess ||
	    (e = hipMalloc(&p->d_wd, p->ntaps * sizeof(double2))) != hipSuccess ||
	    (e = hipMemcpy(p->d_sc, p->sc.data(), S * sizeof(ScaleDesc), hipMemcpyHostToDevice)) != hipSuccess) {
		tspws_hip_plan_destroy(p);
		return fail(e == hipErrorOutOfMemory ? TSPWS_E_NOMEM : TSPWS_E_HIP, "plan_create: device tables", e);
	}
	if (int rc = build_inverse_items(p)) { tspws_hip_plan_destroy(p); return rc; }
	const unsigned nb = (unsigned)((p->ntaps + 255) / 256);
	hipLaunchKernelGGL(k_gen_taps, dim3(nb), dim3(256), 0, 0, p->d_sc, S, type, w0, (unsigned long long)p->ntaps, p->d_w, p->d_wd);
	if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
		tspws_hip_plan_destroy(p);
		return fail(e == hipErrorNoBinaryForGpu ? TSPWS_E_NODEV : TSPWS_E_HIP, "plan_create: tap kernel", e);
	}
	if (p->fwd_kind == 3) {
		if (int rc = upload_mfma_tables(p)) { tspws_hip_plan_destroy(p); return rc; }
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
	for (hipEvent_t e : p->ev_grp) (void)hipEventDestroy(e);
	for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
	if (p->ev_done) (void)hipEventDestroy(p->ev_done);
	if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
	if (p->ev_join) (void)hipEventDestroy(p->ev_join);
	if (p->ev_join2) (void)hipEventDestroy(p->ev_join2);
	if (p->side) (void)hipStreamDestroy(p->side);
	if (p->side2) (void)hipStreamDestroy(p->side2);
	if (p->aux) (void)hipStreamDestroy(p->aux);
	if (p->d_oc) (void)hipFree(p->d_oc);
	if (p->d_ofw) (void)hipFree(p->d_ofw);
	if (p->d_sc_tl) (void)hipFree(p->d_sc_tl);
	if (p->d_tl) (void)hipFree(p->d_tl);
	if (p->d_pairs) (void)hipFree(p->d_pairs);
	if (p->d_bt) (void)hipFree(p->d_bt);
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
// Large host buffers are pinned in place for the duration of the copy: measured on the MI355X box, a pageable 2 GB
// hipMemcpy runs at ~16 GB/s while hipHostRegister (~10 ms per GB) + copy runs at ~57 GB/s.
static bool pin_for_copy(const void *h, size_t bytes)
{
	return bytes >= ((size_t)32 << 20) && hipHostRegister(const_cast<void *>(h), bytes, hipHostRegisterDefault) == hipSuccess;
}

extern "C" int tspws_hip_upload(void *d, const void *h, size_t bytes, void *s)
{
	const bool pinned = pin_for_copy(h, bytes);
	if (!pinned) (void)hipGetLastError(); // a failed registration is not an error: fall back to the pageable path
	hipError_t e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, S_(s));
	if (e == hipSuccess) e = hipStreamSynchronize(S_(s));
	if (pinned) (void)hipHostUnregister(const_cast<void *>(h));
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
// wave / block reductions (wave = 64 lanes)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

// ------------------------------------------------------------------------------------------
// prologue kernels
// ------------------------------------------------------------------------------------------
// fold, ts_pws1f_lib.c:76-86.  grid.y = trace
__global__ void __launch_bounds__(256) k_fold(float *__restrict__ x, size_t max, size_t ld)
{
	float *row = x + (size_t)blockIdx.y * ld;
	const size_t half = max / 2;
	for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < half; n += (size_t)gridDim.x * blockDim.x) {
		float v = row[n];
		v += row[max - 1 - n];
		v *= 0.5f;
		row[max - 1 - n] = v;
		row[n] = v;
	}
}

// mean removal, ts_pws1f_lib.c:159-169: FP64 sum, mean rounded to float, float subtraction.
// One workgroup per trace (the trace is re-read from L2 for the subtraction).
__global__ void __launch_bounds__(1024) k_remove_mean(float *__restrict__ x, size_t max, size_t ld)
{
	__shared__ double part[16];
	__shared__ float meanf;
	float *row = x + (size_t)blockIdx.x * ld;
	double acc = 0;
	for (size_t n = threadIdx.x; n < max; n += blockDim.x) acc += (double)row[n];
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) {
		double t = 0;
		for (unsigned i = 0; i < blockDim.x / 64; i++) t += part[i];
		meanf = (float)(t / (double)max);
	}
	__syncthreads();
	const float m = meanf;
	for (size_t n = threadIdx.x; n < max; n += blockDim.x) row[n] -= m;
}

extern "C" int tspws_hip_fold(float *d_x, size_t mtr, size_t max, size_t ld, void *s)
{
	if (!d_x) return fail(TSPWS_E_ARG, "fold: NULL");
	if (!mtr || max < 2) return 0;
	const unsigned bx = (unsigned)std::min<size_t>((max / 2 + 255) / 256, 64);
	for (size_t t0 = 0; t0 < mtr; t0 += 65535) {
		const unsigned ny = (unsigned)std::min<size_t>(mtr - t0, 65535);
		hipLaunchKernelGGL(k_fold, dim3(bx, ny), dim3(256), 0, S_(s), d_x + t0 * ld, max, ld);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

extern "C" int tspws_hip_remove_mean(float *d_x, size_t mtr, size_t max, size_t ld, void *s)
{
	if (!d_x) return fail(TSPWS_E_ARG, "remove_mean: NULL");
	if (!mtr || !max) return 0;
	for (size_t t0 = 0; t0 < mtr; t0 += (1u << 30)) {
		const unsigned nb = (unsigned)std::min<size_t>(mtr - t0, 1u << 30);
		hipLaunchKernelGGL(k_remove_mean, dim3(nb), dim3(1024), 0, S_(s), d_x + t0 * ld, max, ld);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------------------------------
// stage 1: partial linear stacks (the HBM-streaming kernel)
//
// Work item = (column block, chunk): a chunk is a run of consecutive traces that all add into
// the same destination row.  Every thread owns 4 consecutive samples, walks the chunk's traces
// with independent 16-byte non-temporal loads (8 in flight), accumulates in FP64 and writes one
// partial row per chunk; k_reduce_chunks then adds the few chunk rows of each destination in a
// fixed order, so the result is deterministic (no atomics).
// Algorithmic bytes: 4 per input sample (+ 8 per output sample).
// ------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ void __launch_bounds__(256) k_partial(const float *__restrict__ x, size_t ld, size_t N,
                                                 const Chunk *__restrict__ chunks, double *__restrict__ pc, size_t ldpc)
{
	const Chunk ck = chunks[blockIdx.y];
	const size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
	if (col >= N) return;
	const float *src = x + ck.t0 * ld + col;
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	if (VEC4) {
		typedef float v4f __attribute__((ext_vector_type(4)));
		unsigned t = 0;
		for (; t + 8 <= ck.count; t += 8) {
			v4f v[8];
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)(t + j) * ld));
#pragma unroll
			for (int j = 0; j < 8; j++) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
		}
		for (; t < ck.count; t++) {
			const v4f v = __builtin_nontemporal_load((const v4f *)(src + (size_t)t * ld));
			a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
		}
	} else {
		const unsigned rem = (N - col) < 4 ? (unsigned)(N - col) : 4u;
		for (unsigned t = 0; t < ck.count; t++) {
			const float *r = src + (size_t)t * ld;
			a0 += (double)r[0];
			if (rem > 1) a1 += (double)r[1];
			if (rem > 2) a2 += (double)r[2];
			if (rem > 3) a3 += (double)r[3];
		}
	}
	double *dst = pc + (size_t)blockIdx.y * ldpc + col;
	if (VEC4) {
		*(double2 *)dst = make_double2(a0, a1);
		*(double2 *)(dst + 2) = make_double2(a2, a3);
	} else {
		const unsigned rem = (N - col) < 4 ? (unsigned)(N - col) : 4u;
		dst[0] = a0;
		if (rem > 1) dst[1] = a1;
		if (rem > 2) dst[2] = a2;
		if (rem > 3) dst[3] = a3;
	}
}

// P[row][n] = sum over the row's chunks (in chunk order); rows without chunks become 0.
__global__ void __launch_bounds__(256) k_reduce_chunks(const double *__restrict__ pc, size_t ldpc, const unsigned *__restrict__ row_first,
                                                       double *__restrict__ P, size_t ldP, size_t N)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const unsigned row = blockIdx.y;
	double acc = 0;
	for (unsigned c = row_first[row]; c < row_first[row + 1]; c++) acc += pc[(size_t)c * ldpc + n];
	P[(size_t)row * ldP + n] = acc;
}

// Device copy of a chunk table.  `cached` = the plan's own group table (build_group_chunks): uploaded once, valid only
// after the copy has been enqueued, and a call on another stream waits for that copy through ck_ev.  Any other table
// (masked replicas) is uploaded every time and invalidates the cached device copy, which shares the scratch block.
static int chunk_tables(tspws_hip_plan *p, const std::vector<Chunk> &chunks, const std::vector<unsigned> &row_first, unsigned rows,
                        hipStream_t st, bool cached, Chunk **d_chunks, unsigned **d_rf)
{
	const size_t nck = chunks.size();
	const size_t tab_bytes = nck * sizeof(Chunk) + (rows + 1) * sizeof(unsigned);
	void *d_tab = nullptr;
	int rc;
	if ((rc = scratch(p, SCR_TAB, std::max<size_t>(tab_bytes, 16), &d_tab))) return rc;
	*d_chunks = (Chunk *)d_tab;
	*d_rf = (unsigned *)((char *)d_tab + nck * sizeof(Chunk));
	if (cached && p->ck_dev) {
		if (st != p->ck_stream) HIP_TRY(hipStreamWaitEvent(st, p->ck_ev, 0));
		return 0;
	}
	p->ck_dev = false;
	if (nck) HIP_TRY(hipMemcpyAsync(*d_chunks, chunks.data(), nck * sizeof(Chunk), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(*d_rf, row_first.data(), (rows + 1) * sizeof(unsigned), hipMemcpyHostToDevice, st));
	if (cached) {
		if (!p->ck_ev) HIP_TRY(hipEventCreateWithFlags(&p->ck_ev, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(p->ck_ev, st));
		p->ck_stream = st;
		p->ck_dev = true; // only now: a failed upload must not leave a table that looks valid
	}
	return 0;
}

// Launch the streaming pass for an arbitrary chunk table (rows destinations).
static int run_chunks(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const std::vector<Chunk> &chunks,
                      const std::vector<unsigned> &row_first, unsigned rows, double *d_P, size_t ldP, hipStream_t st, bool cached,
                      unsigned row_begin = 0, unsigned row_end = ~0u)
{
	row_end = std::min(row_end, rows);
	const size_t nck = chunks.size();
	const size_t ldpc = (N + 3) & ~(size_t)3;
	void *d_pc = nullptr;
	Chunk *d_chunks = nullptr;
	unsigned *d_rf = nullptr;
	int rc;
	if ((rc = scratch(p, SCR_CHUNK, std::max<size_t>(nck * ldpc * sizeof(double), 16), &d_pc))) return rc;
	if ((rc = chunk_tables(p, chunks, row_first, rows, st, cached, &d_chunks, &d_rf))) return rc;
	const unsigned bx = (unsigned)((N + 1023) / 1024);
	const bool vec = (N % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)d_x & 15) == 0);
	const size_t ck0 = row_first[row_begin], ck1 = row_first[row_end]; // chunks are sorted by destination row
	// every destination row fed by exactly ONE chunk: the streaming kernel writes the rows themselves, no chunk reduction
	bool direct = ck1 - ck0 == (size_t)(row_end - row_begin) && ck1 - ck0 <= 65535 && (!vec || ldP % 2 == 0);
	for (unsigned r = row_begin; r < row_end && direct; r++) direct = row_first[r + 1] - row_first[r] == 1;
	if (direct) {
		// Rows per launch.  The HBM streams fastest when exactly ONE workgroup per CU marches down the traces and all of them
		// start together: 256 workgroups per launch 0.730 ms (7.2 TB/s), 384 / 640 / 1280 per launch 0.80 / 0.80 / 0.79 ms,
		// 128 (half the CUs) 1.05 ms; a persistent 256-workgroup kernel walking the same items without launch boundaries
		// 0.76 ms -- the boundaries keep the column blocks of a trace row in step, so the chip reads whole 512-KB rows
		// (gpurun_out/sweep_s12.txt, sweep_s13.txt).  Short chunks (class sums of masked replicas) stay in one launch:
		// there the extra launch boundaries would cost more than the rate gains.
		static int wg_target = -1;
		if (wg_target < 0) { const char *e = getenv("TSPWS_STREAM_WGS"); wg_target = e ? std::max(1, atoi(e)) : 256; }
		size_t rows_total = 0;
		for (size_t c = ck0; c < ck1; c++) rows_total += chunks[c].count;
		const bool long_runs = ck1 > ck0 && rows_total / (ck1 - ck0) >= 256;
		const unsigned rpl = long_runs ? std::max(1u, (unsigned)wg_target / std::max(1u, bx)) : 65535u;
		p->last_stream_launches = 0;
		for (size_t c0 = ck0; c0 < ck1; c0 += rpl) {
			p->last_stream_launches++;
			const unsigned ny = (unsigned)std::min<size_t>(ck1 - c0, rpl);
			double *dst = d_P + (size_t)(row_begin + (c0 - ck0)) * ldP;
			if (vec) hipLaunchKernelGGL(k_partial<true>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, dst, ldP);
			else hipLaunchKernelGGL(k_partial<false>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, dst, ldP);
		}
		HIP_TRY(hipGetLastError());
		return 0;
	}
	p->last_stream_launches = 0;
	for (size_t c0 = ck0; c0 < ck1; c0 += 65535) {
		p->last_stream_launches++;
		const unsigned ny = (unsigned)std::min<size_t>(ck1 - c0, 65535);
		if (vec) hipLaunchKernelGGL(k_partial<true>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, (double *)d_pc + c0 * ldpc, ldpc);
		else hipLaunchKernelGGL(k_partial<false>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, (double *)d_pc + c0 * ldpc, ldpc);
	}
	for (unsigned r0 = row_begin; r0 < row_end; r0 += 65535) {
		const unsigned ny = std::min(row_end - r0, 65535u);
		hipLaunchKernelGGL(k_reduce_chunks, dim3((unsigned)((N + 255) / 256), ny), dim3(256), 0, st, (const double *)d_pc, ldpc,
		                   d_rf + r0, d_P + (size_t)r0 * ldP, ldP, N);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// Chunk length of the streaming pass.  Round-2 sweep (10 000 x 131 072, K = 10): one 1000-trace chunk per group (1280
// workgroups, 5 per CU) streams at 0.795-0.80 ms, steadily; 2 / 3-4 chunks per group (2560 / 5120 workgroups) at 0.79-0.83 /
// 0.81-0.89 ms from one process to the next -- long runs per workgroup beat a full complement of waves.  Target: ~1024
// workgroups in all; a group that fits one chunk is written by the streaming kernel directly (no chunk reduction).
static unsigned chunk_len_for(size_t N, size_t mtr)
{
	const size_t colblocks = (N + 1023) / 1024;
	const size_t want = std::max<size_t>(1, 1024 / std::max<size_t>(colblocks, 1));
	size_t len = (mtr + want - 1) / want;
	len = std::max<size_t>(len, 8);
	return (unsigned)std::min<size_t>(len, 1u << 20);
}

// Chunk table of the two-stage streaming pass: every group's run of local traces is cut into equal pieces
// (group of global trace i: floor(i*Kmax/mtr_global), ts_pws1f_lib.c:876).  Cached in the plan.
static void build_group_chunks(tspws_hip_plan *p, size_t mtr_local, size_t first, size_t mtr_global, unsigned Kmax)
{
	if (p->ck_valid && p->ck_mtr == mtr_local && p->ck_first == first && p->ck_glob == mtr_global && p->ck_K == Kmax) return;
	p->ck_dev = false;
	p->chunks.clear();
	p->row_first.assign(Kmax + 1, 0);
	unsigned clen = chunk_len_for(p->N, mtr_local);
	if (const char *e = getenv("TSPWS_CHUNK_LEN")) clen = std::max(1, atoi(e));
	std::vector<std::vector<Chunk>> per(Kmax);
	size_t i = 0;
	while (i < mtr_local) {
		const size_t g = (size_t)floor((double)((first + i) * (size_t)Kmax) / (double)mtr_global);
		size_t j = i + 1;
		while (j < mtr_local && (size_t)floor((double)((first + j) * (size_t)Kmax) / (double)mtr_global) == g) j++;
		const size_t n = j - i, pieces = (n + clen - 1) / clen, base = n / pieces, rem = n % pieces;
		size_t t = i;
		for (size_t k = 0; k < pieces; k++) {
			Chunk c; c.t0 = t; c.count = (unsigned)(base + (k < rem ? 1 : 0)); c.row = (unsigned)g;
			per[std::min<size_t>(g, Kmax - 1)].push_back(c);
			t += c.count;
		}
		i = j;
	}
	for (unsigned g = 0; g < Kmax; g++) {
		p->row_first[g] = (unsigned)p->chunks.size();
		p->chunks.insert(p->chunks.end(), per[g].begin(), per[g].end());
	}
	p->row_first[Kmax] = (unsigned)p->chunks.size();
	p->ck_mtr = mtr_local; p->ck_first = first; p->ck_glob = mtr_global; p->ck_K = Kmax; p->ck_valid = true;
}

extern "C" int tspws_hip_partial_stacks(tspws_hip_plan *p, const float *d_x, size_t ld, size_t mtr_local, size_t first,
                                        size_t mtr_global, unsigned Kmax, double *d_P, size_t ldP, void *stream)
{
	// an empty shard (mtr_local == 0, d_x may be NULL) is legal: its rows become zeros, so that every rank of a
	// trace-sharded call reaches the collective
	if (!p || (!d_x && mtr_local) || !d_P || !Kmax || !mtr_global || first + mtr_local > mtr_global)
		return fail(TSPWS_E_ARG, "partial_stacks: bad argument");
	HIP_TRY(hipSetDevice(p->device));
	build_group_chunks(p, mtr_local, first, mtr_global, Kmax);
	return run_chunks(p, d_x, ld, p->N, p->chunks, p->row_first, Kmax, d_P, ldP, S_(stream), true);
}

extern "C" int tspws_hip_partial_stacks_range(tspws_hip_plan *p, const float *d_x, size_t ld, size_t mtr_local, size_t first,
                                              size_t mtr_global, unsigned Kmax, unsigned g_begin, unsigned g_end, double *d_P, size_t ldP,
                                              void *stream)
{
	if (!p || (!d_x && mtr_local) || !d_P || !Kmax || !mtr_global || g_begin > g_end || g_end > Kmax || first + mtr_local > mtr_global)
		return fail(TSPWS_E_ARG, "partial_stacks_range: bad argument");
	HIP_TRY(hipSetDevice(p->device));
	build_group_chunks(p, mtr_local, first, mtr_global, Kmax);
	return run_chunks(p, d_x, ld, p->N, p->chunks, p->row_first, Kmax, d_P, ldP, S_(stream), true, g_begin, g_end);
}

// ------------------------------------------------------------------------------------------
// forward frame CWT, generic form (any D, any L <= N):
//   Y_s[k] = conj( sum_l x[(k D - c + l) mod N] w_s[l] )          cdotx.c:44-70
// Lanes run along the taps (coalesced x and tap reads), partial sums are combined with
// wave shuffles.  WAVE_PER_OUT: one wave per coefficient; otherwise one 256-thread block.
// ------------------------------------------------------------------------------------------
template <typename TIn, bool BLOCK_PER_OUT>
__global__ void __launch_bounds__(256) k_fwd_generic(const TIn *__restrict__ x, size_t ld, unsigned N, const ScaleDesc *__restrict__ sc,
                                                     unsigned S, const double2 *__restrict__ w, double2 *__restrict__ Y, size_t ncoef,
                                                     unsigned long long first_coef, unsigned long long n_items)
{
	__shared__ double red[8];
	const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned long long item = BLOCK_PER_OUT ? blockIdx.x : (unsigned long long)blockIdx.x * 4 + wv;
	if (item >= n_items) return;
	const unsigned long long ci = first_coef + item;
	const unsigned s = find_scale(sc, S, ci, false);
	const ScaleDesc d = sc[s];
	const unsigned k = (unsigned)(ci - d.coef_off);
	const TIn *xr = x + (size_t)blockIdx.y * ld;
	const double2 *ws = w + d.tap_off;
	long long n0 = (long long)k * d.D - d.c;
	if (n0 < 0) n0 += N;
	double re = 0, im = 0;
	const unsigned stride = BLOCK_PER_OUT ? 256 : 64;
	for (unsigned l = BLOCK_PER_OUT ? threadIdx.x : lane; l < d.L; l += stride) {
		unsigned long long idx = (unsigned long long)n0 + l;
		if (idx >= N) idx -= N;
		const double xv = (double)xr[idx];
		const double2 t = ws[l];
		re = fma(xv, t.x, re);
		im = fma(xv, t.y, im);
	}
	re = wave_sum(re);
	im = wave_sum(im);
	double2 *out = Y + (size_t)blockIdx.y * ncoef + ci;
	if (BLOCK_PER_OUT) {
		if (lane == 0) { red[wv * 2] = re; red[wv * 2 + 1] = im; }
		__syncthreads();
		if (threadIdx.x == 0) {
			re = red[0] + red[2] + red[4] + red[6];
			im = red[1] + red[3] + red[5] + red[7];
			*out = make_double2(re, -im);
		}
	} else if (lane == 0) *out = make_double2(re, -im);
}

template <typename TIn>
static int forward_generic(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_Y, hipStream_t st)
{
	if (!p || !d_x || !d_Y) return fail(TSPWS_E_ARG, "forward: NULL");
	if (!ntr) return 0;
	HIP_TRY(hipSetDevice(p->device));
	// scales are ordered by growing L: coefficients of scales with L > 2048 go block-per-output
	unsigned s_long = p->S;
	for (unsigned s = 0; s < p->S; s++) if (p->sc[s].L > 2048) { s_long = s; break; }
	const unsigned long long n_short = s_long < p->S ? p->sc[s_long].coef_off : p->ncoef;
	const unsigned long long n_long = p->ncoef - n_short;
	for (size_t t0 = 0; t0 < ntr; t0 += 65535) {
		const unsigned ny = (unsigned)std::min<size_t>(ntr - t0, 65535);
		const TIn *xx = d_x + t0 * ld;
		double2 *yy = (double2 *)d_Y + t0 * p->ncoef;
		if (n_long)
			hipLaunchKernelGGL((k_fwd_generic<TIn, true>), dim3((unsigned)n_long, ny), dim3(256), 0, st, xx, ld, p->N, p->d_sc, p->S,
			                   p->d_w, yy, p->ncoef, n_short, n_long);
		if (n_short)
			hipLaunchKernelGGL((k_fwd_generic<TIn, false>), dim3((unsigned)((n_short + 3) / 4), ny), dim3(256), 0, st, xx, ld, p->N,
			                   p->d_sc, p->S, p->d_w, yy, p->ncoef, 0ull, n_short);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// Output of the fused forward + phase-stack kernel: slice j of the launch (traces [j*tps, (j+1)*tps)) leaves the linear and
// phase stacks of every fuse_ok scale in accST / accPS + j*stride ([ncoef] planes).
struct FuseOut {
	double2 *accST = nullptr, *accPS = nullptr;
	size_t stride = 0;
	unsigned tps = 1;
	bool applied = false; // set by forward_parts when the fused kernel ran
};

static bool side_stream_enabled()
{
	static int v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_SIDE_STREAM"); v = (e && *e == '0') ? 0 : 1; }
	return v == 1;
}

static bool fuse_enabled()
{
	static int v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_FUSE"); v = (e && *e == '0') ? 0 : 1; }
	return v == 1;
}

#include "fwd_poly.h"
#include "fwd_lds.h"
#include "fwd_mfma.h"
static_assert(FM_KQCAP == FM_KQCAP_HOST, "tap-step cap");

#include "fwd_oct.h"

// Octaves (runs of scales with the same D and Ns) that the octave-fused kernel takes: D >= 64, every voice eligible for the
// LDS kernel, tap images + x image within the LDS budget.  TSPWS_FWD_OCT=0 leaves them on k_fwd_lds.
static int build_oct_forward(tspws_hip_plan *p, std::vector<char> &is_oct)
{
	p->ofw_host.clear(); p->oct_wgs = 0; p->oct_n = 0; p->oct_lds = 0;
	// opt-in: measured on MI355X the octave-fused kernels do not beat k_fwd_lds yet (forward-transform notes)
	{ const char *e = getenv("TSPWS_FWD_OCT"); if (!e || *e != '1') return 0; }
	p->oct_mode = 2;
	if (const char *e = getenv("TSPWS_OCT_MODE")) p->oct_mode = atoi(e) == 1 ? 1 : 2;
	const bool m2 = p->oct_mode == 2;
	// LDS of a work item with `rows` tap rows, tallest voice qmax, spread of a_v da: mode 1 one image of 64 outputs, mode 2 two
	// team images of 32 outputs (+ pad rows for the operand prefetch)
	auto x_rows = [&](unsigned qmax, unsigned da) { return m2 ? ((31 + qmax + da + 3) & ~3u) : ((63 + qmax + da + 7) & ~7u); };
	auto lds_of = [&](unsigned rows, unsigned xr) {
		return (size_t)rows * 64 * sizeof(double2) + (m2 ? 2 * (size_t)(xr + FT_XPAD) : (size_t)xr) * 64 * sizeof(double);
	};
	const unsigned xr_max = m2 ? FT_XRMAX : FO_XRMAX;
	const size_t lds_max = m2 ? FT_LDS_MAX : FO_LDS_MAX;
	std::vector<OctFwd> tab;
	unsigned wg = 0;
	for (unsigned s = 0; s < p->S;) {
		unsigned e = s + 1;
		while (e < p->S && p->sc[e].D == p->sc[s].D && p->sc[e].Ns == p->sc[s].Ns) e++;
		const unsigned D = p->sc[s].D, Ns = p->sc[s].Ns, nv = e - s;
		bool ok = D >= 64;
		for (unsigned v = s; v < e && ok; v++) ok = p->sc[v].use_lds != 0;
		if (ok) {
			// tap rows of every voice: c = a D + b, tap row q' holds taps (q' - 1) D + rho + b, rho < D
			std::vector<unsigned> a(nv), b(nv), qr(nv);
			for (unsigned v = 0; v < nv; v++) {
				const ScaleDesc &d = p->sc[s + v];
				a[v] = (unsigned)d.c / D; b[v] = (unsigned)d.c % D;
				const long long num = (long long)d.L - 1 - (long long)b[v];
				const long long fl = num >= 0 ? num / (long long)D : -1; // floor division: largest q' - 1 that holds a tap
				qr[v] = ((unsigned)(fl + 2) + 3) & ~3u;
				if (qr[v] > FO_QMAX || (unsigned long long)(a[v] + 1) * D > p->N) ok = false;
			}
			// voice subsets by first-fit-decreasing on the tap rows: an item takes a voice when its tap images plus the x image
			// (whose height depends on the subset's spread of a_v) still fit the LDS budget -- e.g. 4 Morlet voices with
			// 12 / 16 / 16 / 20 rows and 78 KB give {3, 0} and {1, 2}: 32 rows each
			std::vector<unsigned> order(nv);
			for (unsigned v = 0; v < nv; v++) order[v] = v;
			std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return qr[x] > qr[y]; });
			struct Bin { std::vector<unsigned> vs; unsigned rows = 0, amax = 0, amin = ~0u, qmax = 0, xr = 0; };
			std::vector<Bin> bins;
			for (unsigned oi = 0; oi < nv && ok; oi++) {
				const unsigned v = order[oi];
				bool placed = false;
				for (size_t bi = 0; bi <= bins.size() && !placed; bi++) {
					if (bi == bins.size()) bins.emplace_back();
					Bin &bn = bins[bi];
					const unsigned amax2 = std::max(bn.amax, a[v]), amin2 = std::min(bn.amin, a[v]), qmax2 = std::max(bn.qmax, qr[v]);
					const unsigned xr = x_rows(qmax2, amax2 - amin2);
					const size_t lds = lds_of(bn.rows + qr[v], xr);
					if (xr > xr_max || lds > lds_max || bn.vs.size() >= FO_VMAX) {
						if (bn.vs.empty()) { ok = false; bins.pop_back(); break; } // does not even fit alone
						continue;
					}
					bn.vs.push_back(v); bn.rows += qr[v]; bn.amax = amax2; bn.amin = amin2; bn.qmax = qmax2; bn.xr = xr;
					placed = true;
				}
			}
			std::vector<OctFwd> items;
			for (const Bin &bn : bins) {
				if (!ok) break;
				OctFwd o;
				memset(&o, 0, sizeof o);
				o.D = D; o.Ns = Ns; o.MC = (D + 63) / 64; o.nob = (Ns + 63) / 64;
				o.nv = (unsigned)bn.vs.size(); o.amax = bn.amax; o.trows = bn.rows; o.XR = bn.xr;
				unsigned rows = 0;
				for (unsigned n = 0; n < o.nv; n++) {
					const unsigned v = bn.vs[n];
					o.sc[n] = s + v; o.QR[n] = qr[v]; o.trow[n] = rows; o.a[n] = a[v]; o.b[n] = b[v];
					o.L[n] = p->sc[s + v].L; o.tap_off[n] = p->sc[s + v].tap_off;
					rows += qr[v];
				}
				items.push_back(o);
			}
			if (ok) {
				for (OctFwd &o : items) {
					o.wg_off = wg; wg += o.MC * o.nob;
					p->oct_lds = std::max(p->oct_lds, lds_of(o.trows, o.XR));
					tab.push_back(o);
				}
				for (unsigned v = s; v < e; v++) is_oct[v] = 1;
			}
		}
		s = e;
	}
	// chunk-major copies: decimations whose rows are more than 4 KB apart for the input type (and divide N: the circular
	// wrap is then a wrap of the row index)
	for (int ti = 0; ti < 2; ti++) {
		p->cm_n[ti] = 0; p->cm_per_trace[ti] = 0;
		const size_t esz = ti ? sizeof(double) : sizeof(float);
		static int cm_on = -1;
		if (cm_on < 0) { const char *e = getenv("TSPWS_FWD_CM"); cm_on = (e && *e == '1') ? 1 : 0; } // opt-in: measured no faster than the strided windows
		for (OctFwd &o : tab) {
			o.cm_slot[ti] = ~0u; o.cm_pre[ti] = 0;
			if (!cm_on || (size_t)o.D * esz <= 4096 || p->N % o.D != 0) continue;
			unsigned slot = ~0u;
			size_t pre = 0;
			for (unsigned k = 0; k < p->cm_n[ti]; k++) {
				if (p->cm_D[ti][k] == o.D) { slot = k; break; }
				pre += (size_t)p->cm_MC[ti][k] * (p->N / p->cm_D[ti][k]) * 64;
			}
			if (slot == ~0u) {
				if (p->cm_n[ti] >= FO_CMMAX) continue;
				slot = p->cm_n[ti]++;
				p->cm_D[ti][slot] = o.D; p->cm_MC[ti][slot] = o.MC;
				p->cm_per_trace[ti] += (size_t)o.MC * (p->N / o.D) * 64;
			}
			o.cm_slot[ti] = slot; o.cm_pre[ti] = pre;
		}
	}
	p->oct_wgs = wg; p->oct_n = (unsigned)tab.size();
	p->ofw_host.resize(tab.size() * sizeof(OctFwd));
	if (!tab.empty()) memcpy(p->ofw_host.data(), tab.data(), p->ofw_host.size());
	if (getenv("TSPWS_DEBUG"))
		for (const OctFwd &o : tab) fprintf(stderr, "oct sc0=%u nv=%u D=%u Ns=%u MC=%u nob=%u rows=%u XR=%u amax=%u wg_off=%u\n", o.sc[0], o.nv, o.D, o.Ns, o.MC, o.nob, o.trows, o.XR, o.amax, o.wg_off);
	return 0;
}

// the partial-block offsets are known only after the work decomposition: complete the table and upload it
static int upload_oct_forward(tspws_hip_plan *p)
{
	if (!p->oct_n) return 0;
	OctFwd *tab = (OctFwd *)p->ofw_host.data();
	for (unsigned i = 0; i < p->oct_n; i++)
		for (unsigned v = 0; v < tab[i].nv; v++) tab[i].part_off[v] = p->sc[tab[i].sc[v]].part_off;
	HIP_TRY(hipMalloc(&p->d_ofw, p->ofw_host.size()));
	HIP_TRY(hipMemcpy(p->d_ofw, tab, p->ofw_host.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_oct<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FO_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_oct<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FO_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_lds_oct<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FO_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_lds_oct<double, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FO_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_lds_oct<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FO_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_lds_oct<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FO_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_oct2<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS_MAX));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_oct2<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FT_LDS_MAX));
	return 0;
}

#include "fwd_tl.h"

static size_t tl_min_traces()
{
	static long v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_TL_MIN"); v = e ? std::max(1, atoi(e)) : 64; }
	return (size_t)v;
}

static bool tl_enabled()
{
	static int v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_FWD_TL"); v = (e && *e == '0') ? 0 : 1; }
	return v == 1;
}

// Decomposition for many-trace batches (fwd_tl.h): octaves (runs of scales with the same D and Ns) with at least TL_MINNS
// outputs become trace-lane work items (voice subsets of <= TL_VMAX voices), the rest stays on the direct kernel; sc_tl is
// the scale table of that decomposition (partial layout, fused flags, accumulate geometry).
static int build_tl_forward(tspws_hip_plan *p, unsigned FWD_STEPS)
{
	unsigned MINNS = 17, TLSTEPS = 96; // >= 3 of the 4 waves of an output block busy; ~96 residue steps per workgroup (sweeps on 1024 x 32768 and 499 x 16501)
	if (const char *e = getenv("TSPWS_TL_MINNS")) MINNS = (unsigned)std::max(1, atoi(e));
	if (const char *e = getenv("TSPWS_TL_STEPS")) TLSTEPS = (unsigned)std::max(1, atoi(e));
	p->sc_tl = p->sc;
	std::vector<TLItem> items;
	std::vector<char> is_tl(p->S, 0);
	unsigned wg = 0;
	for (unsigned s = 0; s < p->S;) {
		unsigned e = s + 1;
		while (e < p->S && p->sc[e].D == p->sc[s].D && p->sc[e].Ns == p->sc[s].Ns) e++;
		const unsigned D = p->sc[s].D, Ns = p->sc[s].Ns, nv = e - s;
		bool ok = Ns >= MINNS;
		std::vector<unsigned> a(nv), b(nv), qr(nv);
		for (unsigned v = 0; v < nv && ok; v++) {
			const ScaleDesc &d = p->sc[s + v];
			a[v] = (unsigned)d.c / D; b[v] = (unsigned)d.c % D;
			const long long num = (long long)d.L - 1 - (long long)b[v];
			const long long fl = num >= 0 ? num / (long long)D : -1;
			qr[v] = ((unsigned)(fl + 2) + 3) & ~3u;
			if (qr[v] > TL_QMAX) ok = false;
		}
		if (ok) {
			const unsigned nsub = (nv + TL_VMAX - 1) / TL_VMAX, per = (nv + nsub - 1) / nsub;
			std::vector<TLItem> sub;
			for (unsigned v0 = 0; v0 < nv && ok; v0 += per) {
				const unsigned n = std::min(per, nv - v0);
				TLItem o;
				memset(&o, 0, sizeof o);
				o.nv = n; o.D = D; o.Ns = Ns; o.nkb = (Ns + 31) / 32;
				o.nsplit = (D + TL_PMAX - 1) / TL_PMAX; o.pps = (D + o.nsplit - 1) / o.nsplit; o.fused = o.nsplit == 1;
				unsigned amax = 0, amin = ~0u, qmax = 0, rows = 0;
				for (unsigned i = 0; i < n; i++) {
					const unsigned v = v0 + i;
					o.sc[i] = s + v; o.QR[i] = qr[v]; o.trow[i] = rows; o.a[i] = a[v]; o.b[i] = b[v]; o.L[i] = p->sc[s + v].L;
					o.tap_off[i] = p->sc[s + v].tap_off; o.coef_off[i] = p->sc[s + v].coef_off;
					rows += qr[v]; amax = std::max(amax, a[v]); amin = std::min(amin, a[v]); qmax = std::max(qmax, qr[v]);
				}
				o.amax = amax; o.trows = rows;
				o.XR = (31 + qmax + (amax - amin) + 3) & ~3u;
				if (o.XR > TL_XRMAX || rows > TL_NT) ok = false;
				sub.push_back(o);
			}
			if (ok) {
				for (TLItem &o : sub) {
					// a workgroup should walk >= ~64 residue steps: with few residues per output block it takes several blocks
					o.kbw = std::max(1u, std::min(o.nkb, TLSTEPS / std::max(1u, o.pps)));
					o.wg_off = wg; wg += ((o.nkb + o.kbw - 1) / o.kbw) * o.nsplit;
					p->tl_lds = std::max(p->tl_lds, 2 * ((size_t)o.XR * 64 * sizeof(double) + (size_t)o.trows * sizeof(double2)));
					items.push_back(o);
				}
				for (unsigned v = s; v < e; v++) is_tl[v] = 1;
			}
		}
		s = e;
	}
	// scale table of the decomposition: partial layout, direct-kernel waves, accumulate geometry
	unsigned woff = 0, ablk = 0;
	unsigned long long poff = 0;
	for (unsigned s = 0; s < p->S; s++) {
		ScaleDesc &d = p->sc_tl[s];
		d.use_lds = 0; d.use_oct = 0; d.lds_off = 0;
		if (is_tl[s]) {
			d.nsplit = (d.D + TL_PMAX - 1) / TL_PMAX; d.cps = 1;
			d.fuse_ok = d.nsplit == 1 ? 1u : 0u;
		} else { // direct kernel, as in the few-trace table
			const unsigned cps = std::max(1u, (FWD_STEPS + d.Q / 2) / std::max(1u, d.Q));
			d.cps = std::min(cps, d.MC);
			d.nsplit = (d.MC + d.cps - 1) / d.cps;
			d.fuse_ok = 0;
		}
		d.wave_off = woff;
		if (!is_tl[s]) woff += d.ngw * d.nsplit;
		d.part_off = poff;
		if (!(is_tl[s] && d.fuse_ok)) poff += (unsigned long long)d.nsplit * d.Ns; // fused scales never write partials
		d.acc2_off = ablk;
		ablk += d.nsplit > 1 ? (d.Ns + 3) / 4 : (d.Ns + 255) / 256; // split scales: 4 coefficients per block (k_accumulate_parts, many)
	}
	for (TLItem &o : items) for (unsigned i = 0; i < o.nv; i++) o.part_off[i] = p->sc_tl[o.sc[i]].part_off;
	p->tl_n = (unsigned)items.size(); p->tl_wgs = wg; p->tl_waves = woff; p->tl_acc2_blocks = ablk; p->tl_npart = poff;
	if (!p->tl_n) return 0;
	HIP_TRY(hipMalloc(&p->d_sc_tl, p->S * sizeof(ScaleDesc)));
	HIP_TRY(hipMemcpy(p->d_sc_tl, p->sc_tl.data(), p->S * sizeof(ScaleDesc), hipMemcpyHostToDevice));
	HIP_TRY(hipMalloc(&p->d_tl, items.size() * sizeof(TLItem)));
	HIP_TRY(hipMemcpy(p->d_tl, items.data(), items.size() * sizeof(TLItem), hipMemcpyHostToDevice));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_tl<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_tl<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	if (getenv("TSPWS_DEBUG"))
		for (const TLItem &o : items) fprintf(stderr, "tl sc0=%u nv=%u D=%u Ns=%u nkb=%u nsplit=%u pps=%u XR=%u rows=%u fused=%u wg_off=%u\n", o.sc[0], o.nv, o.D, o.Ns, o.nkb, o.nsplit, o.pps, o.XR, o.trows, o.fused, o.wg_off);
	return 0;
}

static int upload_mfma_tables(tspws_hip_plan *p)
{
	const size_t np = p->pairs.size();
	unsigned long long nbt = 0;
	for (const FwdGroup &d : p->pairs) nbt = std::max(nbt, d.bt_off[0] + (unsigned long long)d.MC * d.bper);
	HIP_TRY(hipMalloc(&p->d_pairs, np * sizeof(FwdGroup)));
	HIP_TRY(hipMalloc(&p->d_bt, std::max<unsigned long long>(nbt, 1) * sizeof(double)));
	HIP_TRY(hipMemcpy(p->d_pairs, p->pairs.data(), np * sizeof(FwdGroup), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(p->d_sc, p->sc.data(), p->S * sizeof(ScaleDesc), hipMemcpyHostToDevice)); // nsplit / part_off changed
	for (const FwdGroup &d : p->pairs) {
		unsigned v = d.s0;
		for (unsigned pi = 0; pi < d.NP; pi++) {
			const ScaleDesc &a = p->sc[v], &b = p->sc[v + (d.nv[pi] > 1 ? 1 : 0)];
			const unsigned long long n = (unsigned long long)d.MC * d.Mc * d.Kq[pi] * 16;
			const long long cpp = d.cp - (long long)d.rofs[pi] * d.D;
			hipLaunchKernelGGL(k_build_bt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p->d_bt + d.bt_off[0], (const double2 *)p->d_w, n, d.D,
			                   d.Kq[pi], cpp, d.nv[pi], a.tap_off, a.L, a.c, b.tap_off, b.L, b.c, d.Mc, d.bper, pi ? d.Mc * d.Kq[0] * 16 : 0u);
			v += d.nv[pi];
		}
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}

template <typename TIn>
static int forward_parts(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double2 *d_part, hipStream_t st, FuseOut *fz = nullptr)
{
	if (fz) fz->applied = false;
	if (p->fwd_kind == 3) {
		size_t per_launch = 1u << 20; // keeps the unit counters in 32 bits
		for (const FwdGroup &d : p->pairs) per_launch = std::min<size_t>(per_launch, std::max<size_t>(1, 0x7fffffffu / std::max(1u, d.nob)));
		for (size_t t0 = 0; t0 < ntr; t0 += per_launch) {
			const unsigned nt = (unsigned)std::min<size_t>(ntr - t0, per_launch);
			const unsigned ng = (unsigned)p->pairs.size();
			static int only = -2, spec_on = -1; // debug knobs: time a single group (results are then incomplete); generic kernel only
			if (only == -2) { const char *e = getenv("TSPWS_MFMA_ONLY_GROUP"); only = e ? atoi(e) : -1; }
			if (spec_on < 0) { const char *e = getenv("TSPWS_MFMA_SPEC"); spec_on = (e && *e == '0') ? 0 : 1; }
			auto items_of = [&](const FwdGroup &d) { return (unsigned long long)d.nsplit * (((unsigned long long)nt * d.nob + d.upi - 1) / d.upi); };
			auto spec_ok = [&](const FwdGroup &d) {
				const unsigned rpi = 64 / d.Mc, kqm = std::max(d.Kq[0], d.NP > 1 ? d.Kq[1] : 0u);
				const bool fits = (d.RT + rpi - 1) / rpi <= (4 * d.TQ + 4 * kqm + 6 + rpi - 1) / rpi; // the kernel's staging registers
				return spec_on && fits && d.D % d.Mc == 0 && fwd_mfma_spec_has(d.TQ, d.Mc, d.Kq[0], d.NP > 1 ? d.Kq[1] : 0);
			};
			std::vector<char> done(ng, 0);
			for (unsigned g0 = 0; g0 <= ng; g0++) { // one launch per (Kq0, Kq1) of the specialised groups, then one for the rest
				const bool rest = g0 == ng;
				if (!rest && (done[g0] || !spec_ok(p->pairs[g0]))) continue;
				const unsigned k0 = rest ? 0 : p->pairs[g0].Kq[0], k1 = rest ? 0 : (p->pairs[g0].NP > 1 ? p->pairs[g0].Kq[1] : 0);
				const unsigned tq0 = rest ? 0 : p->pairs[g0].TQ, mc0 = rest ? 0 : p->pairs[g0].Mc;
				FwdOffsets offs;
				unsigned long long items = 0;
				for (unsigned g = 0; g < ng; g++) {
					const FwdGroup &d = p->pairs[g];
					offs.off[g] = (unsigned)items;
					if (only >= 0 && (unsigned)only != g) continue;
					if (done[g]) continue;
					if (!rest && !(spec_ok(d) && d.TQ == tq0 && d.Mc == mc0 && d.Kq[0] == k0 && (d.NP > 1 ? d.Kq[1] : 0) == k1)) continue;
					items += items_of(d);
					done[g] = 1;
				}
				offs.off[ng] = (unsigned)items;
				if (items >= (1ull << 31)) return fail(TSPWS_E_ARG, "forward: too many work items in one launch");
				if (!items) continue;
				if (rest)
					hipLaunchKernelGGL((k_fwd_mfma<TIn>), dim3((unsigned)items), dim3(256), p->mfma_lds, st, d_x + t0 * ld, ld, nt, p->N, p->d_pairs, ng, offs,
					                   p->d_bt, d_part + t0 * p->npart, p->npart);
				else
					fwd_mfma_spec_launch(sizeof(TIn) == 4, tq0, mc0, k0, k1, (unsigned)items, p->mfma_lds, st, d_x + t0 * ld, ld, nt, p->N, p->d_pairs, ng, offs, p->d_bt,
					                     d_part + t0 * p->npart, p->npart);
			}
		}
		HIP_TRY(hipGetLastError());
		return 0;
	}
	// Up to three independent kernels transform disjoint sets of scales: the octave-fused kernel (D >= 64), the LDS kernel
	// (D < 64) and the direct kernel (coarse scales, latency-bound).  They run side by side: two side streams are forked
	// from and joined back into the caller's stream.
	// merged: the octave-fused (mode 1) and the LDS workgroups share ONE launch (k_fwd_lds_oct); TSPWS_OCT_MERGE=0: own launches
	static int merge_env = -1;
	if (merge_env < 0) { const char *e = getenv("TSPWS_OCT_MERGE"); merge_env = (e && *e == '0') ? 0 : 1; }
	const bool merged = merge_env && p->oct_wgs && p->lds_blocks && p->oct_mode == 1 && ntr <= 65535;
	const bool has_lds = p->lds_end() > p->lds_begin(), has_poly = p->wav_end() > p->wav_begin(); // (this call's scale range)
	const int nk = (has_lds ? 1 : 0) + (has_poly ? 1 : 0) + ((p->oct_wgs && !merged) ? 1 : 0);
	const bool both = nk > 1 && side_stream_enabled();
	hipStream_t sp = st, so = st; // streams of the direct and of the octave kernel
	if (both) {
		const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence; // device-local ordering only
		if (!p->side) HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
		if (!p->side2) HIP_TRY(hipStreamCreateWithFlags(&p->side2, hipStreamNonBlocking));
		if (!p->ev_fork) HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, evf));
		if (!p->ev_join) HIP_TRY(hipEventCreateWithFlags(&p->ev_join, evf));
		if (!p->ev_join2) HIP_TRY(hipEventCreateWithFlags(&p->ev_join2, evf));
		HIP_TRY(hipEventRecord(p->ev_fork, st));
		if (has_poly && nk > 1) { HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0)); sp = p->side; }
		if (p->oct_wgs && p->lds_blocks && !merged) { HIP_TRY(hipStreamWaitEvent(p->side2, p->ev_fork, 0)); so = p->side2; }
	}
	constexpr int TIX = sizeof(TIn) == 8 ? 1 : 0;
	TIn *xcm_m = nullptr; // merged launch: chunk-major copies (if enabled) are made on the caller's stream first
	if (merged && p->cm_n[TIX]) {
		void *vx;
		int rcx = scratch(p, SCR_XCM, ntr * p->cm_per_trace[TIX] * sizeof(TIn), &vx);
		if (rcx) return rcx;
		xcm_m = (TIn *)vx;
		ChunkMajor cmj;
		memset(&cmj, 0, sizeof cmj);
		cmj.n = p->cm_n[TIX];
		size_t big = 0;
		for (unsigned k = 0; k < cmj.n; k++) {
			cmj.D[k] = p->cm_D[TIX][k]; cmj.MC[k] = p->cm_MC[TIX][k];
			big = std::max(big, (size_t)cmj.MC[k] * (p->N / cmj.D[k]) * 64);
		}
		hipLaunchKernelGGL((k_chunk_major<TIn>), dim3((unsigned)((big + 255) / 256), (unsigned)ntr, cmj.n), dim3(256), 0, st, d_x, ld, (unsigned)ntr, p->N, cmj, xcm_m);
	}
	if (p->oct_wgs && !merged) {
		unsigned tps = (unsigned)std::min<size_t>(ntr, 32);
		// mode 2: one workgroup per CU, so a single slice is best once half the CUs have work; mode 1: two per CU
		const size_t want = p->oct_mode == 2 ? 128 : 256;
		while (tps > 1 && (size_t)p->oct_wgs * ((ntr + tps - 1) / tps) < want) tps = (tps + 1) / 2;
		if (const char *e = getenv("TSPWS_OCT_TPS")) tps = (unsigned)std::max(1, atoi(e));
		const size_t per_launch = std::min<size_t>((size_t)tps * 65535, 65535);
		constexpr int TI = sizeof(TIn) == 8 ? 1 : 0;
		TIn *xcm = nullptr;
		if (p->cm_n[TI]) { // chunk-major copies of the batch for the far-strided decimations
			void *vx;
			int rcx = scratch(p, SCR_XCM, std::min(ntr, per_launch) * p->cm_per_trace[TI] * sizeof(TIn), &vx);
			if (rcx) return rcx;
			xcm = (TIn *)vx;
		}
		for (size_t t0 = 0; t0 < ntr; t0 += per_launch) {
			const unsigned nt = (unsigned)std::min<size_t>(ntr - t0, per_launch);
			if (xcm) {
				ChunkMajor cmj;
				memset(&cmj, 0, sizeof cmj);
				cmj.n = p->cm_n[TI];
				size_t big = 0;
				for (unsigned k = 0; k < cmj.n; k++) {
					cmj.D[k] = p->cm_D[TI][k]; cmj.MC[k] = p->cm_MC[TI][k];
					big = std::max(big, (size_t)cmj.MC[k] * (p->N / cmj.D[k]) * 64);
				}
				hipLaunchKernelGGL((k_chunk_major<TIn>), dim3((unsigned)((big + 255) / 256), nt, cmj.n), dim3(256), 0, so, d_x + t0 * ld, ld, nt, p->N, cmj, xcm);
			}
			if (p->oct_mode == 2)
				hipLaunchKernelGGL((k_fwd_oct2<TIn>), dim3(p->oct_wgs, (nt + tps - 1) / tps), dim3(FT_NT), p->oct_lds, so, d_x + t0 * ld, ld, nt, tps, p->N,
				                   p->d_ofw, p->oct_n, p->d_w, d_part + t0 * p->npart, p->npart, (const TIn *)xcm, nt);
			else
				hipLaunchKernelGGL((k_fwd_oct<TIn>), dim3(p->oct_wgs, (nt + tps - 1) / tps), dim3(FO_NT), p->oct_lds, so, d_x + t0 * ld, ld, nt, tps, p->N,
				                   p->d_ofw, p->oct_n, p->d_w, d_part + t0 * p->npart, p->npart, (const TIn *)xcm, nt);
		}
	}
	// the direct kernel first: its few hundred long, latency-bound workgroups (no LDS, 88 VGPRs) get their slots and the
	// LDS kernel's workgroups fill in beside them
	const unsigned w0 = p->wav_begin(), w1 = p->wav_end(), b0 = p->lds_begin(), b1 = p->lds_end(); // this call's share of the launch lists
	if (w1 > w0) {
		const unsigned nb = (w1 - w0 + 3) / 4;
		if (ntr == 1) {
			hipLaunchKernelGGL((k_fwd_poly<TIn, 1>), dim3(nb, 1), dim3(256), 0, sp, d_x, ld, 1u, p->N, p->d_sc, p->S, p->d_w, d_part,
			                   p->npart, w1, w0);
		} else {
			for (size_t t0 = 0; t0 < ntr; t0 += 2 * 32768) {
				const unsigned nt = (unsigned)std::min<size_t>(ntr - t0, 2 * 32768);
				hipLaunchKernelGGL((k_fwd_poly<TIn, 2>), dim3(nb, (nt + 1) / 2), dim3(256), 0, sp, d_x + t0 * ld, ld, nt, p->N, p->d_sc,
				                   p->S, p->d_w, d_part + t0 * p->npart, p->npart, w1, w0);
			}
		}
	}
	if (b1 > b0) {
		const bool fuse = fz && fz->accST && p->n_fusable;
		// traces per workgroup: enough slices to fill the GPU (>= ~2048 workgroups), at most 32 traces per slice
		unsigned tps = (unsigned)std::min<size_t>(ntr, 32);
		while (tps > 1 && (size_t)p->lds_blocks * ((ntr + tps - 1) / tps) < 2048) tps = (tps + 1) / 2;
		if (const char *e = getenv("TSPWS_FWD_TPS")) tps = (unsigned)std::max(1, atoi(e));
		if (fuse) tps = fz->tps; // the caller sized the slice planes
		static int rev_env = -1;
		if (rev_env < 0) { const char *e = getenv("TSPWS_FWD_REV"); rev_env = (e && *e == '0') ? 0 : 1; }
		const unsigned rev = (unsigned)rev_env;
		const size_t per_launch = (size_t)tps * 65535;
		if (merged) {
			// octave workgroups: slices of the traces so that a workgroup lasts about as long as an LDS-kernel workgroup
			static int osl = -1;
			if (osl < 0) { const char *e = getenv("TSPWS_OCT_SLICES"); osl = e ? std::max(1, atoi(e)) : 2; }
			const unsigned nsl = (unsigned)std::min<size_t>(ntr, (size_t)osl), otps = (unsigned)((ntr + nsl - 1) / nsl);
			const unsigned oslices = (unsigned)((ntr + otps - 1) / otps);
			const size_t lds = std::max<size_t>(FL_LDS_BYTES, p->oct_lds);
			const dim3 grid(p->oct_wgs * oslices + p->lds_blocks, (unsigned)((ntr + tps - 1) / tps));
			if (fuse)
				hipLaunchKernelGGL((k_fwd_lds_oct<TIn, true>), grid, dim3(FL_NT), lds, st, d_x, ld, (unsigned)ntr, tps, p->N, p->d_sc, p->S, p->d_w, d_part, p->npart,
				                   fz->accST, fz->accPS, fz->stride, p->lds_blocks, p->d_ofw, p->oct_n, p->oct_wgs, oslices, otps, (const TIn *)xcm_m);
			else
				hipLaunchKernelGGL((k_fwd_lds_oct<TIn, false>), grid, dim3(FL_NT), lds, st, d_x, ld, (unsigned)ntr, tps, p->N, p->d_sc, p->S, p->d_w, d_part, p->npart,
				                   (double2 *)nullptr, (double2 *)nullptr, (size_t)0, p->lds_blocks, p->d_ofw, p->oct_n, p->oct_wgs, oslices, otps, (const TIn *)xcm_m);
		} else
		for (size_t t0 = 0; t0 < ntr; t0 += per_launch) {
			const unsigned nt = (unsigned)std::min<size_t>(ntr - t0, per_launch);
			if (fuse)
				hipLaunchKernelGGL((k_fwd_lds<TIn, true>), dim3(b1 - b0, (nt + tps - 1) / tps), dim3(FL_NT), FL_LDS_BYTES, st, d_x + t0 * ld, ld,
				                   nt, tps, p->N, p->d_sc, p->S, p->d_w, d_part + t0 * p->npart, p->npart, fz->accST + (t0 / tps) * fz->stride,
				                   fz->accPS + (t0 / tps) * fz->stride, fz->stride, rev, b0);
			else
				hipLaunchKernelGGL((k_fwd_lds<TIn, false>), dim3(b1 - b0, (nt + tps - 1) / tps), dim3(FL_NT), FL_LDS_BYTES, st, d_x + t0 * ld, ld,
				                   nt, tps, p->N, p->d_sc, p->S, p->d_w, d_part + t0 * p->npart, p->npart, (double2 *)nullptr, (double2 *)nullptr, (size_t)0, rev, b0);
		}
		if (fuse) fz->applied = true;
	}
	if (sp != st) {
		HIP_TRY(hipEventRecord(p->ev_join, sp));
		HIP_TRY(hipStreamWaitEvent(st, p->ev_join, 0));
	}
	if (so != st) {
		HIP_TRY(hipEventRecord(p->ev_join2, so));
		HIP_TRY(hipStreamWaitEvent(st, p->ev_join2, 0));
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

#if FL_TIMING
// debug build only: device buffer of the per-phase wave timers of k_fwd_lds
extern "C" int tspws_hip_fwd_timing(unsigned long long *h_out, int reset)
{
	static unsigned long long *d_buf = nullptr;
	if (!d_buf) {
		HIP_TRY(hipMalloc(&d_buf, 56 * sizeof(unsigned long long)));
		HIP_TRY(hipMemset(d_buf, 0, 56 * sizeof(unsigned long long)));
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(fl_timing_out), &d_buf, sizeof d_buf));
	}
	HIP_TRY(hipDeviceSynchronize());
	if (h_out) HIP_TRY(hipMemcpy(h_out, d_buf, 56 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	if (reset) HIP_TRY(hipMemset(d_buf, 0, 56 * sizeof(unsigned long long)));
	return 0;
}
#endif

// scratch for the split partials of one batch of transformed traces (bigger batches = fewer, fuller launches)
static size_t part_budget_bytes()
{
	static size_t v = 0;
	if (!v) { const char *e = getenv("TSPWS_PART_MB"); v = (size_t)(e ? std::max(16, atoi(e)) : 2048) << 20; }
	return v;
}

static bool use_generic_forward()
{
	static int v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_FWD_GENERIC"); v = (e && *e == '1') ? 1 : 0; }
	return v == 1;
}

template <typename TIn>
static int forward_impl(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_Y, hipStream_t st)
{
	if (!p || !d_x || !d_Y) return fail(TSPWS_E_ARG, "forward: NULL");
	if (!ntr) return 0;
	HIP_TRY(hipSetDevice(p->device));
	if (use_generic_forward()) return forward_generic<TIn>(p, d_x, ntr, ld, d_Y, st);
	const size_t batch = std::min<size_t>(ntr, std::max<size_t>(2, ((part_budget_bytes()) / (p->npart * sizeof(double2))) & ~(size_t)1));
	void *v;
	int rc = scratch(p, SCR_PART, batch * p->npart * sizeof(double2), &v);
	if (rc) return rc;
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const size_t nb = std::min(batch, ntr - t0);
		if ((rc = forward_parts<TIn>(p, d_x + t0 * ld, nb, ld, (double2 *)v, st))) return rc;
		static int nogather = -1; // timing knob for tools/mfma_groups.py: forward kernels only (Y is then not written)
		if (nogather < 0) nogather = getenv("TSPWS_DEBUG_NOGATHER") ? 1 : 0;
		if (!nogather)
		hipLaunchKernelGGL(k_gather_parts, dim3((unsigned)((p->ncoef + 255) / 256), (unsigned)nb), dim3(256), 0, st, (const double2 *)v,
		                   p->npart, p->d_sc, p->S, (double2 *)d_Y + t0 * p->ncoef, p->ncoef);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

extern "C" int tspws_hip_forward_f64(tspws_hip_plan *p, const double *d_x, size_t ntr, size_t ld, double *d_Y, void *s)
{
	return forward_impl<double>(p, d_x, ntr, ld, d_Y, S_(s));
}
extern "C" int tspws_hip_forward_f32(tspws_hip_plan *p, const float *d_x, size_t ntr, size_t ld, double *d_Y, void *s)
{
	return forward_impl<float>(p, d_x, ntr, ld, d_Y, S_(s));
}

// ------------------------------------------------------------------------------------------
// stack accumulation: ST += Y, PS += Y/|Y| unless the quotient is not a unit phasor
// (ts_pws1f_lib.c:489-492).  One thread per coefficient, traces walked in order.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_accumulate(const double2 *__restrict__ Y, size_t ncoef, unsigned ntr, double2 *__restrict__ ST,
                                                    double2 *__restrict__ PS, int zero_first)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= ncoef) return;
	double2 st = zero_first ? make_double2(0, 0) : ST[i];
	double2 ps = zero_first ? make_double2(0, 0) : PS[i];
	for (unsigned b = 0; b < ntr; b++) {
		const double2 v = Y[(size_t)b * ncoef + i];
		st.x += v.x; st.y += v.y;
		add_unit_phasor(ps, v);
	}
	ST[i] = st; PS[i] = ps;
}

extern "C" int tspws_hip_accumulate(tspws_hip_plan *p, const double *d_Y, size_t ntr, double *d_ST, double *d_PS, int zero_first, void *s)
{
	if (!p || !d_ST || !d_PS || (ntr && !d_Y)) return fail(TSPWS_E_ARG, "accumulate: NULL");
	HIP_TRY(hipSetDevice(p->device));
	hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((p->ncoef + 255) / 256)), dim3(256), 0, S_(s), (const double2 *)d_Y, p->ncoef,
	                   (unsigned)ntr, (double2 *)d_ST, (double2 *)d_PS, zero_first);
	HIP_TRY(hipGetLastError());
	return 0;
}

// launches k_accumulate_parts for `nb` transformed traces; fz = what forward_parts left behind (may be NULL / not applied)
static void launch_accumulate(tspws_hip_plan *p, const double2 *part, unsigned nb, double2 *ST, double2 *PS, int zero_first, const FuseOut *fz,
                              unsigned nslices, hipStream_t st, unsigned nbatch = 1, size_t y_part = 0, size_t y_stack = 0, bool tl = false,
                              const WeightArgs *wa = nullptr)
{
	WeightArgs w0; w0.OUT = nullptr; w0.mode = 0; w0.K = w0.M = w0.wu = 0;
	const bool on = fz && fz->applied;
	const bool direct = on && fz->accST == ST; // the single slice went straight into ST / PS
	// tl: the many-trace decomposition's scale table (partial layout, fused flags, block geometry)
	const unsigned a0 = tl ? 0u : p->acc_begin(), a1 = tl ? p->tl_acc2_blocks : p->acc_end();
	if (a1 <= a0) return;
	hipLaunchKernelGGL(k_accumulate_parts, dim3(a1 - a0, nbatch), dim3(256), 0, st, part, tl ? p->tl_npart : p->npart,
	                   tl ? p->d_sc_tl : p->d_sc, p->S, nb, ST, PS, zero_first,
	                   on ? (direct ? 1 : 2) : 0, on ? (const double2 *)fz->accST : nullptr, on ? (const double2 *)fz->accPS : nullptr,
	                   on ? fz->stride : (size_t)0, nslices, y_part, y_stack, tl ? 1 : 0, wa ? *wa : w0, a0);
}

// slice length of the fused forward kernel for a batch of nb traces: whole batch when it is small (two-stage: the K
// partial stacks -> ONE slice that writes ST / PS directly), else 32 traces per slice
static unsigned fuse_tps(size_t nb)
{
	static int forced = -1;
	if (forced < 0) { const char *e = getenv("TSPWS_FUSE_TPS"); forced = e ? std::max(1, atoi(e)) : 0; }
	if (forced) return (unsigned)std::min<size_t>(nb, (size_t)forced);
	return (unsigned)std::min<size_t>(nb, 32);
}

// Many traces (single-stage stacks): trace-lane kernel on the transposed batch (fwd_tl.h); the stacks of the fused scales
// come back as one plane pair per 64-trace block, the split / coarse scales as per-trace partials in the tl layout.
template <typename TIn>
static int stacks_tl(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                     const WeightArgs *wa = nullptr, bool *weighted = nullptr)
{
	int rc;
	void *v;
	// traces per batch: transposed copy <= 1 GiB, at most 4096 (64 plane pairs), a multiple of 64
	size_t batch = std::min<size_t>(4096, std::max<size_t>(64, (((size_t)1 << 30) / ((size_t)p->N * sizeof(TIn))) & ~(size_t)63));
	if (p->tl_npart) batch = std::min(batch, std::max<size_t>(64, (part_budget_bytes() / (p->tl_npart * sizeof(double2))) & ~(size_t)63));
	if (const char *e = getenv("TSPWS_TL_BATCH")) batch = std::max<size_t>(64, (size_t)atoi(e) & ~(size_t)63); // tests: force several batches
	batch = std::min(batch, (ntr + 63) & ~(size_t)63);
	const size_t nblk_max = batch / 64;
	if ((rc = scratch(p, SCR_XT, (size_t)p->N * batch * sizeof(TIn), &v))) return rc;
	TIn *xT = (TIn *)v;
	if ((rc = scratch(p, SCR_FZ, nblk_max * 2 * p->ncoef * sizeof(double2), &v))) return rc;
	double2 *planes = (double2 *)v;
	double2 *part = nullptr;
	if (p->tl_npart) { if ((rc = scratch(p, SCR_PART, batch * p->tl_npart * sizeof(double2), &v))) return rc; part = (double2 *)v; }
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const unsigned nb = (unsigned)std::min(batch, ntr - t0), nblk = (nb + 63) / 64, TP = nblk * 64;
		const TIn *xb = d_x + t0 * ld;
		hipLaunchKernelGGL((k_transpose_traces<TIn>), dim3((p->N + 63) / 64, nblk), dim3(256), 0, st, xb, ld, nb, p->N, TP, xT);
		hipLaunchKernelGGL((k_fwd_tl<TIn>), dim3(p->tl_wgs, nblk), dim3(TL_NT), p->tl_lds, st, (const TIn *)xT, TP, nb, p->N, p->d_tl, p->tl_n, p->d_w,
		                   planes, planes + p->ncoef, 2 * p->ncoef, part, p->tl_npart);
		if (p->tl_waves) { // scales with too few outputs for the trace-lane kernel: direct kernel, tl partial layout
			const unsigned nbw = (p->tl_waves + 3) / 4;
			for (size_t u0 = 0; u0 < nb; u0 += 2 * 32768) {
				const unsigned nt = (unsigned)std::min<size_t>(nb - u0, 2 * 32768);
				hipLaunchKernelGGL((k_fwd_poly<TIn, 2>), dim3(nbw, (nt + 1) / 2), dim3(256), 0, st, xb + u0 * ld, ld, nt, p->N, p->d_sc_tl, p->S, p->d_w,
				                   part + u0 * p->tl_npart, p->tl_npart, p->tl_waves);
			}
		}
		FuseOut fz;
		fz.accST = planes; fz.accPS = planes + p->ncoef; fz.stride = 2 * p->ncoef; fz.tps = 64; fz.applied = true;
		const bool last = t0 + batch >= ntr; // the launch that completes the stacks also weights them (wa)
		launch_accumulate(p, part, nb, (double2 *)d_ST, (double2 *)d_PS, (t0 == 0 && !keep) ? 1 : 0, &fz, nblk, st, 1, 0, 0, true, last && !keep ? wa : nullptr);
		if (last && !keep && wa && wa->OUT && weighted) *weighted = true;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

template <typename TIn>
static int stacks_impl(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, void *s, bool keep = false,
                       const WeightArgs *wa = nullptr, bool *weighted = nullptr)
{ // keep: add to the stacks already in d_ST / d_PS instead of starting from zero
  // wa: weighting to apply where the stacks are completed (same kernel launch); *weighted tells whether that happened
	HIP_TRY(hipSetDevice(p->device));
	hipStream_t st = S_(s);
	if (ntr >= tl_min_traces() && tl_enabled() && fuse_enabled() && p->fwd_kind == 1 && p->tl_n && !use_generic_forward())
		return stacks_tl<TIn>(p, d_x, ntr, ld, d_ST, d_PS, st, keep, wa, weighted);
	if (!ntr) { if (!keep) { HIP_TRY(hipMemsetAsync(d_ST, 0, p->ncoef * 16, st)); HIP_TRY(hipMemsetAsync(d_PS, 0, p->ncoef * 16, st)); } return 0; }
	int rc;
	if (use_generic_forward()) {
		size_t batch = std::max<size_t>(1, ((size_t)256 << 20) / (p->ncoef * sizeof(double2)));
		batch = std::min(batch, ntr);
		void *d_Y = nullptr;
		if ((rc = scratch(p, SCR_Y, batch * p->ncoef * sizeof(double2), &d_Y))) return rc;
		for (size_t t0 = 0; t0 < ntr; t0 += batch) {
			const size_t nb = std::min(batch, ntr - t0);
			if ((rc = forward_generic<TIn>(p, d_x + t0 * ld, nb, ld, (double *)d_Y, st))) return rc;
			if ((rc = tspws_hip_accumulate(p, (const double *)d_Y, nb, d_ST, d_PS, t0 == 0 && !keep, s))) return rc;
		}
		return 0;
	}
	// trace batch sized to keep the partial-coefficient scratch around 256 MiB (even, for the 2-trace tiles)
	const size_t batch = std::min<size_t>(ntr, std::max<size_t>(2, ((part_budget_bytes()) / (p->npart * sizeof(double2))) & ~(size_t)1));
	void *v;
	if ((rc = scratch(p, SCR_PART, batch * p->npart * sizeof(double2), &v))) return rc;
	const bool fuse = fuse_enabled() && p->fwd_kind == 1 && p->n_fusable;
	void *vz = nullptr;
	if (fuse) { // slice planes of the largest batch (unused when the only slice writes ST / PS directly)
		const unsigned tps = fuse_tps(batch);
		const size_t nsl = (batch + tps - 1) / tps;
		if (!(nsl == 1 && !keep && batch >= ntr) && (rc = scratch(p, SCR_FZ, nsl * 2 * p->ncoef * sizeof(double2), &vz))) return rc;
	}
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const size_t nb = std::min(batch, ntr - t0);
		const int zero_first = (t0 == 0 && !keep) ? 1 : 0;
		FuseOut fz;
		unsigned nsl = 0;
		if (fuse) {
			fz.tps = fuse_tps(nb);
			nsl = (unsigned)((nb + fz.tps - 1) / fz.tps);
			if (nsl == 1 && zero_first) { fz.accST = (double2 *)d_ST; fz.accPS = (double2 *)d_PS; fz.stride = 0; }
			else { fz.accST = (double2 *)vz; fz.accPS = (double2 *)vz + p->ncoef; fz.stride = 2 * p->ncoef; }
		}
		if ((rc = forward_parts<TIn>(p, d_x + t0 * ld, nb, ld, (double2 *)v, st, fuse ? &fz : nullptr))) return rc;
		const bool all = nb == ntr && !keep; // one batch holds every trace: the accumulation completes the stacks
		launch_accumulate(p, (const double2 *)v, (unsigned)nb, (double2 *)d_ST, (double2 *)d_PS, zero_first, &fz, nsl, st, 1, 0, 0, false, all ? wa : nullptr);
		if (all && wa && wa->OUT && weighted) *weighted = true;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

extern "C" int tspws_hip_stacks_double(tspws_hip_plan *p, const double *d_P, unsigned K, size_t ldP, double *d_ST, double *d_PS, void *s)
{
	if (!p || !d_P || !d_ST || !d_PS) return fail(TSPWS_E_ARG, "stacks_double: NULL");
	return stacks_impl<double>(p, d_P, K, ldP, d_ST, d_PS, s);
}

extern "C" int tspws_hip_stacks_float(tspws_hip_plan *p, const float *d_x, size_t mtr, size_t ld, double *d_ST, double *d_PS, void *s)
{
	if (!p || !d_x || !d_ST || !d_PS) return fail(TSPWS_E_ARG, "stacks_float: NULL");
	return stacks_impl<float>(p, d_x, mtr, ld, d_ST, d_PS, s);
}

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
	int mode;
	if (wu == 2 && unbiased && K != 1) mode = 3;      // selection rule ts_pws1f_lib.c:226-228, K==1 falls back :972
	else if (wu == 2) mode = 0;
	else if (wu == 1) mode = 1;
	else mode = 2;
	hipLaunchKernelGGL(k_weight, dim3((unsigned)((p->ncoef + 255) / 256)), dim3(256), 0, S_(s), (double2 *)d_OUT, (const double2 *)d_ST,
	                   (const double2 *)d_PS, p->ncoef, mode, (double)K, (double)M, wu, (const double *)nullptr, (size_t)0, (size_t)0);
	HIP_TRY(hipGetLastError());
	return 0;
}

static int weight_mode(double wu, int unbiased, unsigned K)
{
	if (wu == 2 && unbiased && K != 1) return 3; // selection rule ts_pws1f_lib.c:226-228, K==1 falls back :972
	if (wu == 2) return 0;
	if (wu == 1) return 1;
	return 2;
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

// work list of the polyphase inverse: one item group per run of consecutive scales with the same D
static int build_inverse_items(tspws_hip_plan *p)
{
	std::vector<OctDesc> oc;
	unsigned woff = 0;
	p->inv_ngeneric = 0;
	for (unsigned s = 0; s < p->S;) {
		unsigned e = s + 1;
		while (e < p->S && p->sc[e].D == p->sc[s].D) e++;
		static int gen_on = -1; // TSPWS_INV_GEN=0: decimations that do not divide N go to the one-thread-per-sample kernel
		if (gen_on < 0) { const char *ev = getenv("TSPWS_INV_GEN"); gen_on = (ev && *ev == '0') ? 0 : 1; }
		if (!p->sc[s].inv_fast && !gen_on) { p->inv_ngeneric += e - s; s = e; continue; }
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
	p->inv_waves_fast = 0;
	for (size_t i = 0; i < oc.size(); i++) {
		oc[i].wave_off = woff; oc[i].slot = (unsigned)i;
		woff += oc[i].MC * oc[i].ngw;
		if (!oc[i].gen) p->inv_waves_fast = woff;
	}
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

static bool use_generic_inverse()
{
	static int v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_INV_GENERIC"); v = (e && *e == '1') ? 1 : 0; }
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
	if (use_generic_inverse() || p->inv_noct == 0) {
		hipLaunchKernelGGL(k_inverse_generic<NREC>, dim3(nbx, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->S, p->d_wd, x, 0,
		                   (size_t)NREC * p->ncoef, slot);
		return 0;
	}
	const unsigned nslots = p->inv_noct + (p->inv_ngeneric ? 1 : 0);
	void *v;
	int rc = scratch(p, SCR_OBUF, (size_t)nb * nslots * slot * sizeof(double), &v);
	if (rc) return rc;
	double *obuf = (double *)v;
	if (p->inv_waves_fast)
		hipLaunchKernelGGL((k_inv_poly<NREC, false>), dim3((p->inv_waves_fast + 3) / 4, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->d_oc, p->inv_noct,
		                   p->d_wd, obuf, slot, p->inv_waves_fast, (size_t)NREC * p->ncoef, (size_t)nslots * slot, 0u);
	if (p->inv_waves > p->inv_waves_fast)
		hipLaunchKernelGGL((k_inv_poly<NREC, true>), dim3((p->inv_waves - p->inv_waves_fast + 3) / 4, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc,
		                   p->d_oc, p->inv_noct, p->d_wd, obuf, slot, p->inv_waves, (size_t)NREC * p->ncoef, (size_t)nslots * slot, p->inv_waves_fast);
	if (p->inv_ngeneric)
		hipLaunchKernelGGL(k_inverse_generic<NREC>, dim3(nbx, nb), dim3(256), 0, st, Y, p->ncoef, p->N, p->d_sc, p->S, p->d_wd,
		                   obuf + (size_t)p->inv_noct * slot, 1, (size_t)NREC * p->ncoef, (size_t)nslots * slot);
	if (NREC == 2 && nb == 1 && (f_ts || f_ls))
		hipLaunchKernelGGL(k_inv_combine_out, dim3(nbx), dim3(256), 0, st, obuf, slot, nslots, (size_t)p->N, f_ts, f_ls, f_mtr);
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

// Two reconstructions (sets 0 and 1 of Y) from the scales [s_lo, s_hi) ONLY -- whole decimation octaves -- as FP64 partial
// sums x2[2][N]: the reconstruction is a sum over scales, so the shares of disjoint scale ranges add up to the whole.
static int inverse_scales(tspws_hip_plan *p, const double2 *Y, double *x2, hipStream_t st, unsigned s_lo, unsigned s_hi)
{
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

// ------------------------------------------------------------------------------------------
// whole call on device-resident traces
// ------------------------------------------------------------------------------------------
static bool is_two_stage(const t_tsPWS *p, size_t mtr_global) { return !(!p->Kmax || p->Kmax > mtr_global); }
static int class_sums(tspws_hip_plan *pl, unsigned KM, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C, bool with_main,
                      hipStream_t st, size_t first = 0, size_t mtr_local = ~(size_t)0);
static int combine_classes(tspws_hip_plan *pl, unsigned col0, unsigned col1, double *d_P, hipStream_t st);

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
	if (is_two_stage(p, mtr_global)) {
		if (!mtr_local) { HIP_TRY(hipMemsetAsync(buf, 0, nd * sizeof(double), S_(s))); return 0; }
		if (pl->jk_prepared) { // a jackknife of these traces follows: one pass for the groups and all replicas
			const bool fits = first == 0 && mtr_local == mtr_global && mtr_local == pl->jk_mtr && p->Kmax == pl->jk_KM;
			pl->jk_prepared = false;
			if (fits) {
				HIP_TRY(hipSetDevice(pl->device));
				if ((rc = class_sums(pl, p->Kmax, d_x, ld, mtr_local, pl->jk_sel.data(), pl->jk_C, true, S_(s)))) return rc;
				return combine_classes(pl, pl->jk_C, pl->jk_C + 1, buf, S_(s));
			}
		}
		pl->cs.valid = false; // class sums of an earlier prepared call are no longer vouched for
		return tspws_hip_partial_stacks(pl, d_x, ld, mtr_local, first, mtr_global, p->Kmax, buf, pl->N, s);
	}
	if (!mtr_local) { HIP_TRY(hipMemsetAsync(buf, 0, nd * sizeof(double), S_(s))); return 0; }
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
	return stacks_impl<double>(pl, P + (size_t)g_begin * pl->N, g_end - g_begin, pl->N, ST, PS, s, g_begin != 0);
}

static int finish_tail(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, float *d_ls, float *d_ts, void *s, bool weighted)
{
	HIP_TRY(hipSetDevice(pl->device));
	double *OUT, *ST, *PS;
	int rc;
	void *v;
	if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
	const unsigned K = is_two_stage(p, mtr_global) ? p->Kmax : (unsigned)mtr_global;
	if (!weighted && (rc = tspws_hip_weight(pl, OUT, ST, PS, K, (unsigned)mtr_global, p->wu, p->unbiased, s))) return rc;
	if (!use_generic_inverse() && pl->inv_noct) { // set 0 = ICWT(OUT), set 1 = ICWT(ST); the combining kernel writes the floats
		if ((rc = inverse_launch<2>(pl, (const double2 *)OUT, nullptr, S_(s), 1, d_ts, d_ls, (float)(unsigned)mtr_global))) return rc;
		HIP_TRY(hipGetLastError());
		return 0;
	}
	if ((rc = scratch(pl, SCR_X2, 2 * (size_t)pl->N * sizeof(double), &v))) return rc;
	double *x2 = (double *)v;
	if ((rc = tspws_hip_inverse(pl, OUT, 2, x2, s))) return rc; // row 0 = ICWT(OUT), row 1 = ICWT(ST)
	return tspws_hip_epilogue(d_ls, d_ts, x2 + pl->N, x2, pl->N, (unsigned)mtr_global, s);
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
	if (is_two_stage(p, mtr_global)) {
		// all Kmax partial stacks at once: the launch that completes ST / PS also writes the weighted coefficients
		double *OUT, *ST, *PS, *P;
		size_t nd;
		if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
		if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &P, &nd))) return rc;
		WeightArgs wa;
		wa.OUT = (double2 *)OUT; wa.mode = weight_mode(p->wu, p->unbiased, p->Kmax); wa.K = (double)p->Kmax; wa.M = (double)(unsigned)mtr_global; wa.wu = p->wu;
		bool weighted = false;
		if ((rc = stacks_impl<double>(pl, P, p->Kmax, pl->N, ST, PS, s, false, &wa, &weighted))) return rc;
		return finish_tail(pl, p, mtr_global, d_ls, d_ts, s, weighted);
	} else {
		double *OUT, *ST, *PS, *B;
		size_t nd;
		if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
		if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &B, &nd))) return rc;
		HIP_TRY(hipMemcpyAsync(ST, B, 4 * pl->ncoef * sizeof(double), hipMemcpyDeviceToDevice, S_(s)));
	}
	return tspws_hip_stack_finish_tail(pl, p, mtr_global, d_ls, d_ts, s);
}

// ------------------------------------------------------------------------------------------
// Scale-sharded finish stage (multi-GPU).  After the all-reduce every rank holds the same K partial stacks; instead of
// finishing redundantly, rank r transforms, weights and reconstructs only ITS share of the scales -- the reconstruction
// is a sum over scales -- and the ranks add their partial reconstructions (2 N doubles) before the epilogue.
// ------------------------------------------------------------------------------------------
static bool finish_shardable(const tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global)
{
	return is_two_stage(p, mtr_global) && pl->fwd_kind == 1 && !use_generic_forward() && !use_generic_inverse() && pl->inv_noct && !pl->inv_ngeneric &&
	       !pl->oct_wgs && !(p->Kmax >= tl_min_traces() && tl_enabled() && fuse_enabled() && pl->tl_n);
}

// Contiguous, work-balanced share of the scales for `rank` of `world`: whole decimation octaves (the inverse sums the voices
// of an octave in registers), cost = weighted forward MACs.  Returns 0 with [*s_begin, *s_end) (possibly empty), 1 when this plan /
// parameter set has no sharded finish (the caller then finishes as a whole), an error code for bad arguments.
extern "C" int tspws_hip_finish_shard(const tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, unsigned rank, unsigned world, unsigned *s_begin,
                                      unsigned *s_end)
{
	if (!pl || !p || !s_begin || !s_end || !world || rank >= world) return fail(TSPWS_E_ARG, "finish_shard: bad argument");
	if (!finish_shardable(pl, p, mtr_global)) return 1;
	std::vector<unsigned> items(pl->inv_noct);
	for (unsigned i = 0; i < pl->inv_noct; i++) items[i] = i;
	std::sort(items.begin(), items.end(), [&](unsigned a, unsigned b) { return pl->oc_s0[a] < pl->oc_s0[b]; });
	std::vector<double> cost(items.size());
	double total = 0;
	for (size_t k = 0; k < items.size(); k++) {
		double c = 0;
		// forward MACs.  Shares of one or two octaves (world >= 4) are latency-bound -- a launch lasts as long as its
		// longest workgroup -- and the far-decimated octaves then cost about twice as much (tools/shard_finish_timing.py, per
		// octave on the north-star frame: 13 us above the launch-chain floor for D <= 128, ~30 us for D >= 256); shares of
		// many octaves (world 2: 0.148 | 0.157 ms for an even split of the MACs) follow the MAC count
		const double far_w = world >= 4 ? 2.0 : 1.0;
		for (unsigned s = pl->oc_s0[items[k]]; s < pl->oc_s0[items[k]] + pl->oc_nv[items[k]]; s++)
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
		if (lo == ~0u) lo = pl->oc_s0[items[k]];
		hi = pl->oc_s0[items[k]] + pl->oc_nv[items[k]];
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
	for (unsigned i = 0; i < pl->inv_noct; i++) { lo_ok |= pl->oc_s0[i] == s_begin; hi_ok |= pl->oc_s0[i] + pl->oc_nv[i] == s_end; }
	if (!lo_ok || !hi_ok) return fail(TSPWS_E_ARG, "stack_finish_scales: the range must consist of whole decimation octaves");
	double *OUT, *ST, *PS, *P;
	size_t nd;
	int rc;
	if ((rc = finish_block(pl, &OUT, &ST, &PS))) return rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &P, &nd))) return rc;
	WeightArgs wa;
	wa.OUT = (double2 *)OUT; wa.mode = weight_mode(p->wu, p->unbiased, p->Kmax); wa.K = (double)p->Kmax; wa.M = (double)(unsigned)mtr_global; wa.wu = p->wu;
	bool weighted = false;
	pl->rs0 = s_begin; pl->rs1 = s_end; // the finish-stage launches below cover these scales only
	rc = stacks_impl<double>(pl, P, p->Kmax, pl->N, ST, PS, s, false, &wa, &weighted);
	pl->rs0 = pl->rs1 = 0;
	if (rc) return rc;
	// (several forward batches: weight afterwards -- over all coefficients; those of other scales are never read)
	if (!weighted && (rc = tspws_hip_weight(pl, OUT, ST, PS, p->Kmax, (unsigned)mtr_global, p->wu, p->unbiased, s))) return rc;
	return inverse_scales(pl, (const double2 *)OUT, d_x2, st, s_begin, s_end);
}

// ------------------------------------------------------------------------------------------
// Whole call on one GPU, pipelined: the partial stacks are streamed group by group on the caller's stream
// while a second stream transforms each finished group (forward CWT + phase accumulation are FP64-bound, the
// streaming is HBM-bound, so they overlap), then weight / inverses / epilogue.  Same results as
// stack_local + stack_finish (same kernels, same summation order).
// ------------------------------------------------------------------------------------------
// Measured on MI355X (round 2, after the streaming pass went to one workgroup per CU and launch): the transforms now
// co-run, but the streaming stage slows from 0.73 to 0.80-0.85 ms -- 0.98-1.00 ms per call against 0.99-1.01 ms for the
// plain back-to-back schedule, within the box-to-box spread.  It stays opt-in (TSPWS_OVERLAP=1); DESIGN.md section 4.
static bool overlap_enabled()
{
	static int v = -1;
	if (v < 0) { const char *e = getenv("TSPWS_OVERLAP"); v = (e && *e == '1') ? 1 : 0; }
	return v == 1;
}

extern "C" int tspws_hip_profile_begin(tspws_hip_plan *pl, size_t max_calls)
{
	if (!pl) return fail(TSPWS_E_ARG, "profile_begin: NULL");
	HIP_TRY(hipSetDevice(pl->device));
	const size_t need = max_calls * 2;
	while (pl->prof_ev.size() < need) {
		hipEvent_t e;
		HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence)); // timing on, device-scope ordering only: a system-scope release would write the fresh partial stacks back out of L2 (8 us per call)
		pl->prof_ev.push_back(e);
	}
	pl->prof_used = 0;
	return 0;
}

extern "C" int tspws_hip_stream_launches(const tspws_hip_plan *pl) { return pl ? (int)pl->last_stream_launches : 0; }

extern "C" int tspws_hip_profile_end(tspws_hip_plan *pl, double *mean_ms, size_t *ncalls)
{
	if (!pl || !mean_ms || !ncalls) return fail(TSPWS_E_ARG, "profile_end: NULL");
	HIP_TRY(hipSetDevice(pl->device));
	HIP_TRY(hipDeviceSynchronize());
	double tot = 0;
	const size_t n = pl->prof_used / 2;
	for (size_t i = 0; i < n; i++) {
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, pl->prof_ev[2 * i], pl->prof_ev[2 * i + 1]));
		tot += ms;
	}
	*mean_ms = n ? tot / (double)n : 0.0;
	*ncalls = n;
	pl->prof_used = 0;
	return 0;
}

extern "C" int tspws_hip_stack(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, float *d_ls, float *d_ts,
                               void *s)
{
	if (!pl || !p || !d_x || !mtr) return fail(TSPWS_E_ARG, "stack: bad argument");
	HIP_TRY(hipSetDevice(pl->device));
	int rc;
	if (!is_two_stage(p, mtr) || !overlap_enabled()) {
		const bool prof0 = pl->prof_used + 2 <= pl->prof_ev.size();
		if (prof0) HIP_TRY(hipEventRecord(pl->prof_ev[pl->prof_used], S_(s)));
		if ((rc = tspws_hip_stack_local(pl, p, d_x, ld, mtr, 0, mtr, s))) return rc;
		if (prof0) { HIP_TRY(hipEventRecord(pl->prof_ev[pl->prof_used + 1], S_(s))); pl->prof_used += 2; }
		return tspws_hip_stack_finish(pl, p, mtr, d_ls, d_ts, s);
	}
	hipStream_t A = S_(s);
	const unsigned K = p->Kmax;
	const size_t N = pl->N, nc = pl->ncoef;
	if (!pl->aux) {
		int lo = 0, hi = 0; // numerically larger = lower priority
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
		const char *e = getenv("TSPWS_AUX_PRIO");
		const int pr = e ? (atoi(e) > 0 ? lo : atoi(e) < 0 ? hi : 0) : 0; // >0: lowest, <0: highest, 0: default
		HIP_TRY(hipStreamCreateWithPriority(&pl->aux, hipStreamNonBlocking, pr));
	}
	if (!pl->ev_done) HIP_TRY(hipEventCreateWithFlags(&pl->ev_done, hipEventDisableTiming | hipEventDisableSystemFence));
	while (pl->ev_grp.size() < (size_t)K + 1) {
		hipEvent_t e;
		HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
		pl->ev_grp.push_back(e);
	}
	hipStream_t Bq = pl->aux;
	void *v;
	double *P; size_t nd;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr, &P, &nd))) return rc;
	if ((rc = scratch(pl, SCR_OUT, 6 * nc * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *ST = OUT + 2 * nc, *PS = ST + 2 * nc;
	if ((rc = scratch(pl, SCR_X2, 2 * N * sizeof(double), &v))) return rc;
	double *x2 = (double *)v;
	// the aux stream must not start before earlier work on the caller's stream (e.g. a previous call's readers)
	HIP_TRY(hipEventRecord(pl->ev_grp[K], A));
	HIP_TRY(hipStreamWaitEvent(Bq, pl->ev_grp[K], 0));
	const bool prof = pl->prof_used + 2 <= pl->prof_ev.size();
	// groups per hand-over: a multiple of what one streaming launch covers (one workgroup per CU: two groups at N = 131072)
	const unsigned bx = (unsigned)((N + 1023) / 1024);
	unsigned pipe_batch = std::max(1u, 256u / std::max(1u, bx));
	if (const char *e = getenv("TSPWS_PIPE_BATCH")) pipe_batch = (unsigned)std::max(1, atoi(e));
	if (prof) HIP_TRY(hipEventRecord(pl->prof_ev[pl->prof_used], A));
	std::vector<unsigned> sched; // TSPWS_PIPE_SCHED=4,4,2: hand-over sizes in groups (the last one repeats)
	if (const char *e = getenv("TSPWS_PIPE_SCHED")) { for (const char *q = e; *q;) { sched.push_back((unsigned)std::max(1, atoi(q))); while (*q && *q != ',') q++; if (*q == ',') q++; } }
	size_t si = 0;
	for (unsigned g0 = 0, step = pipe_batch; g0 < K; g0 += step) {
		step = sched.empty() ? pipe_batch : sched[std::min(si++, sched.size() - 1)];
		const unsigned g1 = std::min(K, g0 + step);
		if ((rc = tspws_hip_partial_stacks_range(pl, d_x, ld, mtr, 0, mtr, K, g0, g1, P, N, A))) return rc;
		HIP_TRY(hipEventRecord(pl->ev_grp[g0], A));
		HIP_TRY(hipStreamWaitEvent(Bq, pl->ev_grp[g0], 0));
		// transforms + phase stacks of the finished groups beside the streaming of the next ones
		if ((rc = stacks_impl<double>(pl, P + (size_t)g0 * N, g1 - g0, N, ST, PS, Bq, g0 != 0))) return rc;
	}
	if (prof) { HIP_TRY(hipEventRecord(pl->prof_ev[pl->prof_used + 1], A)); pl->prof_used += 2; }
	if ((rc = tspws_hip_weight(pl, OUT, ST, PS, K, (unsigned)mtr, p->wu, p->unbiased, Bq))) return rc;
	if ((rc = tspws_hip_inverse(pl, OUT, 2, x2, Bq))) return rc;
	if ((rc = tspws_hip_epilogue(d_ls, d_ts, x2 + N, x2, N, (unsigned)mtr, Bq))) return rc;
	HIP_TRY(hipEventRecord(pl->ev_done, Bq));
	HIP_TRY(hipStreamWaitEvent(A, pl->ev_done, 0));
	HIP_TRY(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------------------------------
// jackknife (TwoStage_jackknife_float, ts_pws1f_lib.c:719-831)
// ------------------------------------------------------------------------------------------
extern "C" int tspws_jackknife_plan(char *sel, const time_t *tm, size_t mtr, unsigned d, unsigned n, unsigned C)
{
	if (!sel || !tm) return 1;
	if (tm[0] == 0) return -2;
	std::vector<unsigned> bin(mtr), comb(d);
	for (size_t i = 0; i < mtr; i++) {
		struct tm g;
		gmtime_r(tm + i, &g);
		bin[i] = (unsigned)floor((double)(g.tm_yday * (int)n) / 365.); // day-of-year bin, :398-401
	}
	for (unsigned i = 0; i < d; i++) comb[i] = i;
	for (unsigned c = 0; c < C; c++) {
		if (c) { // lexicographic successor of the deleted-bin set, :405-414
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
	return 0;
}

// P_c[g][n] = sum of class sums whose signature sends them to group g of replica c.
// cls_of[(c*Kmax+g)] lists are given as CSR: row_ptr / cols.
__global__ void __launch_bounds__(256) k_combine_classes(const double *__restrict__ cls, size_t ldc, const unsigned *__restrict__ row_ptr,
                                                         const unsigned *__restrict__ cols, double *__restrict__ P, size_t N)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const unsigned row = blockIdx.y;
	double acc = 0;
	for (unsigned j = row_ptr[row]; j < row_ptr[row + 1]; j++) acc += cls[(size_t)cols[j] * ldc + n];
	P[(size_t)row * N + n] = acc;
}

// replica linear stack in the time domain, :799-811: (sum_g P[g]) * (1/K)
__global__ void __launch_bounds__(256) k_jk_linear(const double *__restrict__ P, unsigned Kmax, size_t N, const double *__restrict__ Mv,
                                                   float *__restrict__ out)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	P += (size_t)blockIdx.y * Kmax * N; out += (size_t)blockIdx.y * N; // blockIdx.y = replica
	const double invK = 1. / Mv[blockIdx.y];
	double acc = P[n];
	for (unsigned g = 1; g < Kmax; g++) acc += P[(size_t)g * N + n];
	out[n] = (float)(acc * invK);
}

// ONE pass over the traces for any number of masked two-stage replicas: traces with the same destination group in every
// replica form a class (maximal runs of consecutive traces; equal signatures of separate runs share a class), the
// streaming kernel sums every class once, and each (replica, group) partial stack is a sum of class sums.
// with_main: the plain two-stage groups of ALL traces (ts_pws1f_lib.c:876) are one more signature column.
// Sharded ensembles: d_x holds traces [first, first + mtr_local) of the mtr the selection refers to; signatures come from
// the GLOBAL trace index (a trace's group in a replica is its rank among ALL selected traces, :766), the sums run over the
// shard's traces only, so the rows of all shards add up to the rows of the whole ensemble.
static int class_sums(tspws_hip_plan *pl, unsigned KM, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C, bool with_main,
                      hipStream_t st, size_t first, size_t mtr_local)
{
	tspws_hip_plan::ClassSums &cs = pl->cs;
	cs.valid = false;
	if (mtr_local == ~(size_t)0) { first = 0; mtr_local = mtr; }
	if (first > mtr || mtr_local > mtr - first) return fail(TSPWS_E_ARG, "class sums: shard outside the ensemble");
	const size_t lo = first, hi = first + mtr_local;
	const size_t N = pl->N;
	const unsigned W = C + (with_main ? 1u : 0u);
	// signature of trace i: group index in every replica (0xFFFF = deleted)
	cs.Kc.assign(C, 0);
	for (unsigned c = 0; c < C; c++) for (size_t i = 0; i < mtr; i++) if (h_sel[(size_t)c * mtr + i] == 1) cs.Kc[c]++;
	std::vector<unsigned short> sig((size_t)mtr * W);
	for (unsigned c = 0; c < C; c++) {
		size_t k = 0;
		for (size_t i = 0; i < mtr; i++) {
			if (h_sel[(size_t)c * mtr + i] == 1) {
				sig[i * W + c] = (unsigned short)floor((double)(k * KM) / (double)cs.Kc[c]); // :766
				k++;
			} else sig[i * W + c] = 0xFFFF;
		}
	}
	if (with_main) for (size_t i = 0; i < mtr; i++) sig[i * W + C] = (unsigned short)std::min<size_t>((size_t)floor((double)(i * KM) / (double)mtr), KM - 1);
	cs.sig.clear(); cs.chunks.clear();
	std::vector<std::vector<Chunk>> cls_chunks;
	const unsigned clen = chunk_len_for(N, std::max<size_t>(mtr_local, 1));
	for (size_t i = lo; i < hi;) {
		size_t j = i + 1;
		while (j < hi && !memcmp(&sig[i * W], &sig[j * W], W * sizeof(unsigned short))) j++;
		std::vector<unsigned short> sg(sig.begin() + i * W, sig.begin() + (i + 1) * W);
		size_t id = 0;
		for (; id < cs.sig.size(); id++) if (cs.sig[id] == sg) break;
		if (id == cs.sig.size()) { cs.sig.push_back(sg); cls_chunks.emplace_back(); }
		for (size_t t = i; t < j; t += clen) {
			Chunk c; c.t0 = t - lo; c.count = (unsigned)std::min<size_t>(clen, j - t); c.row = (unsigned)id; // t0: row of d_x
			cls_chunks[id].push_back(c);
		}
		i = j;
	}
	const unsigned ncls = (unsigned)cs.sig.size();
	cs.row_first.assign(ncls + 1, 0);
	for (unsigned id = 0; id < ncls; id++) {
		cs.row_first[id] = (unsigned)cs.chunks.size();
		cs.chunks.insert(cs.chunks.end(), cls_chunks[id].begin(), cls_chunks[id].end());
	}
	cs.row_first[ncls] = (unsigned)cs.chunks.size();
	int rc;
	void *v;
	if ((rc = scratch(pl, SCR_CLS, std::max<size_t>((size_t)ncls * N, 1) * sizeof(double), &v))) return rc;
	if (ncls && (rc = run_chunks(pl, d_x, ld, N, cs.chunks, cs.row_first, ncls, (double *)v, N, st, false))) return rc; // own table: not the cached one
	cs.d_x = d_x; cs.ld = ld; cs.mtr = mtr; cs.first = first; cs.mtr_local = mtr_local; cs.C = C; cs.KM = KM; cs.ncls = ncls; cs.has_main = with_main;
	cs.sel.assign(h_sel, h_sel + (size_t)C * mtr);
	cs.valid = true;
	return 0;
}

// d_P[(col - col0) * KM + g][n] = sum of the class sums whose signature column `col` is g, for col in [col0, col1)
static int combine_classes(tspws_hip_plan *pl, unsigned col0, unsigned col1, double *d_P, hipStream_t st)
{
	tspws_hip_plan::ClassSums &cs = pl->cs;
	const unsigned KM = cs.KM, nrow = (col1 - col0) * KM;
	const size_t N = pl->N;
	cs.rp.assign((size_t)nrow + 1, 0); cs.cols.clear();
	for (unsigned c = col0; c < col1; c++)
		for (unsigned g = 0; g < KM; g++) {
			cs.rp[(size_t)(c - col0) * KM + g] = (unsigned)cs.cols.size();
			for (unsigned id = 0; id < cs.ncls; id++) if (cs.sig[id][c] == g) cs.cols.push_back(id);
		}
	cs.rp[nrow] = (unsigned)cs.cols.size();
	void *v;
	int rc;
	if ((rc = scratch(pl, SCR_JKTAB, (cs.rp.size() + cs.cols.size() + 1) * sizeof(unsigned), &v))) return rc;
	unsigned *d_rp = (unsigned *)v, *d_cols = d_rp + cs.rp.size();
	HIP_TRY(hipMemcpyAsync(d_rp, cs.rp.data(), cs.rp.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
	if (!cs.cols.empty()) HIP_TRY(hipMemcpyAsync(d_cols, cs.cols.data(), cs.cols.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
	for (unsigned r0 = 0; r0 < nrow; r0 += 65535) {
		const unsigned ny = std::min(nrow - r0, 65535u);
		hipLaunchKernelGGL(k_combine_classes, dim3((unsigned)((N + 255) / 256), ny), dim3(256), 0, st, (const double *)pl->scr[SCR_CLS], N, d_rp + r0, d_cols,
		                   d_P + (size_t)r0 * N, N);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// Announce the jackknife that will follow the next two-stage tspws_hip_stack_local of the SAME traces (whole ensemble on this
// device): that call then streams the traces once for its own groups and for every replica, and tspws_hip_jackknife with the
// same selection reuses the class sums instead of streaming the traces again.  The caller must not change the traces
// between the two calls (tspws_main does not).
extern "C" int tspws_hip_jackknife_prepare(tspws_hip_plan *pl, const t_tsPWS *p, const char *h_sel, unsigned C, size_t mtr)
{
	if (!pl || !p) return fail(TSPWS_E_ARG, "jackknife_prepare: NULL");
	pl->jk_prepared = false;
	if (!h_sel || !C || !is_two_stage(p, mtr)) return 0; // nothing to share
	pl->jk_sel.assign(h_sel, h_sel + (size_t)C * mtr);
	pl->jk_C = C; pl->jk_KM = p->Kmax; pl->jk_mtr = mtr;
	pl->jk_prepared = true;
	return 0;
}

static int masked_two_stage(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C,
                            float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s);

extern "C" int tspws_hip_jackknife(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const char *h_sel,
                                   unsigned C, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	if (!pl || !p || !d_x || !h_sel || !d_ls_out || !d_ts_out || !h_mtr_out) return fail(TSPWS_E_ARG, "jackknife: NULL");
	if (!is_two_stage(p, mtr) || !C) return 0; // single-stage variant is an empty stub in the reference (:711-716)
	return masked_two_stage(pl, p, d_x, ld, mtr, h_sel, C, d_ls_out, d_ts_out, h_mtr_out, s);
}

// Rows of the masked replicas: d_P[c * KM + g][N] (SCR_JKP; the replicas' trace counts follow the rows).
static int replica_rows_buffer(tspws_hip_plan *pl, unsigned KM, unsigned C, double **d_P)
{
	void *v;
	int rc = scratch(pl, SCR_JKP, ((size_t)C * KM * pl->N + C) * sizeof(double), &v);
	if (rc) return rc;
	*d_P = (double *)v;
	return 0;
}

// Replicas [c_begin, c_end) from their partial-stack rows: transforms, phase stacks, weights, time-domain linear stacks,
// inverses.  Outputs land in rows c_begin.. of d_ls_out / d_ts_out / h_mtr_out ([C][N] arrays indexed by replica).
static int finish_replicas(tspws_hip_plan *pl, const t_tsPWS *p, double *d_P, const std::vector<size_t> &Kc, unsigned C, unsigned c_begin,
                           unsigned c_end, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	const size_t N = pl->N;
	int rc;
	void *v;
	double *d_Mv = d_P + (size_t)C * KM * N;
	std::vector<double> h_Mv(C);
	for (unsigned c = 0; c < C; c++) h_Mv[c] = (double)Kc[c];
	HIP_TRY(hipMemcpyAsync(d_Mv, h_Mv.data(), C * sizeof(double), hipMemcpyHostToDevice, st));
	// Replicas are processed in batches: ONE forward launch transforms the KM partials of a whole batch of replicas (the
	// kernels fill the GPU far better with 100 traces than with 10), then per replica the phase accumulation and the
	// weight (KM, K_c), and the inverses two replicas at a time.
	const size_t nc = pl->ncoef;
	const unsigned nrep = c_end - c_begin;
	unsigned RB = (unsigned)std::max<size_t>(1, std::min<size_t>(nrep, part_budget_bytes() / std::max<size_t>(1, (size_t)KM * pl->npart * sizeof(double2))));
	if (RB > 1) RB &= ~1u; // pairs for the two-set inverse
	if ((rc = scratch(pl, SCR_PART, (size_t)RB * KM * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	// per replica of the batch: OUT (2 nc doubles) | ST | PS, then the reconstructions
	if ((rc = scratch(pl, SCR_JKOUT, ((size_t)RB * 6 * nc + (size_t)RB * N) * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *STr = OUT + (size_t)RB * 2 * nc, *xr = STr + (size_t)RB * 4 * nc;
	const bool fuse = fuse_enabled() && pl->fwd_kind == 1 && pl->n_fusable;
	for (unsigned c0 = c_begin; c0 < c_end; c0 += RB) {
		const unsigned nr = std::min(RB, c_end - c0);
		// one slice of the fused forward kernel = the KM partial stacks of one replica: its stacks land in the replica's planes
		FuseOut fz;
		fz.accST = (double2 *)STr; fz.accPS = (double2 *)STr + nc; fz.stride = 2 * nc; fz.tps = KM;
		if ((rc = forward_parts<double>(pl, d_P + (size_t)c0 * KM * N, (size_t)nr * KM, N, part, st, fuse ? &fz : nullptr))) return rc;
		for (unsigned j = 0; j < nr; j++) h_mtr_out[c0 + j] = (unsigned)Kc[c0 + j];
		// the replicas of the batch side by side in every launch (grid.y): stacks of the scales the fused kernel left out,
		// weights with each replica's trace count, time-domain linear stacks, inverses two replicas per tap read, outputs
		FuseOut fj = fz; // replica j's slice went straight into its ST / PS planes
		launch_accumulate(pl, (const double2 *)part, KM, (double2 *)STr, (double2 *)STr + nc, 1, &fj, 1, st, nr, (size_t)KM * pl->npart, 2 * nc);
		hipLaunchKernelGGL(k_weight, dim3((unsigned)((nc + 255) / 256), nr), dim3(256), 0, st, (double2 *)OUT, (const double2 *)STr,
		                   (const double2 *)STr + nc, nc, weight_mode(p->wu, p->unbiased, KM), (double)KM, 0.0, p->wu, (const double *)(d_Mv + c0), nc, 2 * nc);
		hipLaunchKernelGGL(k_jk_linear, dim3((unsigned)((N + 255) / 256), nr), dim3(256), 0, st, d_P + (size_t)c0 * KM * N, KM, N, (const double *)(d_Mv + c0),
		                   d_ls_out + (size_t)c0 * N);
		if ((rc = tspws_hip_inverse(pl, OUT, nr, xr, s))) return rc;
		hipLaunchKernelGGL(k_epilogue, dim3((unsigned)((N + 255) / 256), nr), dim3(256), 0, st, (float *)nullptr, d_ts_out + (size_t)c0 * N, (const double *)nullptr,
		                   (const double *)xr, N, 1.0f);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st)); // host tables above go out of scope
	return 0;
}

// All C masked two-stage replicas from ONE pass over the traces (shared by the jackknife and the two-stage
// random subsampling, whose per-replica bodies are identical in the reference: :758-811 and :642-691).
static int masked_two_stage(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const char *h_sel, unsigned C,
                            float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	int rc;
	// class sums: left behind by a prepared tspws_hip_stack_local of these traces with this selection, else streamed now
	tspws_hip_plan::ClassSums &cs = pl->cs;
	const bool reuse = cs.valid && cs.d_x == d_x && cs.ld == ld && cs.mtr == mtr && cs.first == 0 && cs.mtr_local == mtr && cs.C == C && cs.KM == KM &&
	                   cs.sel.size() == (size_t)C * mtr && !memcmp(cs.sel.data(), h_sel, (size_t)C * mtr);
	if (!reuse && (rc = class_sums(pl, KM, d_x, ld, mtr, h_sel, C, false, st))) return rc;
	const std::vector<size_t> Kc = cs.Kc;
	double *d_P;
	if ((rc = replica_rows_buffer(pl, KM, C, &d_P))) return rc;
	if ((rc = combine_classes(pl, 0, C, d_P, st))) return rc;
	cs.valid = false; // one use: the traces may change after this call
	HIP_TRY(hipGetLastError());
	return finish_replicas(pl, p, d_P, Kc, C, 0, C, d_ls_out, d_ts_out, h_mtr_out, s);
}

// ---- trace-sharded jackknife: shard-local rows -> the caller's reduction -> replicas finished where they are owned ----
extern "C" int tspws_hip_jackknife_buffer(tspws_hip_plan *pl, const t_tsPWS *p, unsigned C, double **d_buf, size_t *nd)
{
	if (!pl || !p || !d_buf || !nd) return fail(TSPWS_E_ARG, "jackknife_buffer: NULL");
	if (!p->Kmax || !C) return fail(TSPWS_E_ARG, "jackknife_buffer: two-stage calls with C > 0");
	HIP_TRY(hipSetDevice(pl->device));
	*nd = (size_t)C * p->Kmax * pl->N;
	return replica_rows_buffer(pl, p->Kmax, C, d_buf);
}

extern "C" int tspws_hip_jackknife_local(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr_local, size_t first,
                                         size_t mtr_global, const char *h_sel, unsigned C, void *s)
{
	if (!pl || !p || !h_sel || (!d_x && mtr_local)) return fail(TSPWS_E_ARG, "jackknife_local: NULL");
	if (!is_two_stage(p, mtr_global) || !C) return fail(TSPWS_E_ARG, "jackknife_local: two-stage calls with C > 0");
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const unsigned KM = p->Kmax;
	double *main_rows, *d_P;
	size_t nd;
	int rc;
	if ((rc = tspws_hip_reduce_buffer(pl, p, mtr_global, &main_rows, &nd))) return rc;
	if ((rc = replica_rows_buffer(pl, KM, C, &d_P))) return rc;
	pl->jk_prepared = false;
	// ONE pass over the shard for the plain groups and for every replica (empty shard: all rows zero)
	if ((rc = class_sums(pl, KM, d_x, ld, mtr_global, h_sel, C, true, st, first, mtr_local))) return rc;
	if ((rc = combine_classes(pl, C, C + 1, main_rows, st))) return rc;
	if ((rc = combine_classes(pl, 0, C, d_P, st))) return rc;
	pl->cs.valid = false; // the rows, not the class sums, are what the caller reduces
	HIP_TRY(hipStreamSynchronize(st)); // the CSR tables of the two combine passes share one scratch block and host vectors
	return 0;
}

extern "C" int tspws_hip_jackknife_finish(tspws_hip_plan *pl, const t_tsPWS *p, size_t mtr_global, const char *h_sel, unsigned C, unsigned c_begin,
                                          unsigned c_end, float *d_ls_out, float *d_ts_out, unsigned *h_mtr_out, void *s)
{
	if (!pl || !p || !h_sel || !d_ls_out || !d_ts_out || !h_mtr_out) return fail(TSPWS_E_ARG, "jackknife_finish: NULL");
	if (!is_two_stage(p, mtr_global) || !C || c_begin > c_end || c_end > C) return fail(TSPWS_E_ARG, "jackknife_finish: two-stage calls, 0 <= c_begin <= c_end <= C");
	if (c_begin == c_end) return 0;
	HIP_TRY(hipSetDevice(pl->device));
	std::vector<size_t> Kc(C, 0);
	for (unsigned c = 0; c < C; c++) for (size_t i = 0; i < mtr_global; i++) if (h_sel[(size_t)c * mtr_global + i] == 1) Kc[c]++;
	double *d_P;
	int rc;
	if ((rc = replica_rows_buffer(pl, p->Kmax, C, &d_P))) return rc;
	return finish_replicas(pl, p, d_P, Kc, C, c_begin, c_end, d_ls_out, d_ts_out, h_mtr_out, s);
}

// ------------------------------------------------------------------------------------------
// random subsampling (SubsamplingPlan :355-383, tspws_subsmpl_float :501-610, TwoStage_subsmpl_float :612-709)
// ------------------------------------------------------------------------------------------
extern "C" int tspws_subsampling_plan(char *sel, size_t J, size_t K)
{
	if (!sel) return 1;
	if (K > J) return 2;
	size_t k = 0;
	if (2 * K < J) { // fewer ones than zeros: switch ones on
		memset(sel, 0, J);
		while (k < K) { const size_t j = (size_t)rand() % J; if (!sel[j]) { k++; sel[j] = 1; } }
	} else {         // otherwise switch zeros on
		memset(sel, 1, J);
		K = J - K;
		while (k < K) { const size_t j = (size_t)rand() % J; if (sel[j]) { k++; sel[j] = 0; } }
	}
	return 0;
}

// ST_m += Y_b, PS_m += Y_b/|Y_b| for every mask m that contains trace b; one thread per coefficient, the
// (<= 8) traces of the batch are normalised once and reused for all masks.
__global__ void __launch_bounds__(256) k_accumulate_masked(const double2 *__restrict__ part, size_t npart, const ScaleDesc *__restrict__ sc,
                                                           unsigned S, size_t ncoef, unsigned ntr, const char *__restrict__ sel, size_t mtr,
                                                           size_t t0, unsigned M, double2 *__restrict__ ST, double2 *__restrict__ PS)
{
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].acc_off <= blockIdx.x) lo = mid; else hi = mid;
	}
	const unsigned Ns = sc[lo].Ns, nsplit = sc[lo].nsplit;
	const unsigned k = (blockIdx.x - sc[lo].acc_off) * 256 + threadIdx.x;
	if (k >= Ns) return;
	const size_t i = sc[lo].coef_off + k;
	const double2 *p0 = part + sc[lo].part_off + k;
	double2 v[8], u[8];
#pragma unroll
	for (int b = 0; b < 8; b++) {
		v[b] = make_double2(0, 0); u[b] = make_double2(0, 0);
		if ((unsigned)b < ntr) {
			const double2 *p = p0 + (size_t)b * npart;
			double2 a = p[0];
			for (unsigned sp = 1; sp < nsplit; sp++) { const double2 t = p[(size_t)sp * Ns]; a.x += t.x; a.y += t.y; }
			v[b] = a;
			add_unit_phasor(u[b], a);
		}
	}
	for (unsigned m = 0; m < M; m++) {
		const char *row = sel + (size_t)m * mtr + t0;
		double2 st = ST[(size_t)m * ncoef + i], ps = PS[(size_t)m * ncoef + i];
		bool any = false;
#pragma unroll
		for (int b = 0; b < 8; b++)
			if ((unsigned)b < ntr && row[b] == 1) { st.x += v[b].x; st.y += v[b].y; ps.x += u[b].x; ps.y += u[b].y; any = true; }
		if (any) { ST[(size_t)m * ncoef + i] = st; PS[(size_t)m * ncoef + i] = ps; }
	}
}

// time-domain linear stacks of the subsamples with the reference's FLOAT accumulator, traces in order
// (ts_pws1f_lib.c:538-542), then the float scale W/K (:579-583).  grid.y = mask
__global__ void __launch_bounds__(256) k_sub_linear(const float *__restrict__ x, size_t ld, size_t N, size_t mtr, const char *__restrict__ sel,
                                                    float scale, float *__restrict__ out)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const char *row = sel + (size_t)blockIdx.y * mtr;
	float acc = 0.f;
	for (size_t i = 0; i < mtr; i++)
		if (row[i] == 1) acc = (float)((double)acc + (double)x[i * ld + n]);
	out[(size_t)blockIdx.y * N + n] = acc * scale;
}

extern "C" int tspws_hip_subsample(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, unsigned M,
                                   float *d_ls_out, float *d_ts_out, void *s)
{
	if (!pl || !p || !d_x || !d_ls_out || !d_ts_out) return fail(TSPWS_E_ARG, "subsample: NULL");
	if (!M || !mtr) return 0;
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const size_t K = (size_t)ceil((double)mtr * p->subsmpl_p);
	std::vector<char> sel((size_t)M * mtr);
	for (unsigned m = 0; m < M; m++) tspws_subsampling_plan(sel.data() + (size_t)m * mtr, mtr, K); // same rand() order as the reference
	if (is_two_stage(p, mtr)) {
		std::vector<unsigned> cnt(M);
		return masked_two_stage(pl, p, d_x, ld, mtr, sel.data(), M, d_ls_out, d_ts_out, cnt.data(), s);
	}
	const size_t N = pl->N, nc = pl->ncoef;
	int rc;
	void *v;
	if ((rc = scratch(pl, SCR_SEL, (size_t)M * mtr, &v))) return rc;
	char *d_sel = (char *)v;
	HIP_TRY(hipMemcpyAsync(d_sel, sel.data(), (size_t)M * mtr, hipMemcpyHostToDevice, st));
	if ((rc = scratch(pl, SCR_SUBST, (size_t)M * nc * 2 * sizeof(double2), &v))) return rc;
	double2 *STm = (double2 *)v, *PSm = STm + (size_t)M * nc;
	HIP_TRY(hipMemsetAsync(STm, 0, (size_t)M * nc * 2 * sizeof(double2), st));
	if ((rc = scratch(pl, SCR_PART, 8 * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	for (size_t t0 = 0; t0 < mtr; t0 += 8) {
		const unsigned nb = (unsigned)std::min<size_t>(8, mtr - t0);
		if ((rc = forward_parts<float>(pl, d_x + t0 * ld, nb, ld, part, st))) return rc;
		hipLaunchKernelGGL(k_accumulate_masked, dim3(pl->acc_blocks), dim3(256), 0, st, (const double2 *)part, pl->npart, pl->d_sc, pl->S, nc,
		                   nb, d_sel, mtr, t0, M, STm, PSm);
	}
	const float scale = (float)(1. / (double)K); // fa1 = W[m]/K with W = 1 (:580)
	hipLaunchKernelGGL(k_sub_linear, dim3((unsigned)((N + 255) / 256), M), dim3(256), 0, st, d_x, ld, N, mtr, d_sel, scale, d_ls_out);
	if ((rc = scratch(pl, SCR_JKOUT, (2 * nc + N) * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *xr = OUT + 2 * nc;
	for (unsigned m = 0; m < M; m++) {
		if ((rc = tspws_hip_weight(pl, OUT, (double *)(STm + (size_t)m * nc), (double *)(PSm + (size_t)m * nc), (unsigned)K, (unsigned)K, p->wu,
		                           p->unbiased, s))) return rc;
		if ((rc = tspws_hip_inverse(pl, OUT, 1, xr, s))) return rc;
		if ((rc = tspws_hip_epilogue(nullptr, d_ts_out + (size_t)m * N, nullptr, xr, N, 1, s))) return rc;
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st)); // `sel` goes out of scope
	return 0;
}

// ------------------------------------------------------------------------------------------
// convergence curves (ts_pws1f_lib.c:247-314, similarity :433-449, misfit :452-462)
// ------------------------------------------------------------------------------------------
// out[0] = sum d*r, out[1] = sum d*d, out[2] = sum (d-r)^2, out[3] = sum r*r ; one workgroup, fixed order
__global__ void __launch_bounds__(1024) k_dot4(const double *__restrict__ d, const float *__restrict__ r, size_t N, double *__restrict__ out)
{
	__shared__ double red[16][4];
	double a = 0, b = 0, c = 0, e = 0;
	for (size_t n = threadIdx.x; n < N; n += 1024) {
		const double dv = d[n], rv = (double)r[n], df = dv - rv;
		a = fma(dv, rv, a); b = fma(dv, dv, b); c = fma(df, df, c); e = fma(rv, rv, e);
	}
	a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); e = wave_sum(e);
	if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; red[threadIdx.x >> 6][2] = c; red[threadIdx.x >> 6][3] = e; }
	__syncthreads();
	if (threadIdx.x < 4) {
		double t = 0;
		for (int w = 0; w < 16; w++) t += red[w][threadIdx.x];
		out[threadIdx.x] = t;
	}
}

// running linear stack: d += x_i; d *= (float)(1/(i+1)); metrics; d *= (i+1)   (:288-308, literal FLOAT reciprocal)
// every workgroup writes its partial sums of the three metrics per step; k_conv_lin_reduce adds them in order
__global__ void __launch_bounds__(256) k_conv_linear(const float *__restrict__ x, size_t ld, size_t N, size_t mtr, const float *__restrict__ ref,
                                                     double *__restrict__ partial, float *__restrict__ steps)
{
	__shared__ double red[4][3];
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	const bool live = n < N;
	const double rv = live ? (double)ref[n] : 0.0;
	double d = 0;
	for (size_t i = 0; i < mtr; i++) {
		if (live) d += (double)x[i * ld + n];
		const float inv = (float)(1.0 / (double)(i + 1));
		d *= (double)inv;
		const double df = d - rv;
		double a = live ? d * rv : 0.0, b = live ? d * d : 0.0, c = live ? df * df : 0.0;
		a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; red[threadIdx.x >> 6][2] = c; }
		__syncthreads();
		if (threadIdx.x < 3)
			partial[((size_t)blockIdx.x * mtr + i) * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
		__syncthreads();
		if (steps && live) steps[i * N + n] = (float)d;
		d *= (double)(i + 1);
	}
}

__global__ void __launch_bounds__(256) k_conv_lin_reduce(const double *__restrict__ partial, unsigned nblocks, size_t mtr, double *__restrict__ out)
{
	const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; // index into [mtr][3]
	if (j >= mtr * 3) return;
	double t = 0;
	for (unsigned b = 0; b < nblocks; b++) t += partial[(size_t)b * mtr * 3 + j];
	out[j] = t;
}

extern "C" int tspws_hip_convergence(tspws_hip_plan *pl, const t_tsPWS *p, const float *d_x, size_t ld, size_t mtr, const float *d_ref_ts,
                                     const float *d_ref_ls, double *h_ts_sim, double *h_ts_misfit, double *h_ls_sim, double *h_ls_misfit,
                                     float *d_ts_steps, float *d_ls_steps, void *s)
{
	if (!pl || !p || !d_x || !d_ref_ts || !d_ref_ls || !h_ts_sim || !h_ts_misfit || !h_ls_sim || !h_ls_misfit)
		return fail(TSPWS_E_ARG, "convergence: NULL");
	if (!mtr) return 0;
	HIP_TRY(hipSetDevice(pl->device));
	hipStream_t st = S_(s);
	const size_t N = pl->N, nc = pl->ncoef;
	int rc;
	void *v;
	if ((rc = scratch(pl, SCR_OUT, 6 * nc * sizeof(double), &v))) return rc;
	double *OUT = (double *)v, *ST = OUT + 2 * nc, *PS = ST + 2 * nc;
	if ((rc = scratch(pl, SCR_X2, 2 * N * sizeof(double), &v))) return rc;
	double *xr = (double *)v;
	const unsigned nblk = (unsigned)((N + 255) / 256);
	if ((rc = scratch(pl, SCR_CONV, ((size_t)mtr * 4 + (size_t)nblk * mtr * 3 + mtr * 3) * sizeof(double), &v))) return rc;
	double *d_ts = (double *)v, *d_lpart = d_ts + mtr * 4, *d_lin = d_lpart + (size_t)nblk * mtr * 3;
	if ((rc = scratch(pl, SCR_PART, 2 * pl->npart * sizeof(double2), &v))) return rc;
	double2 *part = (double2 *)v;
	double *P = nullptr;
	if (p->Kmax) { if ((rc = scratch(pl, SCR_P, (size_t)p->Kmax * N * sizeof(double), &v))) return rc; P = (double *)v; }
	for (size_t i = 0; i < mtr; i++) {
		const size_t Tr = i + 1;
		unsigned K;
		if (!p->Kmax || p->Kmax >= Tr) { // incremental single-stage step (tspws_stacks_float_1step, :835-863)
			K = (unsigned)Tr;
			if ((rc = forward_parts<float>(pl, d_x + i * ld, 1, ld, part, st))) return rc;
			launch_accumulate(pl, (const double2 *)part, 1u, (double2 *)ST, (double2 *)PS, i == 0 ? 1 : 0, nullptr, 0, st);
		} else { // two-stage over the first Tr traces, recomputed from scratch like the reference (:266-268)
			K = p->Kmax;
			if ((rc = tspws_hip_partial_stacks(pl, d_x, ld, Tr, 0, Tr, K, P, N, s))) return rc;
			if ((rc = tspws_hip_stacks_double(pl, P, K, N, ST, PS, s))) return rc;
		}
		if ((rc = tspws_hip_weight(pl, OUT, ST, PS, K, (unsigned)Tr, p->wu, p->unbiased, s))) return rc;
		if ((rc = tspws_hip_inverse(pl, OUT, 1, xr, s))) return rc;
		hipLaunchKernelGGL(k_dot4, dim3(1), dim3(1024), 0, st, (const double *)xr, d_ref_ts, N, d_ts + i * 4);
		if (d_ts_steps && (rc = tspws_hip_epilogue(nullptr, d_ts_steps + i * N, nullptr, xr, N, 1, s))) return rc;
	}
	hipLaunchKernelGGL(k_conv_linear, dim3(nblk), dim3(256), 0, st, d_x, ld, N, mtr, d_ref_ls, d_lpart, d_ls_steps);
	hipLaunchKernelGGL(k_conv_lin_reduce, dim3((unsigned)((mtr * 3 + 255) / 256)), dim3(256), 0, st, (const double *)d_lpart, nblk, mtr, d_lin);
	hipLaunchKernelGGL(k_dot4, dim3(1), dim3(1024), 0, st, (const double *)xr, d_ref_ls, N, d_lpart); // only out[3] = sum ref_ls^2 is used
	HIP_TRY(hipGetLastError());
	std::vector<double> hts(mtr * 4), hl(mtr * 3);
	double lsq[4];
	HIP_TRY(hipMemcpyAsync(hts.data(), d_ts, mtr * 4 * sizeof(double), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(hl.data(), d_lin, mtr * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(lsq, d_lpart, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (size_t i = 0; i < mtr; i++) {
		h_ts_sim[i] = hts[i * 4] / sqrt(hts[i * 4 + 1]) / sqrt(hts[i * 4 + 3]);
		h_ts_misfit[i] = hts[i * 4 + 2];
		h_ls_sim[i] = hl[i * 3] / sqrt(hl[i * 3 + 1]) / sqrt(lsq[3]);
		h_ls_misfit[i] = hl[i * 3 + 2];
	}
	return 0;
}

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
