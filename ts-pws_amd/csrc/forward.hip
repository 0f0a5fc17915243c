// forward.hip -- forward frame CWT and the linear / phase stacks in the time-scale domain.
//
//   few traces (the K partial stacks of a two-stage call, the per-trace API): k_fwd_lds (fwd_lds.h) + k_fwd_poly (fwd_poly.h)
//   many traces (single-stage batches >= 64): k_fwd_tl (fwd_tl.h) on the transposed batch
//   check kernel (TSPWS_FWD_GENERIC=1): one wave / workgroup per coefficient
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"
#include <type_traits>

#ifndef FL_PASSES
#define FL_PASSES 2
#endif
#ifndef FL_WAVES
#define FL_WAVES 4
#endif
#ifndef FL_PASSES_FINE
#define FL_PASSES_FINE 1
#endif

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

#include "fwd_poly.h"
#include "fwd_lds.h"
#include "fwd_tl.h"
#include "fwd_gemm.h"
#include "spectral.h"

// TSPWS_ENGINE pins the forward engine of many-trace batches (and of few rows in columns): fir / spectral; unset or "auto": the size rules
// below.  Parsed ONCE per process, here, for every site that asks; any other value is refused by tspws_hip_plan_create (TSPWS_E_ARG) --
// two ranks that spell it differently must not end up with different kernels or collective sequences.
int tspws_engine_pin()
{
	static int eng = -2; // 0 auto, 1 fir, 2 spectral, -1 not a value
	if (eng == -2) {
		const char *e = getenv("TSPWS_ENGINE");
		eng = (!e || !*e || !strcmp(e, "auto")) ? 0 : !strcmp(e, "fir") ? 1 : !strcmp(e, "spectral") ? 2 : -1;
	}
	return eng;
}

// TSPWS_TL_MIN=n forces the many-trace path for batches of >= n traces (tests; read at every call), unset: the rule below
static long tl_min_env()
{
	const char *e = sweep_env("TSPWS_TL_MIN");
	return e ? std::max(1, atoi(e)) : -1;
}

#ifndef FWD_LDS_MIN_NG
#define FWD_LDS_MIN_NG 8 /* output groups (of 8) a scale needs for the LDS-staged kernel */
#endif
#ifndef TL_MINNS0
#define TL_MINNS0 33 /* many-trace batches: octaves with at least this many outputs on the trace-lane kernel */
#endif
#ifndef FWD_STEPS_DEF
#define FWD_STEPS_DEF 32 /* tap steps per wave of the direct kernel, few traces (the K partial stacks of a two-stage call): its waves are chains of dependent
                            loads and there are few of them -- 96 -> 32 steps: 100 x 16384 two-stage 0.108 -> 0.086 ms, 499 x 16501 0.121 -> 0.107, 64 x 65536
                            0.173 -> 0.155, the north star unchanged (tools/experiments/fwd_steps.sh) */
#endif
#ifndef FWD_STEPS_TL
#define FWD_STEPS_TL 96  /* ... beside k_fwd_tl on many traces: there are thousands of waves, longer ones amortise their set-up (cfg2 3.08 vs 3.14 ms at 32) */
#endif
static int build_tl_forward(tspws_hip_plan *p, unsigned FWD_STEPS, unsigned MINNS, unsigned TLSTEPS, TlTable &T, unsigned spec_first = ~0u, unsigned spec_end = ~0u,
                            unsigned nblk_hint = 0);

// Work decomposition of the forward kernels.  Scales with >= 8 output groups and (D >= 64 or D a power of two) run on the
// LDS-staged kernel; the rest (very coarse scales, odd small decimations) on the direct kernel, which aims at ~FWD_STEPS
// tap steps per wave (more splits = shorter dependent load chains).
int tspws_build_forward(tspws_hip_plan *p)
{
	const unsigned S = p->S;
	unsigned FWD_STEPS = FWD_STEPS_DEF;
	if (const char *e = sweep_env("TSPWS_FWD_STEPS")) FWD_STEPS = (unsigned)std::max(8, atoi(e)); // sweeps
	const unsigned R = 8, FL_SLOTS_HOST = FL_WAVES * FL_PASSES;
	unsigned woff = 0, boff = 0;
	unsigned long long poff = 0;
	for (unsigned s = 0; s < S; s++) {
		ScaleDesc &d = p->sc[s];
		d.Q = (d.L + d.D - 1) / d.D;
		unsigned dl = 1, lg = 0;
		while (dl < d.D && dl < 64) { dl <<= 1; lg++; }
		d.DL = dl; d.logDL = lg;
		d.MC = d.D > 64 ? (d.D + 63) / 64 : 1;
		const unsigned NG = (d.Ns + R - 1) / R, GW = 64 / d.DL;
		const bool pow2 = (d.D & (d.D - 1)) == 0;
		d.use_lds = (NG >= FWD_LDS_MIN_NG && (d.D >= 64 || pow2)) ? 1u : 0u;
		d.ngw = (NG + GW - 1) / GW;
		// direct kernel, 64 phase lanes, at least 16 outputs: 16 outputs per thread (half the operand bytes per FMA)
		d.r16 = (!d.use_lds && d.DL == 64 && d.Ns >= 16) ? 1u : 0u;
		if (d.r16) d.ngw = (d.Ns + 15) / 16;
		unsigned cps;
		if (d.use_lds) cps = 1; // one 64-phase chunk per workgroup: its taps stay resident in LDS
		else cps = std::max(1u, (FWD_STEPS + d.Q / 2) / std::max(1u, d.Q));
		d.cps = std::min(cps, d.MC);
		d.nsplit = (d.MC + d.cps - 1) / d.cps;
		d.wave_off = woff; d.lds_off = boff; d.part_off = poff;
		const unsigned slots = (d.D <= 4) ? FL_WAVES * FL_PASSES_FINE : FL_SLOTS_HOST; // as dispatched in k_fwd_lds
		d.lds_bps = (NG + slots * GW - 1) / (slots * GW);
		if (d.use_lds) boff += d.lds_bps * d.nsplit; else woff += d.ngw * d.nsplit;
		poff += (unsigned long long)d.nsplit * d.Ns;
		d.fuse_ok = (d.use_lds && d.nsplit == 1) ? 1u : 0u;
		p->n_fusable += d.fuse_ok;
	}
	p->fwd_waves = woff; p->lds_blocks = boff; p->npart = poff;
	// Tap rows resident in LDS (fwd_lds.h: QT).  A scale with Q <= QT taps per phase keeps its taps in LDS for the whole trace
	// slice and prefetches the next trace's window while it computes; one with more re-stages taps and window per tile and
	// trace.  The default Morlet frames have Q <= 18; the Mexican hat's second voice of every octave has Q = 29 or 30: frames
	// with such scales (24 < Q <= 32) take the QT = 32 instantiation (80 KB of LDS, still two workgroups per CU).
	p->lds_qt = 24;
	for (unsigned s = 0; s < S; s++) if (p->sc[s].use_lds && p->sc[s].Q > 24 && p->sc[s].Q <= 32) p->lds_qt = 32;
	if (const char *e = sweep_env("TSPWS_FWD_QT")) { const int v = atoi(e); if (v == 24 || v == 32) p->lds_qt = (unsigned)v; } // (sweeps)
	// two many-trace decompositions (see stacks_tl for the choice): sweeps on 128 .. 2048 traces x 8192 .. 32768 samples, Morlet
	// and Mexican hat (round 3): batches of >= 12 trace blocks are fastest with the octaves of >= 33 outputs on the trace-lane
	// kernel, smaller batches (and frames with two voices per octave) with >= 129 -- the trace-lane kernel has tl.wgs x blocks
	// workgroups, the direct kernel splits the taps
	// table 0 (>= 12 trace blocks): octaves with >= 33 outputs on the trace-lane kernel, ~96 residue steps per workgroup; table 1 (fewer
	// blocks: the launch has few workgroups): >= 129 outputs, ~48 steps -- tools/experiments/tl_pick.sh and the 2-D sweep of round 4:
	// 499 x 16501 0.893 -> 0.865 ms, 256 x 32768 0.797 -> 0.780, 512 x 16384 0.885 -> 0.806 against (257, 96); cfg2 (table 0) 3.06 vs 3.16 at 48
	if (int rc = build_tl_forward(p, FWD_STEPS_TL, TL_MINNS0, 96, p->tl[0])) return rc;
	unsigned minns1 = 129, tlsteps1 = 48;
	if (const char *e = sweep_env("TSPWS_TLSTEPS")) tlsteps1 = (unsigned)std::max(8, atoi(e)); // sweeps
	if (const char *e = sweep_env("TSPWS_TL_MINNS1")) minns1 = (unsigned)std::max(9, atoi(e)); // sweeps
	return build_tl_forward(p, FWD_STEPS_TL, minns1, tlsteps1, p->tl[1]);
}

// Decomposition for many-trace batches (fwd_tl.h): octaves (runs of scales with the same D and Ns) with at least MINNS
// outputs become trace-lane work items (voice subsets of <= TL_VMAX voices), the rest stays on the direct kernel; T.sc is
// the scale table of that decomposition (partial layout, fused flags, accumulate geometry).
// spec_first, spec_end: the scales [spec_first, spec_end) are left to the spectral engine (spectral.hip) -- no work items here, and the scale
// table marks them as stacked by their producer (one plane pair per 64-trace block, like the trace-lane kernel's fused scales); scales from
// spec_end on (filters too long for the transform window of a frame whose N is not a power of two) are the columns of the matrix-pipe
// kernel (fwd_gemm.h): runs of samples per wave sized for ~2048 waves with nblk_hint trace blocks.
static int build_tl_forward(tspws_hip_plan *p, unsigned FWD_STEPS, unsigned MINNS, unsigned TLSTEPS, TlTable &T, unsigned spec_first, unsigned spec_end,
                            unsigned nblk_hint)
{
	T.minns = MINNS;
	T.sc = p->sc;
	std::vector<TLItem> items;
	std::vector<char> is_tl(p->S, 0);
	unsigned wg = 0;
	const unsigned S_fir = std::min(p->S, spec_first);
	for (unsigned s = 0; s < S_fir;) {
		unsigned e = s + 1;
		while (e < S_fir && p->sc[e].D == p->sc[s].D && p->sc[e].Ns == p->sc[s].Ns) e++;
		const unsigned D = p->sc[s].D, Ns = p->sc[s].Ns, nv = e - s;
		bool ok = Ns >= MINNS;
		std::vector<unsigned> a(nv), b(nv), qr(nv);
		for (unsigned v = 0; v < nv && ok; v++) {
			const ScaleDesc &d = p->sc[s + v];
			a[v] = (unsigned)d.c / D; b[v] = (unsigned)d.c % D;
			const long long num = (long long)d.L - 1 - (long long)b[v];
			const long long fl = num >= 0 ? num / (long long)D : -1;
			qr[v] = ((unsigned)(fl + 2) + 3) & ~3u; // tap rows of a residue image: k_fwd_tl walks them in blocks of 4
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
					T.lds = std::max(T.lds, 2 * ((size_t)o.XR * 64 * sizeof(double) + (size_t)o.trows * sizeof(double2)));
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
	const bool has_gemm = spec_first < p->S && spec_end < p->S;
	static const bool gemm_off = sweep_env("TSPWS_GEMM") && !strcmp(sweep_env("TSPWS_GEMM"), "0"); // sweeps: those scales on the direct kernel
	if (has_gemm && !gemm_off) {
		unsigned ncol = 0;
		for (unsigned s = spec_end; s < p->S; s++) ncol += p->sc[s].Ns;
		T.gcoltiles = (ncol + 15) / 16;
		const unsigned want = (2048 + T.gcoltiles * std::max(1u, nblk_hint) - 1) / (T.gcoltiles * std::max(1u, nblk_hint));
		T.gKS = std::max(1u, std::min(want, std::max(1u, p->N / 128u)));
		if (const char *e = sweep_env("TSPWS_GEMM_KS")) T.gKS = (unsigned)std::max(1, atoi(e)); // sweeps
		T.gKC = (((p->N + T.gKS - 1) / T.gKS) + 3) & ~3u;
		T.gKS = (p->N + T.gKC - 1) / T.gKC; // (no empty runs)
	}
	for (unsigned s = 0; s < p->S; s++) {
		ScaleDesc &d = T.sc[s];
		d.use_lds = 0; d.lds_off = 0;
		const bool is_spec = s >= spec_first && s < spec_end, is_gemm = T.gcoltiles && s >= spec_end;
		if (is_spec) { d.nsplit = 1; d.cps = 1; d.fuse_ok = 1; }
		else if (is_gemm) { d.nsplit = 1; d.cps = 1; d.fuse_ok = 0; } // (k_gemm_reduce leaves one partial per trace and coefficient)
		else if (is_tl[s]) {
			d.nsplit = (d.D + TL_PMAX - 1) / TL_PMAX; d.cps = 1;
			d.fuse_ok = d.nsplit == 1 ? 1u : 0u;
		} else { // direct kernel, as in the few-trace table
			const unsigned cps = std::max(1u, (FWD_STEPS + d.Q / 2) / std::max(1u, d.Q));
			d.cps = std::min(cps, d.MC);
			d.nsplit = (d.MC + d.cps - 1) / d.cps;
			d.fuse_ok = 0;
		}
		d.wave_off = woff;
		if (!is_tl[s] && !is_spec && !is_gemm) woff += d.ngw * d.nsplit;
		d.part_off = poff;
		if (!((is_tl[s] || is_spec) && d.fuse_ok)) poff += (unsigned long long)d.nsplit * d.Ns; // fused scales never write partials
		d.acc2_off = ablk;
		// 4 coefficients per block, a wave each, for every scale whose per-trace partials are added by k_accumulate_parts (`many`): its lanes
		// take every 64th trace.  (One thread per coefficient walked all traces of the batch in dependent round trips: 80 us at 499 x 16501.)
		ablk += (d.nsplit > 1 || !d.fuse_ok) ? (d.Ns + 3) / 4 : (d.Ns + 255) / 256;
	}
	for (TLItem &o : items) for (unsigned i = 0; i < o.nv; i++) o.part_off[i] = T.sc[o.sc[i]].part_off;
	T.n = (unsigned)items.size(); T.wgs = wg; T.waves = woff; T.acc2_blocks = ablk; T.npart = poff;
	if (!T.n && spec_first >= p->S) return 0;
	HIP_TRY(hipMalloc(&T.d_sc, p->S * sizeof(ScaleDesc)));
	HIP_TRY(hipMemcpy(T.d_sc, T.sc.data(), p->S * sizeof(ScaleDesc), hipMemcpyHostToDevice));
	if (T.gcoltiles) {
		std::vector<GemmCol> gc((size_t)T.gcoltiles * 16);
		memset(gc.data(), 0, gc.size() * sizeof(GemmCol));
		size_t i = 0;
		for (unsigned s = spec_end; s < p->S; s++) {
			const ScaleDesc &d = T.sc[s];
			for (unsigned k = 0; k < d.Ns; k++, i++) {
				GemmCol &g = gc[i];
				long long o = ((long long)d.c - (long long)k * (long long)d.D) % (long long)p->N;
				if (o < 0) o += p->N;
				g.L = d.L; g.o = (unsigned)o; g.Ns = d.Ns; g.tap_off = d.tap_off; g.dst = d.part_off + k;
			}
		}
		HIP_TRY(hipMalloc(&T.d_gcols, gc.size() * sizeof(GemmCol)));
		HIP_TRY(hipMemcpy(T.d_gcols, gc.data(), gc.size() * sizeof(GemmCol), hipMemcpyHostToDevice));
	}
	if (!T.n) return 0;
	HIP_TRY(hipMalloc(&T.d_items, items.size() * sizeof(TLItem)));
	HIP_TRY(hipMemcpy(T.d_items, items.data(), items.size() * sizeof(TLItem), hipMemcpyHostToDevice));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_tl<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	HIP_TRY(hipFuncSetAttribute((const void *)k_fwd_tl<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	return 0;
}

int tspws_build_tl_spectral(tspws_hip_plan *p, unsigned s_first, unsigned s_end, unsigned nblk_hint, TlTable &T)
{
	// residue steps per workgroup of the finer octaves: SHORT workgroups (12; the FIR-only decomposition takes 96) -- the spectral chain runs
	// beside this kernel on the side stream and gets the slots they free (cfg2 1.80 -> 1.68 ms, 1024 x 32768 default frame 1.56 -> 1.42; 8: same,
	// 32: 1.77 / 1.48, 192: 2.11 / 1.71; outputs bit-identical)
	unsigned steps = 12;
	if (const char *e = sweep_env("TSPWS_SPEC_TLSTEPS")) steps = (unsigned)std::max(1, atoi(e)); // sweeps: residue steps per workgroup of the finer octaves
	return build_tl_forward(p, FWD_STEPS_TL, TL_MINNS0, steps, T, s_first, s_end, nblk_hint);
}

// ------------------------------------------------------------------------------------------
// few-trace forward transform into split partials (+ fused stacks)
// ------------------------------------------------------------------------------------------
// launch ranges of a scale range: the launch lists are ordered by scale, so a range is one run of workgroups / waves / blocks
struct LaunchRange { unsigned lds0, lds1, wav0, wav1, acc0, acc1; };
static LaunchRange launch_range(const tspws_hip_plan *p, ScaleRange rg)
{
	LaunchRange r;
	r.lds0 = rg.on() ? p->sc[rg.s0].lds_off : 0u;
	r.lds1 = rg.on() && rg.s1 < p->S ? p->sc[rg.s1].lds_off : p->lds_blocks;
	r.wav0 = rg.on() ? p->sc[rg.s0].wave_off : 0u;
	r.wav1 = rg.on() && rg.s1 < p->S ? p->sc[rg.s1].wave_off : p->fwd_waves;
	r.acc0 = rg.on() ? p->sc[rg.s0].acc2_off : 0u;
	r.acc1 = rg.on() && rg.s1 < p->S ? p->sc[rg.s1].acc2_off : p->acc2_blocks;
	return r;
}

// The side stream of the direct kernel.  TSPWS_SIDE_PRIO=1 / -1 creates it with the highest / lowest stream priority (sweeps: does
// the coarse-scale kernel finish inside the main kernel's time when it is dispatched first?); default: plain.
static hipError_t tspws_side_stream(tspws_hip_plan *p)
{
	const char *e = sweep_env("TSPWS_SIDE_PRIO");
	if (!e || !atoi(e)) return hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
	int lo = 0, hi = 0;
	hipError_t rc = hipDeviceGetStreamPriorityRange(&lo, &hi);
	if (rc != hipSuccess) return rc;
	return hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, atoi(e) > 0 ? hi : lo);
}

// Two independent kernels transform disjoint sets of scales: the LDS kernel (FP64-bound) and the direct kernel (coarse
// scales, latency-bound).  They run side by side: a side stream is forked from and joined back into the caller's stream.
template <typename TIn>
static int forward_parts(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double2 *d_part, hipStream_t st, FuseOut *fz, ScaleRange rg)
{
	if (fz) { fz->applied = false; fz->spec_first = p->S; }
	bool spec_join = false;
	hipStream_t fir = st; // stream of the FIR kernels (the caller's, unless a spectral chain of few rows keeps that one: below)
	// Few rows in columns (the K partial stacks of every jackknife replica, resample.hip) whose weighted sets the forward launch completes
	// itself: from FEW_SPEC_MIN rows on, the octaves with D >= 16 go through the spectral engine (lanes = rows; per-column stacks by
	// k_spec_stack_rows) and the FIR kernels below keep the finer ones.  cfg4 (110 rows of 131072 samples, Mexican hat): forward stage
	// 1.22 -> 0.87 ms (docs/history/round-5.md).  TSPWS_ENGINE=fir switches it off.
	if constexpr (std::is_same<TIn, double>::value) {
		if (fz && fz->allow_spec && !rg.on() && fz->fin.OUT && fz->fin.nprev == 0 && ntr <= 512 && fz->tps) {
			static int min_rows = -1;
			static unsigned nsmax_env = 0;
			if (min_rows < 0) {
				const char *e = sweep_env("TSPWS_FEW_SPEC_MIN"); min_rows = e ? std::max(1, atoi(e)) : 64;
				if (const char *m = sweep_env("TSPWS_FEW_NSMAX")) nsmax_env = (unsigned)std::max(2, atoi(m));
			}
			if (ntr >= (size_t)min_rows && tspws_engine_pin() != 1 && !tspws_generic_forward()) {
				// (octaves with at most N / 16 outputs, D >= 16: with D = 8 in the set as well the chain -- two trace blocks, latency-bound -- ends 0.35 ms
				// after the FIR kernels; cfg4 2.025 -> 1.979 ms, N / 32: 2.014, N / 64: 2.069; tools/experiments/r6_cfg4_split.sh)
				const unsigned sf = tspws_spectral_first_scale(p, nsmax_env ? nsmax_env : std::max(512u, (p->N + 15u) / 16u));
				// (a frame whose coarsest filters do not fit the transform window keeps its rows on the FIR kernels: they take ONE run of scales)
				if (sf < p->S && tspws_spectral_end_scale(p) == p->S) {
					SpecDecomp *dc = nullptr;
					int rc;
					if ((rc = tspws_spectral_decomp(p, sf, (unsigned)((ntr + 63) / 64), &dc, true))) return rc;
					// the spectral chain (transforms through HBM: bandwidth-bound) on its own stream beside the FIR kernels of the finer octaves
					// (FP64-bound), joined before this function returns.  The FIR kernels start BEHIND the chain's transposition: k_fwd_lds takes
					// all of a CU's LDS (two 80-KB workgroups), and a transposition that has to wait for its 33 KB starves (mid-round, with the
					// 0.16-ms transposition and the FIR kernels launched first, the pair took 1.70 ms side by side against 1.05 one after the
					// other); the chain's other kernels need no LDS to speak of and fill the register space the FIR kernels leave.  cfg4 with
					// the end-of-round kernels on one box: 2.08-2.11 ms one after the other, 2.05 side by side with the FIR kernels first,
					// 2.03-2.04 in this order (docs/history/experiments/r5_rows_parallel.sh; TSPWS_SPEC_PARALLEL=0, sweeps build: one after the other).
					static const bool serial = sweep_env("TSPWS_SPEC_PARALLEL") && !strcmp(sweep_env("TSPWS_SPEC_PARALLEL"), "0");
					hipEvent_t behind_tr = nullptr;
					if (!serial && sf > 0) {
						// The CHAIN stays on the caller's stream (it is the longer branch, and every hand-over to another stream costs ~20 us before
						// the first kernel there starts); the FIR kernels go to a second stream of PLAIN priority.  (A least urgent FIR stream gains
						// another 0.03 ms at cfg4 when the plan is alone in the process -- the chain's short workgroups take the slots the long FIR
						// workgroups free -- but with the streams of a SECOND plan alive in the process any non-default priority costs 0.4-0.5 ms:
						// 2.00 / 2.46 ms least urgent, 2.10 / 2.56 most urgent, 2.03 / 2.03 plain, 2.10 / 2.10 one after the other;
						// docs/history/experiments/r5_cfg4_robust.sh, 150 calls each.)  TSPWS_XS_PRIO (sweeps): -1 least, 1 most urgent
						const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
						if (!p->xs) {
							int plo = 0, phi = 0;
							HIP_TRY(hipDeviceGetStreamPriorityRange(&plo, &phi)); // (plo = least urgent)
							const char *e = sweep_env("TSPWS_XS_PRIO");
							HIP_TRY(hipStreamCreateWithPriority(&p->xs, hipStreamNonBlocking, e ? (atoi(e) > 0 ? phi : atoi(e) < 0 ? plo : 0) : 0));
						}
						if (!p->ev_xs1) HIP_TRY(hipEventCreateWithFlags(&p->ev_xs1, evf));
						if (!p->ev_xs2) HIP_TRY(hipEventCreateWithFlags(&p->ev_xs2, evf));
						behind_tr = fz->ev_mid ? fz->ev_mid : p->ev_xs2;
					}
					if ((rc = tspws_spectral_rows_f64(p, dc, (const double *)d_x, ld, (unsigned)ntr, fz->tps, *fz, st, behind_tr))) return rc;
					fz->spec_first = sf;
					if (behind_tr) { // side by side: the FIR kernels (and the caller's readers of the rows: ev_mid) wait for the transposition only
						HIP_TRY(hipStreamWaitEvent(p->xs, behind_tr, 0));
						fz->mid_recorded = behind_tr == fz->ev_mid;
						fir = p->xs;
					} else if (fz->ev_mid) { HIP_TRY(hipEventRecord(fz->ev_mid, st)); fz->mid_recorded = true; } // one after the other: behind the whole chain
					if (sf == 0) { fz->applied = true; return 0; } // (every scale went that way)
					rg.s0 = 0; rg.s1 = sf;
					spec_join = fir != st;
				}
			}
		}
	}
	const LaunchRange lr = launch_range(p, rg);
	const bool has_lds = lr.lds1 > lr.lds0, has_poly = lr.wav1 > lr.wav0;
	hipStream_t sp = fir; // stream of the direct kernel
#if FL_TIMING || FL_ABLATE
	static const bool serial = sweep_env("TSPWS_FWD_SERIAL") != nullptr; // debug builds: the two kernels one after the other
#else
	constexpr bool serial = false;
#endif
	if (has_lds && has_poly && !serial) {
		const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence; // device-local ordering only
		if (!p->side) HIP_TRY(tspws_side_stream(p));
		if (!p->ev_fork) HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, evf));
		if (!p->ev_join) HIP_TRY(hipEventCreateWithFlags(&p->ev_join, evf));
		// fork: the side stream waits for the producer of d_x -- its launch carried the event (plan->le.ready) or a record here
		hipEvent_t ready = p->le.ready;
		if (!ready || fir != st) { ready = p->ev_fork; HIP_TRY(hipEventRecord(ready, fir)); }
		HIP_TRY(hipStreamWaitEvent(p->side, ready, 0));
		sp = p->side;
	}
	p->le.ready = nullptr; // (valid for the first transforms after the producer only)
	// the direct kernel first: its few hundred long, latency-bound workgroups (no LDS, 116 VGPRs) get their slots and the
	// LDS kernel's workgroups fill in beside them
	if (has_poly) {
		const unsigned nb = (lr.wav1 - lr.wav0 + 3) / 4;
		if (ntr == 1) {
			hipLaunchKernelGGL((k_fwd_poly<TIn, 1>), dim3(nb, 1), dim3(256), 0, sp, d_x, ld, 1u, p->N, p->d_sc, p->S, p->d_w, d_part,
			                   p->npart, lr.wav1, lr.wav0);
		} else {
			for (size_t t0 = 0; t0 < ntr; t0 += 2 * 32768) {
				const unsigned nt = (unsigned)std::min<size_t>(ntr - t0, 2 * 32768);
				hipLaunchKernelGGL((k_fwd_poly<TIn, 2>), dim3(nb, (nt + 1) / 2), dim3(256), 0, sp, d_x + t0 * ld, ld, nt, p->N, p->d_sc,
				                   p->S, p->d_w, d_part + t0 * p->npart, p->npart, lr.wav1, lr.wav0);
			}
		}
	}
	if (has_lds) {
		const bool fuse = fz && fz->accST && p->n_fusable;
		// traces per workgroup: enough slices to fill the GPU (>= ~2048 workgroups), at most 32 traces per slice
		unsigned tps = (unsigned)std::min<size_t>(ntr, 32);
		while (tps > 1 && (size_t)p->lds_blocks * ((ntr + tps - 1) / tps) < 2048) tps = (tps + 1) / 2;
		if (fuse) tps = fz->tps; // the caller sized the slice planes
		const size_t per_launch = (size_t)tps * 65535;
		for (size_t t0 = 0; t0 < ntr; t0 += per_launch) {
			const unsigned nt = (unsigned)std::min<size_t>(ntr - t0, per_launch);
			const dim3 grid(lr.lds1 - lr.lds0, (nt + tps - 1) / tps);
			double2 *aS = fuse ? fz->accST + (t0 / tps) * fz->stride : nullptr, *aP = fuse ? fz->accPS + (t0 / tps) * fz->stride : nullptr;
			const size_t astr = fuse ? fz->stride : 0;
			FuseFinal ff;
			if (fuse && fz->fin.OUT && t0 == 0 && ntr <= per_launch) ff = fz->fin; // (slice indices of the final step are the launch's own: one launch)
#define FL_LAUNCH(F, QT) hipLaunchKernelGGL((k_fwd_lds<TIn, F, QT>), grid, dim3(FL_NT), FL_LDS_BYTES_(QT), fir, d_x + t0 * ld, ld, nt, tps, p->N, p->d_sc, p->S, \
			                            p->d_w, d_part + t0 * p->npart, p->npart, aS, aP, astr, lr.lds0, ff)
			if (p->lds_qt == 32) { if (fuse) FL_LAUNCH(true, 32); else FL_LAUNCH(false, 32); }
			else { if (fuse) FL_LAUNCH(true, 24); else FL_LAUNCH(false, 24); }
#undef FL_LAUNCH
		}
		if (fuse) fz->applied = true;
	}
	if (sp != fir) {
		HIP_TRY(hipEventRecord(p->ev_join, sp));
		HIP_TRY(hipStreamWaitEvent(fir, p->ev_join, 0));
	}
	if (spec_join) {
		if (fz && fz->defer_fir_join) fz->fir_stream = p->xs; // (the caller enqueues more behind the FIR kernels and joins: tspws_join_fir_stream)
		else {
			HIP_TRY(hipEventRecord(p->ev_xs1, p->xs));
			HIP_TRY(hipStreamWaitEvent(st, p->ev_xs1, 0));
		}
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

int tspws_join_fir_stream(tspws_hip_plan *p, hipStream_t fir, hipStream_t st)
{
	HIP_TRY(hipEventRecord(p->ev_xs1, fir));
	HIP_TRY(hipStreamWaitEvent(st, p->ev_xs1, 0));
	return 0;
}

int tspws_forward_parts_f32(tspws_hip_plan *p, const float *d_x, size_t ntr, size_t ld, double2 *d_part, hipStream_t st, FuseOut *fz, ScaleRange rg)
{
	return forward_parts<float>(p, d_x, ntr, ld, d_part, st, fz, rg);
}
int tspws_forward_parts_f64(tspws_hip_plan *p, const double *d_x, size_t ntr, size_t ld, double2 *d_part, hipStream_t st, FuseOut *fz, ScaleRange rg)
{
	return forward_parts<double>(p, d_x, ntr, ld, d_part, st, fz, rg);
}

#if FL_ABLATE
// debug build only: only the log2(D) classes of k_fwd_lds named by the hex mask TSPWS_FWD_CLASSES run (results wrong: timing ablation)
extern "C" int tspws_hip_fwd_ablate(void)
{
	if (const char *e = sweep_env("TSPWS_FWD_CLASSES")) {
		const unsigned m = (unsigned)strtoul(e, nullptr, 16);
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(fl_class_mask), &m, sizeof m));
	}
	return 0;
}
#endif

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
size_t tspws_part_budget_bytes()
{
	static size_t v = 0;
	if (!v) { const char *e = getenv("TSPWS_PART_MB"); v = (size_t)(e ? std::max(16, atoi(e)) : 2048) << 20; }
	return v;
}

bool tspws_generic_forward()
{
	static int v = -1;
	if (v < 0) { const char *e = sweep_env("TSPWS_FWD_GENERIC"); v = (e && *e == '1') ? 1 : 0; }
	return v == 1;
}

bool tspws_fused_forward(const tspws_hip_plan *p) { return p->n_fusable != 0; }

// Does a batch of ntr traces go to the trace-lane kernel?  Round-3 sweeps (tools/cfg_bench.py, 64 .. 4096 traces x 8192 .. 131072
// samples, both paths forced): below ~7 M samples per batch, or with fewer than three trace blocks, the few-trace kernels win
// (64 x 32768: 0.37 vs 0.85 ms; 192 x 32768: 0.88 vs 1.01; 256 x 16501: 0.86 vs 1.01), above it the trace-lane kernel
// (320 x 32768: 1.24 vs 1.40; 1024 x 32768: 3.03 vs 4.19; 256 x 131072: 3.60 vs 4.10); frames with two voices per octave
// (Mexican hat) stay on the few-trace kernels at every size measured (1024 x 32768: 3.66 vs 4.63; 4096 x 16384: 8.5 vs 9.9) --
// a trace-lane work item shares its staged rows among the voices of an octave.
// Engine of a many-trace batch: the first scale of the spectral set (S: every scale on the FIR kernels).  TSPWS_ENGINE=fir / spectral
// pins it (spectral: every batch that has such a set takes the many-trace path, whatever its size).  Default: batches of >= 64 traces and
// >= 1 M samples (or >= 256 traces) send the octaves with D >= 32 (at most max(512, N / 32) outputs; two-voice frames: D >= 16)
// through the spectrum -- sweeps over 128 .. 4096 traces x 8192 .. 131072 samples (tools/experiments/
// r5_spec12.sh, default Morlet): 1024 x 32768 2.73 -> 1.81 ms, 2048 x 16384 2.83 -> 1.74, 512 x 65536 3.05 -> 2.02, 1024 x 131072 10.96 ->
// 7.63 (N_s <= 4096), 4096 x 8192 2.38 -> 1.65; below that size the FIR kernels win (256 x 8192: 0.31 vs 0.38 ms); outputs
// bit-identical at every size.  TSPWS_SPEC_NSMAX overrides the octave bound.
static bool many_trace_size(const tspws_hip_plan *p, size_t ntr) { return ntr >= 128 && (double)ntr * (double)p->N >= 7.0 * 1048576.0; }
// ... and the size from which the spectral engine (with the trace-lane kernel for the finer octaves) beats the few-trace kernels: a full
// block of 64 traces and >= 1 M samples, or >= 256 traces of any frame that has a spectral set (docs/history/experiments/r5_thresh.sh, default
// Morlet, FIR / spectral in ms: 64 x 32768 0.284 / 0.233, 64 x 16384 0.170 / 0.154, 256 x 2048 0.153 / 0.115, 512 x 1024 0.212 / 0.112,
// 256 x 8192 0.293 / 0.170, 1024 x 4096 0.565 / 0.246; below: 48 x 32768 0.226 / 0.244, 64 x 8192 0.106 / 0.128, 128 x 4096 0.113 / 0.119,
// 32 x 131072 0.562 / 0.693 -- fewer than 64 traces leave lanes of the trace blocks idle)
static bool spectral_size(const tspws_hip_plan *p, size_t ntr) { return ntr >= 64 && ((double)ntr * (double)p->N >= 1048576.0 || ntr >= 256); }
unsigned tspws_spectral_choice(const tspws_hip_plan *p, size_t ntr)
{
	const int eng = tspws_engine_pin(); // 0 auto, 1 fir, 2 spectral
	static unsigned nsmax_env = ~0u;
	if (nsmax_env == ~0u) { const char *m = sweep_env("TSPWS_SPEC_NSMAX"); nsmax_env = m ? (unsigned)std::max(2, atoi(m)) : 0u; }
	if (eng == 1 || tspws_generic_forward() || !ntr) return p->S;
	if (eng == 0 && !spectral_size(p, ntr)) return p->S;
	// (two voices per octave -- the Mexican hat --: the trace-lane kernel is at its weakest there (it shares its staged rows among the voices of
	// an octave), so one octave more goes through the spectrum: 1024 x 32768 Mexican hat 2.47 ms on the few-trace kernels, 1.56 / 1.52 / 1.54 ms
	// with N_s <= 1024 / 2048 / 4096; 4096 x 8192 2.72 -> 1.51-1.54; 512 x 65536 2.52 -> 1.73 with N_s <= 4096)
	const unsigned dmin = p->V > 2 ? 32u : 16u; // (N_s = ceil(N / D): the bound that admits D >= dmin for every N)
	return tspws_spectral_first_scale(p, nsmax_env ? nsmax_env : std::max(512u, (p->N + dmin - 1) / dmin));
}

static int spectral_transpose_t(tspws_hip_plan *p, const float *d_x, size_t ld, unsigned ntr, float *xT, unsigned TP, hipStream_t st) { return tspws_spectral_transpose_f32(p, d_x, ld, ntr, xT, TP, st); }
static int spectral_transpose_t(tspws_hip_plan *p, const double *d_x, size_t ld, unsigned ntr, double *xT, unsigned TP, hipStream_t st) { return tspws_spectral_transpose_f64(p, d_x, ld, ntr, xT, TP, st); }
static int spectral_run_t(tspws_hip_plan *p, SpecDecomp *dc, const float *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st)
{ return tspws_spectral_run_f32(p, dc, xT, TP, ntr, ST, PS, stride, Y, st); }
static int spectral_run_t(tspws_hip_plan *p, SpecDecomp *dc, const double *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st)
{ return tspws_spectral_run_f64(p, dc, xT, TP, ntr, ST, PS, stride, Y, st); }

bool tspws_many_trace_path(const tspws_hip_plan *p, size_t ntr)
{
	if (tspws_generic_forward()) return false;
	if (tspws_spectral_choice(p, ntr) < p->S) return true; // (the spectral engine works on the transposed batch of this path)
	if (!(p->tl[0].n || p->tl[1].n)) return false;
	const long forced = tl_min_env();
	if (forced > 0) return ntr >= (size_t)forced;
	return p->V > 2 && many_trace_size(p, ntr); // (tools/experiments/tl_threshold.sh: 128 x 65536 0.94 vs 0.86 ms, 160 x 32768 0.63 vs 0.69, 1024 x 4096 0.58 vs 0.70, 499 x 16501 1.17 vs 0.99)
}

template <typename TIn>
static int forward_impl(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_Y, hipStream_t st)
{
	if (!p || !d_x || !d_Y) return fail(TSPWS_E_ARG, "forward: NULL");
	if (!ntr) return 0;
	HIP_TRY(hipSetDevice(p->device));
	if (tspws_generic_forward()) return forward_generic<TIn>(p, d_x, ntr, ld, d_Y, st);
	const size_t batch = std::min<size_t>(ntr, std::max<size_t>(2, ((tspws_part_budget_bytes()) / (p->npart * sizeof(double2))) & ~(size_t)1));
	void *v;
	int rc = scratch(p, SCR_PART, batch * p->npart * sizeof(double2), &v);
	if (rc) return rc;
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const size_t nb = std::min(batch, ntr - t0);
		if ((rc = forward_parts<TIn>(p, d_x + t0 * ld, nb, ld, (double2 *)v, st, nullptr, ScaleRange()))) return rc;
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

// Per-trace coefficients of the spectral set through the spectral engine (parity tests, the per-trace API): batches of <= 4096 traces
// are transposed like the many-trace stacks, the last inverse pass writes the traces' own coefficients.
template <typename TIn>
static int forward_spectral(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_Y, unsigned nsmax, hipStream_t st)
{
	if (!p || !d_x || !d_Y) return fail(TSPWS_E_ARG, "forward_spectral: NULL");
	if (!ntr) return 0;
	HIP_TRY(hipSetDevice(p->device));
	const unsigned sf = tspws_spectral_first_scale(p, nsmax);
	if (sf >= p->S) return fail(TSPWS_E_ARG, "forward_spectral: this frame has no spectral set");
	SpecDecomp *dc = nullptr;
	int rc;
	if ((rc = tspws_spectral_decomp(p, sf, (unsigned)std::min<size_t>((ntr + 63) / 64, 64), &dc))) return rc;
	const size_t batch = std::min<size_t>(4096, (ntr + 63) & ~(size_t)63);
	void *v;
	if ((rc = scratch(p, SCR_XT, (size_t)p->N * batch * sizeof(TIn), &v))) return rc;
	TIn *xT = (TIn *)v;
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const unsigned nb = (unsigned)std::min(batch, ntr - t0), nblk = (nb + 63) / 64, TP = nblk * 64;
		if ((rc = spectral_transpose_t(p, d_x + t0 * ld, ld, nb, xT, TP, st))) return rc;
		if ((rc = spectral_run_t(p, dc, (const TIn *)xT, TP, nb, nullptr, nullptr, 0, (double2 *)d_Y + t0 * p->ncoef, st))) return rc;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

extern "C" unsigned tspws_hip_spectral_first_scale(const tspws_hip_plan *p, unsigned nsmax) { return p ? tspws_spectral_first_scale(p, nsmax) : 0u; }
extern "C" unsigned tspws_hip_spectral_end_scale(const tspws_hip_plan *p) { return p ? tspws_spectral_end_scale(p) : 0u; }
extern "C" unsigned tspws_hip_spectral_transform_length(const tspws_hip_plan *p)
{
	unsigned NT = 0, e, c;
	if (p) tspws_spectral_geometry(p, &NT, &e, &c);
	return NT;
}
extern "C" unsigned tspws_hip_spectral_choice(const tspws_hip_plan *p, size_t ntr) { return p ? tspws_spectral_choice(p, ntr) : 0u; }
extern "C" int tspws_hip_forward_spectral_f64(tspws_hip_plan *p, const double *d_x, size_t ntr, size_t ld, double *d_Y, unsigned nsmax, void *s)
{
	return forward_spectral<double>(p, d_x, ntr, ld, d_Y, nsmax, S_(s));
}
extern "C" int tspws_hip_forward_spectral_f32(tspws_hip_plan *p, const float *d_x, size_t ntr, size_t ld, double *d_Y, unsigned nsmax, void *s)
{
	return forward_spectral<float>(p, d_x, ntr, ld, d_Y, nsmax, S_(s));
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

// launches k_accumulate_parts for `nb` transformed traces; fz = what the forward launch left behind (may be NULL / not applied)
void tspws_launch_accumulate(tspws_hip_plan *p, const double2 *part, unsigned nb, double2 *ST, double2 *PS, int zero_first, const FuseOut *fz,
                             unsigned nslices, hipStream_t st, unsigned nbatch, size_t y_part, size_t y_stack, const TlTable *tl, const WeightArgs *wa,
                             ScaleRange rg, const AccExtra *ex)
{
	const size_t trace_stride = ex ? ex->trace_stride : 0;
	const WeightArgs w0;
	const bool on = fz && fz->applied;
	const bool direct = on && fz->accST == ST; // the single slice went straight into ST / PS
	// tl: the many-trace decomposition's scale table (partial layout, fused flags, block geometry)
	const LaunchRange lr = launch_range(p, rg);
	const unsigned a0 = tl ? 0u : lr.acc0, a1 = tl ? tl->acc2_blocks : lr.acc1;
	if (a1 <= a0) return;
	auto kern = (wa && wa->OUTP) ? k_accumulate_parts<true> : k_accumulate_parts<false>;
	hipLaunchKernelGGL(kern, dim3(a1 - a0, nbatch), dim3(256), 0, st, part, trace_stride ? trace_stride : (tl ? tl->npart : p->npart),
	                   (const ScaleDesc *)(tl ? tl->d_sc : p->d_sc), p->S, nb, ST, PS, zero_first,
	                   (ex && ex->fused_done) ? 3 : on ? (direct ? 1 : 2) : 0, on ? (const double2 *)fz->accST : nullptr, on ? (const double2 *)fz->accPS : nullptr,
	                   on ? fz->stride : (size_t)0, nslices, y_part, y_stack, tl ? 1 : 0, wa ? *wa : w0, a0, ex ? ex->y_fz : (size_t)0,
	                   ex ? ex->rowmap : (const unsigned *)nullptr);
}

unsigned tspws_first_unfused_scale(const tspws_hip_plan *p)
{
	unsigned s = 0;
	while (s < p->S && p->sc[s].fuse_ok) s++;
	return s;
}

// slice length of the fused forward kernel for a batch of nb traces: whole batch when it is small (two-stage: the K
// partial stacks -> ONE slice that writes ST / PS directly), else 32 traces per slice
// Short frames have few workgroups per slice (N = 8192: 72; the chip holds 512 at a time): slices are halved until the launch has
// ~256 workgroups (TSPWS_FUSE_WGS), down to three traces per slice (TSPWS_FUSE_MINTPS) -- also the ten partial stacks of a
// two-stage call on a short frame: 64 x 8192 two-stage 0.073 -> 0.064 ms, 30 x 4096 0.070 -> 0.059, 64 x 2048 0.068 -> 0.055;
// single-stage 64 x 8192 0.152 -> 0.106, 30 x 4096 0.127 -> 0.065 (tools/experiments/fuse_slices.sh).  From N = 16501 up the K
// partial stacks are one slice again (208 workgroups and more).  The slices' plane pairs are added in slice order by the accumulation.
static unsigned fuse_tps(const tspws_hip_plan *p, size_t nb)
{
	static int target = -1;
	if (target < 0) { const char *e = sweep_env("TSPWS_FUSE_WGS"); target = e ? std::max(1, atoi(e)) : 256; }
	unsigned tps = (unsigned)std::min<size_t>(nb, 32);
	static int mintps = -1;
	if (mintps < 0) { const char *e = sweep_env("TSPWS_FUSE_MINTPS"); mintps = e ? std::max(1, atoi(e)) : 3; }
	while (tps > 1 && (tps + 1) / 2 >= (unsigned)mintps && (size_t)std::max(1u, p->lds_blocks) * ((nb + tps - 1) / tps) < (size_t)target) tps = (tps + 1) / 2;
	return std::max(1u, tps);
}

// Many traces (single-stage stacks): trace-lane kernel on the transposed batch (fwd_tl.h); the stacks of the fused scales
// come back as one plane pair per 64-trace block, the split / coarse scales as per-trace partials in the tl layout.
template <typename TIn>
static int stacks_tl(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                     const WeightArgs *wa, bool *weighted)
{
	int rc;
	void *v;
	// decomposition: many trace blocks and more than two voices per octave -> tl[0], else tl[1] (tspws_build_forward)
	unsigned pick = ((ntr + 63) / 64 >= 12 && p->V > 2) ? 0u : 1u;
	if (const char *e = sweep_env("TSPWS_TL_PICK")) pick = atoi(e) ? 1u : 0u; // sweeps: force a decomposition
	// ... or the spectral engine for the far-decimated octaves (spectral.hip) with its own decomposition of the rest
	SpecDecomp *dc = nullptr;
	const unsigned spec_first = tspws_spectral_choice(p, ntr);
	if (spec_first < p->S && (rc = tspws_spectral_decomp(p, spec_first, (unsigned)std::min<size_t>((ntr + 63) / 64, 64), &dc))) return rc;
	const TlTable &T = dc ? dc->T : p->tl[p->tl[pick].n ? pick : 1u - pick]; // (a short frame may leave one of them without trace-lane items)
	// traces per batch: transposed copy <= 1 GiB, at most 4096 (64 plane pairs), a multiple of 64
	size_t batch = std::min<size_t>(4096, std::max<size_t>(64, (((size_t)1 << 30) / ((size_t)p->N * sizeof(TIn))) & ~(size_t)63));
	if (T.npart) batch = std::min(batch, std::max<size_t>(64, (tspws_part_budget_bytes() / (T.npart * sizeof(double2))) & ~(size_t)63));
	if (const char *e = sweep_env("TSPWS_TL_BATCH")) batch = std::max<size_t>(64, (size_t)atoi(e) & ~(size_t)63); // tests: force several batches
	batch = std::min(batch, (ntr + 63) & ~(size_t)63);
	const size_t nblk_max = batch / 64;
	if ((rc = scratch(p, SCR_XT, (size_t)p->N * batch * sizeof(TIn), &v))) return rc;
	TIn *xT = (TIn *)v;
	if ((rc = scratch(p, SCR_FZ, nblk_max * 2 * p->ncoef * sizeof(double2), &v))) return rc;
	double2 *planes = (double2 *)v;
	double2 *part = nullptr, *gsum = nullptr;
	if (T.npart) { if ((rc = scratch(p, SCR_PART, batch * T.npart * sizeof(double2), &v))) return rc; part = (double2 *)v; }
	if (T.gcoltiles) { if ((rc = scratch(p, SCR_GEMM, (size_t)T.gKS * batch * T.gcoltiles * 16 * sizeof(double2), &v))) return rc; gsum = (double2 *)v; }
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const unsigned nb = (unsigned)std::min(batch, ntr - t0), nblk = (nb + 63) / 64, TP = nblk * 64;
		const TIn *xb = d_x + t0 * ld;
		// the direct kernel (scales with too few outputs for the trace-lane kernel: latency-bound, tl partial layout) on the side
		// stream: it reads the traces themselves, so it starts with the transposition and runs beside the trace-lane kernel
		hipStream_t sp = st;
		bool poly_join = false;
		// order of the matrix-pipe kernel for the coarsest scales (fwd_gemm.h): 1 = on the caller's stream BEHIND the trace-lane kernel (default: the
		// chain is the longer branch, and its transform passes are at their slowest with a second kernel beside them at the start -- 499 x 16501:
		// 0.631 ms; 500 x 20000: 0.731), 0 = in front of the trace-lane kernel (0.662 / 0.769), 2 = on a third stream beside both (0.644 / 0.777);
		// TSPWS_GEMM_ORDER, sweeps (tools/experiments/r6_gemm_order.sh)
		static const int gemm_order = sweep_env("TSPWS_GEMM_ORDER") ? atoi(sweep_env("TSPWS_GEMM_ORDER")) : 1;
		auto launch_gemm = [&](hipStream_t gs) {
			hipLaunchKernelGGL((k_fwd_gemm<TIn>), dim3((T.gcoltiles + 3) / 4, T.gKS, nblk), dim3(256), 0, gs, (const TIn *)xT, TP, nb, p->N,
			                   (const GemmCol *)T.d_gcols, T.gcoltiles, T.gKC, (const double2 *)p->d_w, gsum);
			const size_t nred = (size_t)nb * T.gcoltiles * 16;
			hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, gs, (const double2 *)gsum, TP, nb, T.gcoltiles * 16, T.gKS,
			                   (const GemmCol *)T.d_gcols, part, T.npart);
		};
		if ((T.waves || (T.gcoltiles && gemm_order == 2)) && dc) {
			// the direct kernel of a spectral decomposition: the scales whose filters are too long for the transform window (N not a power of two:
			// the clipped scales of the shipped example's frame, ~9 % of its FIR work) or middle octaves that fit neither the trace-lane kernel nor
			// the set.  It reads the traces themselves: a third stream, forked here, joined before the accumulation -- beside the chain and
			// the trace-lane kernel
			const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
			if (!p->xs) HIP_TRY(hipStreamCreateWithFlags(&p->xs, hipStreamNonBlocking));
			if (!p->ev_xs0) HIP_TRY(hipEventCreateWithFlags(&p->ev_xs0, evf));
			if (!p->ev_xs1) HIP_TRY(hipEventCreateWithFlags(&p->ev_xs1, evf));
			HIP_TRY(hipEventRecord(p->ev_xs0, st)); // (after the previous batch's accumulation: `part` is free again)
			HIP_TRY(hipStreamWaitEvent(p->xs, p->ev_xs0, 0));
			const unsigned nbw = (T.waves + 3) / 4;
			for (size_t u0 = 0; T.waves && u0 < nb; u0 += 2 * 32768) {
				const unsigned nt = (unsigned)std::min<size_t>(nb - u0, 2 * 32768);
				hipLaunchKernelGGL((k_fwd_poly<TIn, 2>), dim3(nbw, (nt + 1) / 2), dim3(256), 0, p->xs, xb + u0 * ld, ld, nt, p->N, T.d_sc, p->S, p->d_w,
				                   part + u0 * T.npart, T.npart, T.waves);
			}
			poly_join = true;
		} else if (T.waves) {
			const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
			if (!p->side) HIP_TRY(tspws_side_stream(p));
			if (!p->ev_fork) HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, evf));
			if (!p->ev_join) HIP_TRY(hipEventCreateWithFlags(&p->ev_join, evf));
			HIP_TRY(hipEventRecord(p->ev_fork, st)); // (after the previous batch's accumulation: `part` is free again)
			HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
			sp = p->side;
			const unsigned nbw = (T.waves + 3) / 4;
			for (size_t u0 = 0; u0 < nb; u0 += 2 * 32768) {
				const unsigned nt = (unsigned)std::min<size_t>(nb - u0, 2 * 32768);
				hipLaunchKernelGGL((k_fwd_poly<TIn, 2>), dim3(nbw, (nt + 1) / 2), dim3(256), 0, sp, xb + u0 * ld, ld, nt, p->N, T.d_sc, p->S, p->d_w,
				                   part + u0 * T.npart, T.npart, T.waves);
			}
		}
		// (a batch with a spectral set: the transposition also leaves the traces' maxima for the chain)
		if (dc) { if ((rc = spectral_transpose_t(p, xb, ld, nb, xT, TP, st))) return rc; }
		else hipLaunchKernelGGL((k_transpose_traces<TIn>), dim3((p->N + 63) / 64, nblk), dim3(256), 0, st, xb, ld, nb, p->N, TP, xT);
		// the spectral chain (transforms through HBM / MALL: bandwidth-bound) beside the trace-lane kernel (FP64-bound) on the side stream
		static const bool spec_serial = sweep_env("TSPWS_SPEC_SERIAL") != nullptr; // sweeps: one after the other
		if (dc && ((T.n && !spec_serial) || (T.gcoltiles && gemm_order == 2))) {
			const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
			if (!p->side) HIP_TRY(tspws_side_stream(p));
			if (!p->ev_fork) HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, evf));
			if (!p->ev_join) HIP_TRY(hipEventCreateWithFlags(&p->ev_join, evf));
			HIP_TRY(hipEventRecord(p->ev_fork, st)); // (after the transposition)
			if (T.n && !spec_serial) { HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0)); sp = p->side; }
			if (T.gcoltiles && gemm_order == 2) { HIP_TRY(hipStreamWaitEvent(p->xs, p->ev_fork, 0)); launch_gemm(p->xs); }
		}
		if (T.gcoltiles && gemm_order == 0) launch_gemm(st);
		if (dc && (rc = spectral_run_t(p, dc, (const TIn *)xT, TP, nb, planes, planes + p->ncoef, 2 * p->ncoef, nullptr, dc && T.n && !spec_serial ? sp : st))) return rc;
		if (T.n)
			hipLaunchKernelGGL((k_fwd_tl<TIn>), dim3(T.wgs, nblk), dim3(TL_NT), T.lds, st, (const TIn *)xT, TP, nb, p->N, T.d_items, T.n, p->d_w,
			                   planes, planes + p->ncoef, 2 * p->ncoef, part, T.npart);
		if (T.gcoltiles && gemm_order == 1) launch_gemm(st);
		if (sp != st) {
			HIP_TRY(hipEventRecord(p->ev_join, sp));
			HIP_TRY(hipStreamWaitEvent(st, p->ev_join, 0));
		}
		if (poly_join) {
			HIP_TRY(hipEventRecord(p->ev_xs1, p->xs));
			HIP_TRY(hipStreamWaitEvent(st, p->ev_xs1, 0));
		}
		FuseOut fz;
		fz.accST = planes; fz.accPS = planes + p->ncoef; fz.stride = 2 * p->ncoef; fz.tps = 64; fz.applied = true;
		const bool last = t0 + batch >= ntr; // the launch that completes the stacks also weights them (wa)
		tspws_launch_accumulate(p, part, nb, (double2 *)d_ST, (double2 *)d_PS, (t0 == 0 && !keep) ? 1 : 0, &fz, nblk, st, 1, 0, 0, &T, last && !keep ? wa : nullptr, ScaleRange());
		if (last && !keep && wa && wa->OUT && weighted) *weighted = true;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}


template <typename TIn>
static int stacks_impl(tspws_hip_plan *p, const TIn *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                       const WeightArgs *wa, bool *weighted, ScaleRange rg)
{ // keep: add to the stacks already in d_ST / d_PS instead of starting from zero
  // wa: weighting to apply where the stacks are completed (same kernel launch); *weighted tells whether that happened
	HIP_TRY(hipSetDevice(p->device));
	if (weighted) *weighted = false;
	if (!rg.on() && tspws_many_trace_path(p, ntr)) return stacks_tl<TIn>(p, d_x, ntr, ld, d_ST, d_PS, st, keep, wa, weighted);
	if (!ntr) { if (!keep) { HIP_TRY(hipMemsetAsync(d_ST, 0, p->ncoef * 16, st)); HIP_TRY(hipMemsetAsync(d_PS, 0, p->ncoef * 16, st)); } return 0; }
	int rc;
	if (tspws_generic_forward()) {
		size_t batch = std::max<size_t>(1, ((size_t)256 << 20) / (p->ncoef * sizeof(double2)));
		batch = std::min(batch, ntr);
		void *d_Y = nullptr;
		if ((rc = scratch(p, SCR_Y, batch * p->ncoef * sizeof(double2), &d_Y))) return rc;
		for (size_t t0 = 0; t0 < ntr; t0 += batch) {
			const size_t nb = std::min(batch, ntr - t0);
			if ((rc = forward_generic<TIn>(p, d_x + t0 * ld, nb, ld, (double *)d_Y, st))) return rc;
			if ((rc = tspws_hip_accumulate(p, (const double *)d_Y, nb, d_ST, d_PS, t0 == 0 && !keep, (void *)st))) return rc;
		}
		return 0;
	}
	// trace batch sized to the partial-coefficient scratch budget (even, for the 2-trace tiles)
	const size_t batch = std::min<size_t>(ntr, std::max<size_t>(2, ((tspws_part_budget_bytes()) / (p->npart * sizeof(double2))) & ~(size_t)1));
	void *v;
	if ((rc = scratch(p, SCR_PART, batch * p->npart * sizeof(double2), &v))) return rc;
	const bool fuse = p->n_fusable != 0;
	void *vz = nullptr;
	if (fuse) { // slice planes of the largest batch (unused when the only slice writes ST / PS directly)
		const unsigned tps = fuse_tps(p, batch);
		size_t nsl = (batch + tps - 1) / tps;
		if (const size_t last = ntr % batch) nsl = std::max(nsl, (last + fuse_tps(p, last) - 1) / fuse_tps(p, last)); // (a shorter last batch may take shorter slices)
		if (!(nsl == 1 && !keep && batch >= ntr) && (rc = scratch(p, SCR_FZ, nsl * 2 * p->ncoef * sizeof(double2), &vz))) return rc;
	}
	for (size_t t0 = 0; t0 < ntr; t0 += batch) {
		const size_t nb = std::min(batch, ntr - t0);
		const int zero_first = (t0 == 0 && !keep) ? 1 : 0;
		FuseOut fz;
		unsigned nsl = 0;
		if (fuse) {
			fz.tps = fuse_tps(p, nb);
			nsl = (unsigned)((nb + fz.tps - 1) / fz.tps);
			if (nsl == 1 && zero_first) { fz.accST = (double2 *)d_ST; fz.accPS = (double2 *)d_PS; fz.stride = 0; }
			else { fz.accST = (double2 *)vz; fz.accPS = (double2 *)vz + p->ncoef; fz.stride = 2 * p->ncoef; }
		}
		const bool all = nb == ntr && !keep; // one batch holds every trace: the accumulation completes the stacks
		if ((rc = forward_parts<TIn>(p, d_x + t0 * ld, nb, ld, (double2 *)v, st, fuse ? &fz : nullptr, rg))) return rc;
		tspws_launch_accumulate(p, (const double2 *)v, (unsigned)nb, (double2 *)d_ST, (double2 *)d_PS, zero_first, &fz, nsl, st, 1, 0, 0, nullptr, all ? wa : nullptr, rg);
		if (all && wa && wa->OUT && weighted) *weighted = true;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

int tspws_stacks_f32(tspws_hip_plan *p, const float *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                     const WeightArgs *wa, bool *weighted, ScaleRange rg)
{
	return stacks_impl<float>(p, d_x, ntr, ld, d_ST, d_PS, st, keep, wa, weighted, rg);
}
int tspws_stacks_f64(tspws_hip_plan *p, const double *d_x, size_t ntr, size_t ld, double *d_ST, double *d_PS, hipStream_t st, bool keep,
                     const WeightArgs *wa, bool *weighted, ScaleRange rg)
{
	return stacks_impl<double>(p, d_x, ntr, ld, d_ST, d_PS, st, keep, wa, weighted, rg);
}

extern "C" int tspws_hip_stacks_double(tspws_hip_plan *p, const double *d_P, unsigned K, size_t ldP, double *d_ST, double *d_PS, void *s)
{
	if (!p || !d_P || !d_ST || !d_PS) return fail(TSPWS_E_ARG, "stacks_double: NULL");
	return stacks_impl<double>(p, d_P, K, ldP, d_ST, d_PS, S_(s), false, nullptr, nullptr, ScaleRange());
}

extern "C" int tspws_hip_stacks_float(tspws_hip_plan *p, const float *d_x, size_t mtr, size_t ld, double *d_ST, double *d_PS, void *s)
{
	if (!p || !d_x || !d_ST || !d_PS) return fail(TSPWS_E_ARG, "stacks_float: NULL");
	return stacks_impl<float>(p, d_x, mtr, ld, d_ST, d_PS, S_(s), false, nullptr, nullptr, ScaleRange());
}
