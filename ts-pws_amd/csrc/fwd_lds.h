// fwd_lds.h -- LDS-staged polyphase forward frame CWT for gfx950 (included by forward.hip).
//
// Same decomposition as fwd_poly.h (thread = 8 consecutive outputs of one phase, sliding register window),
// but the operands come from LDS: a 256-thread workgroup owns 8 "group slots" (4 waves x 2 passes) of ONE
// scale (and one 64-phase chunk when D >= 64), keeps that scale's taps RESIDENT in LDS and walks a slice of
// traces: the x window of trace t+1 is fetched into registers while trace t is computed out of LDS, so the
// global-memory latency hides behind the FMAs and the tap staging / scale set-up is paid once per workgroup.
//
// Why it is written the way it is: a wave64 v_fma_f64 costs 4 cycles and so does almost every other 64-bit VALU op, so
// address arithmetic, predicates and pointer math compete with the FP64 FMAs for the same issue slots.  Hence: the body
// is templated on log2(D) so every LDS address in the FMA loop is `one base VGPR + immediate`; the window loads use a
// uniform base pointer plus ONE opaque 32-bit offset per thread (the compiler otherwise hoists all row offsets out of
// the trace loop as 64-bit pairs: 56 VGPRs and spills); the prefetched window stays raw in registers (conversion and the
// idle-lane mask happen at the LDS store); the lane reductions run on the VALU (permlane swaps + DPP: no LDS traffic);
// the workgroup barriers order LDS only (fl_lds_barrier below).
//
//   LOGD = 6 (D >= 64, any D): slot = output group g0+slot, lanes = 64 consecutive phases of the chunk;
//             LDS x image  xL[row][64]   row j <-> sample (g0*8 + qa + j) D + m0 + lane - c   (88 rows)
//             LDS taps     tL[q][64]     q <-> tap (qa + q) D + m0 + lane                      (24 rows)
//   LOGD < 6 (D = 2^LOGD):     lanes = (group, phase); a slot is 64/D groups; the x window of the workgroup is one
//             contiguous sample range stored with D doubles of padding per 8 D samples, so the 64/D groups of
//             a wave fall on distinct banks (group stride 9 D doubles).
//   Q <= 24 taps per phase: taps resident, one stage per trace.  Q > 24: balanced tiles of <= 24, taps re-staged.
//
// Tunables (compile time): FL_WAVES x FL_PASSES group slots per workgroup (FL_PASSES_FINE for D <= 4), FL_BATCH steps per
// LDS read burst (FL_PREFETCH: the next burst's operands are requested first).
// Measurements (MI355X, the ten 131072-sample north-star transforms):
//   round 1: 4x2 slots 166 us; 8x1 forced to 128 VGPRs 177 us; 4x1 with 16-row tap tiles (3 workgroups per CU) 178 us.
//   round 2 (-DFL_TIMING=1, tools/fwd_timing.py; PMC in profiles/r02_pmc_counters.txt): 138 us alone, 177-187 us with
//   k_fwd_poly beside it.  Of the issued VALU instructions only 52 % are FMAs (48 M vs 25 M wave-instructions): lane
//   reductions (~200 per wave and trace), window addressing (~100), phase normalisation, register moves; the VALU is
//   active 51 % of the SIMD cycles, half of that in FMAs.  The D >= 64 workgroups spend 20-25 % of their cycles in the
//   FMA passes: their 45 KB window (FP64 partial stacks) takes as long to pull in (~16 B/cycle per CU) as the FMAs take.
//   One dense FMA-issuing wave per SIMD would be enough for ~85 % of the FP64 rate (tools/fma64_issue.hip) -- the time
//   goes to everything around the FMA passes, not to the issue rate.
//   Timing ablations (results wrong, whole-call time on the ten transforms, direct kernel not running beside): the
//   D >= 64 workgroups cost 62 us, the D < 64 ones 57 us, one after the other.  Of the 62 us: prologue (scale search,
//   descriptor, tap staging, first window) 2, the per-trace skeleton (two barriers, LDS image store, window loads) 20,
//   lane reductions + phase normalisation + partial stores 20, FMA passes with their LDS reads 20.  The window loads are
//   NOT on the critical path: contiguous rows instead of D-strided ones, or no window loads at all, change the call by
//   0-3 us (the phase timers charge the waiting to whichever phase meets the barrier).  Hence the two changes that did
//   pay: buffer loads (no address VALU work) and the 16-output window (half the LDS reads per FMA): -4 and -8 us.
//
// Used for scales with at least 8 output groups (N_s >= 64) and D >= 64 or a power of two; everything else
// (very coarse scales whose parallelism is only in the taps, odd small decimations) stays on k_fwd_poly.
#pragma once

#define FL_R 8
#ifndef FL_WAVES
#define FL_WAVES 4                              /* waves per workgroup */
#endif
#ifndef FL_PASSES
#define FL_PASSES 2                             /* group slots per wave (workgroup = FL_WAVES x FL_PASSES slots) */
#endif
#ifndef FL_BATCH
#define FL_BATCH 2                              /* tap steps per burst: the next burst's LDS reads are requested before this burst's FMAs (FL_PREFETCH); 4 + prefetch spills */
#endif
#ifndef FL_PASSES_FINE
#define FL_PASSES_FINE 1                        /* group slots per wave for D <= 4: the fused kernel keeps 8/D coefficients' stacks per lane and slot */
#endif
#ifndef FL_PREFETCH
#define FL_PREFETCH 1                           /* 1: LDS operands of the next burst are requested before this burst's FMAs */
#endif
#ifndef FL_WIDE
#define FL_WIDE 1                               /* D >= 64: the two slots of a wave as ONE 16-output window (half the LDS reads per FMA) */
#endif
#define FL_NT (64 * FL_WAVES)                   /* threads per workgroup */
#define FL_SLOTS (FL_WAVES * FL_PASSES)
// QT = tap rows resident in LDS (taps per phase a workgroup handles without re-staging): a template parameter of the kernel,
// chosen per plan (fl_pick_qt).  24: the default Morlet frames (Q <= 18), 70 KB of LDS per workgroup.  32: frames with 24 < Q <= 32
// -- the Mexican hat's second voice of every octave has Q = 29 or 30 --, exactly 80 KB, so that two workgroups still share a CU
// (round 4: cfg4's 110 transforms 1.36 -> 1.22 ms; with 64 doubles of slack on top, i.e. ONE workgroup per CU, 1.66 ms).
#define FL_TAPS_BYTES_(QT) ((QT) * 64 * 16)      /* 24 KiB / 32 KiB */
#define FL_XROWS_(QT) (FL_SLOTS * FL_R + (QT))   /* rows staged when D >= 64 (FL_SLOTS*R + QT - 1 needed); multiple of FL_WAVES */
#define FL_XSMALL_(QT) ((FL_SLOTS * 512 + ((QT) - 1) * 32 + FL_NT - 1) / FL_NT) /* D < 64: FL_NT-sample columns staged (window <= FL_SLOTS*512 + (QT - 1) D) */
#define FL_XPAD (FL_NT * 9 / 8)                 /* padded LDS distance of two columns */
#define FL_SCR 528                              /* reduction scratch per wave: 8 rows x 65 (+8) doubles */
#define FL_MAX2(a, b) ((a) > (b) ? (a) : (b))
#define FL_XSLACK_(QT) ((QT) <= 24 ? 64 : 0)    /* doubles past the image (reads of prefetched rows that are never used stay inside the allocation; beyond it LDS reads return zeros) */
#define FL_X_ALLOC_(QT) (FL_MAX2(FL_MAX2(FL_XROWS_(QT) * 64, FL_XSMALL_(QT) * FL_XPAD), FL_WAVES * FL_SCR) + FL_XSLACK_(QT))
#define FL_LDS_BYTES_(QT) (FL_TAPS_BYTES_(QT) + FL_X_ALLOC_(QT) * 8)
static_assert(FL_LDS_BYTES_(32) * 2 <= 160 * 1024, "two workgroups of the QT = 32 kernel must fit the 160 KB of a CU");
// Every row that is CONSUMED lies inside the image: the last wave's 16-output window ends at row 16 (FL_WAVES - 1) + QT + 14 (k_fwd_lds, wide
// form).  Only the look-ahead of the LAST batch (operands requested before the loop knows it has ended, never used) may read past the
// allocation -- inside the slack for QT = 24, beyond the workgroup's LDS for QT = 32, where the hardware returns zeros.
static_assert(16 * (FL_WAVES - 1) + 24 + 14 < FL_XROWS_(24) && 16 * (FL_WAVES - 1) + 32 + 14 < FL_XROWS_(32), "a consumed window row would lie outside the staged image");
static_assert(FL_XROWS_(24) * 64 <= FL_X_ALLOC_(24) - FL_XSLACK_(24) && FL_XROWS_(32) * 64 <= FL_X_ALLOC_(32) - FL_XSLACK_(32), "the staged image must fit its allocation");

#ifndef FL_TIMING
#define FL_TIMING 0                             /* 1: per-phase s_memtime totals per LOGD class (tools/fwd_timing.py) */
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every wave would wait at each
// barrier for its global stores of the previous trace's partials (and any load in flight) to complete -- measured with
// the FL_TIMING hooks: ~13 k of the ~26 k shader cycles per wave and trace of the D >= 64 workgroups.  Global memory
// needs no ordering here: the prefetched window is consumed through registers (the compiler's own vmcnt waits).
__device__ __forceinline__ void fl_lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef FL_ABLATE
#define FL_ABLATE 0                             /* 1: debug builds can switch parts of the kernel off (TSPWS_FWD_CLASSES, tools/fwd_bench.py): results wrong, only the clock counts */
#endif
#if FL_ABLATE
// bits 0-6: the workgroups of log2(D) class c run (6: D >= 64); for the D >= 64 workgroups: 0x100 lane reduction and partial
// stores only after the last trace, 0x200 no FMA passes, 0x400 no LDS image store, 0x800 no window prefetch; every class: 0x1000 the two
// barriers only in front of EVEN traces (what a trace pair per thread would share; racy, only the clock counts)
__device__ unsigned fl_class_mask = 0x7fu;
#endif
#if FL_TIMING
__device__ unsigned long long *fl_timing_out; // [7 classes][8]: set-up, barrier 1, stage, barrier 2, FMA, reduce, traces, waves
#endif

struct FlWrapNo { static constexpr bool value = true; };   // run<NW>: no circular wrap in any window of the workgroup
struct FlWrapYes { static constexpr bool value = false; };

// FUSE: scales without phase splits (D <= 64) do not write per-trace coefficients at all: the workgroup owns its
// coefficients for the whole trace slice, so it phase-normalises each one right after the lane reduction and keeps
// ST += Y, PS += Y/|Y| (ts_pws1f_lib.c:489-492) in registers, in trace order; one store per coefficient and slice at
// the end (accST / accPS are the slice's [ncoef] planes).  Split scales (D > 64) still write their partials.
template <typename TIn, int LOGD, bool FUSE, int PASSES, int QT>
__device__ __forceinline__ void fwd_lds_body(const TIn *__restrict__ x0, const size_t ld, const unsigned ntr, const unsigned N, const ScaleDesc &d,
                                             const double2 *__restrict__ ws, double *__restrict__ pout0, const size_t npart,
                                             const unsigned chunk, const unsigned bb, double2 *tL, double *xL,
                                             double2 *__restrict__ accST, double2 *__restrict__ accPS, const FuseFinal &ff, const unsigned slice)
{
	static_assert(QT % FL_R == 0 && QT % FL_WAVES == 0 && QT % FL_BATCH == 0, "QT: a multiple of the window length, the waves and the burst");
	constexpr int R = FL_R;
	constexpr int TPW = QT / FL_WAVES;                        // tap rows staged per wave (D >= 64)
	constexpr int TSMALL = (QT * 32 + FL_NT - 1) / FL_NT;     // tap values staged per thread (D < 64: <= QT * 32 taps)
	constexpr bool SMALL = LOGD < 6;
	constexpr int LG = SMALL ? LOGD : 0;
	constexpr unsigned DC = 1u << (SMALL ? LOGD : 6);     // D when SMALL
	constexpr unsigned GW = SMALL ? (64u >> LG) : 1u;     // groups per wave-slot
	constexpr int SLOTS = FL_WAVES * PASSES;               // group slots of this workgroup
	// WIDE (D >= 64, two slots per wave): the wave's two slots are ADJACENT output groups and share one 16-output sliding
	// window -- one x row and one tap row per 32 FMAs instead of per 16 (the tap row alone is a 1-KB LDS read per wave).
	constexpr bool WIDE = !SMALL && PASSES == 2 && FL_WIDE;
	constexpr int XROWS = SLOTS * FL_R + QT;            // rows staged when D >= 64
	constexpr int NXV = SMALL ? (SLOTS * 512 + (QT - 1) * 32 + FL_NT - 1) / FL_NT : XROWS / FL_WAVES; // x values staged per thread
	const unsigned D = SMALL ? DC : d.D;
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned g0 = bb * (unsigned)SLOTS * GW;      // first output group of this workgroup
	const unsigned lane_m = SMALL ? (lane & (DC - 1)) : lane;
	const unsigned lane_g = SMALL ? (lane >> LG) : 0;
	const unsigned m0 = SMALL ? 0u : chunk * 64u;
	const unsigned m = m0 + lane;

	const unsigned ntile = (d.Q + QT - 1) / QT, qt = (d.Q + ntile - 1) / ntile; // balanced tiles
	const bool resident = ntile == 1;

	// ---- trace-independent staging geometry -------------------------------------------------
	// SMALL: contiguous window of NXV*256 samples starting at base(qa); LARGE: XROWS rows of 64 lanes
	auto x_base = [&](unsigned qa) -> long long {
		return SMALL ? ((long long)g0 * R + qa) * DC - d.c : ((long long)g0 * R + qa) * D + m0 - d.c;
	};
	const bool full = SMALL ? true : (m0 + 63 < D);
	// The prefetched window stays RAW in registers (TIn: half the registers for float traces) and every load of a window
	// is independent of the others: conversion to double and the idle-lane mask are applied when the window is written
	// to LDS, one trace later -- nothing between the loads waits for memory.
	const bool mok = SMALL ? true : (m < D);
	// Addressing: uniform trace pointer + one 32-bit element offset per thread that walks the window; the empty asm keeps
	// the compiler from hoisting all NXV offsets (trace-independent) out of the trace loop as 64-bit register pairs.
	// NW (no circular wrap anywhere in this workgroup's windows -- all but the blocks at the seam): uniform row offsets
	// (SGPRs, advanced on the scalar unit) + one constant 32-bit lane offset, i.e. the loads need no VALU address arithmetic
	// at all -- a wave's address VALU ops queue behind the other wave's 4-cycle FMAs (a load took 100-300 cycles to
	// issue).  The two cases are separate instantiations of the trace loop (run<NW> below): inside one function the compiler
	// merges their loads and computes per-lane 64-bit addresses for both.  Wrap: one opaque 32-bit element offset per
	// thread that walks the window (the empty asm keeps the compiler from hoisting all NXV offsets, which are trace-
	// independent, out of the trace loop as 64-bit register pairs).
	const unsigned qa_last = (ntile - 1) * qt;
	const bool nw_all = (unsigned long long)N * sizeof(TIn) <= 0xFFFFFFFFull && // (a trace must fit a buffer resource's 32-bit byte range)
	                    (SMALL ? (x_base(0) >= 0 && x_base(qa_last) + NXV * FL_NT <= (long long)N)
	                           : (full && x_base(0) >= 0 && x_base(qa_last) + (long long)(XROWS - 1) * D + 63 < (long long)N));
	auto load_x = [&](auto NWT, TIn (&xv)[NXV], const TIn *__restrict__ xt, unsigned qa) {
		const long long base = x_base(qa);
		if constexpr (decltype(NWT)::value) {
			// buffer loads: resource = this trace (uniform, rebuilt per trace on the scalar unit), voffset = the lane's constant
			// byte offset, soffset = the row's uniform byte offset -- `buffer_load v, v_lane, s[rsrc], s_row offen`
			const unsigned ubase = SMALL ? (unsigned)base : (unsigned)(base + (long long)wv * D);
			const unsigned stp = (SMALL ? (unsigned)FL_NT : (unsigned)FL_WAVES * D) * (unsigned)sizeof(TIn);
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)xt, 0, N * (unsigned)sizeof(TIn), 0x00020000);
			const unsigned vb = (SMALL ? tid : lane) * (unsigned)sizeof(TIn);
			unsigned so = __builtin_amdgcn_readfirstlane(ubase) * (unsigned)sizeof(TIn);
#pragma unroll
			for (int i = 0; i < NXV; i++) {
				if constexpr (sizeof(TIn) == 8) xv[i] = __builtin_bit_cast(TIn, __builtin_amdgcn_raw_buffer_load_b64(rs, vb, so, 0));
				else xv[i] = __builtin_bit_cast(TIn, __builtin_amdgcn_raw_buffer_load_b32(rs, vb, so, 0));
				so += stp;
			}
		} else {
			// circular seam and / or a last chunk with idle phase lanes (they read a valid address, store_x zeroes them)
			unsigned off = SMALL ? wrap_index(base + tid, N) : wrap_index(base + (long long)wv * D + (mok ? lane : 0), N);
			const unsigned wstp = SMALL ? (unsigned)FL_NT % N : (unsigned)(((unsigned long long)FL_WAVES * D) % N);
			asm volatile("" : "+v"(off));
#pragma unroll
			for (int i = 0; i < NXV; i++) {
				xv[i] = xt[off];
				off += wstp; if (off >= N) off -= N;
			}
		}
	};
	auto store_x = [&](const TIn (&xv)[NXV]) {
		if (SMALL) {
			const unsigned pt = tid + ((tid >> (3 + LG)) << LG); // padded index of element tid; +FL_XPAD per FL_NT elements
#pragma unroll
			for (int i = 0; i < NXV; i++) xL[pt + FL_XPAD * i] = (double)xv[i];
		} else {
			double *xdst = xL + wv * 64 + lane;
			if (full) {
#pragma unroll
				for (int i = 0; i < NXV; i++) xdst[FL_NT * i] = (double)xv[i];
			} else {
#pragma unroll
				for (int i = 0; i < NXV; i++) xdst[FL_NT * i] = mok ? (double)xv[i] : 0.0;
			}
		}
	};
	auto stage_taps = [&](unsigned qa, unsigned qn) {
		if (SMALL) { // qn * D <= QT * 32 values, natural order
			double2 tv[TSMALL];
#pragma unroll
			for (int i = 0; i < TSMALL; i++) {
				const unsigned e = tid + (unsigned)FL_NT * (unsigned)i, l = qa * DC + e;
				tv[i] = (e < qn * DC && l < d.L) ? ws[l] : make_double2(0.0, 0.0);
			}
#pragma unroll
			for (int i = 0; i < TSMALL; i++) if (tid + FL_NT * i < QT * 64) tL[tid + FL_NT * i] = tv[i];
		} else {     // rows wv, wv+4, ... of the QT-row tile
			double2 tv[TPW];
			const unsigned l0 = (qa + wv) * D + m;
			if (qn == QT && full && (qa + QT - 1u) * D + m0 + 63 < d.L) { // a full tile whose every tap exists
#pragma unroll
				for (int i = 0; i < TPW; i++) tv[i] = ws[l0 + (unsigned)FL_WAVES * D * (unsigned)i];
			} else {
#pragma unroll
				for (int i = 0; i < TPW; i++) {
					const unsigned q = wv + (unsigned)FL_WAVES * (unsigned)i, l = l0 + (unsigned)FL_WAVES * D * (unsigned)i;
					tv[i] = (m < D && q < qn && l < d.L) ? ws[l] : make_double2(0.0, 0.0);
				}
			}
			double2 *tdst = tL + wv * 64 + lane;
#pragma unroll
			for (int i = 0; i < TPW; i++) tdst[FL_NT * i] = tv[i];
		}
	};

	if (resident) stage_taps(0, d.Q); // made visible by the barrier in front of the first compute
	auto run = [&](auto NWT) {
	TIn xv[NXV];
	if (resident) load_x(NWT, xv, x0, 0);

	constexpr int NACC = FUSE ? (SMALL ? (LOGD <= 3 ? (8 >> LOGD) : 1) : 1) : 1; // complex coefficients per lane and pass
	const bool fuse = FUSE && d.nsplit == 1;
	double2 fst[PASSES][NACC], fps[PASSES][NACC];
	if (FUSE) {
#pragma unroll
		for (int p = 0; p < PASSES; p++)
#pragma unroll
			for (int i = 0; i < NACC; i++) { fst[p][i] = make_double2(0.0, 0.0); fps[p][i] = make_double2(0.0, 0.0); }
	}

#if FL_TIMING
	unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define FL_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tm[i] += n_ - tc; tc = n_; } while (0)
#else
#define FL_STAMP(i) do { } while (0)
#endif
	FL_STAMP(0); // set-up: descriptor, tap staging, first x loads
	for (unsigned t = 0; t < ntr; t++) {
		const TIn *xt = x0 + (size_t)t * ld;
		double ar[PASSES][R], ai[PASSES][R]; // set by the first tap step of the first tile (a product instead of 0 + product)

		for (unsigned qa = 0; qa < d.Q; qa += qt) {
			const unsigned qn = (d.Q - qa) < qt ? (d.Q - qa) : qt;
#if FL_ABLATE
			if (!((fl_class_mask & 0x1000u) && (t & 1)))
#endif
			fl_lds_barrier(); // everyone is done reading the previous image (x rows, taps)
			FL_STAMP(1); // barrier 1 (+ accumulator reset)
			if (!resident) { stage_taps(qa, qn); load_x(NWT, xv, xt, qa); }
#if FL_ABLATE
			if (!(!SMALL && (fl_class_mask & 0x400u)))
#endif
			store_x(xv);
			FL_STAMP(2); // wait for the prefetched x (vmcnt) + LDS stores
#if FL_ABLATE
			if (!((fl_class_mask & 0x1000u) && (t & 1)))
#endif
			fl_lds_barrier();
#if FL_ABLATE
			if (!(!SMALL && (fl_class_mask & 0x800u)))
#endif
			if (resident && t + 1 < ntr) load_x(NWT, xv, xt + ld, 0); // next trace's window flies while this one is computed
			FL_STAMP(3); // barrier 2 + issue of the next prefetch
			// ------------------------------------------------------------------ compute: two passes (group slots) per wave
#if FL_ABLATE
			if (!SMALL && (fl_class_mask & 0x200u)) continue;
#endif
			if constexpr (WIDE) {
				constexpr int R2 = 2 * R, RING = R2 + 2 * FL_BATCH;
				const double *xb = xL + wv * (R2 * 64) + lane; // slots 2 wv, 2 wv + 1: outputs (g0 + 2 wv) R .. + 15
				const double2 *tb = tL + lane;
				// the sliding window is a ring of registers: the operands of the next burst land in the slots they are used from
				// (no copies); every index below is a compile-time constant after unrolling
				double xw[RING];
#pragma unroll
				for (int j = 0; j < R2 - 1; j++) xw[j] = xb[j * 64];
				double2 tn[2][FL_BATCH];
#pragma unroll
				for (int u = 0; u < FL_BATCH; u++) { xw[(u + R2 - 1) % RING] = xb[(u + R2 - 1) * 64]; tn[0][u] = tb[u * 64]; }
				const bool first_tile = qa == 0;
#pragma unroll
				for (int h = 0; h < QT / FL_BATCH; h++) {
					if ((unsigned)(h * FL_BATCH) < qn) {
						if (h + 1 < QT / FL_BATCH) {
#pragma unroll
							for (int u = 0; u < FL_BATCH; u++) {
								xw[((h + 1) * FL_BATCH + u + R2 - 1) % RING] = xb[((h + 1) * FL_BATCH + u + R2 - 1) * 64]; // (last row read: 16 wv + QT + 14 < XROWS)
								tn[(h + 1) & 1][u] = tb[((h + 1) * FL_BATCH + u) * 64];
							}
							asm volatile("" ::: "memory"); // the scheduler otherwise sinks these reads to just in front of their FMAs
						}
#pragma unroll
						for (int u = 0; u < FL_BATCH; u++) {
							const int sidx = h * FL_BATCH + u; // compile-time after unrolling
							if (sidx == 0 && first_tile) {
#pragma unroll
								for (int r = 0; r < R2; r++) {
									ar[r / R][r % R] = xw[r % RING] * tn[0][0].x;
									ai[r / R][r % R] = xw[r % RING] * tn[0][0].y;
								}
							} else {
#pragma unroll
								for (int r = 0; r < R2; r++) {
									ar[r / R][r % R] = fma(xw[(sidx + r) % RING], tn[h & 1][u].x, ar[r / R][r % R]);
									ai[r / R][r % R] = fma(xw[(sidx + r) % RING], tn[h & 1][u].y, ai[r / R][r % R]);
								}
							}
						}
					}
				}
			} else
#pragma unroll
			for (int p = 0; p < PASSES; p++) {
				const unsigned slot = (unsigned)p * (unsigned)FL_WAVES + wv;
				const double *xb;  // one base per pass; every read below is base + compile-time offset
				const double2 *tb;
				if (SMALL) {
					const unsigned gl = slot * GW + lane_g;            // group within the workgroup
					xb = xL + gl * (R * DC) + gl * DC + lane_m;       // padded: + D per 8 D samples
					tb = tL + lane_m;
				} else {
					xb = xL + slot * (R * 64) + lane;
					tb = tL + lane;
				}
				constexpr unsigned XS = SMALL ? DC : 64u;              // row stride in doubles (before padding)
#define FL_XOFF(j) (SMALL ? (unsigned)(j) * DC + ((unsigned)(j) >> 3) * DC : (unsigned)(j) * 64u)
#if FL_PREFETCH
				// software pipeline: the operands of burst h+1 are requested before the FMAs of burst h, so a wave's own FMA
				// stream covers its LDS latency (one wave per SIMD sustains ~85 % of the FP64 rate when its stream is dense:
				// tools/fma64_issue.hip).  Prefetching past qn reads rows / taps that exist in LDS but are never used.
				// The sliding window is a ring of registers (see the wide form above).
				constexpr int RING = R + 2 * FL_BATCH;
				double xw[RING];
#pragma unroll
				for (int j = 0; j < R - 1; j++) xw[j] = xb[FL_XOFF(j)];
				double2 tn[2][FL_BATCH];
#pragma unroll
				for (int u = 0; u < FL_BATCH; u++) { xw[(u + R - 1) % RING] = xb[FL_XOFF(u + R - 1)]; tn[0][u] = tb[u * XS]; }
				const bool first_tile = qa == 0;
#pragma unroll
				for (int h = 0; h < QT / FL_BATCH; h++) {
					if ((unsigned)(h * FL_BATCH) < qn) {
						if (h + 1 < QT / FL_BATCH) {
#pragma unroll
							for (int u = 0; u < FL_BATCH; u++) {
								xw[((h + 1) * FL_BATCH + u + R - 1) % RING] = xb[FL_XOFF((h + 1) * FL_BATCH + u + R - 1)];
								tn[(h + 1) & 1][u] = tb[((h + 1) * FL_BATCH + u) * XS];
							}
							asm volatile("" ::: "memory"); // the scheduler otherwise sinks these reads to just in front of their FMAs
						}
#pragma unroll
						for (int u = 0; u < FL_BATCH; u++) {
							const int sidx = h * FL_BATCH + u; // compile-time after unrolling
							if (sidx == 0 && first_tile) {
#pragma unroll
								for (int r = 0; r < R; r++) {
									ar[p][r] = xw[r % RING] * tn[0][0].x;
									ai[p][r] = xw[r % RING] * tn[0][0].y;
								}
							} else {
#pragma unroll
								for (int r = 0; r < R; r++) {
									ar[p][r] = fma(xw[(sidx + r) % RING], tn[h & 1][u].x, ar[p][r]);
									ai[p][r] = fma(xw[(sidx + r) % RING], tn[h & 1][u].y, ai[p][r]);
								}
							}
						}
					}
				}
#else
				double xw[R];
#pragma unroll
				for (int j = 0; j < R - 1; j++) xw[j] = xb[FL_XOFF(j)];
				if (qa == 0) {
#pragma unroll
					for (int r = 0; r < R; r++) { ar[p][r] = 0; ai[p][r] = 0; }
				}
#pragma unroll
				for (int h = 0; h < QT / FL_BATCH; h++) { // LDS reads of FL_BATCH steps are issued together, then their FMAs
					if ((unsigned)(h * FL_BATCH) < qn) {
						double xn[FL_BATCH];
						double2 tn[FL_BATCH];
#pragma unroll
						for (int u = 0; u < FL_BATCH; u++) {
							xn[u] = xb[FL_XOFF(h * FL_BATCH + u + R - 1)];
							tn[u] = tb[(h * FL_BATCH + u) * XS];
						}
#pragma unroll
						for (int u = 0; u < FL_BATCH; u++) {
							// no per-step guard: taps past the filter end are staged as exact zeros and the x image always
							// holds the rows of a whole burst, so a burst is straight-line code (per-step branches turned
							// the window registers into phi copies and cost an issue bubble each)
							const int sidx = h * FL_BATCH + u; // compile-time after unrolling
							xw[(sidx + R - 1) % R] = xn[u];
#pragma unroll
							for (int r = 0; r < R; r++) {
								ar[p][r] = fma(xw[(sidx + r) % R], tn[u].x, ar[p][r]);
								ai[p][r] = fma(xw[(sidx + r) % R], tn[u].y, ai[p][r]);
							}
						}
					}
				}
#endif
#undef FL_XOFF
			}
		}

		FL_STAMP(4); // FMA passes
		// ------------------------------------------------------------------ combine the phase lanes
		if (FUSE && fuse) { // keep the running stacks in registers
#pragma unroll
			for (int p = 0; p < PASSES; p++) {
				double v[2 * R];
#pragma unroll
				for (int r = 0; r < R; r++) { v[2 * r] = ar[p][r]; v[2 * r + 1] = ai[p][r]; }
				if (SMALL) {
					int n;
					unsigned first;
					valu_rs_cplx<LG>(v, lane, n, first);
#pragma unroll
					for (int i = 0; i < NACC; i++) {
						const double2 y = make_double2(v[2 * i], -v[2 * i + 1]); // conj
						fst[p][i].x += y.x; fst[p][i].y += y.y;
#ifdef FL_NONORM /* timing ablation (results wrong): what does the phase normalisation cost inside this kernel? */
						fps[p][i].x += y.x; fps[p][i].y -= y.y;
#else
						add_unit_phasor(fps[p][i], y);
#endif
					}
				} else {
					double re, im;
					unsigned first;
					valu_reduce_cplx64(v, lane, re, im, first);
					const double2 y = make_double2(re, -im);
					fst[p][0].x += y.x; fst[p][0].y += y.y;
#ifdef FL_NONORM
					fps[p][0].x += y.x; fps[p][0].y -= y.y;
#else
					add_unit_phasor(fps[p][0], y);
#endif
				}
			}
			FL_STAMP(5); // lane reduction + phase normalisation
			continue;
		}
#if FL_ABLATE
		if (!SMALL && (fl_class_mask & 0x100u) && t + 1 < ntr) continue;
#endif
		// ------------------------------------------------------------------ store the split partial
		double *pout = pout0 + (size_t)t * npart * 2;
		if (SMALL) {
#pragma unroll
			for (int p = 0; p < PASSES; p++) {
				constexpr int NV = 2 * R;
				double v[NV];
#pragma unroll
				for (int r = 0; r < R; r++) { v[2 * r] = ar[p][r]; v[2 * r + 1] = ai[p][r]; }
				int n;
				unsigned first;
				valu_rs16<LG>(v, lane, n, first); // permlane / DPP exchanges over the D lanes of each group (no LDS traffic)
				constexpr unsigned dup_mask = LOGD == 5 ? 0x1u : 0u; // D = 32: the last bit (0) was a butterfly, odd lanes duplicate
				const unsigned g = g0 + ((unsigned)p * (unsigned)FL_WAVES + wv) * GW + lane_g;
				if (!(lane & dup_mask)) {
#pragma unroll
					for (int i = 0; i < NV; i++) {
						if (i < n) {
							const unsigned id = first + i, ri = id & 1, r = id >> 1;
							const unsigned k = g * R + r;
							if (k < d.Ns) pout[(size_t)k * 2 + ri] = ri ? -v[i] : v[i]; // conj
						}
					}
				}
			}
		} else {
			// 64-lane reduction on the VALU (valu_reduce16: permlane swaps + DPP, no LDS traffic, no extra barrier):
			// every lane ends with element o = 8*b5 + 4*b4 + 2*b3 + b2 of (re0, im0, re1, im1, ...); one lane per quad stores
#pragma unroll
			for (int p = 0; p < PASSES; p++) {
				double v[2 * R];
#pragma unroll
				for (int r = 0; r < R; r++) { v[2 * r] = ar[p][r]; v[2 * r + 1] = ai[p][r]; }
				const double sum = valu_reduce16(v, lane);
				const unsigned o = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
				const unsigned k = (g0 + (WIDE ? wv * (unsigned)PASSES + (unsigned)p : (unsigned)p * (unsigned)FL_WAVES + wv)) * R + (o >> 1);
				if ((lane & 3) == 0 && k < d.Ns) pout[(size_t)k * 2 + (o & 1)] = (o & 1) ? -sum : sum; // conj
			}
		}
	}
	if (FUSE && fuse) { // one store per coefficient: the slice's linear and phase stacks
		constexpr unsigned dup_mask = SMALL ? (LOGD == 5 ? 0x3u : LOGD == 4 ? 0x1u : 0u) : 0x7u; // butterfly bits: duplicates
		// element offset of the lane's first value, as left by valu_rs_cplx / valu_reduce_cplx64 (a lane whose bit is set
		// keeps the upper half at every reduce-scatter stage)
		unsigned first = 0;
		if (SMALL) {
			if (LOGD >= 5) first += (lane & 16) ? 8u : 0u;
			if (LOGD >= 4) first += (lane & 8) ? (LOGD == 5 ? 4u : 8u) : 0u;
			if (LOGD >= 3) first += (lane & 4) ? (LOGD == 5 ? 2u : LOGD == 4 ? 4u : 8u) : 0u;
			if (LOGD >= 2 && LOGD <= 4) first += (lane & 2) ? (LOGD == 4 ? 2u : LOGD == 3 ? 4u : 8u) : 0u;
			if (LOGD >= 1 && LOGD <= 3) first += (lane & 1) ? (LOGD == 3 ? 2u : LOGD == 2 ? 4u : 8u) : 0u;
			// LOGD == 0: no exchange at all, the lane keeps its 16 values (first = 0)
		} else first = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2;
		if (!(lane & dup_mask)) {
#pragma unroll
			for (int p = 0; p < PASSES; p++) {
				const unsigned g = SMALL ? g0 + ((unsigned)p * (unsigned)FL_WAVES + wv) * GW + lane_g
				                         : g0 + (WIDE ? wv * (unsigned)PASSES + (unsigned)p : (unsigned)p * (unsigned)FL_WAVES + wv);
#pragma unroll
				for (int i = 0; i < NACC; i++) {
					const unsigned k = g * R + (first >> 1) + (unsigned)i;
					if (k < d.Ns) {
						const size_t ci = d.coef_off + k;
						if (ff.OUT) { // complete the column's stacks here: earlier stages first (stage order), then this slice; weights on the spot
							double2 st = fst[p][i], ps = fps[p][i];
							if (ff.nprev) {
								double2 s0 = make_double2(0.0, 0.0), p0 = make_double2(0.0, 0.0);
								for (unsigned q = 0; q < ff.nprev; q++) {
									const double2 a = ff.pST[q * ff.pair_stride + slice * ff.slice_stride + ci], b = ff.pPS[q * ff.pair_stride + slice * ff.slice_stride + ci];
									s0.x += a.x; s0.y += a.y; p0.x += b.x; p0.y += b.y;
								}
								st.x = s0.x + st.x; st.y = s0.y + st.y; ps.x = p0.x + ps.x; ps.y = p0.y + ps.y;
							}
							ff.OUT[slice * ff.out_stride + ci] = weight_value(st, ps, ff.mode, ff.K, ff.Mv ? ff.Mv[slice] : ff.M, ff.wu);
							if ((int)slice == ff.keep_slice) ff.keepST[ci] = st;
						} else { accST[ci] = fst[p][i]; accPS[ci] = fps[p][i]; }
					}
				}
			}
		}
	}
#if FL_TIMING
	FL_STAMP(5);
	if (lane == 0 && fl_timing_out) {
#pragma unroll
		for (int i = 0; i < 6; i++) atomicAdd(&fl_timing_out[(SMALL ? LOGD : 6) * 8 + i], tm[i]);
		atomicAdd(&fl_timing_out[(SMALL ? LOGD : 6) * 8 + 6], (unsigned long long)ntr);
		atomicAdd(&fl_timing_out[(SMALL ? LOGD : 6) * 8 + 7], 1ull);
	}
#endif
	}; // run
	if (nw_all) run(FlWrapNo()); else run(FlWrapYes());
#undef FL_STAMP
}

// grid = (workgroups of all LDS scales, trace slices); a workgroup handles traces [slice*tps, min(ntr, (slice+1)*tps))
// FUSE: accST / accPS + slice * acc_stride are the [ncoef] planes that receive the slice's stacks of the unsplit scales.
// one workgroup of the LDS kernel: `bid` = its index among the lds_blocks workgroups of a trace slice, `slice` = trace slice
template <typename TIn, bool FUSE, int QT>
__device__ __forceinline__ void fwd_lds_workgroup(const unsigned bid, const unsigned slice, char *smem, const TIn *__restrict__ x, size_t ld,
                                                  unsigned ntr, unsigned tps, unsigned N, const ScaleDesc *__restrict__ sc, unsigned S,
                                                  const double2 *__restrict__ w, double2 *__restrict__ part, size_t npart,
                                                  double2 *__restrict__ accST, double2 *__restrict__ accPS, size_t acc_stride, const FuseFinal &ff)
{
	double2 *tL = (double2 *)smem;
	double *xL = (double *)(smem + FL_TAPS_BYTES_(QT));
	// scale of this workgroup: last s with lds_off[s] <= bid among the scales that use this kernel
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].lds_off <= bid) lo = mid; else hi = mid;
	}
	const ScaleDesc d = sc[lo];
	const unsigned wl = bid - d.lds_off;
	const unsigned chunk = wl / d.lds_bps, bb = wl - chunk * d.lds_bps; // one 64-phase chunk per workgroup (split == chunk)
	const unsigned t0 = slice * tps;
	const unsigned nt = (ntr - t0) < tps ? (ntr - t0) : tps;
	const TIn *x0 = x + (size_t)t0 * ld;
	const double2 *ws = w + d.tap_off;
	double *pout0 = (double *)(part + (size_t)t0 * npart + d.part_off + (size_t)chunk * d.Ns);
	double2 *aS = FUSE ? accST + (size_t)slice * acc_stride : nullptr, *aP = FUSE ? accPS + (size_t)slice * acc_stride : nullptr;
#if FL_ABLATE
	if (!((fl_class_mask >> (d.D >= 64 ? 6u : d.logDL)) & 1u)) return;
#endif
	if (d.D >= 64) { fwd_lds_body<TIn, 6, FUSE, FL_PASSES, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); return; }
	switch (d.logDL) {
	case 0: fwd_lds_body<TIn, 0, FUSE, FL_PASSES_FINE, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); break; // D = 1 (Mexican hat, uni): 8 coefficients per lane
	case 1: fwd_lds_body<TIn, 1, FUSE, FL_PASSES_FINE, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); break;
	case 2: fwd_lds_body<TIn, 2, FUSE, FL_PASSES_FINE, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); break;
	case 3: fwd_lds_body<TIn, 3, FUSE, FL_PASSES, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); break;
	case 4: fwd_lds_body<TIn, 4, FUSE, FL_PASSES, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); break;
	default: fwd_lds_body<TIn, 5, FUSE, FL_PASSES, QT>(x0, ld, nt, N, d, ws, pout0, npart, chunk, bb, tL, xL, aS, aP, ff, slice); break;
	}
}

// grid = (workgroups of all LDS scales, trace slices); a workgroup handles traces [slice*tps, min(ntr, (slice+1)*tps))
// FUSE: accST / accPS + slice * acc_stride are the [ncoef] planes that receive the slice's stacks of the unsplit scales.
#ifdef FL_MAXVGPR
#define FL_VGPR_ATTR __attribute__((amdgpu_num_vgpr(FL_MAXVGPR)))
#else
#define FL_VGPR_ATTR
#endif
template <typename TIn, bool FUSE, int QT>
__global__ void __launch_bounds__(FL_NT, 2) FL_VGPR_ATTR k_fwd_lds(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned tps, unsigned N,
                                                 const ScaleDesc *__restrict__ sc, unsigned S, const double2 *__restrict__ w,
                                                 double2 *__restrict__ part, size_t npart, double2 *__restrict__ accST,
                                                 double2 *__restrict__ accPS, size_t acc_stride, unsigned bid0, FuseFinal ff)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	// Workgroups are dispatched in blockIdx order and a launch is a few rounds of ~46 us workgroups, so the LAST round sets
	// the tail: walk the scales from coarse to fine -- the fine scales (D <= 4: one slot per wave, half the work) finish
	// the launch with short workgroups.
	// bid0: first workgroup of the launch in the plan's list (a launch may cover a sub-range of the scales: sharded finish)
	const unsigned bid = bid0 + (gridDim.x - 1u - blockIdx.x);
	fwd_lds_workgroup<TIn, FUSE, QT>(bid, blockIdx.y, smem, x, ld, ntr, tps, N, sc, S, w, part, npart, accST, accPS, acc_stride, ff);
}
