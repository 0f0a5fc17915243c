// fwd_tl.h -- "trace-lane" forward frame CWT + phase stack for MANY traces (included by forward.hip).
//
//   Y_s[k] = conj( sum_l x[(k D - c_s + l) mod N] w_s[l] )                  (cdotx.c:44-70)
//   ST += Y,  PS += Y / |Y|                                                 (ts_pws1f_lib.c:489-492)
//
// The single-stage stack (tspws_stacks_float, ts_pws1f_lib.c:466-499) transforms EVERY trace: the batch dimension is
// large, and with it the decomposition of k_fwd_lds (lanes = phases of one trace, a cross-lane reduction and two
// workgroup barriers per trace and scale) leaves half the FP64 pipe idle (FL_TIMING: ~50 % of the wave cycles in the FMA
// passes).  Here the 64 lanes of a wave are 64 TRACES:
//
//   input   = the batch transposed once, xT[n][trace] (k_transpose_traces): a row is the same sample of every trace
//   thread  = (trace, 8 consecutive outputs k of every voice of the octave): all taps of a filter are walked
//             SEQUENTIALLY, residue class by residue class (sample n = j D + rho), so
//               * there is no cross-lane reduction: a lane owns its trace's coefficients from the first tap to the last,
//               * the taps are the same for all lanes: one broadcast LDS read per step,
//               * the voices of the octave share the staged rows (same D, hence the same samples; see fwd_oct.h for the
//                 row / tap-row bookkeeping: c_v = a_v D + b_v, tap row q' holds tap (q' - 1) D + rho + b_v)
//   wave    = one group of 8 outputs; workgroup = 4 waves = 32 consecutive outputs of one (octave, 64-trace block)
//   step    = one residue rho: the rows (j0 + i) D + rho, i < XR, of the 64 traces (XR x 256 B) and the voices' tap rows
//             go to LDS (double-buffered: the next residue's rows travel while this one is computed, ONE barrier per
//             residue), then every wave runs its sliding 8-row register window over them: 16 FMAs per (x read, tap read)
//   output  = D <= TL_PMAX residues: the workgroup walks all of them, normalises its lanes' coefficients and reduces the
//             64 traces on the VALU into the block's ST / PS planes (slice planes of the fused path: FuseOut);
//             larger D: the residues are split over several workgroups whose per-trace partial sums go to a partial
//             buffer (tl layout) and k_accumulate_parts finishes them.
//
// Works for any D and N (circular wrap per row), float and double input, partially filled trace blocks (idle lanes
// contribute exact zeros and are skipped by the phase stack like the reference's all-zero traces).
#pragma once

#define TL_NT 256
#ifndef TL_VMAX
#define TL_VMAX 3            /* voices per work item: V = 4 as 2 + 2 (twice the workgroups; 499 x 16501 1.16 -> 1.00 ms, 2048 x 8192 1.56 -> 1.48), V = 5 as 3 + 2 either way */
#endif
#ifndef TL_WGS
#define TL_WGS 2             /* workgroups per CU the register budget is set for */
#endif
#define TL_QMAX 32           /* tap rows per voice */
#define TL_XRMAX 72          /* staged rows per residue (multiple of 4): 32 outputs + tap rows + spread of a_v */
#define TL_PMAX 256          /* residues one workgroup walks */
#define TL_NXV (TL_XRMAX * 64 / TL_NT)

struct TLItem {
	unsigned nv;                 // voices
	unsigned D, Ns;
	unsigned nkb;                // blocks of 32 outputs
	unsigned nsplit, pps;        // residue splits, residues per split
	unsigned XR;                 // staged rows (multiple of 4)
	unsigned amax;
	unsigned trows;              // tap rows of all voices
	unsigned fused;              // 1: single split, stacks written to the slice planes
	unsigned wg_off;             // first workgroup of the item in the launch
	unsigned kbw;                // consecutive output blocks one workgroup walks (small D: few residues per block)
	unsigned sc[TL_VMAX], QR[TL_VMAX], trow[TL_VMAX], a[TL_VMAX], b[TL_VMAX], L[TL_VMAX];
	unsigned long long tap_off[TL_VMAX], coef_off[TL_VMAX], part_off[TL_VMAX]; // part_off: tl partial layout [nsplit][Ns]
};

// xT[n][t] = x[t][n]; TP = traces padded to a multiple of 64 (pad lanes are zero).  64 x 64 tiles through LDS.
template <typename TIn>
__global__ void __launch_bounds__(256) k_transpose_traces(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N, unsigned TP,
                                                         TIn *__restrict__ xT)
{
	__shared__ TIn tile[64][65];
	const unsigned n0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
	const unsigned tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	for (unsigned r = ty; r < 64; r += 4) { // r = trace within the tile, tx = sample
		const unsigned t = t0 + r, n = n0 + tx;
		tile[r][tx] = (t < ntr && n < N) ? x[(size_t)t * ld + n] : (TIn)0;
	}
	__syncthreads();
	for (unsigned r = ty; r < 64; r += 4) { // r = sample within the tile, tx = trace
		const unsigned n = n0 + r;
		if (n < N) xT[(size_t)n * TP + t0 + tx] = tile[tx][r];
	}
}

template <typename TIn>
__global__ void __launch_bounds__(TL_NT, TL_WGS) k_fwd_tl(const TIn *__restrict__ xT, unsigned TP, unsigned ntr, unsigned N,
                                                     const TLItem *__restrict__ items, unsigned nitems, const double2 *__restrict__ w,
                                                     double2 *__restrict__ accST, double2 *__restrict__ accPS, size_t acc_stride,
                                                     double2 *__restrict__ part, size_t npart)
{
	constexpr int R = 8;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const unsigned bid = gridDim.x - 1u - blockIdx.x; // coarse (long) items first
	unsigned lo = 0, hi = nitems;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (items[mid].wg_off <= bid) lo = mid; else hi = mid;
	}
	const TLItem *__restrict__ o = items + lo;
	const unsigned D = o->D, Ns = o->Ns, nv = o->nv, XR = o->XR, amax = o->amax, trows = o->trows;
	const unsigned wl = bid - o->wg_off;
	const unsigned nkbg = (o->nkb + o->kbw - 1) / o->kbw; // workgroups per split
	const unsigned split = wl / nkbg, kb0 = (wl - split * nkbg) * o->kbw;
	const unsigned kb1 = (kb0 + o->kbw < o->nkb) ? kb0 + o->kbw : o->nkb;
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned tb = blockIdx.y;               // 64-trace block
	const unsigned rho_lo = split * o->pps, rho_hi = (rho_lo + o->pps < D) ? rho_lo + o->pps : D;
	const unsigned nph = rho_hi - rho_lo;
	// LDS: two buffers of { x rows [XR][64] TIn -- RAW, converted at the read --, tap rows [trows] double2 }
	constexpr unsigned ROWB = 64 * sizeof(TIn);
	const size_t buf_bytes = (size_t)XR * ROWB + (size_t)trows * sizeof(double2);
	auto xbuf = [&](unsigned b) { return (const TIn *)(smem + b * buf_bytes); };
	auto tbuf = [&](unsigned b) { return (double2 *)(smem + b * buf_bytes + (size_t)XR * ROWB); };

	// ---- staging of one step (output block kb, residue rho) --------------------------------------------------------------
	// The rows go from global memory STRAIGHT into LDS (global_load_lds_dwordx4: 1 KB per wave-instruction, no staging
	// registers, no LDS store pass): a lane fetches VEC consecutive traces of a row, LPR lanes cover a row, so the 64 lanes
	// of a wave fill RPW consecutive LDS rows (the destination is wave-uniform base + lane x 16 B: rows are contiguous),
	// the workgroup RPP rows per instruction.  XR is a multiple of 4, hence a wave-instruction is valid or idle as a whole.
	constexpr int VEC = 16 / (int)sizeof(TIn), LPR = 64 / VEC, RPW = 64 / LPR, RPP = TL_NT / LPR, NLD = (TL_XRMAX + RPP - 1) / RPP;
	typedef __attribute__((address_space(3))) void lds_void;
	typedef __attribute__((address_space(1))) const void glb_void;
	const unsigned srow = tid / LPR, ssub = (tid % LPR) * VEC; // first row and first trace of this thread's fetches
	const TIn *xcol = xT + (size_t)tb * 64 + ssub;
	unsigned tv = 0, tq = 0;
	const bool tap_thread = tid < trows;
	if (tap_thread) { while (tv + 1 < nv && o->trow[tv + 1] <= tid) tv++; tq = tid - o->trow[tv]; }
	const unsigned long long tap_base = o->tap_off[tv];
	const unsigned tap_L = o->L[tv], tap_b = o->b[tv];
	const unsigned stp = (unsigned)(((unsigned long long)RPP * D) % N);
	// Row / tap position of the stage to be loaded next, kept INCREMENTALLY: a residue step moves every row by one sample,
	// a block step by 32 D - (nph - 1)  (a `% N` per step would be a 64-bit software division of ~130 instructions).
	unsigned n_st;        // sample of this thread's first row: ((32 kb - amax - 1 + srow) D + rho) mod N
	long long l_st;       // tap index of this thread's tap-image element: (tq - 1) D + rho + b
	{
		long long s0 = ((long long)kb0 * 32 - (long long)amax - 1 + (long long)srow) * D + rho_lo;
		s0 %= (long long)N; if (s0 < 0) s0 += N;
		n_st = (unsigned)s0;
		l_st = ((long long)tq - 1) * D + rho_lo + tap_b;
	}
	const unsigned blk_stp = (unsigned)((32ull * D + (unsigned long long)N - (unsigned long long)((nph - 1) % N)) % N);
	auto next_residue = [&]() { n_st += 1u; if (n_st >= N) n_st -= N; l_st += 1; };
	auto next_block = [&]() { n_st += blk_stp; if (n_st >= N) n_st -= N; l_st -= (long long)(nph - 1); };
	auto load_stage = [&](double2 &tp, const unsigned b) {
		// sample of image row i: (j0 + i) D + rho  (mod N), j0 = 32 kb - amax - 1; this thread: rows srow, srow + RPP, ...
		unsigned n = n_st;
		asm volatile("" : "+v"(n)); // keeps the compiler from hoisting every row offset out of the step loop
		char *dst = smem + b * buf_bytes + (size_t)(wv * RPW) * ROWB; // this wave's first row (wave-uniform)
#pragma unroll
		for (int i = 0; i < NLD; i++) {
			if ((unsigned)i * RPP + wv * RPW < XR)
				__builtin_amdgcn_global_load_lds((glb_void *)(xcol + (size_t)n * TP), (lds_void *)(dst + (size_t)i * RPP * ROWB), 16, 0, 0);
			n += stp;
			if (n >= N) n -= N;
		}
		if (tap_thread) tp = (l_st >= 0 && l_st < (long long)tap_L) ? w[tap_base + (unsigned long long)l_st] : make_double2(0.0, 0.0);
	};
	// own rows landed + the tap image element written: after the workgroup barrier that follows, buffer b is complete
	auto store_stage = [&](const double2 tp, const unsigned b) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (tap_thread) tbuf(b)[tid] = tp;
	};

	// per-voice descriptors in registers: read once, not in every residue step
	unsigned vQR[TL_VMAX], vrow[TL_VMAX]; // (the tap images are packed: trow[v + 1] = trow[v] + QR[v])
#pragma unroll
	for (int v = 0; v < TL_VMAX; v++) {
		const bool on = (unsigned)v < nv;
		vQR[v] = on ? o->QR[v] : 0u; vrow[v] = on ? amax - o->a[v] : 0u;
	}
	double ar[TL_VMAX][R], ai[TL_VMAX][R];
	const unsigned t = tb * 64 + lane;            // this lane's trace
	const bool tlive = t < ntr;
	const unsigned o16 = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1); // element left by valu_reduce16
	double2 *pS = accST + (size_t)tb * acc_stride, *pP = accPS + (size_t)tb * acc_stride;

	// finished outputs of block kb: residue split -> per-trace partial sums [trace][part_off + split Ns + k]; else phase-
	// normalise per lane and add the 64 traces of the block (VALU reduce-scatter, 16 values at a time) into its planes
	auto emit = [&](const unsigned kb) {
		const unsigned kbase = kb * 32u + wv * R;
		if (kbase >= Ns) return;
		if (!o->fused) {
			if (tlive) {
#pragma unroll
				for (int v = 0; v < TL_VMAX; v++) {
					if ((unsigned)v < nv) {
						double2 *dst = part + (size_t)t * npart + o->part_off[v] + (size_t)split * Ns;
#pragma unroll
						for (int r = 0; r < R; r++) if (kbase + r < Ns) dst[kbase + r] = make_double2(ar[v][r], -ai[v][r]); // conj
					}
				}
			}
			return;
		}
#pragma unroll
		for (int v = 0; v < TL_VMAX; v++) {
			if ((unsigned)v < nv) {
				double st16[2 * R], ps16[2 * R];
#pragma unroll
				for (int r = 0; r < R; r++) {
					const double2 y = tlive ? make_double2(ar[v][r], -ai[v][r]) : make_double2(0.0, 0.0); // conj
					double2 u = make_double2(0.0, 0.0);
					add_unit_phasor(u, y); // zero / non-finite quotients are skipped like the reference (:491-492)
					st16[2 * r] = y.x; st16[2 * r + 1] = y.y;
					ps16[2 * r] = u.x; ps16[2 * r + 1] = u.y;
				}
				const double sm = valu_reduce16(st16, lane), q = valu_reduce16(ps16, lane);
				const unsigned k = kbase + (o16 >> 1);
				if ((lane & 3) == 0 && k < Ns) {
					((double *)(pS + o->coef_off[v] + k))[o16 & 1] = sm;
					((double *)(pP + o->coef_off[v] + k))[o16 & 1] = q;
				}
			}
		}
	};

	double2 tp = make_double2(0.0, 0.0);
	load_stage(tp, 0);
	unsigned cur = 0;
#if FL_TIMING
	unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter(), nsteps = 0;
#define TL_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tm[i] += n_ - tc; tc = n_; } while (0)
#else
#define TL_STAMP(i) do { } while (0)
#endif
	// flat sequence of steps (kb, rho): the next step's rows travel while this one is computed, across block boundaries too
	for (unsigned kb = kb0; kb < kb1; kb++) {
#pragma unroll
		for (int v = 0; v < TL_VMAX; v++)
#pragma unroll
			for (int r = 0; r < R; r++) { ar[v][r] = 0; ai[v][r] = 0; }
		const bool group_live = kb * 32u + wv * R < Ns;
		for (unsigned ph = 0; ph < nph; ph++) {
			TL_STAMP(5); // (end of the previous step's output / loop overhead)
			store_stage(tp, cur);
			TL_STAMP(2); // wait for the rows + LDS stores
			fl_lds_barrier(); // buffer `cur` complete; everyone is done with the other buffer (computed one step ago)
			TL_STAMP(1); // barrier
			if (ph + 1 < nph) { next_residue(); load_stage(tp, cur ^ 1u); }
			else if (kb + 1 < kb1) { next_block(); load_stage(tp, cur ^ 1u); }
			TL_STAMP(3); // issue of the next step's loads
#if FL_TIMING
			nsteps++;
#endif
			if (group_live) {
				const TIn *xi = xbuf(cur) + (size_t)wv * R * 64 + lane;
				const double2 *tbv0 = tbuf(cur);
#pragma unroll
				for (int v = 0; v < TL_VMAX; v++) {
					if ((unsigned)v < nv) {
						const unsigned QR = vQR[v];
						const TIn *xb = xi + (size_t)vrow[v] * 64;
						const double2 *tbv = tbv0;
						tbv0 += QR;
						double xw[R];
#pragma unroll
						for (int j = 0; j < R - 1; j++) xw[j] = (double)xb[j * 64];
						// Four tap steps at a time ("half turn" of the ring of R window registers): the LDS operands of a half turn are
						// requested one half turn AHEAD (two operand sets, A and B), the x values stay RAW until their step -- the conversion
						// is the move into the ring slot, placed (scheduling barriers) where the slot's old value has died.  As a hoisted
						// conversion + copy, and with the second half of the loop body conditional, the compiler spent 14 register moves
						// per 128 FMAs: 8 % of the kernel's VALU instructions (profiles/r04_valu_mix.txt).
#define TL_READ(S, hh) do { \
							_Pragma("unroll") for (int u = 0; u < 4; u++) { \
								xr##S[u] = xb[((hh) * 4 + u + R - 1) * 64]; \
								tn##S[u] = tbv[(hh) * 4 + u]; /* same address in every lane: broadcast */ \
							} \
						} while (0)
#define TL_HALF(S, h) do { \
							_Pragma("unroll") for (int u = 0; u < 4; u++) { \
								const int sidx = (h) * 4 + u; \
								__builtin_amdgcn_sched_barrier(0); \
								xw[(sidx + R - 1) % R] = (double)xr##S[u]; \
								_Pragma("unroll") for (int r = 0; r < R; r++) { \
									ar[v][r] = fma(xw[(sidx + r) % R], tn##S[u].x, ar[v][r]); \
									ai[v][r] = fma(xw[(sidx + r) % R], tn##S[u].y, ai[v][r]); \
								} \
							} \
							__builtin_amdgcn_sched_barrier(0); \
						} while (0)
						// QR is a multiple of 4.  Whole turns of the ring (8 steps) in the loop, so that the window registers are where they
						// were after every iteration; the odd half turn after it, where the window dies.  The last prefetch reads four rows /
						// taps past the voice's (inside the stage buffers or, beyond the allocation, zeros): never used.
						TIn xrA[4], xrB[4];
						double2 tnA[4], tnB[4];
						TL_READ(A, 0);
						unsigned sb = 0;
						for (; sb + R <= QR; sb += R) {
							TL_READ(B, 1);
							TL_HALF(A, 0);
							TL_READ(A, 2);
							TL_HALF(B, 1);
							xb += R * 64; tbv += R;
						}
						if (sb < QR) TL_HALF(A, 0);
#undef TL_READ
#undef TL_HALF
					}
				}
			}
			TL_STAMP(4); // FMA passes
			cur ^= 1u;
		}
		emit(kb);
	}
#if FL_TIMING
	TL_STAMP(5);
	if (lane == 0 && fl_timing_out) { // class 0 of the timing table; "traces" = residue steps here
#pragma unroll
		for (int i = 0; i < 6; i++) atomicAdd(&fl_timing_out[i], tm[i]);
		atomicAdd(&fl_timing_out[6], nsteps);
		atomicAdd(&fl_timing_out[7], 1ull);
	}
#endif
#undef TL_STAMP
}
