// fwd_lds.h -- LDS-staged polyphase forward frame CWT for gfx950 (included by tspws_hip.hip).
//
// Same decomposition as fwd_poly.h (thread = 8 consecutive outputs of one phase, sliding register window),
// but the operands come from LDS: a 256-thread workgroup owns 8 "group slots" (4 waves x 2 passes) of ONE
// scale and ONE trace, stages the x window and the taps all of its slots need with one burst of independent
// coalesced loads, and then runs the FMA loop out of LDS with compile-time LDS offsets.
//
// Why it is written the way it is (PMC, round 1): on gfx950 a wave64 v_fma_f64 costs 4 cycles and an integer
// VALU op 2, so address arithmetic, predicates and 64-bit pointer math -- not the FP64 pipe -- were >75 % of the
// issued VALU instructions of the first versions.  Hence: the body is templated on log2(D) so every LDS
// address in the FMA loop is `one base VGPR + immediate`; staging has a predicate-free fast path (window fully
// inside the trace, all taps inside the filter) and a generic slow path for the few workgroups at the circular
// seam / filter end; the 64-lane phase reduction of D >= 64 goes through an LDS transpose (1 write + 1 read per
// value) instead of 5-instruction shuffle stages.
//
//   LOGD = 6 (D >= 64, any D): slot = output group g0+slot, lanes = 64 consecutive phases of the chunk;
//             LDS x image  xL[row][64]   row j <-> sample (g0*8 + qa + j) D + m0 + lane - c   (80 rows)
//             LDS taps     tL[q][64]     q <-> tap (qa + q) D + m0 + lane
//   LOGD < 6 (D = 2^LOGD):     lanes = (group, phase); a slot is 64/D groups; the x window of the workgroup is one
//             contiguous sample range stored with D doubles of padding per 8 D samples, so the 64/D groups of
//             a wave fall on distinct banks (group stride 9 D doubles).
//   stages  = (64-phase chunk) x (balanced tile of <= 16 taps-per-phase); accumulators live across stages.
//
// Used for scales with at least 8 output groups (N_s >= 64) and D >= 64 or a power of two; everything else
// (very coarse scales whose parallelism is only in the taps, odd small decimations) stays on k_fwd_poly.
#pragma once

#define FL_R 8
#define FL_QT 16
#define FL_TAPS_BYTES (FL_QT * 64 * 16)         /* 16 KiB */
#define FL_X_DOUBLES 5760                       /* >= 80*64, >= 4608*9/8 and the 4 x 1056 reduction scratch */
#define FL_LDS_BYTES (FL_TAPS_BYTES + FL_X_DOUBLES * 8)

template <typename TIn, int LOGD>
__device__ __forceinline__ void fwd_lds_body(const TIn *__restrict__ xt, const unsigned N, const ScaleDesc &d, const double2 *__restrict__ ws,
                                             double *__restrict__ pout, const unsigned split, const unsigned bb, double2 *tL, double *xL)
{
	constexpr int R = FL_R;
	constexpr bool SMALL = LOGD < 6;
	constexpr unsigned DC = 1u << (SMALL ? LOGD : 6);     // D when SMALL
	constexpr unsigned GW = SMALL ? (64u >> LOGD) : 1u;   // groups per wave-slot
	const unsigned D = SMALL ? DC : d.D;
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned g0 = bb * 8u * GW;                     // first output group of this workgroup
	const unsigned lane_m = SMALL ? (lane & (DC - 1)) : lane;
	const unsigned lane_g = SMALL ? (lane >> (SMALL ? LOGD : 0)) : 0;

	double ar[2][R], ai[2][R];
#pragma unroll
	for (int p = 0; p < 2; p++)
#pragma unroll
		for (int r = 0; r < R; r++) { ar[p][r] = 0; ai[p][r] = 0; }

	// taps-per-phase are cut into equal tiles of at most FL_QT (Q = 17 -> 9 + 8, not 16 + 1)
	const unsigned ntile = (d.Q + FL_QT - 1) / FL_QT, qt = (d.Q + ntile - 1) / ntile;
	const unsigned nchunks = SMALL ? 1u : d.cps;
	for (unsigned ci = 0; ci < nchunks; ci++) {
		const unsigned chunk = split * d.cps + ci;
		if (!SMALL && chunk >= d.MC) break;
		const unsigned m0 = SMALL ? 0u : chunk * 64u;
		for (unsigned qa = 0; qa < d.Q; qa += qt) {
			const unsigned qn = (d.Q - qa) < qt ? (d.Q - qa) : qt;
			__syncthreads(); // everyone is done reading the previous stage
			// ------------------------------------------------------------------ stage
			if (SMALL) {
				// contiguous window; always stage 18*256 samples (>= the (8 GW R + qn - 1) D needed)
				const long long base = ((long long)g0 * R + qa) * DC - d.c;
				const unsigned pt = tid + ((tid >> (3 + (SMALL ? LOGD : 0))) << (SMALL ? LOGD : 0)); // padded index of element tid; +288 per 256
				double xv[18];
				if (base >= 0 && base + 18 * 256 <= (long long)N) { // fast path: no circular wrap
					const TIn *src = xt + (unsigned)base + tid;
#pragma unroll
					for (int i = 0; i < 18; i++) xv[i] = (double)src[256 * i];
				} else {
					unsigned idx = wrap_index(base + tid, N);
					const unsigned step = 256u % N;
#pragma unroll
					for (int i = 0; i < 18; i++) {
						xv[i] = (double)xt[idx];
						idx += step; if (idx >= N) idx -= N;
					}
				}
				double2 tv[2]; // taps: qn * D <= 512 values, natural order
#pragma unroll
				for (int i = 0; i < 2; i++) {
					const unsigned e = tid + 256u * (unsigned)i, l = qa * DC + e;
					tv[i] = (e < qn * DC && l < d.L) ? ws[l] : make_double2(0.0, 0.0);
				}
#pragma unroll
				for (int i = 0; i < 18; i++) xL[pt + 288 * i] = xv[i];
#pragma unroll
				for (int i = 0; i < 2; i++) tL[tid + 256 * i] = tv[i];
			} else {
				// thread (wv, lane) stages rows wv, wv+4, ..., wv+76 of its own lane column (80 rows >= 8R+qn-1+8)
				const unsigned m = m0 + lane;
				const long long s_first = ((long long)g0 * R + qa) * D + m0 - d.c;          // row 0, lane 0
				const long long s_last = s_first + 79ll * D + 63;                           // row 79, lane 63
				const bool full = m0 + 63 < D;
				double xv[20];
				if (full && s_first >= 0 && s_last < (long long)N) { // fast path
					const TIn *src = xt + (unsigned)(s_first + (long long)wv * D) + lane;
					const unsigned stride = 4u * D;
#pragma unroll
					for (int i = 0; i < 20; i++) xv[i] = (double)src[(size_t)stride * i];
				} else {
					const bool mok = m < D;
					unsigned idx = wrap_index(s_first + (long long)wv * D + (mok ? lane : 0), N);
					const unsigned step = (unsigned)((4ull * D) % N);
#pragma unroll
					for (int i = 0; i < 20; i++) {
						xv[i] = mok ? (double)xt[idx] : 0.0;
						idx += step; if (idx >= N) idx -= N;
					}
				}
				double2 tv[4];
				const unsigned l0 = (qa + wv) * D + m;
				if (full && (qa + 15u) * D + m0 + 63 < d.L) { // every tap of the 16-row tile exists
#pragma unroll
					for (int i = 0; i < 4; i++) tv[i] = ws[l0 + 4u * D * (unsigned)i];
				} else {
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const unsigned q = wv + 4u * (unsigned)i, l = l0 + 4u * D * (unsigned)i;
						tv[i] = (m < D && q < qn && l < d.L) ? ws[l] : make_double2(0.0, 0.0);
					}
				}
				double *xdst = xL + wv * 64 + lane;
#pragma unroll
				for (int i = 0; i < 20; i++) xdst[256 * i] = xv[i];
				double2 *tdst = tL + wv * 64 + lane;
#pragma unroll
				for (int i = 0; i < 4; i++) tdst[256 * i] = tv[i];
			}
			__syncthreads();
			// ------------------------------------------------------------------ compute: two passes (group slots) per wave
#pragma unroll
			for (int p = 0; p < 2; p++) {
				const unsigned slot = (unsigned)p * 4u + wv;
				// one base per pass; every read below is base + compile-time offset
				const double *xb;
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
				double xw[R];
#pragma unroll
				for (int j = 0; j < R - 1; j++) xw[j] = xb[FL_XOFF(j)];
#pragma unroll
				for (int h = 0; h < FL_QT / R; h++) {
					if ((unsigned)(h * R) < qn) {
						double xn[R];
						double2 tn[R];
#pragma unroll
						for (int u = 0; u < R; u++) {
							xn[u] = xb[FL_XOFF(h * R + u + R - 1)];
							tn[u] = tb[(h * R + u) * XS];
						}
#pragma unroll
						for (int u = 0; u < R; u++) {
							if ((unsigned)(h * R + u) < qn) {
								xw[(u + R - 1) % R] = xn[u];
#pragma unroll
								for (int r = 0; r < R; r++) {
									ar[p][r] = fma(xw[(u + r) % R], tn[u].x, ar[p][r]);
									ai[p][r] = fma(xw[(u + r) % R], tn[u].y, ai[p][r]);
								}
							}
						}
					}
				}
#undef FL_XOFF
			}
		}
	}

	// ---------------------------------------------------------------------- combine the phase lanes, store the split partial
	if (SMALL) {
#pragma unroll
		for (int p = 0; p < 2; p++) {
			constexpr int NV = 2 * R;
			double v[NV];
#pragma unroll
			for (int r = 0; r < R; r++) { v[2 * r] = ar[p][r]; v[2 * r + 1] = ai[p][r]; }
			int n = NV;
			unsigned first = 0;
			ReduceScatter<NV, 0>::run(v, (unsigned)(SMALL ? LOGD : 0), lane, n, first);
			constexpr unsigned dup_mask = LOGD > 4 ? 0x10u : 0u; // NV = 16: bits 0..3 scatter, bit 4 (D = 32) duplicates
			const unsigned g = g0 + ((unsigned)p * 4u + wv) * GW + lane_g;
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
		// 64-lane reduction through a wave-private LDS transpose: row i (stride 65 doubles: conflict-free both
		// ways) holds value i of every lane; lane (o = lane&15, quarter = lane>>4) sums 16 entries of row o.
		__syncthreads(); // all waves are done with the x image
		double *scr = xL + wv * 1056;
#pragma unroll
		for (int p = 0; p < 2; p++) {
#pragma unroll
			for (int r = 0; r < R; r++) { scr[(2 * r) * 65 + lane] = ar[p][r]; scr[(2 * r + 1) * 65 + lane] = ai[p][r]; }
			const unsigned o = lane & 15, qd = lane >> 4;
			const double *src = scr + o * 65 + qd * 16;
			double sum = src[0];
#pragma unroll
			for (int t = 1; t < 16; t++) sum += src[t];
			sum += __shfl_xor(sum, 16, 64);
			sum += __shfl_xor(sum, 32, 64);
			const unsigned k = (g0 + (unsigned)p * 4u + wv) * R + (o >> 1);
			if (qd == 0 && k < d.Ns) pout[(size_t)k * 2 + (o & 1)] = (o & 1) ? -sum : sum; // conj
		}
	}
}

template <typename TIn>
__global__ void __launch_bounds__(256) k_fwd_lds(const TIn *__restrict__ x, size_t ld, unsigned N, const ScaleDesc *__restrict__ sc,
                                                 unsigned S, const double2 *__restrict__ w, double2 *__restrict__ part, size_t npart)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	double2 *tL = (double2 *)smem;
	double *xL = (double *)(smem + FL_TAPS_BYTES);
	// scale of this workgroup: last s with lds_off[s] <= blockIdx.x among the scales that use this kernel
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].lds_off <= blockIdx.x) lo = mid; else hi = mid;
	}
	const ScaleDesc d = sc[lo];
	const unsigned wl = blockIdx.x - d.lds_off;
	const unsigned split = wl / d.lds_bps, bb = wl - split * d.lds_bps;
	const TIn *xt = x + (size_t)blockIdx.y * ld;
	const double2 *ws = w + d.tap_off;
	double *pout = (double *)(part + (size_t)blockIdx.y * npart + d.part_off + (size_t)split * d.Ns);
	if (d.D >= 64) { fwd_lds_body<TIn, 6>(xt, N, d, ws, pout, split, bb, tL, xL); return; }
	switch (d.logDL) {
	case 0: fwd_lds_body<TIn, 0>(xt, N, d, ws, pout, split, bb, tL, xL); break;
	case 1: fwd_lds_body<TIn, 1>(xt, N, d, ws, pout, split, bb, tL, xL); break;
	case 2: fwd_lds_body<TIn, 2>(xt, N, d, ws, pout, split, bb, tL, xL); break;
	case 3: fwd_lds_body<TIn, 3>(xt, N, d, ws, pout, split, bb, tL, xL); break;
	case 4: fwd_lds_body<TIn, 4>(xt, N, d, ws, pout, split, bb, tL, xL); break;
	default: fwd_lds_body<TIn, 5>(xt, N, d, ws, pout, split, bb, tL, xL); break;
	}
}
