// fwd_mfma.h -- forward frame CWT on the FP64 matrix pipe of gfx950 (included by tspws_hip.hip).
//
// Why: tools/fma64_peak.hip and tools/mfma64_peak.hip (MI355X): a v_fma_f64 stream sustains 23 / 42 / 49 / 60 TFLOP/s
// at 1 / 2 / 4 / 16 waves per SIMD, while v_mfma_f64_4x4x4_4b_f64 sustains 72-76 TFLOP/s already at ONE wave per
// SIMD (16.5 cycles per instruction, 256 MACs each).  The decimating FIR bank maps onto that instruction without
// padding the contraction:
//
//   Y_v[n] = conj( sum_l x[(n D - c_v + l) mod N] w_v[l] )                      (cdotx.c:44-70)
//
//   * two voices of an octave (same D) form the 4 columns of B: j = (voice, re/im); the pair is aligned at an origin
//     cp >= c_v, its common tap index is l' = (4 kappa + k) D + m  (phase m, tap row 4 kappa + k),
//   * for a fixed phase the operand A is the Hankel matrix of the row sequence X_m[r] = x[r D + m - origin]:
//         A[i][k] = X_m[n + i + 4 kappa + k],      B[k][j] = w_j[(4 kappa + k) D + m]
//     so with TQ tiles of 4 outputs    C[a] += A[a + kappa] * B[kappa]     (a < TQ, kappa < Kq tap steps):
//     the A operand slides -- TQ + Kq - 1 LDS reads feed TQ Kq instructions -- and the sum over the phases happens in
//     the accumulator: no cross-lane reduction of the FIR at all,
//   * the 4 independent blocks of the instruction are 4 UNITS -- a unit is (trace, block of 4 TQ outputs) -- that share
//     the B tile (same phase, same taps) and bring their own rows: every lane ends up with finished outputs of one
//     unit, which it stores directly.
//
// Lane maps of v_mfma_f64_4x4x4_4b_f64 (tools/mfma64_layout.hip, measured):
//   A: lane = 16 k + 4 blk + i      B: lane = 16 k + 4 blk + j      C/D: lane = 16 i + 4 blk + j
//
// Work item = one workgroup of 4 waves: (group of <= 2 voice pairs of one octave, split = run of phase chunks, run of
// units).  A wave owns quads of 4 consecutive units.  The B tiles of the split are staged into LDS by the whole
// workgroup, a sub-split of <= 16 KB at a time (the next one waits in registers); between those barriers the waves run
// independently: per step (quad, chunk of Mc <= 4 phases) a wave stages the rows of its 4 units into its private image
// img[unit][phase][row] (unit pitch = 8 mod 32 doubles: the four units of an operand read hit disjoint banks), which
// both pairs read (their origins differ by whole rows), after fetching the rows of its NEXT step into registers.
//
// (This file is the GENERIC kernel: tap steps are run-time values.  Groups whose shape has a specialised kernel run on
// fwd_mfma_spec.hip instead -- see its header for the current numbers; the figures below are for this kernel alone.)
// Status (round 1): parity-green on every frame of the test suite (tests/test_hip_parity.py::test_forward_matrix_pipe_kernel)
// but OPT-IN (TSPWS_FWD_KERNEL=mfma): 10 x 131072 north-star transforms take 294 us against 198 us on the VALU kernels
// (k_fwd_lds + k_fwd_poly); 1024 x 32768 single-stage 8.0 ms against 6.1 ms.  PMC (profiles/r01_mfma_forward_pmc.txt):
// 9.2-10.5 M matrix instructions = 68 us of pipe time, i.e. the pipe is 25-30 % busy; per matrix instruction a wave still
// issues 2.3 VALU + 1.8 SALU + 0.4 LDS instructions and spends a third of its life in s_waitcnt, and the 10 KB image per
// wave caps the occupancy at 2 waves per SIMD.  Steady-state ablation (tools/mfma_sweep.sh, 256 traces, one octave: 448 us
// against 127 us of pipe time): without the (phase, pair) products 175 us, i.e. the products cost 273 us = 2.1x their
// pipe time (operand reads are not overlapped with the previous product's instructions); without the global x loads
// -93 us, without the staging writes -98 us (both include the wait for the prefetched rows: with Mc = 4 a row segment
// is 32 bytes -- 16 for float input -- so every 128-byte line is fetched four to eight times, from HBM once the traces
// outgrow the L2), without the stores -49 us.  Next steps: software-pipeline the operand reads of consecutive products,
// stage full lines (a workgroup-shared 16-phase row tile), trim the address arithmetic, float images for float input.
// Ablation macros: FM_ABL_NOX, FM_ABL_NOSTAGE, FM_ABL_NOLDSREAD, FM_ABL_NOMULT, FM_ABL_NOSTORE (make EXTRA_HIPFLAGS=-D...).
#pragma once

#include "fwd_mfma_types.h"

// B tiles of one pair inside its group's table, zero outside the filters / past phase D:
//   bt[chunk * chunk_stride + pair_off + ((ml Kq + kappa) 16 + 4 k + j)],   m = chunk Mc + ml
// (the tiles of both pairs of a chunk are adjacent, so the tiles of a run of chunks are ONE contiguous range)
__global__ void __launch_bounds__(256) k_build_bt(double *__restrict__ bt, const double2 *__restrict__ w, unsigned long long n, unsigned D,
                                                  unsigned Kq, long long cp, unsigned nv, unsigned long long tap0, unsigned L0, int c0,
                                                  unsigned long long tap1, unsigned L1, int c1, unsigned Mc, unsigned chunk_stride,
                                                  unsigned pair_off)
{
	const unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x; // over [chunk][ml][kappa][16]
	if (e >= n) return;
	const unsigned j = (unsigned)(e & 3), k = (unsigned)((e >> 2) & 3);
	const unsigned long long t = e >> 4;  // (chunk Mc + ml) Kq + kappa
	const unsigned kappa = (unsigned)(t % Kq);
	const unsigned long long m = t / Kq;
	const unsigned long long chunk = m / Mc;
	const unsigned ml = (unsigned)(m - chunk * Mc);
	const unsigned v = j >> 1;
	double val = 0.0;
	if (v < nv && m < D) {
		const long long lp = (long long)(4 * kappa + k) * D + (long long)m;
		const long long l = lp - (cp - (long long)(v ? c1 : c0));
		const unsigned L = v ? L1 : L0;
		if (l >= 0 && l < (long long)L) {
			const double2 tv = w[(v ? tap1 : tap0) + (unsigned long long)l];
			val = (j & 1) ? tv.y : tv.x;
		}
	}
	bt[chunk * chunk_stride + pair_off + ((unsigned long long)ml * Kq + kappa) * 16 + 4 * k + j] = val;
}

//   A: lane (k, blk, i) = X^{unit blk}_m[4 id + i + k + r_p]
//   B: lane (k, blk, j) = w_j[(4 kappa + k) D + m]              (the same tile in every block)
//   C: lane (i, blk, j) = output 4 a + i of unit blk, column j
// FIRST: the accumulators start from zero (first chunk of a unit) -- the zero is the instruction's C operand, so there
// is no separate clearing code and no copy at the loop head.
template <int TQ, int KQ, bool FIRST>
__device__ __forceinline__ void fm_pair_mult(double (&C)[TQ], const double *__restrict__ ap, const double *__restrict__ bp)
{
	double A[TQ + KQ - 1], B[KQ];
#ifdef FM_ABL_NOLDSREAD
#pragma unroll
	for (int kap = 0; kap < KQ; kap++) B[kap] = (double)(size_t)bp + kap;
#pragma unroll
	for (int id = 0; id < TQ + KQ - 1; id++) A[id] = (double)(size_t)ap + id;
#else
#pragma unroll
	for (int kap = 0; kap < KQ; kap++) B[kap] = bp[kap * 16];
#pragma unroll
	for (int id = 0; id < TQ + KQ - 1; id++) A[id] = ap[4 * id];
#endif
#pragma unroll
	for (int kap = 0; kap < KQ; kap++)
#pragma unroll
		for (int a = 0; a < TQ; a++)
			C[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[a + kap], B[kap], (FIRST && kap == 0) ? 0.0 : C[a], 0, 0, 0);
}

template <int TQ, bool FIRST>
__device__ __forceinline__ void fm_pair_mult_kq(double (&C)[TQ], const double *__restrict__ ap, const double *__restrict__ bp, const unsigned Kq)
{
	switch (Kq) {
	case 1: fm_pair_mult<TQ, 1, FIRST>(C, ap, bp); break;
	case 2: fm_pair_mult<TQ, 2, FIRST>(C, ap, bp); break;
	case 3: fm_pair_mult<TQ, 3, FIRST>(C, ap, bp); break;
	case 4: fm_pair_mult<TQ, 4, FIRST>(C, ap, bp); break;
	case 5: fm_pair_mult<TQ, 5, FIRST>(C, ap, bp); break;
	case 6: fm_pair_mult<TQ, 6, FIRST>(C, ap, bp); break;
	case 7: fm_pair_mult<TQ, 7, FIRST>(C, ap, bp); break;
	default: fm_pair_mult<TQ, 8, FIRST>(C, ap, bp); break;
	}
}

#define FM_BREG 8 /* B doubles per thread of a staged sub-split (<= 16 KB per workgroup) */

template <typename TIn, int TQ>
__device__ __forceinline__ void fwd_mfma_wg_body(const TIn *__restrict__ x, const size_t ld, const unsigned ntr, const unsigned N,
                                                 const FwdGroup &d, const double *__restrict__ bt, double2 *__restrict__ part,
                                                 const size_t npart, const unsigned split, const unsigned uc, double *__restrict__ smem)
{
	constexpr int NIMAX = (4 * TQ + 4 * FM_KQCAP + 3 + 15) / 16;  // staging iterations per unit at 4 phases x 16 rows per iteration
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned l_hi = lane >> 4, l_blk = (lane >> 2) & 3, l_lo = lane & 3;
	const unsigned Mc = d.Mc, P = d.P, Pu = d.Pu, nob = d.nob, D = d.D, Ns = d.Ns;
	const unsigned Kq0 = d.Kq[0], Kq1 = d.Kq[1];
	const bool two = d.NP > 1;
	const unsigned ml = lane & (Mc - 1), rs = lane >> d.logMc, RPI = 64u >> d.logMc;
	const unsigned NI = (d.RT + RPI - 1) / RPI;
	double *Bl = smem;                                       // staged B tiles of the current sub-split
	double *img = smem + d.bl_doubles + wv * (4 * Pu);       // this wave's image: [unit][phase][row]
	const double *ap0 = img + l_blk * Pu + l_lo + l_hi + d.rofs[0]; // A: i = lane & 3, k = lane >> 4, unit = blk
	const double *ap1 = img + l_blk * Pu + l_lo + l_hi + d.rofs[1];
	const unsigned b_lane = l_hi * 4 + l_lo;                 // B: k = lane >> 4, j = lane & 3
	double *stp = img + ml * P + rs;
	const unsigned rowstep = (unsigned)(((unsigned long long)RPI * D) % N);

	// chunks of this split, cut into sub-splits of css chunks
	const unsigned ch0 = split * d.cps;
	const unsigned nch = (d.MC - ch0) < d.cps ? (d.MC - ch0) : d.cps;
	const unsigned css = d.css, nss = (nch + css - 1) / css;
	// quads of this wave: units ubase + 4 (wv + 4 i) + b, b = 0..3; several quads only when there is one sub-split
	const unsigned U = ntr * nob, ubase = uc * d.upi;
	unsigned nquad = 0;
	{
		const unsigned uend = (ubase + d.upi) < U ? (ubase + d.upi) : U;
		if (ubase + 4 * wv < uend) nquad = (uend - ubase - 4 * wv + 15) / 16;
	}

	// ---- unit descriptors, one per lane: lane = 4 i + b ----
	unsigned s_first = 0, s_t = 0, s_blk = 0, s_valid = 0;
	if (lane < 4 * nquad) {
		const unsigned i = lane >> 2, b = lane & 3;
		unsigned u = ubase + 4 * (wv + 4 * i) + b;
		s_valid = u < U ? 1u : 0u;
		if (u >= U) u = U - 1; // rows of a valid unit; nothing is stored
		s_t = u / nob; s_blk = u - s_t * nob;
		const unsigned long long N64 = N;
		const unsigned long long cpm = (unsigned long long)(((d.cp % (long long)N64) + (long long)N64) % (long long)N64);
		const unsigned long long v = ((unsigned long long)s_blk * 4 * TQ % N64) * (D % N64) + (unsigned long long)ch0 * Mc % N64 + N64 - cpm;
		s_first = (unsigned)(v % N64); // sample of image row 0, phase 0 of the split's first chunk
	}
	const unsigned lane_c = (unsigned)(((unsigned long long)rs * D + ml) % N);
	const unsigned chstep = Mc % N;

	double xv[4][NIMAX];
	// rows of step (quad i, chunk offset coff = (ch Mc) mod N) -> registers.  Straight-line: independent loads (the few
	// past NI re-read valid addresses and are never stored), nothing conditional in between
	auto fetch = [&](unsigned i, unsigned coff) {
#pragma unroll
		for (int b = 0; b < 4; b++) {
			unsigned first = (unsigned)__builtin_amdgcn_readlane((int)s_first, (int)(4 * i + b)) + coff;
			if (first >= N) first -= N;
			const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)s_t, (int)(4 * i + b));
			const TIn *__restrict__ xt = x + (size_t)t * ld;
			unsigned idx = first + lane_c;
			if (idx >= N) idx -= N;
#pragma unroll
			for (int it = 0; it < NIMAX; it++) {
#ifdef FM_ABL_NOX
				xv[b][it] = (double)idx;
#else
				xv[b][it] = (double)xt[idx];
#endif
				idx += rowstep; if (idx >= N) idx -= N;
			}
		}
	};

	// B tiles of a sub-split: one contiguous run of the group's table ([chunk][pair][phase][kappa][16])
	double breg[FM_BREG];
	const unsigned bper0 = Mc * Kq0 * 16, bper = d.bper;
	auto bfetch = [&](unsigned ss) {
		const unsigned c0 = ss * css, nc = (nch - c0) < css ? (nch - c0) : css;
		const unsigned n = nc * bper;
		const double *src = bt + d.bt_off[0] + (size_t)(ch0 + c0) * bper;
#pragma unroll
		for (int i = 0; i < FM_BREG; i++) {
			const unsigned e = (unsigned)i * 256 + tid;
			breg[i] = src[e < n ? e : n - 1]; // the clamped tail is never read back
		}
	};

	double C0[TQ], C1[TQ];
	// the step being fetched: order = for ss, for quad, for chunk of the sub-split
	unsigned f_i = 0, f_ch = 0, f_coff = 0; // f_ch: chunk relative to the split
	bool f_more = nquad > 0;
	if (f_more) fetch(0, 0);
	bfetch(0);
	for (unsigned ss = 0; ss < nss; ss++) {
		const unsigned c0 = ss * css, ncs = (nch - c0) < css ? (nch - c0) : css;
		__syncthreads(); // every wave is done with the previous sub-split's tiles
#pragma unroll
		for (int i = 0; i < FM_BREG; i++) { const unsigned e = (unsigned)i * 256 + tid; if (e < d.bl_doubles) Bl[e] = breg[i]; }
		__syncthreads();
		if (ss + 1 < nss) bfetch(ss + 1);
		for (unsigned i = 0; i < nquad; i++) {
			for (unsigned ch = 0; ch < ncs; ch++) {
				const bool first_chunk = c0 + ch == 0, last_chunk = c0 + ch + 1 == nch;
				const bool rag = (ch0 + c0 + ch) * Mc + Mc > D; // phases past D are staged as zeros (they must not inject Inf * 0)
				__builtin_amdgcn_wave_barrier();
#ifndef FM_ABL_NOSTAGE
				if (rag) {
					const bool ok = (ch0 + c0 + ch) * Mc + ml < D;
#pragma unroll
					for (int b = 0; b < 4; b++)
#pragma unroll
						for (int it = 0; it < NIMAX; it++) if ((unsigned)it < NI) stp[(unsigned)b * Pu + (unsigned)it * RPI] = ok ? xv[b][it] : 0.0;
				} else {
#pragma unroll
					for (int b = 0; b < 4; b++)
#pragma unroll
						for (int it = 0; it < NIMAX; it++) if ((unsigned)it < NI) stp[(unsigned)b * Pu + (unsigned)it * RPI] = xv[b][it]; // rows past RT land in the plane's slack
				}
#else
				if (xv[0][0] == 1.234e300) stp[0] = xv[1][0] + xv[2][0] + xv[3][0];
#endif
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				// next step in the order (ss, quad, chunk)
				if (f_more) {
					const unsigned fss = f_ch / css; // sub-split of the fetched step
					const unsigned fend = ((fss + 1) * css < nch) ? (fss + 1) * css : nch;
					if (++f_ch == fend) {
						if (++f_i == nquad) { f_i = 0; if (fend == nch) f_more = false; }
						else f_ch = fss * css;
					}
					f_coff = f_ch * chstep; // <= D + Mc
					while (f_coff >= N) f_coff -= N;
					if (f_more) fetch(f_i, f_coff);
				}
#ifndef FM_ABL_NOMULT
				const double *bl0 = Bl + ch * bper + b_lane, *bl1 = bl0 + bper0;
				if (first_chunk) {
					fm_pair_mult_kq<TQ, true>(C0, ap0, bl0, Kq0);
					if (two) fm_pair_mult_kq<TQ, true>(C1, ap1, bl1, Kq1);
					for (unsigned ph = 1; ph < Mc; ph++) {
						fm_pair_mult_kq<TQ, false>(C0, ap0 + ph * P, bl0 + ph * Kq0 * 16, Kq0);
						if (two) fm_pair_mult_kq<TQ, false>(C1, ap1 + ph * P, bl1 + ph * Kq1 * 16, Kq1);
					}
				} else {
					for (unsigned ph = 0; ph < Mc; ph++) {
						fm_pair_mult_kq<TQ, false>(C0, ap0 + ph * P, bl0 + ph * Kq0 * 16, Kq0);
						if (two) fm_pair_mult_kq<TQ, false>(C1, ap1 + ph * P, bl1 + ph * Kq1 * 16, Kq1);
					}
				}
#endif
				if (last_chunk) { // C: lane = 16 i + 4 blk + j holds output 4 a + i of unit blk, column j
					const int src = (int)((4 * i + l_blk) << 2); // descriptor lane of this lane's unit (byte address for bpermute)
					const unsigned t = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)s_t);
					const unsigned blk = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)s_blk);
					const unsigned valid = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)s_valid);
					const unsigned j = l_lo, v = j >> 1;
					const unsigned nb = blk * 4 * TQ + l_hi;
					double2 *pt = part + (size_t)t * npart + (size_t)split * Ns;
#pragma unroll
					for (int p = 0; p < 2; p++) {
						if (p == 0 || two) {
							double *dst = (double *)(pt + d.po[2 * p + v]) + (j & 1) + (size_t)nb * 2;
							const bool act = valid && v < d.nv[p];
#pragma unroll
							for (int a = 0; a < TQ; a++) {
								const double val = p ? C1[a] : C0[a];
#ifdef FM_ABL_NOSTORE
								if (act && val == 1.234e300) dst[8 * a] = val;
#else
								if (act && nb + 4 * (unsigned)a < Ns) dst[8 * a] = (j & 1) ? -val : val; // conj
#endif
							}
						}
					}
				}
			}
		}
	}
}

template <typename TIn>
__global__ void __launch_bounds__(256) k_fwd_mfma(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N,
                                                  const FwdGroup *__restrict__ pd, unsigned ngroups, const FwdOffsets offs,
                                                  const double *__restrict__ bt, double2 *__restrict__ part, size_t npart)
{
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	double *smem = (double *)smem_raw;
	unsigned lo = 0, hi = ngroups;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (offs.off[mid] <= blockIdx.x) lo = mid; else hi = mid;
	}
	const FwdGroup &d = pd[lo];
	const unsigned il = blockIdx.x - offs.off[lo];
	const unsigned nuc = (ntr * d.nob + d.upi - 1) / d.upi;
	const unsigned uc = il % nuc, split = il / nuc;
	switch (d.TQ) {
	case 1: fwd_mfma_wg_body<TIn, 1>(x, ld, ntr, N, d, bt, part, npart, split, uc, smem); break;
	case 2: fwd_mfma_wg_body<TIn, 2>(x, ld, ntr, N, d, bt, part, npart, split, uc, smem); break;
	case 4: fwd_mfma_wg_body<TIn, 4>(x, ld, ntr, N, d, bt, part, npart, split, uc, smem); break;
	default: fwd_mfma_wg_body<TIn, 8>(x, ld, ntr, N, d, bt, part, npart, split, uc, smem); break;
	}
}
