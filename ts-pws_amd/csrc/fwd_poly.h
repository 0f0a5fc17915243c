// fwd_poly.h -- register-tiled polyphase forward frame CWT for gfx950 (included by forward.hip).
//
//   Y_s[k] = conj( sum_l x[(k D - c + l) mod N] w_s[l] )                  (cdotx.c:44-70)
//
// Decomposition.  Write l = q D + m (phase m in [0,D), q in [0,Q), Q = ceil(L/D)).  Output k at tap
// (q,m) reads x[(k+q) D + m - c]: for a fixed phase the decimating FIR is an ordinary stride-1
// correlation between the "row" sequence j -> x[j D + m - c] and the Q-tap sub-filter q -> w[q D + m].
//
//   thread  = (output group g of R consecutive k, phase m, B traces)
//   lanes   = consecutive phases m  (so a wave reads 64 consecutive samples / taps: coalesced),
//             DL = min(nextpow2(D),64) lanes per group, 64/DL groups per wave
//   window  = R rows of x per trace held in registers, advanced one row per tap (R-way reuse of x,
//             R*B-way reuse of the tap); the q loop is unrolled R times so the rotation is static
//   phases  = D > 64: a thread walks `cps` chunks of 64 phases; the D/64 chunks of a scale are split
//             over `nsplit` waves so that every wave runs a similar number of steps (long filters at
//             coarse scales have few outputs: the parallelism comes from the taps)
//   reduce  = the DL phase-lanes of a group are combined with a shuffle reduce-scatter (each of the
//             2*B*R partial sums crosses the wave once), split partials go to `part` and are summed
//             by the accumulate / gather kernels in split order (deterministic, no atomics).
//
// FP64 FMA only; MFMA unused by design (skinny FIRs).  Works for any D and any N (explicit wrap).
#pragma once

#define FWD_R 8

__device__ __forceinline__ unsigned wrap_index(long long idx, unsigned N)
{
	if (idx < 0) idx += N;
	while (idx >= (long long)N) idx -= N;
	return (unsigned)idx;
}

// Reduce-scatter over the low `nbits` lane bits.  Stage J (lane bit J) halves the live value count:
// a lane whose bit is set keeps the upper half, its partner the lower half, each adds what the other
// sends.  All register indices are compile-time.  On return the lane holds `n` finished sums which are
// entries first .. first+n-1 of the original vector; once a single value is left the remaining bits are
// plain butterflies (those lanes end up with duplicates).
template <int NV, int J>
struct ReduceScatter {
	static __device__ __forceinline__ void run(double (&v)[NV], unsigned nbits, unsigned lane, int &n, unsigned &first)
	{
		constexpr int NCUR = NV >> J;
		if constexpr (NCUR > 1) {
			if ((unsigned)J < nbits) {
				constexpr int half = NCUR / 2;
				const unsigned mask = 1u << J;
				const bool up = (lane & mask) != 0;
#pragma unroll
				for (int i = 0; i < half; i++) {
					const double send = up ? v[i] : v[i + half];
					const double keep = up ? v[i + half] : v[i];
					v[i] = keep + __shfl_xor(send, (int)mask, 64);
				}
				n = half;
				if (up) first += (unsigned)half;
				ReduceScatter<NV, J + 1>::run(v, nbits, lane, n, first);
			}
		} else {
			for (unsigned bit = (unsigned)J; bit < nbits; bit++) v[0] += __shfl_xor(v[0], 1 << bit, 64);
		}
	}
};

#include "lane_reduce.h"

// ---- reduce-scatter of 16 doubles over the low LOGD lane bits on the VALU -------------------------------------------
// Same bookkeeping as ReduceScatter (a lane whose bit is set keeps the upper half of the live values; on return the
// lane holds `n` finished sums, elements first .. first+n-1), but the lane bits are visited from the highest down and
// every exchange is a permlane swap (bit 4) or a DPP move (bits 3..0): no ds_bpermute, i.e. no LDS-pipe traffic.
template <int NCUR, int CTRL_SEL>
__device__ __forceinline__ void dpp_rs_stage(double (&v)[16], const bool up)
{
	constexpr int half = NCUR / 2;
#pragma unroll
	for (int i = 0; i < half; i++) {
		const double send = up ? v[i] : v[i + half], keep = up ? v[i + half] : v[i];
		v[i] = keep + dpp_mov_f64(send, CTRL_SEL);
	}
}

template <int LOGD>
__device__ __forceinline__ void valu_rs16(double (&v)[16], const unsigned lane, int &n, unsigned &first)
{
	static_assert(LOGD >= 0 && LOGD <= 5, "group of at most 32 phase lanes");
	n = 16; first = 0;
	if constexpr (LOGD >= 5) { // bit 4: 16-lane rows, swap form (no selects)
#pragma unroll
		for (int i = 0; i < 8; i++) v[i] = swap_add_f64<true>(v[i], v[i + 8]);
		n = 8; if (lane & 16) first += 8;
	}
	if constexpr (LOGD >= 4) { // bit 3: row_ror:8
		if constexpr (LOGD >= 5) dpp_rs_stage<8, 0>(v, (lane & 8) != 0); else dpp_rs_stage<16, 0>(v, (lane & 8) != 0);
		n >>= 1; if (lane & 8) first += (unsigned)n;
	}
	if constexpr (LOGD >= 3) { // bit 2: row_half_mirror
		if constexpr (LOGD >= 5) dpp_rs_stage<4, 1>(v, (lane & 4) != 0);
		else if constexpr (LOGD == 4) dpp_rs_stage<8, 1>(v, (lane & 4) != 0);
		else dpp_rs_stage<16, 1>(v, (lane & 4) != 0);
		n >>= 1; if (lane & 4) first += (unsigned)n;
	}
	if constexpr (LOGD >= 2) { // bit 1: quad xor 2
		if constexpr (LOGD >= 5) dpp_rs_stage<2, 2>(v, (lane & 2) != 0);
		else if constexpr (LOGD == 4) dpp_rs_stage<4, 2>(v, (lane & 2) != 0);
		else if constexpr (LOGD == 3) dpp_rs_stage<8, 2>(v, (lane & 2) != 0);
		else dpp_rs_stage<16, 2>(v, (lane & 2) != 0);
		n >>= 1; if (lane & 2) first += (unsigned)n;
	}
	if constexpr (LOGD >= 1) { // bit 0: quad xor 1 (a plain butterfly once a single value is left)
		if constexpr (LOGD >= 5) v[0] += dpp_mov_f64(v[0], 3);
		else {
			if constexpr (LOGD == 4) dpp_rs_stage<2, 3>(v, (lane & 1) != 0);
			else if constexpr (LOGD == 3) dpp_rs_stage<4, 3>(v, (lane & 1) != 0);
			else if constexpr (LOGD == 2) dpp_rs_stage<8, 3>(v, (lane & 1) != 0);
			else dpp_rs_stage<16, 3>(v, (lane & 1) != 0);
			n >>= 1; if (lane & 1) first += (unsigned)n;
		}
	}
}

// ---- reductions that keep (re, im) pairs together ------------------------------------------------------------------
// The fused forward + phase-stack kernel normalises every coefficient right after the phase reduction, so a lane must end
// with BOTH components of its coefficients.  Same exchanges as valu_rs16 / valu_reduce16, but the reduce-scatter stops at
// one complex value (two doubles) and the remaining lane bits are plain butterflies on both components.
// On return the lane holds n >= 2 finished sums = elements first .. first+n-1 (first even) of (re0, im0, re1, im1, ...);
// lanes that differ only in the butterfly bits hold duplicates.
template <int LOGD>
__device__ __forceinline__ void valu_rs_cplx(double (&v)[16], const unsigned lane, int &n, unsigned &first)
{
	static_assert(LOGD >= 0 && LOGD <= 5, "group of at most 32 phase lanes");
	if constexpr (LOGD <= 3) { valu_rs16<LOGD>(v, lane, n, first); return; }
	n = 16; first = 0;
	if constexpr (LOGD == 5) { // bit 4: 16-lane rows
#pragma unroll
		for (int i = 0; i < 8; i++) v[i] = swap_add_f64<true>(v[i], v[i + 8]);
		n = 8; if (lane & 16) first += 8;
		dpp_rs_stage<8, 0>(v, (lane & 8) != 0); n = 4; if (lane & 8) first += 4;  // bit 3
		dpp_rs_stage<4, 1>(v, (lane & 4) != 0); n = 2; if (lane & 4) first += 2;  // bit 2 (mirror pairing)
		// bits 1, 0 (and the mirror's flip of them): butterflies -- after both every lane of the quad holds the sum
		v[0] += dpp_mov_f64(v[0], 2); v[1] += dpp_mov_f64(v[1], 2);
		v[0] += dpp_mov_f64(v[0], 3); v[1] += dpp_mov_f64(v[1], 3);
	} else {                   // LOGD == 4
		dpp_rs_stage<16, 0>(v, (lane & 8) != 0); n = 8; if (lane & 8) first += 8; // bit 3
		dpp_rs_stage<8, 1>(v, (lane & 4) != 0); n = 4; if (lane & 4) first += 4;  // bit 2
		dpp_rs_stage<4, 2>(v, (lane & 2) != 0); n = 2; if (lane & 2) first += 2;  // bit 1
		v[0] += dpp_mov_f64(v[0], 3); v[1] += dpp_mov_f64(v[1], 3);               // bit 0
	}
}

// 64 phase lanes: every lane ends with the complex element  first/2 = 4*b5 + 2*b4 + b3  in (re, im); the eight lanes of a
// half row hold the same pair.
__device__ __forceinline__ void valu_reduce_cplx64(const double (&v)[16], const unsigned lane, double &re, double &im, unsigned &first)
{
	double s8[8], s4[4];
#pragma unroll
	for (int i = 0; i < 8; i++) s8[i] = swap_add_f64<false>(v[i], v[i + 8]);   // lane bit 5
#pragma unroll
	for (int i = 0; i < 4; i++) s4[i] = swap_add_f64<true>(s8[i], s8[i + 4]);  // lane bit 4
	const bool up3 = (lane & 8) != 0;
	double s2[2];
#pragma unroll
	for (int i = 0; i < 2; i++) {                                              // lane bit 3
		const double send = up3 ? s4[i] : s4[i + 2], keep = up3 ? s4[i + 2] : s4[i];
		s2[i] = keep + dpp_mov_f64(send, 0);
	}
	// bits 2, 1, 0: mirror + two quad butterflies = the sum over the eight lanes of the half row
#pragma unroll
	for (int i = 0; i < 2; i++) {
		s2[i] += dpp_mov_f64(s2[i], 1);
		s2[i] += dpp_mov_f64(s2[i], 2);
		s2[i] += dpp_mov_f64(s2[i], 3);
	}
	re = s2[0]; im = s2[1];
	first = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2;
}

// One wave of the direct kernel: R outputs per thread (8, or 16 for the scales flagged r16: 64 phase lanes and at least 16
// outputs -- one x value and one tap per 2 R FMAs and trace, i.e. half the operand bytes per FMA; the kernel is bound by
// the L2 operand stream: without the tap loads the call is 10 us shorter), BS tap steps per load block.
// BUF: the operands come through raw buffer loads -- resource = the trace / the scale's taps (uniform: SGPRs), ONE 32-bit byte offset
// per lane and stream.  The circular wrap of a row is add / subtract / unsigned min on that offset, a tap past the filter end or of an
// idle phase lane is an offset beyond the resource's range (the load returns zeros): 4 integer VALU ops per tap step instead of ~13
// (64-bit address per trace, index select, compare, four selects that zero the tap) beside the step's 16 B FMAs -- every VALU op
// competes with the 4-cycle FMAs (profiles/r04_valu_mix.txt: 63 % of this kernel's VALU instructions were FMAs on cfg2).
template <typename TIn, int B, int R, int BS, bool BUF>
__device__ __forceinline__ void fwd_poly_wave(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N, const ScaleDesc &d,
                                              const double2 *__restrict__ w, double2 *__restrict__ part, size_t npart, const unsigned wl,
                                              const unsigned lane)
{
	static_assert(R % BS == 0, "a ring of R window registers is walked in blocks of BS steps");
	const unsigned split = wl / d.ngw, gb = wl - split * d.ngw;
	const unsigned lane_m = lane & (d.DL - 1);
	const unsigned g = gb * (64u / d.DL) + (lane >> d.logDL);
	const unsigned k0 = g * R;
	const double2 *ws = w + d.tap_off;

	const TIn *xb[B];
#pragma unroll
	for (int b = 0; b < B; b++) {
		unsigned t = blockIdx.y * B + b;
		if (t >= ntr) t = ntr - 1;
		xb[b] = x + (size_t)t * ld;
	}
	double ar[B][R], ai[B][R];
#pragma unroll
	for (int b = 0; b < B; b++)
#pragma unroll
		for (int r = 0; r < R; r++) { ar[b][r] = 0; ai[b][r] = 0; }

	const unsigned Dw = d.D % N; // row advance, pre-reduced so one conditional subtraction re-wraps (D may exceed N on tiny traces)
	__amdgpu_buffer_rsrc_t rx[B], rw;
	const unsigned Nb = N * (unsigned)sizeof(TIn), Dwb = Dw * (unsigned)sizeof(TIn), D16 = d.D * 16u;
	if constexpr (BUF) {
#pragma unroll
		for (int b = 0; b < B; b++) rx[b] = __builtin_amdgcn_make_buffer_rsrc((void *)xb[b], 0, Nb, 0x00020000);
		rw = __builtin_amdgcn_make_buffer_rsrc((void *)ws, 0, d.L * 16u, 0x00020000);
	}
	auto ldx = [&](const int b, const unsigned row) -> double { // row: sample index, or its byte offset (BUF)
		if constexpr (BUF) {
			if constexpr (sizeof(TIn) == 8) return (double)__builtin_bit_cast(TIn, __builtin_amdgcn_raw_buffer_load_b64(rx[b], row, 0, 0));
			else return (double)__builtin_bit_cast(TIn, __builtin_amdgcn_raw_buffer_load_b32(rx[b], row, 0, 0));
		} else return (double)xb[b][row];
	};
	for (unsigned ci = 0; ci < d.cps; ci++) {
		const unsigned chunk = split * d.cps + ci;
		if (chunk >= d.MC) break;
		const unsigned m = chunk * 64 + lane_m;
		const bool mvalid = m < d.D;
		const unsigned mm = mvalid ? m : 0; // idle phase lanes read valid addresses, contribute nothing
		// row j of this thread is x[(k0 + j) D + m - c  (mod N)]; keep the wrapped index incrementally
		unsigned row = wrap_index((long long)k0 * d.D + mm - d.c, N);
		if constexpr (BUF) row *= (unsigned)sizeof(TIn);
		double xw[B][R];
#pragma unroll
		for (int j = 0; j < R - 1; j++) {
#pragma unroll
			for (int b = 0; b < B; b++) xw[b][j] = ldx(b, row);
			if constexpr (BUF) { row += Dwb; row = min(row, row - Nb); }
			else { row += Dw; if (row >= N) row -= N; }
		}
		unsigned l = mm;
		unsigned tb = mvalid ? mm * 16u : 0x80000000u; // BUF: byte offset of the next tap (idle lanes: out of range for good)
		for (unsigned q = 0; q < d.Q; q += R) {
#pragma unroll
			for (int h = 0; h < R / BS; h++) {
				if (q + (unsigned)(h * BS) < d.Q) { // wave-uniform
					// issue every load of the block first (BS rows per trace + BS taps), then BS tap steps of FMAs
					const unsigned nsteps = d.Q - q - (unsigned)(h * BS); // >= BS for a full block
					double xn[B][BS];
					double2 tp[BS];
#pragma unroll
					for (int u = 0; u < BS; u++) {
#pragma unroll
						for (int b = 0; b < B; b++) xn[b][u] = ldx(b, row);
						if constexpr (BUF) {
							row += Dwb; row = min(row, row - Nb);
							const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rw, tb, 0, 0); // past the filter end / idle lane: zeros
							tp[u] = make_double2(__hiloint2double((int)raw[1], (int)raw[0]), __hiloint2double((int)raw[3], (int)raw[2]));
							tb += D16;
						} else {
							row += Dw; if (row >= N) row -= N;
							const unsigned lu = l + (unsigned)(h * BS + u) * d.D;
							const bool ok = mvalid && lu < d.L;
							tp[u] = ws[ok ? lu : 0];
							if (!ok) tp[u] = make_double2(0.0, 0.0); // taps past the filter end / idle lanes contribute exactly nothing
						}
					}
#pragma unroll
					for (int u = 0; u < BS; u++) {
						if ((unsigned)u < nsteps) { // wave-uniform: only the last, partial block skips steps
							constexpr int dummy = 0; (void)dummy;
							const int sidx = h * BS + u; // compile-time after unrolling
#pragma unroll
							for (int b = 0; b < B; b++) xw[b][(sidx + R - 1) % R] = xn[b][u];
#pragma unroll
							for (int b = 0; b < B; b++)
#pragma unroll
								for (int r = 0; r < R; r++) {
									ar[b][r] = fma(xw[b][(sidx + r) % R], tp[u].x, ar[b][r]);
									ai[b][r] = fma(xw[b][(sidx + r) % R], tp[u].y, ai[b][r]);
								}
						}
					}
				}
			}
			l += (unsigned)R * d.D;
		}
	}

	// ---- combine the DL phase lanes of each group ----
	if (d.logDL == 6) { // 64 phase lanes (every coarse scale): VALU reduction, one trace at a time, no LDS traffic
		const unsigned o = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
#pragma unroll
		for (int b = 0; b < B; b++) {
#pragma unroll
			for (int half = 0; half < R / 8; half++) { // 16 values (8 complex outputs) per reduction
				double v16[16];
#pragma unroll
				for (int r = 0; r < 8; r++) { v16[2 * r] = ar[b][half * 8 + r]; v16[2 * r + 1] = ai[b][half * 8 + r]; }
				const double sum = valu_reduce16(v16, lane);
				const unsigned t = blockIdx.y * B + b, k = k0 + (unsigned)(half * 8) + (o >> 1);
				if ((lane & 3) == 0 && t < ntr && k < d.Ns) {
					double *dst = (double *)(part + (size_t)t * npart + d.part_off + (size_t)split * d.Ns);
					dst[(size_t)k * 2 + (o & 1)] = (o & 1) ? -sum : sum; // conj
				}
			}
		}
		return;
	}
	if constexpr (R == 8) {
	// fewer than 64 phase lanes per group: shuffle reduce-scatter over lane bits 0..logDL-1
	constexpr int NV = 2 * B * R;
	double v[NV];
#pragma unroll
	for (int b = 0; b < B; b++)
#pragma unroll
		for (int r = 0; r < R; r++) { v[(b * R + r) * 2] = ar[b][r]; v[(b * R + r) * 2 + 1] = ai[b][r]; }
	int n = NV;
	unsigned first = 0; // index of v[0] in the original numbering
	ReduceScatter<NV, 0>::run(v, d.logDL, lane, n, first);
	// lane bits at or above log2(NV) were plain butterflies: those lanes carry duplicates, the lowest writes
	unsigned dup_mask = 0;
	for (unsigned bit = 0; bit < d.logDL; bit++) if ((NV >> bit) <= 1) dup_mask |= 1u << bit;
	if (lane & dup_mask) return;
	double *pout[B];
#pragma unroll
	for (int b = 0; b < B; b++) {
		const unsigned t = blockIdx.y * B + b;
		pout[b] = (t < ntr) ? (double *)(part + (size_t)t * npart + d.part_off + (size_t)split * d.Ns) : nullptr;
	}
#pragma unroll
	for (int i = 0; i < NV; i++) {
		if (i < n) {
			const unsigned id = first + i;          // ((b*R + r)*2 + ri)
			const unsigned ri = id & 1, r = (id >> 1) % R, b = (id >> 1) / R;
			const unsigned k = k0 + r;
			if (k < d.Ns) {
				double *dst = nullptr;
#pragma unroll
				for (int bb = 0; bb < B; bb++) if (b == (unsigned)bb) dst = pout[bb];
				if (dst) dst[(size_t)k * 2 + ri] = ri ? -v[i] : v[i]; // conj
			}
		}
	}
	} // R == 8 (scales with fewer than 64 phase lanes are never flagged r16)
}

template <typename TIn, int B>
__global__ void __launch_bounds__(256) k_fwd_poly(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N,
                                                  const ScaleDesc *__restrict__ sc, unsigned S, const double2 *__restrict__ w,
                                                  double2 *__restrict__ part, size_t npart, unsigned total_waves, unsigned wid0 = 0)
{
	const unsigned lane = threadIdx.x & 63;
	// readfirstlane makes the wave index provably uniform: the scale lookup and the whole descriptor then
	// live in SGPRs (scalar loads, scalar branches) instead of VGPRs.  wid0 / total_waves: the launch's range of the wave list
	const unsigned wid = wid0 + blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (wid >= total_waves) return;
	// scale of this wave: last s with wave_off[s] <= wid
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].wave_off <= wid) lo = mid; else hi = mid;
	}
	const ScaleDesc d = sc[lo];
	// buffer-load form: byte offsets of a trace and of the scale's taps (+ the steps past its end a wave may request) stay below 2^31
	// (float traces: the many-trace single-stage batches, where this kernel sets the call's duration beside k_fwd_tl -- cfg2 3.27 -> 3.12 ms
	// with the register-move-free k_fwd_tl loop; FP64 partial stacks run it beside k_fwd_lds, off the critical path: cfg3 unchanged, cfg4 + 0.6 %)
	const bool buf = sizeof(TIn) == 4 && (unsigned long long)N * sizeof(TIn) < 0x80000000ull && ((unsigned long long)d.L + (unsigned long long)(d.Q + 16u) * d.D + 64u) * 16ull < 0x80000000ull;
	if constexpr (sizeof(TIn) == 4) {
		if (buf) {
			if (d.r16) fwd_poly_wave<TIn, B, 16, 4, true>(x, ld, ntr, N, d, w, part, npart, wid - d.wave_off, lane);
			else fwd_poly_wave<TIn, B, FWD_R, FWD_R, true>(x, ld, ntr, N, d, w, part, npart, wid - d.wave_off, lane);
			return;
		}
	}
	if (d.r16) fwd_poly_wave<TIn, B, 16, 4, false>(x, ld, ntr, N, d, w, part, npart, wid - d.wave_off, lane);
	else fwd_poly_wave<TIn, B, FWD_R, FWD_R, false>(x, ld, ntr, N, d, w, part, npart, wid - d.wave_off, lane);
}

// Y[b][coef] = sum over the scale's split partials (plain coefficient layout; API / tests)
__global__ void __launch_bounds__(256) k_gather_parts(const double2 *__restrict__ part, size_t npart, const ScaleDesc *__restrict__ sc,
                                                      unsigned S, double2 *__restrict__ Y, size_t ncoef)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= ncoef) return;
	const unsigned s = find_scale(sc, S, i, false);
	const ScaleDesc d = sc[s];
	const double2 *p = part + (size_t)blockIdx.y * npart + d.part_off + (i - d.coef_off);
	double2 a = p[0];
	for (unsigned sp = 1; sp < d.nsplit; sp++) { const double2 t = p[(size_t)sp * d.Ns]; a.x += t.x; a.y += t.y; }
	Y[(size_t)blockIdx.y * ncoef + i] = a;
}

// ST += sum_b Y_b ; PS += sum_b Y_b/|Y_b|  straight from the split partials (ts_pws1f_lib.c:489-492); scales the fused
// forward kernel stacked itself (fuse_ok) are skipped or combined from its slice planes.
// One block works on ONE scale (acc2_off[s] = its first block), so the scale lookup and descriptor reads are
// wave-uniform scalar work.  wa.OUT != nullptr (the call finishes the stacks: every trace of a single batch): the thread
// that completes a coefficient also writes its weighted value -- no separate pass over the coefficients.
// Scales without splits: 256 coefficients per block, one thread each.  Split scales (coarse scales: few coefficients,
// up to 32 partials each): 4 coefficients per block, a wave each -- otherwise a handful of threads would walk hundreds of
// dependent-latency loads and set the kernel's duration.
// PREFIX: the convergence curves' instantiation (weighted coefficients after every trace: WeightArgs::OUTP); the stacks' own
// instantiation carries none of its registers and code.
template <bool PREFIX>
__global__ void __launch_bounds__(256) k_accumulate_parts(const double2 *__restrict__ part, size_t npart, const ScaleDesc *__restrict__ sc,
                                                          unsigned S, unsigned ntr, double2 *__restrict__ ST, double2 *__restrict__ PS,
                                                          int zero_first, int fused, const double2 *__restrict__ fzST,
                                                          const double2 *__restrict__ fzPS, size_t fz_stride, unsigned nslices,
                                                          size_t y_part, size_t y_stack, int many, WeightArgs wa, unsigned blk0,
                                                          size_t y_fz, const unsigned *__restrict__ rowmap)
{
	const unsigned bx = blk0 + blockIdx.x; // (blk0: the launch may cover a sub-range of the scales)
	// blockIdx.y = independent stack (jackknife replica): its ntr transformed traces start y_part further in `part`, its
	// ST / PS y_stack further (the fused forward kernel wrote the fuse_ok scales there directly: fused == 1)
	part += (size_t)blockIdx.y * y_part; ST += (size_t)blockIdx.y * y_stack; PS += (size_t)blockIdx.y * y_stack;
	// y_fz: the slice planes of stack blockIdx.y start y_fz further (stacks whose slices interleave: the staged masked replicas);
	// rowmap (few-trace branches): transformed trace b of stack y sits rowmap[y ntr + b] trace strides into `part` (y_part unused)
	if (fzST) { fzST += (size_t)blockIdx.y * y_fz; fzPS += (size_t)blockIdx.y * y_fz; }
	if (rowmap) rowmap += (size_t)blockIdx.y * ntr;
	if (wa.Mv) wa.M = wa.Mv[blockIdx.y]; // (replicas side by side: each with its own trace count and weighted-coefficient set)
	// stacks that are only needed as weighted coefficients are not written (few-trace branches below; zero_first callers only)
	const bool planes = wa.planes_batch == -1 || (int)blockIdx.y == wa.planes_batch;
	if (wa.OUT) wa.OUT += (size_t)blockIdx.y * wa.out_stride;
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].acc2_off <= bx) lo = mid; else hi = mid;
	}
	const unsigned Ns = sc[lo].Ns, nsplit = sc[lo].nsplit;
#ifdef ACC_ABLATE
	if (ACC_ABLATE == 1 && fused && sc[lo].fuse_ok) return; // timing ablation: no work for the scales the forward kernel stacked
	if (ACC_ABLATE == 2 && !(fused && sc[lo].fuse_ok)) return; // ... only those
#endif
	if (fused && sc[lo].fuse_ok) {
		// the forward kernel already stacked this scale: fused == 1, straight into ST / PS (nothing left to do);
		// fused == 2, one plane pair per trace slice, added here in slice order; fused == 3: it also completed and weighted them
		if (fused == 3 || (fused == 1 && !wa.OUT)) return;
		const unsigned k = (bx - sc[lo].acc2_off) * 256u + threadIdx.x;
		if (k >= Ns) return;
		const size_t i = sc[lo].coef_off + k;
		if (fused == 1) { wa.OUT[i] = weight_value(ST[i], PS[i], wa.mode, wa.K, wa.M, wa.wu); return; }
		double2 st = make_double2(0, 0), ps = make_double2(0, 0);
		if (!zero_first) { st = ST[i]; ps = PS[i]; }
		for (unsigned j = 0; j < nslices; j++) {
			const double2 a = fzST[(size_t)j * fz_stride + i], b = fzPS[(size_t)j * fz_stride + i];
			st.x += a.x; st.y += a.y; ps.x += b.x; ps.y += b.y;
		}
		if (planes) { ST[i] = st; PS[i] = ps; }
		if (wa.OUT) wa.OUT[i] = weight_value(st, ps, wa.mode, wa.K, wa.M, wa.wu);
		return;
	}
	const bool wide = nsplit > 1 || (many && !sc[lo].fuse_ok); // (the many-trace tables give every unfused scale the wave-per-coefficient geometry)
	if (wide && many) {
		// many traces, few coefficients (coarse / residue-split scales of a single-stage batch; table geometry: 4 coefficients
		// per block): a wave per coefficient, its 64 lanes take every 64th TRACE (all splits of it), then a wave reduction --
		// the dependent-load chain is ntr / 64 long and there are Ns / 4 blocks instead of Ns / 32
		const unsigned k = (bx - sc[lo].acc2_off) * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63;
		if (k >= Ns) return;
		const size_t i = sc[lo].coef_off + k;
		const double2 *p0 = part + sc[lo].part_off + k;
		double2 st = make_double2(0, 0), ps = make_double2(0, 0);
		for (unsigned b = lane; b < ntr; b += 64) {
			const double2 *p = p0 + (size_t)b * npart;
			double2 v = make_double2(0.0, 0.0);
			for (unsigned sp = 0; sp < nsplit; sp++) { const double2 t = p[(size_t)sp * Ns]; v.x += t.x; v.y += t.y; }
			st.x += v.x; st.y += v.y;
			add_unit_phasor(ps, v);
		}
		st.x = wave_sum(st.x); st.y = wave_sum(st.y); ps.x = wave_sum(ps.x); ps.y = wave_sum(ps.y);
		if (lane == 0) {
			if (!zero_first) { const double2 a = ST[i], b = PS[i]; st.x += a.x; st.y += a.y; ps.x += b.x; ps.y += b.y; }
			ST[i] = st; PS[i] = ps;
			if (wa.OUT) wa.OUT[i] = weight_value(st, ps, wa.mode, wa.K, wa.M, wa.wu);
		}
		return;
	}
	if (wide) {
		// few traces, few coefficients, up to 32 partials each (table geometry: 4 coefficients per block): a wave per
		// coefficient, its lanes are (8 traces) x (8 lanes that share the partials of a trace: up to four independent loads per
		// lane and round, combined with three shuffles).  The traces of a round are then added in trace order from their
		// groups' lanes (v_readlane: the sums are wave-uniform) -- ntr / 8 dependent memory round trips instead of ntr, the
		// additions and their order are those of the one-trace-at-a-time form.
		const unsigned k = (bx - sc[lo].acc2_off) * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63;
		if (k >= Ns) return;
		const unsigned sub = lane & 7, bt = lane >> 3;
		const size_t i = sc[lo].coef_off + k;
		const double2 *p0 = part + sc[lo].part_off + k;
		double2 st = make_double2(0, 0), ps = make_double2(0, 0);
		if (!zero_first) { st = ST[i]; ps = PS[i]; }
		for (unsigned b0 = 0; b0 < ntr; b0 += 8) {
			const bool on = b0 + bt < ntr;
			const unsigned tb_ = on ? b0 + bt : 0u;
			const double2 *p = p0 + (size_t)(rowmap ? rowmap[tb_] : tb_) * npart;
			double2 v = make_double2(0.0, 0.0);
			for (unsigned sp = sub; sp < nsplit; sp += 32) {
				double2 t[4];
#pragma unroll
				for (int j = 0; j < 4; j++) t[j] = (on && sp + (unsigned)j * 8u < nsplit) ? p[(size_t)(sp + (unsigned)j * 8u) * Ns] : make_double2(0.0, 0.0);
#pragma unroll
				for (int j = 0; j < 4; j++) { v.x += t[j].x; v.y += t[j].y; }
			}
#pragma unroll
			for (int o = 1; o < 8; o <<= 1) { v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); }
			const unsigned nb = ntr - b0 < 8u ? ntr - b0 : 8u;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				if ((unsigned)j < nb) {
					const double2 vj = make_double2(readlane_f64(v.x, 8 * j), readlane_f64(v.y, 8 * j));
					st.x += vj.x; st.y += vj.y;
					add_unit_phasor(ps, vj);
					if (PREFIX && wa.OUTP && lane == 0) {
						const unsigned cnt = wa.k0 + b0 + (unsigned)j + 1u;
						wa.OUTP[(size_t)(b0 + (unsigned)j) * wa.outp_stride + i] = weight_value(st, ps, cnt == 1 ? wa.mode1 : wa.mode, (double)cnt, (double)cnt, wa.wu);
					}
				}
			}
		}
		if (lane == 0) {
			if (planes) { ST[i] = st; PS[i] = ps; }
			if (wa.OUT) wa.OUT[i] = weight_value(st, ps, wa.mode, wa.K, wa.M, wa.wu);
		}
		return;
	}
	// scales without splits that the forward kernel did not stack itself: one thread per coefficient
	const unsigned k = (bx - sc[lo].acc2_off) * 256u + threadIdx.x;
	if (k >= Ns) return;
	const size_t i = sc[lo].coef_off + k;
	const double2 *p0 = part + sc[lo].part_off + k;
	double2 st = make_double2(0, 0), ps = make_double2(0, 0);
	if (!zero_first) { st = ST[i]; ps = PS[i]; }
	for (unsigned b0 = 0; b0 < ntr; b0 += 4) { // four traces' loads in flight (independent addresses); the additions keep the trace order
		double2 v[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const unsigned tb_ = b0 + (unsigned)j < ntr ? b0 + (unsigned)j : ntr - 1u;
			v[j] = p0[(size_t)(rowmap ? rowmap[tb_] : tb_) * npart];
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const unsigned b = b0 + (unsigned)j;
			if (b < ntr) {
				st.x += v[j].x; st.y += v[j].y;
				add_unit_phasor(ps, v[j]);
				if (PREFIX && wa.OUTP) {
					const unsigned cnt = wa.k0 + b + 1u;
					wa.OUTP[(size_t)b * wa.outp_stride + i] = weight_value(st, ps, cnt == 1 ? wa.mode1 : wa.mode, (double)cnt, (double)cnt, wa.wu);
				}
			}
		}
	}
	if (planes) { ST[i] = st; PS[i] = ps; }
	if (wa.OUT) wa.OUT[i] = weight_value(st, ps, wa.mode, wa.K, wa.M, wa.wu);
}
