// fwd_mfma_spec.hip -- matrix-pipe forward kernels specialised on the tap steps of both voice pairs.
//
// Same work decomposition, image layout and B tables as fwd_mfma.h (read its header first); the difference is the
// product loop.  tools/mfma_inner.hip (MI355X): with TQ = 8 tiles per unit, compile-time tap steps and the operands of
// the NEXT (phase, pair) product read from LDS while the current one multiplies, the products alone sustain 64-68
// TFLOP/s (82-87 % of the 78.6 peak) at one to four waves per SIMD; read-then-multiply reaches 55-65.  The runtime
// switch on the tap steps inside the loops of fwd_mfma.h costs most of that, hence one kernel per (KQ0, KQ1):
// a plan uses a handful of them.  Variants: (TQ, Mc) = (8,4) for every tap-step pair up to 8, (4,4), (2,4), (8,2) up to 5;
// anything else (D = 1, ragged phase chunks, one tap step) stays on the generic kernel of fwd_mfma.h.
//
// Measured (round 1, MI355X): a wave of a (4,5) group lives 18 us for 864 matrix instructions (s_memtime breakdown with
// -DFS_TIMING: products 56 %, x fetch 11 %, staging 8 %, stores 5 %, barriers 3 %, set-up 15 %), i.e. two waves per SIMD
// keep the pipe ~70 % busy.  1024 x 32768 single-stage (float input, V = 5) in one batch: 3.05 ms for the nine octaves
// with TQ = 8, Mc = 4 (61 % pipe utilisation, 70 % of the issued MACs useful: the fifth voice has no partner) against
// 3.85 ms for ten octaves on k_fwd_lds; whole call 5.9-6.1 ms vs 5.7 ms on the VALU kernels.  North-star (10 traces,
// TSPWS_MFMA_TARGET=640): four launches, 24 + 134 + 19 + 25 = 203 us vs 187 us on the VALU kernels, and the accumulate
// kernel pays 64 instead of 26 us for the extra phase splits of the coarse scales.  Still opt-in.  Next: run the small
// (TQ, Mc) launches beside the big one, reduce the splits of the coarse scales before the accumulate kernel, keep
// workgroups resident across work items (the 10-trace problem is 6 rounds of 23 us workgroups: set-up and tails).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include "fwd_mfma_types.h"

namespace {

template <int TQ, int KQ>
__device__ __forceinline__ void fs_load(double (&A)[TQ + KQ - 1], double (&B)[KQ], const double *__restrict__ ap, const double *__restrict__ bp)
{
#pragma unroll
	for (int k = 0; k < KQ; k++) B[k] = bp[k * 16];
#pragma unroll
	for (int i = 0; i < TQ + KQ - 1; i++) A[i] = ap[4 * i];
}

// FIRST: the accumulators start from zero -- the zero is the instruction's C operand
template <int TQ, int KQ, bool FIRST>
__device__ __forceinline__ void fs_mult(double (&C)[TQ], const double (&A)[TQ + KQ - 1], const double (&B)[KQ])
{
#pragma unroll
	for (int k = 0; k < KQ; k++)
#pragma unroll
		for (int a = 0; a < TQ; a++) C[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[a + k], B[k], (FIRST && k == 0) ? 0.0 : C[a], 0, 0, 0);
}

// the MC phases of a chunk, both pairs, software pipelined (KQ1 == 0: one pair, its phases alternate two operand sets)
template <int TQ, int MC, int KQ0, int KQ1, bool FIRST>
__device__ __forceinline__ void fs_chunk(double (&C0)[TQ], double (&C1)[TQ], const double *__restrict__ ap0, const double *__restrict__ ap1,
                                         const double *__restrict__ bl0, const double *__restrict__ bl1, const unsigned P)
{
	if constexpr (KQ1 > 0) {
		double A0[TQ + KQ0 - 1], B0[KQ0], A1[TQ + KQ1 - 1], B1[KQ1];
		fs_load<TQ, KQ0>(A0, B0, ap0, bl0);
#pragma unroll
		for (int ph = 0; ph < MC; ph++) {
			fs_load<TQ, KQ1>(A1, B1, ap1 + ph * P, bl1 + ph * KQ1 * 16);
			if (ph == 0) fs_mult<TQ, KQ0, FIRST>(C0, A0, B0); else fs_mult<TQ, KQ0, false>(C0, A0, B0);
			if (ph < MC - 1) fs_load<TQ, KQ0>(A0, B0, ap0 + (ph + 1) * P, bl0 + (ph + 1) * KQ0 * 16);
			if (ph == 0) fs_mult<TQ, KQ1, FIRST>(C1, A1, B1); else fs_mult<TQ, KQ1, false>(C1, A1, B1);
		}
	} else {
		static_assert(MC % 2 == 0, "phase pairs");
		double Aa[TQ + KQ0 - 1], Ba[KQ0], Ab[TQ + KQ0 - 1], Bb[KQ0];
		fs_load<TQ, KQ0>(Aa, Ba, ap0, bl0);
#pragma unroll
		for (int ph = 0; ph < MC; ph += 2) {
			fs_load<TQ, KQ0>(Ab, Bb, ap0 + (ph + 1) * P, bl0 + (ph + 1) * KQ0 * 16);
			if (ph == 0) fs_mult<TQ, KQ0, FIRST>(C0, Aa, Ba); else fs_mult<TQ, KQ0, false>(C0, Aa, Ba);
			if (ph + 2 < MC) fs_load<TQ, KQ0>(Aa, Ba, ap0 + (ph + 2) * P, bl0 + (ph + 2) * KQ0 * 16);
			fs_mult<TQ, KQ0, false>(C0, Ab, Bb);
		}
	}
}

#define FS_BREG 8 /* B doubles per thread of a staged sub-split (<= 16 KB per workgroup), as in fwd_mfma.h */

template <typename TIn, int TQ, int Mc, int KQ0, int KQ1>
__global__ void __launch_bounds__(256) k_fwd_mfma_t(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N, const FwdGroup *__restrict__ pd,
                                                    unsigned ngroups, const FwdOffsets offs, const double *__restrict__ bt, double2 *__restrict__ part,
                                                    size_t npart)
{
	constexpr int RPI = 64 / Mc, LOGMC = Mc == 4 ? 2 : 1;
	constexpr int KQM = KQ0 > KQ1 ? KQ0 : KQ1;
	constexpr int NIMAX = (4 * TQ + 4 * KQM + 6 + RPI - 1) / RPI; // staging iterations per unit (rows staged <= 4 TQ + 4 KQ + 3 + row offset <= 3)
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	double *smem = (double *)smem_raw;
	unsigned lo = 0, hi = ngroups;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (offs.off[mid] <= blockIdx.x) lo = mid; else hi = mid;
	}
	const FwdGroup &d = pd[lo];
	const unsigned il = blockIdx.x - offs.off[lo];
	const unsigned nob = d.nob, D = d.D, Ns = d.Ns, P = d.P, Pu = d.Pu;
	const unsigned nuc = (ntr * nob + d.upi - 1) / d.upi;
	const unsigned uc = il % nuc, split = il / nuc;

	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned l_hi = lane >> 4, l_blk = (lane >> 2) & 3, l_lo = lane & 3;
	const unsigned ml = lane & (Mc - 1), rs = lane >> LOGMC;
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
				xv[b][it] = (double)xt[idx];
				idx += rowstep; if (idx >= N) idx -= N;
			}
		}
	};

	// B tiles of a sub-split: one contiguous run of the group's table ([chunk][pair][phase][kappa][16])
	double breg[FS_BREG];
	constexpr unsigned bper0 = Mc * KQ0 * 16, bper = Mc * (KQ0 + KQ1) * 16;
	auto bfetch = [&](unsigned ss) {
		const unsigned c0 = ss * css, nc = (nch - c0) < css ? (nch - c0) : css;
		const unsigned n = nc * bper;
		const double *src = bt + d.bt_off[0] + (size_t)(ch0 + c0) * bper;
#pragma unroll
		for (int i = 0; i < FS_BREG; i++) {
			const unsigned e = (unsigned)i * 256 + tid;
			breg[i] = src[e < n ? e : n - 1]; // the clamped tail is never read back
		}
	};

#ifdef FS_TIMING
	unsigned long long t_stage = 0, t_fetch = 0, t_mult = 0, t_store = 0, t_sync = 0, t_all = __builtin_amdgcn_s_memtime(), t_a, nstep = 0, t_real = wall_clock64();
#define FS_T0() t_a = __builtin_amdgcn_s_memtime()
#define FS_T1(acc) acc += __builtin_amdgcn_s_memtime() - t_a
#else
#define FS_T0()
#define FS_T1(acc)
#endif
	double C0[TQ], C1[TQ];
	// the step being fetched: order = for ss, for quad, for chunk of the sub-split
	unsigned f_i = 0, f_ch = 0; // f_ch: chunk relative to the split
	bool f_more = nquad > 0;
	if (f_more) fetch(0, 0);
	bfetch(0);
	for (unsigned ss = 0; ss < nss; ss++) {
		const unsigned c0 = ss * css, ncs = (nch - c0) < css ? (nch - c0) : css;
		FS_T0();
		__syncthreads(); // every wave is done with the previous sub-split's tiles
#pragma unroll
		for (int i = 0; i < FS_BREG; i++) { const unsigned e = (unsigned)i * 256 + tid; if (e < d.bl_doubles) Bl[e] = breg[i]; }
		__syncthreads();
		if (ss + 1 < nss) bfetch(ss + 1);
		FS_T1(t_sync);
		for (unsigned i = 0; i < nquad; i++) {
			for (unsigned ch = 0; ch < ncs; ch++) {
				const bool first_chunk = c0 + ch == 0, last_chunk = c0 + ch + 1 == nch;
				FS_T0();
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (int b = 0; b < 4; b++)
#pragma unroll
					for (int it = 0; it < NIMAX; it++) if ((unsigned)it < NI) stp[(unsigned)b * Pu + (unsigned)it * RPI] = xv[b][it]; // rows past RT land in the plane's slack
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				FS_T1(t_stage);
				FS_T0();
				// next step in the order (ss, quad, chunk)
				if (f_more) {
					const unsigned fss = f_ch / css; // sub-split of the fetched step
					const unsigned fend = ((fss + 1) * css < nch) ? (fss + 1) * css : nch;
					if (++f_ch == fend) {
						if (++f_i == nquad) { f_i = 0; if (fend == nch) f_more = false; }
						else f_ch = fss * css;
					}
					unsigned coff = f_ch * Mc; // <= D + Mc
					while (coff >= N) coff -= N;
					if (f_more) fetch(f_i, coff);
				}
				FS_T1(t_fetch);
				FS_T0();
				const double *bl0 = Bl + ch * bper + b_lane, *bl1 = bl0 + bper0;
				if (first_chunk) fs_chunk<TQ, Mc, KQ0, KQ1, true>(C0, C1, ap0, ap1, bl0, bl1, P);
				else fs_chunk<TQ, Mc, KQ0, KQ1, false>(C0, C1, ap0, ap1, bl0, bl1, P);
#ifdef FS_TIMING
				asm volatile("s_nop 0" :: "v"(C0[TQ - 1]), "v"(C1[TQ - 1])); // the last accumulators are done
				nstep++;
#endif
				FS_T1(t_mult);
				FS_T0();
				if (last_chunk) { // C: lane = 16 i + 4 blk + j holds output 4 a + i of unit blk, column j
					const int src = (int)((4 * i + l_blk) << 2); // descriptor lane of this lane's unit (byte address for bpermute)
					const unsigned t = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)s_t);
					const unsigned blk = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)s_blk);
					const unsigned valid = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)s_valid);
					const unsigned j = l_lo, v = j >> 1;
					const unsigned nb = blk * 4 * TQ + l_hi;
					double2 *pt = part + (size_t)t * npart + (size_t)split * Ns;
					{
						double *dst = (double *)(pt + d.po[v]) + (j & 1) + (size_t)nb * 2;
						const bool act = valid && v < d.nv[0];
#pragma unroll
						for (int a = 0; a < TQ; a++) if (act && nb + 4 * (unsigned)a < Ns) dst[8 * a] = (j & 1) ? -C0[a] : C0[a]; // conj
					}
					if constexpr (KQ1 > 0) {
						double *dst = (double *)(pt + d.po[2 + v]) + (j & 1) + (size_t)nb * 2;
						const bool act = valid && v < d.nv[1];
#pragma unroll
						for (int a = 0; a < TQ; a++) if (act && nb + 4 * (unsigned)a < Ns) dst[8 * a] = (j & 1) ? -C1[a] : C1[a]; // conj
					}
				}
				FS_T1(t_store);
			}
		}
	}
#ifdef FS_TIMING
	if (tid == 0 && (blockIdx.x % 997) == 5)
		printf("blk %u grp %u steps %llu: real100MHz %llu all %llu | sync %llu stage %llu fetch %llu mult %llu store %llu (memtime ticks)\n", blockIdx.x, lo, nstep,
		       wall_clock64() - t_real, __builtin_amdgcn_s_memtime() - t_all, t_sync, t_stage, t_fetch, t_mult, t_store);
#endif
}

typedef void (*kern_d)(const double *, size_t, unsigned, unsigned, const FwdGroup *, unsigned, const FwdOffsets, const double *, double2 *, size_t);
typedef void (*kern_f)(const float *, size_t, unsigned, unsigned, const FwdGroup *, unsigned, const FwdOffsets, const double *, double2 *, size_t);

// variants: 0 = (TQ 8, Mc 4) for KQ0 in 2..8, KQ1 in {0, 2..8}; 1 = (4, 4), 2 = (2, 4), 3 = (8, 2) for KQ0 in 2..5, KQ1 in {0, 2..5}
struct Tables {
	kern_d td[4][9][9];
	kern_f tf[4][9][9];
};

template <int V, int TQ, int MC, int KMAX, int K0, int K1> struct Fill {
	static void run(Tables &t)
	{
		t.td[V][K0][K1] = k_fwd_mfma_t<double, TQ, MC, K0, K1>;
		t.tf[V][K0][K1] = k_fwd_mfma_t<float, TQ, MC, K0, K1>;
		if constexpr (K1 < KMAX) Fill<V, TQ, MC, KMAX, K0, (K1 == 0 ? 2 : K1 + 1)>::run(t);
		else if constexpr (K0 < KMAX) Fill<V, TQ, MC, KMAX, K0 + 1, 0>::run(t);
	}
};

const Tables &tables()
{
	static Tables t;
	static bool init = false;
	if (!init) {
		for (int v = 0; v < 4; v++) for (int a = 0; a < 9; a++) for (int b = 0; b < 9; b++) { t.td[v][a][b] = nullptr; t.tf[v][a][b] = nullptr; }
		Fill<0, 8, 4, 8, 2, 0>::run(t);
		Fill<1, 4, 4, 5, 2, 0>::run(t);
		Fill<2, 2, 4, 5, 2, 0>::run(t);
		Fill<3, 8, 2, 5, 2, 0>::run(t);
		init = true;
	}
	return t;
}

int variant_of(unsigned tq, unsigned mc) { return (tq == 8 && mc == 4) ? 0 : (tq == 4 && mc == 4) ? 1 : (tq == 2 && mc == 4) ? 2 : (tq == 8 && mc == 2) ? 3 : -1; }

} // namespace

int fwd_mfma_spec_has(unsigned tq, unsigned mc, unsigned kq0, unsigned kq1)
{
	const int v = variant_of(tq, mc);
	return v >= 0 && kq0 < 9 && kq1 < 9 && tables().td[v][kq0][kq1] != nullptr;
}

int fwd_mfma_spec_launch(int is_float, unsigned tq, unsigned mc, unsigned kq0, unsigned kq1, unsigned items, size_t lds, void *stream, const void *x,
                         size_t ld, unsigned ntr, unsigned N, const FwdGroup *pd, unsigned ngroups, const FwdOffsets &offs, const double *bt, void *part,
                         size_t npart)
{
	if (!fwd_mfma_spec_has(tq, mc, kq0, kq1) || !items) return 0;
	const int v = variant_of(tq, mc);
	if (getenv("TSPWS_DEBUG")) {
		int nb = -1;
		if (is_float) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tables().tf[v][kq0][kq1], 256, lds);
		else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tables().td[v][kq0][kq1], 256, lds);
		fprintf(stderr, "mfma spec TQ %u Mc %u (%u,%u): %u items, %zu B LDS, %d workgroups per CU\n", tq, mc, kq0, kq1, items, lds, nb);
	}
	if (is_float)
		hipLaunchKernelGGL(tables().tf[v][kq0][kq1], dim3(items), dim3(256), lds, (hipStream_t)stream, (const float *)x, ld, ntr, N, pd, ngroups, offs, bt,
		                   (double2 *)part, npart);
	else
		hipLaunchKernelGGL(tables().td[v][kq0][kq1], dim3(items), dim3(256), lds, (hipStream_t)stream, (const double *)x, ld, ntr, N, pd, ngroups, offs, bt,
		                   (double2 *)part, npart);
	return 1;
}
