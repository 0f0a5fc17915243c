// stream.hip -- trace prologue (fold, mean removal) and stage 1 of the two-stage stack: the HBM-streaming partial-stack pass.
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"
#include <hip/hip_ext.h>

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
		if (t < ck.count) { // the remainder as ONE batch (rows past the end re-read the last row and are not added): one round trip, not up to seven
			const unsigned nv = ck.count - t, last = ck.count - 1u;
			v4f v[8];
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)(t + (unsigned)j < ck.count ? t + (unsigned)j : last) * ld));
#pragma unroll
			for (int j = 0; j < 8; j++)
				if ((unsigned)j < nv) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
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

// Running sums instead of chunk sums (masked replicas, resample.hip): segment `seg` of the launch walks ITS runs of traces in
// trace order without ever resetting its accumulators and stores a snapshot after every run -- snap[r] = sum of the segment's
// traces up to the end of run r (segment 0 starts from the sum of the `ncarry` rows carry[.]: the total of everything before the
// launch, so its snapshots are prefix sums of the whole ensemble).  A sum over any run of consecutive traces is then a difference
// of two snapshots (plus a segment base), whatever the number of runs in between: k_combine_terms.  Long sequential walks like the
// plain pass's (one workgroup per column block and segment), no chunk rows to reduce.
// grid = bx column blocks x segments; runs of segment s: [seg_first[s], seg_first[s + 1]).
#ifndef PREFIX_STORE
#define PREFIX_STORE 0
#endif
template <bool VEC4>
__global__ void __launch_bounds__(256) k_prefix_walk(const float *__restrict__ x, size_t ld, size_t N, const Chunk *__restrict__ runs,
                                                     const unsigned *__restrict__ seg_first, unsigned bx, double *__restrict__ snap, size_t ldpc,
                                                     const unsigned *__restrict__ carry, unsigned ncarry)
{
	const unsigned seg = blockIdx.x / bx, cb = blockIdx.x - seg * bx;
	unsigned ri = seg_first[seg];
	const unsigned rend = seg_first[seg + 1];
	const size_t col = ((size_t)cb * 256 + threadIdx.x) * 4;
	if (ri >= rend || col >= N) return;
	const unsigned rem = (N - col) < 4 ? (unsigned)(N - col) : 4u;
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	if (seg == 0)
		for (unsigned k = 0; k < ncarry; k++) {
			const double *c = snap + (size_t)carry[k] * ldpc + col;
			a0 += c[0];
			if (rem > 1) a1 += c[1];
			if (rem > 2) a2 += c[2];
			if (rem > 3) a3 += c[3];
		}
	if (VEC4) {
		// the plain pass's loop (eight independent 16-byte non-temporal loads in flight, then their additions), run after run; the
		// last, short batch of a run is a batch like the others (rows past the end re-read the last row and are not added)
		typedef float v4f __attribute__((ext_vector_type(4)));
		for (; ri < rend; ri++) {
			const Chunk ck = runs[ri];
			const float *src = x + ck.t0 * ld + col;
			unsigned t = 0;
			for (; t + 8 <= ck.count; t += 8) {
				v4f v[8];
#pragma unroll
				for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)(t + j) * ld));
#pragma unroll
				for (int j = 0; j < 8; j++) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
			}
			if (t < ck.count) {
				const unsigned nv = ck.count - t, last = ck.count - 1u;
				v4f v[8];
#pragma unroll
				for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)(t + (unsigned)j < ck.count ? t + (unsigned)j : last) * ld));
#pragma unroll
				for (int j = 0; j < 8; j++)
					if ((unsigned)j < nv) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
			}
#if PREFIX_STORE == 3
			double *dst = snap + (size_t)(ri + 1 == rend ? ri : (ri & 1u)) * ldpc + col; // timing ablation: the stores hit two rows only (L2 absorbs them)
#else
			double *dst = snap + (size_t)ri * ldpc + col;
#endif
#if PREFIX_STORE == 2
			if (ri + 1 == rend) { *(double2 *)dst = make_double2(a0, a1); *(double2 *)(dst + 2) = make_double2(a2, a3); } // timing ablation
#elif PREFIX_STORE == 1
			typedef double v2d __attribute__((ext_vector_type(2)));
			v2d s0 = {a0, a1}, s1 = {a2, a3};
			__builtin_nontemporal_store(s0, (v2d *)dst);
			__builtin_nontemporal_store(s1, (v2d *)(dst + 2));
#else
			*(double2 *)dst = make_double2(a0, a1);
			*(double2 *)(dst + 2) = make_double2(a2, a3);
#endif
		}
		return;
	}
	for (; ri < rend; ri++) {
		const Chunk ck = runs[ri];
		const float *src = x + ck.t0 * ld + col;
		for (unsigned t = 0; t < ck.count; t++) {
			const float *r = src + (size_t)t * ld;
			a0 += (double)r[0];
			if (rem > 1) a1 += (double)r[1];
			if (rem > 2) a2 += (double)r[2];
			if (rem > 3) a3 += (double)r[3];
		}
		double *dst = snap + (size_t)ri * ldpc + col;
		dst[0] = a0;
		if (rem > 1) dst[1] = a1;
		if (rem > 2) dst[2] = a2;
		if (rem > 3) dst[3] = a3;
	}
}

// Few columns (<= ROWS_WMAX: the jackknife of cfg4 has 10 replicas + the plain stack): the rows THEMSELVES from the walk.  A
// workgroup owns 1024 samples of every trace of ITS segment of the stage and keeps one running sum per column in registers; after
// every run of traces (one signature) the run's sum is added to the columns the run belongs to (`member`, a bit per column --
// uniform branches, once per ~30 traces, not once per trace: the round-3 kernel that added every trace to every column was
// VALU-bound at 2.4 ms), and a column whose group ends there stores its sum as that group's row and starts over (`flush`).  No
// snapshots (their 326 MB of writes cost the prefix walk 0.25 ms), no combining pass over ~10 snapshots per row.
// Two workgroups per column block (grid.y): 128 alone would leave half the CUs without a stream (1.05 ms for the plain pass at 128
// workgroups), and narrower workgroups lose (512 samples: 1.08-1.19 ms -- a workgroup that reads 4 KB of every row keeps one
// channel stream to itself).  The stage's runs are cut in two SEGMENTS of similar trace counts: segment A starts from the sums the
// previous stage left (carry), segment B from zero -- the first row a column stores in B lacks what A had collected for it
// (`tail`), and k_seg_fix adds it (or, for a column that stores nothing in B, hands tail + B's sum on to the next stage): W rows of
// fix-up per stage instead of a second set of rows and a pass that adds the two (alternate runs in two halves: 0.97 + 0.05 ms).
#define ROWS_WMAX 16
// WM: columns the instantiation carries running sums for (12: the jackknife n = 10, d = 1 with its plain stack -- 32 registers less than
// 16, which NL = 16 loads in flight take); NL: independent 16-byte loads in flight per lane
// The stores of a flush go through LDS to a FIFTH wave of the workgroup (VEC4 form).  gfx950 counts loads and stores in ONE in-order
// counter (vmcnt), so a wave that stores a group's rows cannot consume any later load before the memory system has acknowledged those stores
// -- under the read stream that takes 15-25 us per flush: 2 % of the bytes cost the walk 0.14 of its 0.86 ms (tools/read_ceiling.hip:
// the same walk without the stores 0.72-0.73 ms).  The four loading waves put the finished sums into LDS (W x 1024 doubles) and go on; the
// writer wave walks the same run descriptors, takes every flush out of LDS and stores it, one contiguous KB per instruction, and waits for
// nobody's loads.  Two barriers per flush: B "LDS is free again" (the writer has read the previous flush), A "the sums are in LDS".
#define ROWS_WRITER 64 /* threads of the writer wave behind the 256 loading threads */
#ifndef ROWS_ABL
#define ROWS_ABL 0 /* timing ablations (results wrong): 1 the writer stores nothing, 2 no flush at all (no LDS copies, no barriers, no writer) */
#endif
typedef double rows_v2d __attribute__((ext_vector_type(2)));
// a run descriptor into scalar registers (behind the barriers' memory clobbers the compiler reads the list with vector loads: every
// descriptor field a VGPR, the segment's trace count compared lane by lane)
__device__ __forceinline__ RunDesc rows_run(const RunDesc *__restrict__ runs, unsigned i)
{
	const RunDesc r = runs[i];
	RunDesc u;
	u.t0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(r.t0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r.t0);
	u.count = (unsigned)__builtin_amdgcn_readfirstlane((int)r.count);
	u.member = (unsigned)__builtin_amdgcn_readfirstlane((int)r.member);
	u.flush = (unsigned)__builtin_amdgcn_readfirstlane((int)r.flush);
	u.frow = (unsigned)__builtin_amdgcn_readfirstlane((int)r.frow);
	u.pad[0] = u.pad[1] = 0;
	return u;
}
__device__ __forceinline__ void rows_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool VEC4, int WM, int NL>
__global__ void __launch_bounds__(256 + (VEC4 ? ROWS_WRITER : 0)) k_rows_walk(const float *__restrict__ x, size_t ld, size_t N, const RunDesc *__restrict__ runs, unsigned qa0,
                                                   unsigned qm, unsigned qb1, unsigned W, const unsigned *__restrict__ flush_rows,
                                                   double *__restrict__ rows, const double *__restrict__ carry_in, double *__restrict__ endA,
                                                   double *__restrict__ endB)
{
	extern __shared__ __attribute__((aligned(16))) double wl[]; // VEC4: [W][1024] sums on their way to the writer
	const unsigned seg = blockIdx.y;
	const unsigned r0 = seg ? qm : qa0, r1 = seg ? qb1 : qm;
	if (VEC4 && threadIdx.x >= 256) { // ---- the writer wave
		if (ROWS_ABL == 2) return;
		const unsigned lane = threadIdx.x - 256;
		const size_t c0 = (size_t)blockIdx.x * 1024;
		for (unsigned ri = r0; ri < r1; ri++) {
			const unsigned fl = runs[ri].flush;
			if (!fl) continue;
			unsigned fr = runs[ri].frow;
			rows_lds_barrier(); // B (the LDS reads of the previous flush have landed: lgkmcnt(0) above)
			rows_lds_barrier(); // A
#pragma unroll 1
			for (unsigned c = 0; c < (unsigned)WM; c++) {
				if (!((fl >> c) & 1u)) continue;
				double *dst = rows + (size_t)flush_rows[fr++] * N + c0;
				const double *srcl = wl + (size_t)c * 1024;
				double2 v[8];
#pragma unroll
				for (int k = 0; k < 8; k++) v[k] = *(const double2 *)(srcl + (k * 64 + lane) * 2);
				if (ROWS_ABL != 1)
#pragma unroll
				for (int k = 0; k < 8; k++) // (N % 4 == 0; non-temporal: the rows leave in the burst instead of dribbling out of L2 between the reads)
					if (c0 + (size_t)(k * 64 + lane) * 2 < N) __builtin_nontemporal_store((rows_v2d){v[k].x, v[k].y}, (rows_v2d *)(dst + (k * 64 + lane) * 2));
			}
		}
		return;
	}
	size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
	const bool live = col < N; // (VEC4: lanes past the end ride along on column 0 -- they meet the barriers -- and write nothing)
	if (!live) { if (!VEC4) return; col = 0; }
	const unsigned rem = (N - col) < 4 ? (unsigned)(N - col) : 4u;
	double P[WM][4];
#pragma unroll
	for (int c = 0; c < WM; c++) {
#pragma unroll
		for (int k = 0; k < 4; k++) P[c][k] = 0;
		if (!seg && carry_in && (unsigned)c < W) {
			const double *s = carry_in + (size_t)c * N + col;
#pragma unroll
			for (int k = 0; k < 4; k++) if ((unsigned)k < rem) P[c][k] = s[k];
		}
	}
	typedef float v4f __attribute__((ext_vector_type(4)));
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	// the end of a run: its sum goes to the columns it belongs to; a column whose group ends there hands its row to the writer and starts over
	auto run_end = [&](const RunDesc &rd) {
		unsigned fr = rd.frow;
		if (VEC4 && rd.flush && ROWS_ABL != 2) rows_lds_barrier(); // B: the writer has taken the previous flush out of LDS
#pragma unroll
		for (int c = 0; c < WM; c++) {
			if ((rd.member >> c) & 1u) { P[c][0] += a0; P[c][1] += a1; P[c][2] += a2; P[c][3] += a3; } // (wave-uniform)
			if ((rd.flush >> c) & 1u) {
				if (VEC4 && ROWS_ABL == 2) { }
				else if (VEC4) {
					double *dl = wl + (size_t)c * 1024 + threadIdx.x * 4;
					*(double2 *)dl = make_double2(P[c][0], P[c][1]); *(double2 *)(dl + 2) = make_double2(P[c][2], P[c][3]);
				} else {
					double *dst = rows + (size_t)flush_rows[fr++] * N + col;
#pragma unroll
					for (int k = 0; k < 4; k++) if ((unsigned)k < rem) dst[k] = P[c][k];
				}
#pragma unroll
				for (int k = 0; k < 4; k++) P[c][k] = 0;
			}
		}
		if (VEC4 && rd.flush && ROWS_ABL != 2) rows_lds_barrier(); // A: the sums are in LDS
		a0 = 0; a1 = 0; a2 = 0; a3 = 0;
	};
	if (VEC4) {
		// The runs of a segment are consecutive traces, so the loads need not care where a run ends: ONE stream of batches of NL rows over
		// the whole segment (NL independent 16-byte non-temporal loads in flight, then their additions -- the plain pass's loop), a run's end
		// handled inside the batch that holds its last rows and the next run's first, the next run's descriptor requested a run ahead.  (Rounds 4-5: a load loop per run -- every run end, once per
		// ~30 traces at cfg4, re-read a partial batch and waited for the next descriptor with nothing in flight: 0.80 ms without any flush
		// against 0.73 of the same walk in one run, tools/read_ceiling.hip.)  The additions keep their order, the rows their last bit.
		if (r0 < r1) {
			const RunDesc first = rows_run(runs, r0), lastd = rows_run(runs, r1 - 1);
			const unsigned long long total = lastd.t0 + lastd.count - first.t0; // traces of the segment
			const float *src = x + first.t0 * ld + col;
			unsigned ri = r0;
			RunDesc rd = first, rn = rows_run(runs, r0 + 1 < r1 ? r0 + 1 : r0);
			unsigned left = rd.count;
			unsigned long long t = 0;
			bool over = false;
			while (!over && t < total) {
				// whole batches inside the current run: the plain pass's loop, nothing else in it
				for (; left >= (unsigned)NL; left -= NL, t += NL) {
					v4f v[NL];
#pragma unroll
					for (int j = 0; j < NL; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)(t + (unsigned)j) * ld));
#pragma unroll
					for (int j = 0; j < NL; j++) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
				}
				// the batch with the run's last rows (< NL, possibly none) and the first rows of what follows: once per run
				v4f v[NL];
#pragma unroll
				for (int j = 0; j < NL; j++) { // (past the end of the segment: the last row again, not added)
					const unsigned long long tj = t + (unsigned)j < total ? t + (unsigned)j : total - 1;
					v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)tj * ld));
				}
				const unsigned nb = total - t < (unsigned long long)NL ? (unsigned)(total - t) : (unsigned)NL;
				unsigned j0 = 0;
#pragma unroll 1
				do {
					const unsigned n = left < nb - j0 ? left : nb - j0, j1 = j0 + n;
#pragma unroll
					for (int j = 0; j < NL; j++) { // rows [j0, j1) of the batch belong to the current run (uniform branches: with selects every row costs
						// a convert, two selects and an add per round of this loop, and with one wave per SIMD nothing hides them)
						if (__builtin_amdgcn_readfirstlane((int)((unsigned)j >= j0 && (unsigned)j < j1))) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
					}
					left -= n; j0 = j1;
					if (left == 0) {
						run_end(rd);
						if (++ri >= r1) { over = true; break; } // (the runs tile the segment: nothing is left when the last one ends)
						rd = rn; left = rd.count;
						rn = rows_run(runs, ri + 1 < r1 ? ri + 1 : ri);
					}
				} while (j0 < nb);
				t += nb;
			}
		}
	} else {
		for (unsigned ri = r0; ri < r1; ri++) {
			const RunDesc rd = runs[ri];
			const float *src = x + rd.t0 * ld + col;
			for (unsigned t = 0; t < rd.count; t++) {
				const float *r = src + (size_t)t * ld;
				a0 += (double)r[0];
				if (rem > 1) a1 += (double)r[1];
				if (rem > 2) a2 += (double)r[2];
				if (rem > 3) a3 += (double)r[3];
			}
			run_end(rd);
		}
	}
	double *end = seg ? endB : endA; // the live sums of the segment
	if (end && live) {
#pragma unroll
		for (int c = 0; c < WM; c++)
			if ((unsigned)c < W) {
				double *d = end + (size_t)c * N + col;
#pragma unroll
				for (int k = 0; k < 4; k++) if ((unsigned)k < rem) d[k] = P[c][k];
			}
	}
}

// after a stage walked in two segments: column c (blockIdx.y) -- the first row it stored in segment B gets what segment A had
// collected for it (fix_row[c]), or, if it stored nothing in B, the next stage inherits tail + B's sum
__global__ void __launch_bounds__(256) k_seg_fix(const double *__restrict__ tailA, const double *__restrict__ endB, const unsigned *__restrict__ fix_row,
                                                 double *__restrict__ rows, double *__restrict__ carry, size_t N)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const unsigned c = blockIdx.y, fr = fix_row[c];
	const double t = tailA[(size_t)c * N + n], e = endB[(size_t)c * N + n];
	if (fr != ~0u) { rows[(size_t)fr * N + n] = t + rows[(size_t)fr * N + n]; carry[(size_t)c * N + n] = e; }
	else carry[(size_t)c * N + n] = t + e;
}

// one stage of the direct walk: runs [q0, q1), cut at qm into two segments (qm == q1: one segment); d_blk = [tail | endB | carry], W rows
// each; carry_in / carry_out: the stage starts from / leaves the carry rows
int tspws_rows_walk_launch(const float *d_x, size_t ld, size_t N, const RunDesc *d_runs, unsigned q0, unsigned qm, unsigned q1, unsigned W,
                           const unsigned *d_flush_rows, const unsigned *d_fix_row, double *d_rows, double *d_blk, int carry_in, int carry_out, hipStream_t st)
{
	if (W > ROWS_WMAX) return fail(TSPWS_E_ARG, "rows_walk: too many columns");
	const unsigned grid = (unsigned)((N + 1023) / 1024);
	const bool vec = (N % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)d_x & 15) == 0);
	double *tail = d_blk, *endB = d_blk + (size_t)W * N, *carry = d_blk + (size_t)2 * W * N;
	const bool two = qm < q1 && qm > q0;
	if (!two) qm = q1;
	// one segment: its live sums are the carry; two: tail + endB, merged by k_seg_fix
	double *endA = two ? tail : (carry_out ? carry : nullptr);
	static int nl = -1; // sweeps: loads in flight of the W <= 12 instantiation
	if (nl < 0) { const char *e = sweep_env("TSPWS_WALK_NL"); nl = e ? atoi(e) : 8; } // (cfg4: 0.888 ms with 8, 0.903 with 16; the 16-column form: 0.909)
	const dim3 g2(grid, two ? 2 : 1);
	const double *cin = carry_in ? carry : nullptr;
	// (the VEC4 form: 256 loading threads + the writer wave, W x 8 KB of LDS for the sums on their way out -- 128 KB at 16 columns)
#define ROWS_WALK(V, WMV, NLV) do { \
		const size_t lds = (V) ? (size_t)(WMV) * 1024 * sizeof(double) : 0; \
		if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)k_rows_walk<V, WMV, NLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); /* (per device: every launch) */ \
		hipLaunchKernelGGL((k_rows_walk<V, WMV, NLV>), g2, dim3(256 + ((V) ? ROWS_WRITER : 0)), lds, st, d_x, ld, N, d_runs, q0, qm, q1, W, d_flush_rows, d_rows, cin, endA, endB); \
	} while (0)
	if (!vec) ROWS_WALK(false, 16, 8);
	else if (W <= 12 && nl == 16) ROWS_WALK(true, 12, 16);
	else if (W <= 12) ROWS_WALK(true, 12, 8);
	else ROWS_WALK(true, 16, 8);
#undef ROWS_WALK
	if (two) hipLaunchKernelGGL(k_seg_fix, dim3((unsigned)((N + 255) / 256), W), dim3(256), 0, st, (const double *)tail, (const double *)endB, d_fix_row, d_rows, carry, N);
	HIP_TRY(hipGetLastError());
	return 0;
}
unsigned tspws_rows_walk_wmax() { return ROWS_WMAX; }

// P[row][n] = sum_j coef[j] snap[idx[j]][n] over the row's terms [row_ptr[row], row_ptr[row + 1]) -- in list order; no terms: 0
__global__ void __launch_bounds__(256) k_combine_terms(const double *__restrict__ snap, size_t ldpc, const unsigned *__restrict__ row_ptr,
                                                       const unsigned *__restrict__ idx, const float *__restrict__ coef, double *__restrict__ P, size_t N)
{
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const unsigned row = blockIdx.y;
	double acc = 0;
	for (unsigned j = row_ptr[row]; j < row_ptr[row + 1]; j++) acc = fma((double)coef[j], snap[(size_t)idx[j] * ldpc + n], acc);
	P[(size_t)row * N + n] = acc;
}

// host side of the two kernels above: launches on device-resident tables (the caller uploaded them)
int tspws_prefix_launch(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const Chunk *d_runs, const unsigned *d_seg_first, unsigned nseg,
                        size_t nruns_total, const unsigned *d_carry, unsigned ncarry, double **d_snap, size_t *ldpc_out, hipStream_t st)
{
	const size_t ldpc = (N + 3) & ~(size_t)3;
	void *v = nullptr;
	int rc;
	if ((rc = scratch(p, SCR_CHUNK, std::max<size_t>(nruns_total * ldpc * sizeof(double), 16), &v))) return rc;
	*d_snap = (double *)v; *ldpc_out = ldpc;
	if (!nseg) return 0;
	const unsigned bx = (unsigned)((N + 1023) / 1024);
	const bool vec = (N % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)d_x & 15) == 0);
	if (vec) hipLaunchKernelGGL(k_prefix_walk<true>, dim3(bx * nseg), dim3(256), 0, st, d_x, ld, N, d_runs, d_seg_first, bx, (double *)v, ldpc, d_carry, ncarry);
	else hipLaunchKernelGGL(k_prefix_walk<false>, dim3(bx * nseg), dim3(256), 0, st, d_x, ld, N, d_runs, d_seg_first, bx, (double *)v, ldpc, d_carry, ncarry);
	HIP_TRY(hipGetLastError());
	return 0;
}

void tspws_combine_terms_launch(const double *d_snap, size_t ldpc, const unsigned *d_row_ptr, const unsigned *d_idx, const float *d_coef, unsigned nrows,
                                double *d_P, size_t N, hipStream_t st)
{
	for (unsigned r0 = 0; r0 < nrows; r0 += 65535) {
		const unsigned ny = std::min(nrows - r0, 65535u);
		hipLaunchKernelGGL(k_combine_terms, dim3((unsigned)((N + 255) / 256), ny), dim3(256), 0, st, d_snap, ldpc, d_row_ptr + r0, d_idx, d_coef, d_P + (size_t)r0 * N, N);
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

// Launch the streaming pass for an arbitrary chunk table (rows destinations): table upload + launches.
int tspws_run_chunks(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const std::vector<Chunk> &chunks,
                     const std::vector<unsigned> &row_first, unsigned rows, double *d_P, size_t ldP, hipStream_t st, bool cached,
                     unsigned row_begin, unsigned row_end)
{
	int rc;
	if ((rc = tspws_chunks_upload(p, chunks, row_first, rows, st, cached))) return rc;
	return tspws_chunks_launch(p, d_x, ld, N, chunks, row_first, rows, d_P, ldP, st, row_begin, row_end);
}

// Device copy of a chunk table into the plan's table block SCR_TAB (see chunk_tables).  cached: the plan's own group table (uploaded once,
// later calls only order their stream behind that copy); any other table is uploaded every time.  Either way SCR_TAB no longer holds
// whatever a masked-replica call may have left there, so that call's generation mark is reset (its own tables live in SCR_JKTAB).
int tspws_chunks_upload(tspws_hip_plan *p, const std::vector<Chunk> &chunks, const std::vector<unsigned> &row_first, unsigned rows, hipStream_t st,
                        bool cached)
{
	Chunk *d_chunks = nullptr;
	unsigned *d_rf = nullptr;
	p->jk_gen = 0;
	return chunk_tables(p, chunks, row_first, rows, st, cached, &d_chunks, &d_rf);
}

// The launches of the streaming pass for the destination rows [row_begin, row_end) of a table that tspws_chunks_upload has put
// into the plan's table block (the host copies give the launch geometry).
int tspws_chunks_launch(tspws_hip_plan *p, const float *d_x, size_t ld, size_t N, const std::vector<Chunk> &chunks,
                        const std::vector<unsigned> &row_first, unsigned rows, double *d_P, size_t ldP, hipStream_t st,
                        unsigned row_begin, unsigned row_end)
{
	row_end = std::min(row_end, rows);
	const size_t nck = chunks.size();
	const size_t ldpc = (N + 3) & ~(size_t)3;
	void *d_pc = nullptr;
	int rc;
	if ((rc = scratch(p, SCR_CHUNK, std::max<size_t>(nck * ldpc * sizeof(double), 16), &d_pc))) return rc;
	if (!p->scr[SCR_TAB]) return fail(TSPWS_E_ARG, "chunks_launch: no table uploaded");
	Chunk *d_chunks = (Chunk *)p->scr[SCR_TAB];
	unsigned *d_rf = (unsigned *)((char *)p->scr[SCR_TAB] + nck * sizeof(Chunk));
	if (row_begin >= row_end) return 0;
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
		const unsigned wg_target = 256;
		size_t rows_total = 0;
		for (size_t c = ck0; c < ck1; c++) rows_total += chunks[c].count;
		const bool long_runs = ck1 > ck0 && rows_total / (ck1 - ck0) >= 256;
		const unsigned rpl = long_runs ? std::max(1u, wg_target / std::max(1u, bx)) : 65535u;
		p->last_stream_launches = 0;
		for (size_t c0 = ck0; c0 < ck1; c0 += rpl) {
			p->last_stream_launches++;
			const unsigned ny = (unsigned)std::min<size_t>(ck1 - c0, rpl);
			double *dst = d_P + (size_t)(row_begin + (c0 - ck0)) * ldP;
			// the call's events ride on the first / last launch (tspws_hip_stack armed them: plan->le)
			hipEvent_t e0 = nullptr, e1 = nullptr;
			if (c0 == ck0) { e0 = p->le.first_start; p->le.first_start = nullptr; }
			if (c0 + rpl >= ck1) { e1 = p->le.last_stop; p->le.last_stop = nullptr; p->le.ready = e1; }
			if (e0 || e1) {
				if (vec) hipExtLaunchKernelGGL(k_partial<true>, dim3(bx, ny), dim3(256), 0, st, e0, e1, 0, d_x, ld, N, (const Chunk *)(d_chunks + c0), dst, ldP);
				else hipExtLaunchKernelGGL(k_partial<false>, dim3(bx, ny), dim3(256), 0, st, e0, e1, 0, d_x, ld, N, (const Chunk *)(d_chunks + c0), dst, ldP);
			} else if (vec) hipLaunchKernelGGL(k_partial<true>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, dst, ldP);
			else hipLaunchKernelGGL(k_partial<false>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, dst, ldP);
		}
		HIP_TRY(hipGetLastError());
		return 0;
	}
	p->last_stream_launches = 0;
	for (size_t c0 = ck0; c0 < ck1; c0 += 65535) {
		p->last_stream_launches++;
		const unsigned ny = (unsigned)std::min<size_t>(ck1 - c0, 65535);
		hipEvent_t e0 = nullptr;
		if (c0 == ck0) { e0 = p->le.first_start; p->le.first_start = nullptr; }
		if (e0) {
			if (vec) hipExtLaunchKernelGGL(k_partial<true>, dim3(bx, ny), dim3(256), 0, st, e0, nullptr, 0, d_x, ld, N, (const Chunk *)(d_chunks + c0), (double *)d_pc + c0 * ldpc, ldpc);
			else hipExtLaunchKernelGGL(k_partial<false>, dim3(bx, ny), dim3(256), 0, st, e0, nullptr, 0, d_x, ld, N, (const Chunk *)(d_chunks + c0), (double *)d_pc + c0 * ldpc, ldpc);
		} else if (vec) hipLaunchKernelGGL(k_partial<true>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, (double *)d_pc + c0 * ldpc, ldpc);
		else hipLaunchKernelGGL(k_partial<false>, dim3(bx, ny), dim3(256), 0, st, d_x, ld, N, d_chunks + c0, (double *)d_pc + c0 * ldpc, ldpc);
	}
	for (unsigned r0 = row_begin; r0 < row_end; r0 += 65535) {
		const unsigned ny = std::min(row_end - r0, 65535u);
		hipEvent_t e1 = nullptr;
		if (r0 + 65535u >= row_end) { e1 = p->le.last_stop; p->le.last_stop = nullptr; p->le.ready = e1; }
		if (e1) hipExtLaunchKernelGGL(k_reduce_chunks, dim3((unsigned)((N + 255) / 256), ny), dim3(256), 0, st, nullptr, e1, 0, (const double *)d_pc, ldpc,
		                              (const unsigned *)(d_rf + r0), d_P + (size_t)r0 * ldP, ldP, N);
		else hipLaunchKernelGGL(k_reduce_chunks, dim3((unsigned)((N + 255) / 256), ny), dim3(256), 0, st, (const double *)d_pc, ldpc,
		                        d_rf + r0, d_P + (size_t)r0 * ldP, ldP, N);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// Chunk length of the streaming pass.  Round-2 sweep (10 000 x 131 072, K = 10): one 1000-trace chunk per group (1280
// workgroups, 5 per CU) streams at 0.795-0.80 ms, steadily; 2 / 3-4 chunks per group (2560 / 5120 workgroups) at 0.79-0.83 /
// 0.81-0.89 ms from one process to the next -- long runs per workgroup beat a full complement of waves.  Target: ~1024
// workgroups in all; a group that fits one chunk is written by the streaming kernel directly (no chunk reduction).
unsigned tspws_chunk_len_for(size_t N, size_t mtr)
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
	const unsigned clen = tspws_chunk_len_for(p->N, mtr_local);
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
	return tspws_run_chunks(p, d_x, ld, p->N, p->chunks, p->row_first, Kmax, d_P, ldP, S_(stream), true);
}

extern "C" int tspws_hip_partial_stacks_range(tspws_hip_plan *p, const float *d_x, size_t ld, size_t mtr_local, size_t first,
                                              size_t mtr_global, unsigned Kmax, unsigned g_begin, unsigned g_end, double *d_P, size_t ldP,
                                              void *stream)
{
	if (!p || (!d_x && mtr_local) || !d_P || !Kmax || !mtr_global || g_begin > g_end || g_end > Kmax || first + mtr_local > mtr_global)
		return fail(TSPWS_E_ARG, "partial_stacks_range: bad argument");
	HIP_TRY(hipSetDevice(p->device));
	build_group_chunks(p, mtr_local, first, mtr_global, Kmax);
	return tspws_run_chunks(p, d_x, ld, p->N, p->chunks, p->row_first, Kmax, d_P, ldP, S_(stream), true, g_begin, g_end);
}
