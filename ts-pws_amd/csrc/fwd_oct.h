// fwd_oct.h -- octave-fused forward frame CWT for the decimations D >= 64 (included by tspws_hip.hip).
//
//   Y_s[k] = conj( sum_l x[(k D - c_s + l) mod N] w_s[l] )                  (cdotx.c:44-70)
//
// Why: the voices of an octave share the decimation D (wavelet_def_v7.c:325-339), hence the SAME samples: with
// sample n = j D + rho (row j, residue rho) and c_v = a_v D + b_v, output k of voice v meets tap
//   l = n - k D + c_v = (q' - 1) D + rho + b_v     at row  j = k + q' - 1 - a_v,   q' = 0 .. QR-1.
// k_fwd_lds stages one x window per (scale, 64-phase chunk, 64 outputs): 46 KB of operands for 10 k wave-cycles of FMAs,
// i.e. ~29 FMAs per loaded sample, which for FP64 partial stacks (8 B/sample) is the L2 -> CU bandwidth, not the FP64
// pipe (FL_TIMING hooks: the D >= 64 workgroups spent 20 % of their cycles in the FMA passes, the rest waiting for the
// window and at the barriers around it).  Here ONE x window serves every voice of the octave:
//
//   workgroup = (octave, chunk of 64 residues rho, block of 64 outputs k), 512 threads = 8 waves, one per CU
//   LDS       = tap images of ALL voices  T_v[q'][lane]  (resident for the whole trace slice)  +  x image [XR][64]
//               image row i <-> sample (j0 + i) D + rho,  j0 = k0 - amax - 1;  output k, tap row q' of voice v reads
//               image row (k - k0) + q' + (amax - a_v): a constant row offset per voice, nothing else changes
//   wave w    = output group w (8 consecutive k), loops over the voices: sliding 8-row register window, 16 FMAs per
//               (x read, tap read); the 64 residue lanes are reduced on the VALU (valu_reduce16) and the partial of
//               chunk ci goes to part[trace][scale][ci][k] exactly like k_fwd_lds (the accumulate / gather kernels
//               are unchanged)
//   traces    = the workgroup walks its slice; the next trace's window is prefetched RAW into registers while this
//               one is computed; barriers order LDS only (fl_lds_barrier)
//
// 4 voices -> 3.6x fewer window loads, LDS stores and barriers per FMA than the per-scale kernel.
#pragma once

// Work item of the table = a SUBSET of the voices of one octave (all of them when their tap images fit next to the x image,
// else e.g. pairs): WAVES = 8 -> 512 threads, one workgroup per CU, every voice of the octave; WAVES = 4 -> 256 threads,
// 2 group passes per wave, <= 78 KB so that TWO workgroups share a CU and one computes while the other stages / waits at
// its barriers (a workgroup-wide barrier puts all its waves into the same phase: with one workgroup per CU the staging
// and barrier phases are dead time for the FP64 pipe -- FL_TIMING: 8.5 k of 23.7 k cycles per trace).
#ifndef FO_WAVES
#define FO_WAVES 4
#endif
#define FO_NT (64 * FO_WAVES)
#define FO_PASSES (8 / FO_WAVES) /* output groups per wave: a workgroup always covers 64 outputs */
#define FO_VMAX 8          /* voices per work item */
#define FO_QMAX 32         /* tap rows per voice (multiple of 4) */
#define FO_XRMAX 112       /* x image rows (multiple of 8) */
#define FO_NXV (FO_XRMAX / FO_WAVES)
#define FO_LDS_MAX ((FO_WAVES == 8 ? 156 : 78) * 1024)
#define FO_NPART 4         /* slices in which the next window is requested (one per compute pass) */
#ifndef FO_SPREAD
#define FO_SPREAD 0        /* 1: one slice per compute pass instead of one burst (measured slower) */
#endif
#define FO_CMMAX 8         /* decimations with a chunk-major copy */

struct OctFwd {
	unsigned nv;                // voices of this item
	unsigned D, Ns;
	unsigned MC, nob;           // 64-residue chunks, 64-output blocks
	unsigned XR;                // x image rows
	unsigned amax, wg_off;      // max_v a_v; first workgroup of the item in the launch
	unsigned trows;             // tap rows of all voices (sum of QR)
	unsigned pad0;
	unsigned cm_slot[2];        // [float, double input]: chunk-major copy this item reads (k_chunk_major), ~0u: the traces themselves
	unsigned long long cm_pre[2]; // elements per trace of the slots in front of it
	unsigned sc[FO_VMAX];       // scale index of each voice
	unsigned QR[FO_VMAX], trow[FO_VMAX]; // tap rows of each voice (multiple of 4), first row of its image
	unsigned a[FO_VMAX], b[FO_VMAX], L[FO_VMAX];
	unsigned long long tap_off[FO_VMAX], part_off[FO_VMAX];
};

// one workgroup: `bid` = its index among the workgroups of a trace slice, `slice` = trace slice
template <typename TIn>
__device__ __forceinline__ void fwd_oct_workgroup(const unsigned bid, const unsigned slice, char *smem, const TIn *__restrict__ x, size_t ld,
                                                  unsigned ntr, unsigned tps, unsigned N, const OctFwd *__restrict__ oc, unsigned noct,
                                                  const double2 *__restrict__ w, double2 *__restrict__ part, size_t npart,
                                                  const TIn *__restrict__ xcm, unsigned ntr_all)
{
	constexpr int R = 8;
	unsigned lo = 0, hi = noct;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (oc[mid].wg_off <= bid) lo = mid; else hi = mid;
	}
	const OctFwd *__restrict__ o = oc + lo;
	const unsigned D = o->D, Ns = o->Ns, nv = o->nv, XR = o->XR, amax = o->amax, trows = o->trows;
	const unsigned wl = bid - o->wg_off;
	const unsigned ci = wl / o->nob, ob = wl - ci * o->nob;
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned rho0 = ci * 64u;
	const bool mok = rho0 + lane < D;
	const bool full = rho0 + 63u < D;
	const unsigned k0 = ob * 64u;
	double2 *tL = (double2 *)smem;                                      // [trows][64]: voice v owns rows trow[v] .. trow[v] + QR[v] - 1
	double *xL = (double *)(smem + (size_t)trows * 64 * sizeof(double2)); // [XR][64]

	const unsigned t0 = slice * tps;
	const unsigned nt = (ntr - t0) < tps ? (ntr - t0) : tps;
	const TIn *x0 = x + (size_t)t0 * ld;

	// ---- tap images (zero outside the filter / for idle residue lanes) -----------------------------------------------
	for (unsigned idx = tid; idx < trows * 64u; idx += FO_NT) {
		const unsigned row = idx >> 6, ln = idx & 63u;
		unsigned v = 0;
		while (v + 1 < nv && o->trow[v + 1] <= row) v++;
		const unsigned q = row - o->trow[v];
		const long long l = ((long long)q - 1) * D + rho0 + ln + o->b[v];
		const bool ok = (rho0 + ln < D) && l >= 0 && l < (long long)o->L[v];
		tL[idx] = ok ? w[o->tap_off[v] + (unsigned long long)l] : make_double2(0.0, 0.0);
	}

	// ---- x window: rows wv, wv + 8, ... of the image; sample (j0 + i) D + rho, circular -------------------------------
	const unsigned nxr = XR / FO_WAVES; // rows per wave (XR is a multiple of 8)
	const long long base = ((long long)k0 - (long long)amax - 1 + (long long)wv) * D + rho0; // lane 0 of this wave's first row
	const unsigned idx0 = wrap_index(base + (mok ? lane : 0), N);                            // idle lanes read lane 0's sample
	const unsigned step = (unsigned)(((unsigned long long)FO_WAVES * D) % N);
	const bool nowrap = base >= 0 && base + (long long)(nxr - 1) * FO_WAVES * D + 63 < (long long)N; // wave-uniform
	// Chunk-major input (o->cm): rows of one residue chunk are contiguous, [trace][chunk][row j][64], so the window is one
	// contiguous block instead of 512-byte rows D samples apart.  Why: a CU's L1 TLB covers ~1 MB; the windows of two
	// workgroups with row strides above ~4 KB span more and every row load then waits for a TLB refill (tools/
	// stride_probe.hip: 22 row loads issue in 1.4 k cycles up to 4 KB stride, 5.9 k beyond).  k_chunk_major writes the copy.
	constexpr int TI = sizeof(TIn) == 8 ? 1 : 0;
	const bool cm = xcm != nullptr && o->cm_slot[TI] != ~0u;
	const unsigned NJ = Ns;                                   // rows per chunk (D | N in this mode)
	const long long j0w = (long long)k0 - (long long)amax - 1 + (long long)wv; // first image row of this wave, as a trace row
	unsigned jw = 0;
	bool cm_nowrap = false;
	if (cm) {
		long long jj = j0w % (long long)NJ; if (jj < 0) jj += NJ;
		jw = (unsigned)jj;
		cm_nowrap = jw + (nxr - 1) * FO_WAVES < NJ;
	}
	// Branch-free: all FO_NXV register slots are always loaded / stored; slots past the image (i >= nxr, wave-uniform)
	// repeat its last row -- a per-slot guard would put every load into its own basic block.  Addressing is a uniform
	// base pointer + one 32-bit element offset per thread that walks the rows; the empty asm makes that offset opaque per
	// trace, otherwise the compiler hoists all FO_NXV row offsets (they do not depend on the trace) out of the trace loop
	// as 64-bit values: 56 VGPRs, spills, and waits in the middle of the load sequence.
	// The window of the next trace is requested in FO_NPART slices spread over the compute passes: issued in one burst,
	// the 22-28 loads of every wave of both workgroups exceed what a CU keeps in flight and the issue itself stalls for
	// ~6 k cycles (FL_TIMING) -- spread out, the requests flow while the FMAs run.
	const TIn *row0 = nullptr;
	unsigned off = 0, stp = 0, mod = 0;
	auto load_begin = [&](const unsigned t) {
		if (cm) {
			row0 = xcm + (size_t)ntr_all * o->cm_pre[TI] + ((size_t)(t0 + t) * o->MC + ci) * (size_t)NJ * 64;
			off = jw * 64u + lane; stp = FO_WAVES * 64u; mod = cm_nowrap ? 0u : NJ * 64u;
		} else {
			row0 = x0 + (size_t)t * ld;
			off = idx0; stp = nowrap ? FO_WAVES * D : step; mod = nowrap ? 0u : N;
		}
		asm volatile("" : "+v"(off));
	};
	auto load_part = [&](TIn (&xv)[FO_NXV], const int part) { // slots [part * FO_NXV / FO_NPART, (part + 1) * FO_NXV / FO_NPART)
#pragma unroll
		for (int i = 0; i < FO_NXV; i++) {
			if (i * FO_NPART / FO_NXV != part) continue; // compile-time after unrolling
			xv[i] = row0[off];
			off += (unsigned)(i + 1) < nxr ? stp : 0u; // scalar select: stop at the last row
			if (mod && off >= mod) off -= mod;
		}
	};
	auto store_x = [&](const TIn (&xv)[FO_NXV]) {
		double *xdst = xL + wv * 64 + lane;
		if (full) {
#pragma unroll
			for (int i = 0; i < FO_NXV; i++) {
				const unsigned ii = (unsigned)i < nxr ? (unsigned)i : nxr - 1;
				xdst[FO_NT * ii] = (double)xv[i];
			}
		} else {
#pragma unroll
			for (int i = 0; i < FO_NXV; i++) {
				const unsigned ii = (unsigned)i < nxr ? (unsigned)i : nxr - 1;
				xdst[FO_NT * ii] = mok ? (double)xv[i] : 0.0;
			}
		}
	};

	TIn xv[FO_NXV];
	load_begin(0);
#pragma unroll
	for (int pt = 0; pt < FO_NPART; pt++) load_part(xv, pt);
	const unsigned o16 = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1); // element left by valu_reduce16
	const bool group_live = k0 + wv * R < Ns; // waves past the last output group only help with the staging (pass p: group p * WAVES + wv)

#if FL_TIMING
	unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define FO_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tm[i] += n_ - tc; tc = n_; } while (0)
#else
#define FO_STAMP(i) do { } while (0)
#endif
	FO_STAMP(0);
	for (unsigned t = 0; t < nt; t++) {
		fl_lds_barrier(); // everyone is done reading the previous image (the first one also publishes nothing yet)
		FO_STAMP(1);
		store_x(xv);
		FO_STAMP(2);
		fl_lds_barrier(); // image (and, the first time, the tap images) visible
		FO_STAMP(3);
#ifndef FO_ABL
#define FO_ABL 0 /* ablation builds: 1 = no window loads after the first trace, 2 = every window from trace 0 (L2-warm) */
#endif
		const bool more = t + 1 < nt && FO_ABL != 1;
		if (more) load_begin(FO_ABL == 2 ? 0 : t + 1);
		int parts_done = 0;
		if (more && !FO_SPREAD) {
#pragma unroll
			for (int pt = 0; pt < FO_NPART; pt++) load_part(xv, pt);
			parts_done = FO_NPART;
		}
		FO_STAMP(0); // (timing build: the issue of the prefetch is booked under "setup")
		if (!group_live) {
			if (more && FO_SPREAD) {
#pragma unroll
				for (int pt = 0; pt < FO_NPART; pt++) load_part(xv, pt);
			}
			continue;
		}
		for (unsigned pv = 0; pv < (unsigned)FO_PASSES * nv; pv++) {
			if (more && FO_SPREAD && pv < (unsigned)FO_NPART) { // one slice of the next window per pass
				switch (pv) {
				case 0: load_part(xv, 0); break;
				case 1: load_part(xv, 1); break;
				case 2: load_part(xv, 2); break;
				default: load_part(xv, 3); break;
				}
				parts_done = (int)pv + 1;
			}
			const unsigned ps = FO_PASSES == 1 ? 0u : pv / nv, v = FO_PASSES == 1 ? pv : pv - ps * nv;
			const unsigned grp = ps * FO_WAVES + wv; // output group of this pass
			if (k0 + grp * R >= Ns) break;
			const unsigned QR = o->QR[v];
			const double *xb = xL + ((size_t)grp * R + (amax - o->a[v])) * 64 + lane;
			const double2 *tb = tL + (size_t)o->trow[v] * 64 + lane;
			double ar[R], ai[R];
#pragma unroll
			for (int r = 0; r < R; r++) { ar[r] = 0; ai[r] = 0; }
			double xw[R];
#pragma unroll
			for (int j = 0; j < R - 1; j++) xw[j] = xb[j * 64];
			for (unsigned sb = 0; sb < QR; sb += R) { // blocks of 8 tap rows: the window rotation is static inside a block
#pragma unroll
				for (int h = 0; h < 2; h++) {
					if (h == 0 || sb + 4u < QR) { // QR is a multiple of 4
						double xn[4];
						double2 tn[4];
#pragma unroll
						for (int u = 0; u < 4; u++) {
							xn[u] = xb[(h * 4 + u + R - 1) * 64];
							tn[u] = tb[(h * 4 + u) * 64];
						}
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const int sidx = h * 4 + u;
							xw[(sidx + R - 1) % R] = xn[u];
#pragma unroll
							for (int r = 0; r < R; r++) {
								ar[r] = fma(xw[(sidx + r) % R], tn[u].x, ar[r]);
								ai[r] = fma(xw[(sidx + r) % R], tn[u].y, ai[r]);
							}
						}
					}
				}
				xb += R * 64; tb += R * 64;
			}
			FO_STAMP(4);
			double v16[2 * R];
#pragma unroll
			for (int r = 0; r < R; r++) { v16[2 * r] = ar[r]; v16[2 * r + 1] = ai[r]; }
			const double sum = valu_reduce16(v16, lane);
			const unsigned kout = k0 + grp * R + (o16 >> 1);
			if ((lane & 3) == 0 && kout < Ns) {
				double *pout = (double *)(part + (size_t)(t0 + t) * npart + o->part_off[v] + (size_t)ci * Ns);
				pout[(size_t)kout * 2 + (o16 & 1)] = (o16 & 1) ? -sum : sum; // conj
			}
			FO_STAMP(5);
		}
		if (more) { // fewer passes than slices (or the group loop ended early): the rest of the window
			if (parts_done <= 0) load_part(xv, 0);
			if (parts_done <= 1) load_part(xv, 1);
			if (parts_done <= 2) load_part(xv, 2);
			if (parts_done <= 3) load_part(xv, 3);
		}
	}
#if FL_TIMING
	if (lane == 0 && fl_timing_out && group_live) { // class 0 of the timing table (k_fwd_lds never uses it when D = 1 is absent)
#pragma unroll
		for (int i = 0; i < 6; i++) atomicAdd(&fl_timing_out[i], tm[i]);
		atomicAdd(&fl_timing_out[6], (unsigned long long)nt);
		atomicAdd(&fl_timing_out[7], 1ull);
	}
#endif
#undef FO_STAMP
}


template <typename TIn>
__global__ void __launch_bounds__(FO_NT, 2) k_fwd_oct(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned tps, unsigned N,
                                                      const OctFwd *__restrict__ oc, unsigned noct, const double2 *__restrict__ w,
                                                      double2 *__restrict__ part, size_t npart, const TIn *__restrict__ xcm,
                                                      unsigned ntr_all)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	fwd_oct_workgroup<TIn>(blockIdx.x, blockIdx.y, smem, x, ld, ntr, tps, N, oc, noct, w, part, npart, xcm, ntr_all);
}

#if FO_WAVES == 4
// ONE launch for the octave-fused workgroups (D >= 64) and the LDS-kernel workgroups (D < 64): same block size, LDS = the
// larger of the two, no streams / events between them and the block scheduler balances the two kinds.  The octave
// workgroups (long) come first, `oct_slices` trace slices of them; the LDS workgroups follow, coarse scales first.
template <typename TIn, bool FUSE>
__global__ void __launch_bounds__(FL_NT, 2) k_fwd_lds_oct(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned tps, unsigned N,
                                                          const ScaleDesc *__restrict__ sc, unsigned S, const double2 *__restrict__ w,
                                                          double2 *__restrict__ part, size_t npart, double2 *__restrict__ accST,
                                                          double2 *__restrict__ accPS, size_t acc_stride, unsigned lds_blocks,
                                                          const OctFwd *__restrict__ oc, unsigned noct, unsigned oct_wgs, unsigned oct_slices,
                                                          unsigned oct_tps, const TIn *__restrict__ xcm)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const unsigned noctb = oct_wgs * oct_slices;
	if (blockIdx.x < noctb) {
		if (blockIdx.y) return; // the octave workgroups slice the traces themselves
		const unsigned sl = blockIdx.x / oct_wgs;
		fwd_oct_workgroup<TIn>(blockIdx.x - sl * oct_wgs, sl, smem, x, ld, ntr, oct_tps, N, oc, noct, w, part, npart, xcm, ntr);
		return;
	}
	const unsigned bid = lds_blocks - 1u - (blockIdx.x - noctb); // coarse scales first
	fwd_lds_workgroup<TIn, FUSE>(bid, blockIdx.y, smem, x, ld, ntr, tps, N, sc, S, w, part, npart, accST, accPS, acc_stride);
}
#endif

// k_fwd_oct2: the same octave-fused transform as a two-team software pipeline.
//
// Why: with FL_TIMING the window load turned out to be bound by what one CU pulls in (~16 B/cycle: 90 KB for the two
// workgroups of a CU = 5-6 k cycles per trace), as long as the FMA passes of a trace (5 k cycles per wave), and every
// wave of both workgroups did it at the same moment -- the FP64 pipe idled through the loads, then two waves per SIMD
// shared it.  Here ONE 512-thread workgroup per CU is split into two TEAMS of four waves (32 outputs each, own x image,
// all tap images shared).  Workgroup barriers cut the time into intervals; in every interval one team COMPUTES its
// trace (one FMA-ing wave per SIMD: a dense v_fma_f64 stream of a single wave sustains ~85 % of the pipe,
// tools/fma64_issue.hip) while the other STAGES its next window (LDS writes, then the loads of the window after that,
// which have two intervals to arrive):
//
//     interval      0        1        2        3        4     ...
//     team 0     stage t0  comp t0  stage t1  comp t1  stage t2
//     team 1        -      stage t0  comp t0  stage t1  comp t1
//
// All voices of the octave share one window (3.6x fewer window bytes per FMA than the per-scale kernel) and the loads
// of one team run beside the FMAs of the other.
#define FT_NT 512
#define FT_XRMAX 64        /* rows of a team image (multiple of 4) */
#define FT_NXV (FT_XRMAX / 4)
#define FT_LDS_MAX (156 * 1024)
#define FT_XPAD 8          /* rows past an image that the operand prefetch may touch */

template <typename TIn>
__global__ void __launch_bounds__(FT_NT, 2) k_fwd_oct2(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned tps, unsigned N,
                                                       const OctFwd *__restrict__ oc, unsigned noct, const double2 *__restrict__ w,
                                                       double2 *__restrict__ part, size_t npart, const TIn *__restrict__ xcm,
                                                       unsigned ntr_all)
{
	constexpr int R = 8;
	constexpr int TI = sizeof(TIn) == 8 ? 1 : 0;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	unsigned lo = 0, hi = noct;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (oc[mid].wg_off <= blockIdx.x) lo = mid; else hi = mid;
	}
	const OctFwd *__restrict__ o = oc + lo;
	const unsigned D = o->D, Ns = o->Ns, nv = o->nv, XR = o->XR, amax = o->amax, trows = o->trows;
	const unsigned wl = blockIdx.x - o->wg_off;
	const unsigned ci = wl / o->nob, ob = wl - ci * o->nob;
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const unsigned team = wv >> 2, tw = wv & 3;
	const unsigned rho0 = ci * 64u;
	const bool mok = rho0 + lane < D;
	const bool full = rho0 + 63u < D;
	const unsigned kt0 = ob * 64u + team * 32u; // first output of the team
	double2 *tL = (double2 *)smem;                                                            // [trows][64]
	double *xLt = (double *)(smem + (size_t)trows * 64 * sizeof(double2)) + (size_t)team * (XR + FT_XPAD) * 64; // this team's [XR][64]

	const unsigned t0 = blockIdx.y * tps;
	const unsigned nt = (ntr - t0) < tps ? (ntr - t0) : tps;
	const TIn *x0 = x + (size_t)t0 * ld;

	for (unsigned idx = tid; idx < trows * 64u; idx += FT_NT) { // tap images, zero outside the filter / for idle residue lanes
		const unsigned row = idx >> 6, ln = idx & 63u;
		unsigned v = 0;
		while (v + 1 < nv && o->trow[v + 1] <= row) v++;
		const unsigned q = row - o->trow[v];
		const long long l = ((long long)q - 1) * D + rho0 + ln + o->b[v];
		const bool ok = (rho0 + ln < D) && l >= 0 && l < (long long)o->L[v];
		tL[idx] = ok ? w[o->tap_off[v] + (unsigned long long)l] : make_double2(0.0, 0.0);
	}

	// ---- window of the team: image row i <-> sample (kt0 - amax - 1 + i) D + rho; wave tw owns rows tw, tw + 4, ... ------
	const unsigned nxr = XR / 4;
	const long long j0w = (long long)kt0 - (long long)amax - 1 + (long long)tw;
	const long long base = j0w * D + rho0;
	const unsigned idx0 = wrap_index(base + (mok ? lane : 0), N);
	const unsigned step = (unsigned)((4ull * D) % N);
	const bool nowrap = base >= 0 && base + (long long)(nxr - 1) * 4 * D + 63 < (long long)N;
	const bool cm = xcm != nullptr && o->cm_slot[TI] != ~0u;
	const unsigned NJ = Ns;
	unsigned jw = 0;
	bool cm_nowrap = false;
	if (cm) {
		long long jj = j0w % (long long)NJ; if (jj < 0) jj += NJ;
		jw = (unsigned)jj;
		cm_nowrap = jw + (nxr - 1) * 4 < NJ;
	}
	auto load_x = [&](TIn (&xv)[FT_NXV], const unsigned t) {
		const TIn *row0;
		unsigned off, stp, mod;
		if (cm) {
			row0 = xcm + (size_t)ntr_all * o->cm_pre[TI] + ((size_t)(t0 + t) * o->MC + ci) * (size_t)NJ * 64;
			off = jw * 64u + lane; stp = 4u * 64u; mod = cm_nowrap ? 0u : NJ * 64u;
		} else {
			row0 = x0 + (size_t)t * ld;
			off = idx0; stp = nowrap ? 4u * D : step; mod = nowrap ? 0u : N;
		}
		asm volatile("" : "+v"(off)); // see k_fwd_oct
#pragma unroll
		for (int i = 0; i < FT_NXV; i++) {
			xv[i] = row0[off];
			off += (unsigned)(i + 1) < nxr ? stp : 0u;
			if (mod && off >= mod) off -= mod;
		}
	};
	auto store_x = [&](const TIn (&xv)[FT_NXV]) {
		double *xdst = xLt + tw * 64 + lane;
#pragma unroll
		for (int i = 0; i < FT_NXV; i++) {
			const unsigned ii = (unsigned)i < nxr ? (unsigned)i : nxr - 1;
			xdst[256 * ii] = (full || mok) ? (double)xv[i] : 0.0;
		}
	};

	const unsigned o16 = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1); // element left by valu_reduce16
	const unsigned kout = kt0 + tw * R + (o16 >> 1);
	const bool writer = (lane & 3) == 0 && kout < Ns;
	const bool group_live = kt0 + tw * R < Ns;
	const bool team_live = kt0 < Ns; // a team past the last outputs neither stages nor computes

	auto compute = [&](const unsigned t) {
		for (unsigned v = 0; v < nv; v++) {
			const unsigned QR = o->QR[v];
			const double *xb = xLt + ((size_t)tw * R + (amax - o->a[v])) * 64 + lane;
			const double2 *tb = tL + (size_t)o->trow[v] * 64 + lane;
			double ar[R], ai[R];
#pragma unroll
			for (int r = 0; r < R; r++) { ar[r] = 0; ai[r] = 0; }
			double xw[R];
#pragma unroll
			for (int j = 0; j < R - 1; j++) xw[j] = xb[j * 64];
			// operands of the next burst are requested before the FMAs of this one: the only computing wave of its SIMD
			// must cover its own LDS latency.  Reads past QR touch rows that exist in LDS and are never used.
			double xn[2][4];
			double2 tn[2][4];
#pragma unroll
			for (int u = 0; u < 4; u++) { xn[0][u] = xb[(u + R - 1) * 64]; tn[0][u] = tb[u * 64]; }
			for (unsigned sb = 0; sb < QR; sb += R) { // QR is a multiple of 4
#pragma unroll
				for (int u = 0; u < 4; u++) { xn[1][u] = xb[(4 + u + R - 1) * 64]; tn[1][u] = tb[(4 + u) * 64]; }
				asm volatile("" ::: "memory"); // the scheduler otherwise sinks these reads to just in front of their FMAs
#pragma unroll
				for (int u = 0; u < 4; u++) {
					xw[(u + R - 1) % R] = xn[0][u];
#pragma unroll
					for (int r = 0; r < R; r++) {
						ar[r] = fma(xw[(u + r) % R], tn[0][u].x, ar[r]);
						ai[r] = fma(xw[(u + r) % R], tn[0][u].y, ai[r]);
					}
				}
#pragma unroll
				for (int u = 0; u < 4; u++) { xn[0][u] = xb[(8 + u + R - 1) * 64]; tn[0][u] = tb[(8 + u) * 64]; }
				asm volatile("" ::: "memory");
				if (sb + 4u < QR) {
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int sidx = 4 + u;
						xw[(sidx + R - 1) % R] = xn[1][u];
#pragma unroll
						for (int r = 0; r < R; r++) {
							ar[r] = fma(xw[(sidx + r) % R], tn[1][u].x, ar[r]);
							ai[r] = fma(xw[(sidx + r) % R], tn[1][u].y, ai[r]);
						}
					}
				}
				xb += R * 64; tb += R * 64;
			}
			double v16[2 * R];
#pragma unroll
			for (int r = 0; r < R; r++) { v16[2 * r] = ar[r]; v16[2 * r + 1] = ai[r]; }
			const double sum = valu_reduce16(v16, lane);
			if (writer) {
				double *pout = (double *)(part + (size_t)(t0 + t) * npart + o->part_off[v] + (size_t)ci * Ns);
				pout[(size_t)kout * 2 + (o16 & 1)] = (o16 & 1) ? -sum : sum; // conj
			}
		}
	};

	TIn xv[FT_NXV];
	if (team_live) load_x(xv, 0);
	const int nint = 2 * (int)nt + 1;
#if FL_TIMING
	unsigned long long tm[6] = {0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define FT_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tm[i] += n_ - tc; tc = n_; } while (0)
#else
#define FT_STAMP(i) do { } while (0)
#endif
	for (int it = 0; it < nint; it++) {
		fl_lds_barrier();
		FT_STAMP(1); // barrier wait (= the rest of the other team's interval)
		const int rel = it - (int)team; // team 1 runs one interval behind
		if (rel < 0 || !team_live) continue;
		const unsigned t = (unsigned)rel >> 1;
		if (t >= nt) continue;
		if ((rel & 1) == 0) { // stage: the window requested two intervals ago goes to LDS, the next one is requested
			store_x(xv);
			FT_STAMP(2); // wait for the window + LDS stores
			if (t + 1 < nt) load_x(xv, t + 1);
			FT_STAMP(3); // issue of the next window's loads
		} else if (group_live) { compute(t); FT_STAMP(4); }
	}
#if FL_TIMING
	if (lane == 0 && fl_timing_out && group_live) {
#pragma unroll
		for (int i = 0; i < 6; i++) atomicAdd(&fl_timing_out[i], tm[i]);
		atomicAdd(&fl_timing_out[6], (unsigned long long)nt);
		atomicAdd(&fl_timing_out[7], 1ull);
	}
#endif
#undef FT_STAMP
}

// Chunk-major copies of a batch of traces for the decimations whose rows are far apart (see load_x above):
//   dst[slot][trace][ci][j][lane] = x[trace][j D + 64 ci + lane]   (0 where 64 ci + lane >= D), D | N, j < N / D.
// One thread per destination element group: reads are 256-byte runs, writes fully coalesced.
struct ChunkMajor { unsigned n, D[FO_CMMAX], MC[FO_CMMAX]; };

template <typename TIn>
__global__ void __launch_bounds__(256) k_chunk_major(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N, ChunkMajor cmj,
                                                     TIn *__restrict__ dst)
{
	const unsigned slot = blockIdx.z, tr = blockIdx.y;
	const unsigned D = cmj.D[slot], MC = cmj.MC[slot], NJ = N / D;
	size_t off = 0; // elements of the earlier slots
	for (unsigned s = 0; s < slot; s++) off += (size_t)ntr * cmj.MC[s] * (N / cmj.D[s]) * 64;
	const size_t per_trace = (size_t)MC * NJ * 64;
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; // element of [ci][j][lane]
	if (e >= per_trace) return;
	const unsigned lane = (unsigned)(e & 63);
	const size_t row = e >> 6;                // ci * NJ + j
	const unsigned ci = (unsigned)(row / NJ), j = (unsigned)(row - (size_t)ci * NJ);
	const unsigned rho = ci * 64 + lane;
	TIn v = 0;
	if (rho < D) v = x[(size_t)tr * ld + (size_t)j * D + rho];
	dst[off + (size_t)tr * per_trace + e] = v;
}
