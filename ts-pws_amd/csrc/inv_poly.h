// inv_poly.h -- register-tiled polyphase inverse frame transform (real part) for gfx950.
//
//   x^[n] = sum_s gain_s D_s sum_{l : D_s | (n - cd_s + l)} Re( conj(wd_s[l]) Y_s[(n - cd_s + l)/D_s] )
//                                                   (wavelet_v7.c:138-147, cdotx.c:305-340 / :176-211)
//
// Decimations that divide N: the zero-stuffed grid is circular and the seam restart of cdotx.c:331-332 coincides with
// plain mod-N_s indexing.  Other decimations: the three-frame form in the kernel (same thread mapping).  Write n - cd = a D + rho; only taps
// l = q D + mu, mu = (-rho) mod D, land on the grid and hit coefficient a + q + (rho > 0).  For a fixed
// output phase the inverse is therefore a stride-1 correlation between the coefficient row and a Q-tap
// sub-filter -- the same sliding-window structure as the forward kernel, but with NO reduction across
// lanes: every thread owns its outputs.
//
//   work item = (octave, output phase n0 in [0,D), group of R outputs n0 + (gR+r) D, NREC coefficient sets)
//   lanes     = consecutive n0 (taps and outputs coalesced; the coefficient row is wave-uniform when D >= 64)
//   all V voices of the octave (same D) are summed in registers; octave sums go to obuf[octave][rec][N]
//   and k_inv_combine adds the <= J octave rows (+ the generic-path row) in a fixed order.
//
// Round-1 measurements (north-star frame, 2 coefficient sets: 3584 waves, all resident at once, 50 us): a wave's life is
// ~13 dependent memory round trips (window preload + blocks of 8 tap steps, for each of the 4 voices) of ~3 us each --
// 24 vector loads per lane and block from 14 lock-stepped waves per CU -- while its 1.9 k FMAs need < 4 us.  Tried and
// not faster: the 4 voices on 4 waves with an LDS reduction (49 us: twice the waves no longer fit in one round), all
// taps of a voice hoisted into registers (152 VGPRs -> two rounds, 112 us), the partial last block as one guarded
// block (50 us), and an LDS-staged variant (two waves per workgroup, per voice one bulk load of the tap tile and of the
// coefficient windows, steps out of LDS: 45.6 us at 166 VGPRs / two rounds, 52 us capped to 128 VGPRs).  Its timers
// put the steps themselves at 24 us per wave with four waves per SIMD: the 6.9 M wave-FMAs of the two reconstructions
// are 13 us of pipe time at the full v_fma_f64 rate, ~30 us at the rate this chip sustains at that occupancy -- the
// kernel is closer to its arithmetic bound than the round-trip count suggests.
#pragma once

#define INV_R 8
#ifndef INV_UNIFORM
#define INV_UNIFORM 1                           /* 1: D >= 64: the coefficient windows of a wave through the scalar unit (the stack's pair 49 -> 44 us, cfg4 with its batched replicas 3.61 -> 3.48 ms) */
#endif

#ifndef FL_ABLATE
#define FL_ABLATE 0
#endif
#if FL_ABLATE
__device__ unsigned inv_class_mask = 0x7fu;   // debug builds: bit c = the octaves of lanes-per-phase class log2(DL) = c run (6: D >= 64); TSPWS_INV_CLASSES
#endif

// LDS-staged form for the finely decimated octaves (D < 64; round 4).  There a wave's lanes are (output group, phase) and the
// coefficient windows of adjacent groups start R = 8 coefficients apart: a per-lane 16-byte load touches up to 64 different
// cache lines per wave (D = 1), and with two coefficient sets + the tap that is ~144 address-path cycles per tap step against
// 128 FMA cycles on each of the four SIMDs -- the batched reconstructions of the jackknife (cfg4: 12 at once) ran at 27 % of the
// FP64 pipe.  Here each wave copies the coefficient window of a block of INV_QB tap steps into its own slab of LDS with coalesced
// loads (row e at e + e / 8: the sliding windows of the lanes then fall on distinct banks) and the steps read it from there.
// Same additions in the same order as the per-lane form: bit-identical outputs.
#define INV_QB 32                               /* tap steps per staged block */
#define INV_WROWS (64 * INV_R + INV_QB + 16)    /* rows of a wave's window: <= 512 outputs + the block's steps + the sliding window + shifts */
#define INV_WPAD(e) ((e) + ((e) >> 3))
#define INV_SLAB (INV_WROWS + INV_WROWS / 8 + 2)

struct OctDesc {
	unsigned s0, nv;        // first scale, voices
	unsigned D, Ns;
	unsigned DL, logDL, MC; // lanes per phase block, 64-phase chunks
	unsigned ngw;           // group-blocks per chunk
	unsigned wave_off;
	unsigned slot;          // row of obuf
	unsigned gen, pad1;     // gen: D does not divide N (three-frame form, see k_inv_poly)
};

// GEN = the three-frame form for octaves whose D does not divide N (its own instantiation: the masked loads and 64-bit
// index arithmetic would otherwise cost the common case registers and time; the host launches the two classes separately)
template <int NREC, bool GEN, bool LDSW = false>
__global__ void __launch_bounds__(256) k_inv_poly(const double2 *__restrict__ Y, size_t ncoef, unsigned N, const ScaleDesc *__restrict__ sc,
                                                  const OctDesc *__restrict__ oc, unsigned noct, const double2 *__restrict__ wd,
                                                  double *__restrict__ obuf, size_t slot_stride, unsigned total_waves, size_t y_coef,
                                                  size_t y_obuf, unsigned wave_base)
{
	Y += (size_t)blockIdx.y * y_coef; obuf += (size_t)blockIdx.y * y_obuf; // blockIdx.y = independent reconstruction set
	constexpr int R = INV_R;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wid = wave_base + blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (wid >= total_waves) return;
	unsigned lo = 0, hi = noct;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (oc[mid].wave_off <= wid) lo = mid; else hi = mid;
	}
	const OctDesc o = oc[lo];
#if FL_ABLATE
	if (!((inv_class_mask >> o.logDL) & 1u)) return;
#endif
	const unsigned wl = wid - o.wave_off;
	const unsigned chunk = wl / o.ngw, gb = wl - chunk * o.ngw;
	const unsigned n0 = chunk * 64 + (lane & (o.DL - 1));
	const unsigned g = gb * (64u / o.DL) + (lane >> o.logDL);
	const bool live = n0 < o.D && g * R < o.Ns;
	const unsigned n0c = n0 < o.D ? n0 : 0;
	const unsigned D = o.D, Ns = o.Ns;

	double acc[NREC][R];
#pragma unroll
	for (int c = 0; c < NREC; c++)
#pragma unroll
		for (int r = 0; r < R; r++) acc[c][r] = 0;

	if constexpr (LDSW) {
		// (this instantiation is launched for the waves of the octaves with D <= INV_LDS_MAXD whose D divides N only)
		__shared__ __attribute__((aligned(16))) double2 slab_all[4][NREC][INV_SLAB];
		double2 (*slab)[INV_SLAB] = slab_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
		const unsigned GW = 64u >> o.logDL;
		for (unsigned v = 0; v < o.nv; v++) {
			const ScaleDesc d = sc[o.s0 + v];
			const double2 *ws = wd + d.tap_off;
			const double2 *ys = Y + d.coef_off;
			// n - cd = a D + rho for n = n0 + (gR + r) D
			const long long t = (long long)n0c - d.cd;
			long long a = t >= 0 ? t / D : -((-t + D - 1) / D);
			const unsigned rho = (unsigned)(t - a * (long long)D);
			const unsigned mu = rho ? D - rho : 0;
			a += (long long)g * R + (rho ? 1 : 0);
			// the wave's window starts at the coefficient of its first group's phase 0 (the smallest index any lane uses)
			const long long tb = -(long long)d.cd;
			const long long ab = (tb >= 0 ? tb / D : -((-tb + D - 1) / D)) + (long long)(gb * GW) * R;
			const unsigned off = (unsigned)(a - ab);   // <= GW R + 1
			const double gD = d.gain * (double)D;
			unsigned l = mu;
			for (unsigned q0 = 0; q0 < d.Q; q0 += INV_QB) {
				const unsigned nq = d.Q - q0 < (unsigned)INV_QB ? d.Q - q0 : (unsigned)INV_QB;
				const unsigned nq8 = (nq + R - 1) / R * R;
				const unsigned need = GW * R + 2 + nq8 + R; // rows the block's steps can touch
				long long sm = (ab + (long long)q0) % (long long)Ns;
				if (sm < 0) sm += Ns;
				const unsigned smu = (unsigned)sm;
				const bool one_wrap = Ns >= need; // (the window wraps at most once: one conditional subtraction instead of a software division per row)
				// (the previous block's reads of this slab are complete: LDS operations of a wave execute in order)
				for (unsigned e = lane; e < need; e += 64) {
					unsigned row = smu + e;
					if (one_wrap) { if (row >= Ns) row -= Ns; } else row %= Ns;
#pragma unroll
					for (int c = 0; c < NREC; c++) slab[c][INV_WPAD(e)] = ys[(size_t)c * ncoef + row];
				}
				double2 yw[NREC][R];
#pragma unroll
				for (int j = 0; j < R - 1; j++) {
#pragma unroll
					for (int c = 0; c < NREC; c++) yw[c][j] = slab[c][INV_WPAD(off + (unsigned)j)];
				}
				for (unsigned qq = 0; qq < nq8; qq += R) {
					// the block's R taps first (independent loads; taps of steps past the block / past the filter are exact zeros and add
					// nothing), then its window rows, then straight-line FMAs
					double2 tp[R];
#pragma unroll
					for (int u = 0; u < R; u++) {
						const unsigned lu = l + (unsigned)u * D;
						const bool on = qq + (unsigned)u < nq && lu < d.L;
						tp[u] = ws[on ? lu : 0u];
						tp[u].x = on ? tp[u].x * gD : 0.0; tp[u].y = on ? tp[u].y * gD : 0.0;
					}
#pragma unroll
					for (int u = 0; u < R; u++) {
						const unsigned e = off + qq + (unsigned)u + (unsigned)(R - 1);
#pragma unroll
						for (int c = 0; c < NREC; c++) yw[c][(u + R - 1) % R] = slab[c][INV_WPAD(e)];
#pragma unroll
						for (int c = 0; c < NREC; c++)
#pragma unroll
							for (int r = 0; r < R; r++)
								acc[c][r] = fma(tp[u].x, yw[c][(u + r) % R].x, fma(tp[u].y, yw[c][(u + r) % R].y, acc[c][r]));
					}
					l += (unsigned)R * D; // (a partial last block leaves l past its steps: the next voice starts over)
				}
			}
		}
	} else if constexpr (GEN) {
		// D does not divide N (cdotx.c:313-337: the zero-stuffed grid restarts at the circular seam).  With the raw position
		// P = n - cd + l in (-N, 2N), frame f = floor(P / N) and in-frame position p = P - f N, a tap is on the grid iff
		// D | p and then meets coefficient p / D.  Per frame this is the ordinary polyphase correlation for the shifted
		// output position n - f N, WITHOUT any wrap of the coefficient index: indices outside [0, Ns) belong to another frame
		// and are masked.  Frames -1 / +1 only matter next to the seam: a wave skips a frame none of its lanes can reach.
		for (unsigned v = 0; v < o.nv; v++) {
			const ScaleDesc d = sc[o.s0 + v];
			const double2 *ws = wd + d.tap_off;
			const double2 *ys = Y + d.coef_off;
			const double gD = d.gain * (double)D;
			for (int f = -1; f <= 1; f++) {
				const long long t = (long long)n0c - d.cd - (long long)f * (long long)N;
				long long a = t >= 0 ? t / D : -((-t + D - 1) / D);
				const unsigned rho = (unsigned)(t - a * (long long)D);
				const unsigned mu = rho ? D - rho : 0;
				a += (long long)g * R + (rho ? 1 : 0);          // coefficient index of output r at step q: a + q + r
				// steps that can meet a coefficient of this frame: 0 <= a + q + r < Ns for some r, and tap mu + q D < L
				const long long qlo = a + (R - 1) < 0 ? -(a + (R - 1)) : 0;
				const long long qhi_c = (long long)Ns - 1 - a;   // last step with a + q <= Ns - 1
				const long long qhi_t = mu < d.L ? (long long)((d.L - 1 - mu) / D) : -1;
				const bool any = live && qlo <= qhi_c && qlo <= qhi_t;
				if (!__any(any)) continue;
				long long idx = a;                               // coefficient index of window slot 0
				double2 yw[NREC][R];
#pragma unroll
				for (int j = 0; j < R - 1; j++) {
					const bool in = idx >= 0 && idx < (long long)Ns;
#pragma unroll
					for (int c = 0; c < NREC; c++) yw[c][j] = in ? ys[(size_t)c * ncoef + (size_t)(in ? idx : 0)] : make_double2(0.0, 0.0);
					idx++;
				}
				unsigned l = mu;
				for (unsigned q = 0; q < d.Q + 1; q += R) {
#pragma unroll
					for (int u = 0; u < R; u++) {
						const bool in = idx >= 0 && idx < (long long)Ns;
#pragma unroll
						for (int c = 0; c < NREC; c++)
							yw[c][(u + R - 1) % R] = in ? ys[(size_t)c * ncoef + (size_t)(in ? idx : 0)] : make_double2(0.0, 0.0);
						idx++;
						double2 tp = ws[l < d.L ? l : d.L - 1];
						if (l < d.L) {
							tp.x *= gD; tp.y *= gD;
#pragma unroll
							for (int c = 0; c < NREC; c++)
#pragma unroll
								for (int r = 0; r < R; r++)
									acc[c][r] = fma(tp.x, yw[c][(u + r) % R].x, fma(tp.y, yw[c][(u + r) % R].y, acc[c][r]));
						}
						l += D;
					}
				}
			}
		}
	} else if (INV_UNIFORM && o.DL == 64) {
	// D >= 64: the 64 lanes of a wave are consecutive output phases of ONE group block, so their coefficient windows are the
	// same rows -- up to a shift by one row for the lanes behind the point where n - cd crosses a multiple of D.  The rows are
	// fetched ONCE per wave through the scalar unit (wave-uniform window in SGPRs) and the shifted lanes run one tap step late
	// (their tap index starts at mu - D; taps outside [0, L) do nothing): per step one vector load (the lane's tap) instead of
	// three.  Every lane executes the same additions in the same order as in the per-lane form below.
	const unsigned gu = gb; // 64 / DL = 1 group per wave
	for (unsigned v = 0; v < o.nv; v++) {
		const ScaleDesc d = sc[o.s0 + v];
		const double2 *ws = wd + d.tap_off;
		const double2 *ys = Y + d.coef_off;
		const long long t = (long long)n0c - d.cd;
		const long long a = t >= 0 ? t / D : -((-t + D - 1) / D);
		const unsigned rho = (unsigned)(t - a * (long long)D);
		const unsigned mu = rho ? D - rho : 0;
		const int idx = (int)a + (rho ? 1 : 0);                    // lane 0 has the smallest n0, hence the smallest index
		const int A = __builtin_amdgcn_readfirstlane(idx);
		const unsigned shift = n0 < D ? (unsigned)(idx - A) : 0u;  // 0 or 1
		long long am = ((long long)A + (long long)gu * R) % (long long)Ns;
		if (am < 0) am += Ns;
		unsigned row = (unsigned)am;                               // uniform: coefficient index of window slot 0
		const double gD = d.gain * (double)D;
		double2 yw[NREC][R];
#pragma unroll
		for (int j = 0; j < R - 1; j++) {
#pragma unroll
			for (int c = 0; c < NREC; c++) yw[c][j] = ys[(size_t)c * ncoef + row];
			if (++row == Ns) row = 0;
		}
		unsigned l = mu - shift * D;                               // (wraps below zero for the first step of a shifted lane)
		for (unsigned q = 0; q < d.Q + 1; q += R) {
#pragma unroll
			for (int u = 0; u < R; u++) {
#pragma unroll
				for (int c = 0; c < NREC; c++) yw[c][(u + R - 1) % R] = ys[(size_t)c * ncoef + row];
				if (++row == Ns) row = 0;
				double2 tp = ws[l < d.L ? l : d.L - 1];
				if (l < d.L) {
					tp.x *= gD; tp.y *= gD;
#pragma unroll
					for (int c = 0; c < NREC; c++)
#pragma unroll
						for (int r = 0; r < R; r++)
							acc[c][r] = fma(tp.x, yw[c][(u + r) % R].x, fma(tp.y, yw[c][(u + r) % R].y, acc[c][r]));
				}
				l += D;
			}
		}
	}
	} else {
	for (unsigned v = 0; v < o.nv; v++) {
		const ScaleDesc d = sc[o.s0 + v];
		const double2 *ws = wd + d.tap_off;
		const double2 *ys = Y + d.coef_off;
		// n - cd = a D + rho for n = n0 + (gR + r) D
		const long long t = (long long)n0c - d.cd;
		long long a = t >= 0 ? t / D : -((-t + D - 1) / D);
		const unsigned rho = (unsigned)(t - a * (long long)D);
		const unsigned mu = rho ? D - rho : 0;
		a += (long long)g * R + (rho ? 1 : 0);
		long long am = a % (long long)Ns;
		if (am < 0) am += Ns;
		unsigned row = (unsigned)am;               // coefficient index of window slot 0
		const double gD = d.gain * (double)D;
		double2 yw[NREC][R];
#pragma unroll
		for (int j = 0; j < R - 1; j++) {
#pragma unroll
			for (int c = 0; c < NREC; c++) yw[c][j] = ys[(size_t)c * ncoef + row];
			if (++row == Ns) row = 0;
		}
		unsigned l = mu, q = 0;
		for (; q + R <= d.Q; q += R) {
#pragma unroll
			for (int u = 0; u < R; u++) {
#pragma unroll
				for (int c = 0; c < NREC; c++) yw[c][(u + R - 1) % R] = ys[(size_t)c * ncoef + row];
				if (++row == Ns) row = 0;
				double2 tp = ws[l < d.L ? l : d.L - 1];
				if (l < d.L) {
					tp.x *= gD; tp.y *= gD;
#pragma unroll
					for (int c = 0; c < NREC; c++)
#pragma unroll
						for (int r = 0; r < R; r++)
							acc[c][r] = fma(tp.x, yw[c][(u + r) % R].x, fma(tp.y, yw[c][(u + r) % R].y, acc[c][r]));
				}
				l += D;
			}
		}
		for (; q < d.Q; q++) {
#pragma unroll
			for (int c = 0; c < NREC; c++) yw[c][R - 1] = ys[(size_t)c * ncoef + row];
			if (++row == Ns) row = 0;
			double2 tp = ws[l < d.L ? l : d.L - 1];
			if (l < d.L) {
				tp.x *= gD; tp.y *= gD;
#pragma unroll
				for (int c = 0; c < NREC; c++)
#pragma unroll
					for (int r = 0; r < R; r++) acc[c][r] = fma(tp.x, yw[c][r].x, fma(tp.y, yw[c][r].y, acc[c][r]));
			}
#pragma unroll
			for (int c = 0; c < NREC; c++)
#pragma unroll
				for (int r = 0; r < R - 1; r++) yw[c][r] = yw[c][r + 1];
			l += D;
		}
	}
	}
	if (!live) return;
	double *dst = obuf + (size_t)o.slot * slot_stride;
#pragma unroll
	for (int r = 0; r < R; r++) {
		const unsigned k = g * R + r;
		const size_t n = (size_t)n0 + (size_t)k * D;
		if (k < Ns && n < N) { // (n < N matters only when D does not divide N: the last row is partial)
#pragma unroll
			for (int c = 0; c < NREC; c++) dst[(size_t)c * N + n] = acc[c][r];
		}
	}
}

// x^[rec][n] = sum of the obuf rows (fixed order)
__global__ void __launch_bounds__(256) k_inv_combine(const double *__restrict__ obuf, size_t slot_stride, unsigned nslots, size_t total,
                                                     double *__restrict__ xout, size_t y_obuf, size_t y_out)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	obuf += (size_t)blockIdx.y * y_obuf; xout += (size_t)blockIdx.y * y_out;
	double a = obuf[i];
	for (unsigned s = 1; s < nslots; s++) a += obuf[(size_t)s * slot_stride + i];
	xout[i] = a;
}

// The stack's two reconstructions (set 0 = ICWT(OUT), set 1 = ICWT(ST)) summed over the obuf rows in the same fixed order as
// k_inv_combine and cast straight to the float outputs (epilogue, ts_pws1f_lib.c:233-241: ls is a FLOAT division).
__global__ void __launch_bounds__(256) k_inv_combine_out(const double *__restrict__ obuf, size_t slot_stride, unsigned nslots, size_t N,
                                                         float *__restrict__ ts, float *__restrict__ ls, float mtr)
{
	// blockIdx.y = the reconstruction (0: ts-PWS, 1: linear stack); four rows in flight, added in row order (short frames have one row
	// per SCALE: 44-56 rows of a few thousand samples -- a chain of dependent loads per thread otherwise)
	const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (n >= N) return;
	const double *src = obuf + (size_t)blockIdx.y * N + n;
	double a = src[0];
	unsigned s = 1;
	for (; s + 4 <= nslots; s += 4) {
		double v[4];
#pragma unroll
		for (int j = 0; j < 4; j++) v[j] = src[(size_t)(s + (unsigned)j) * slot_stride];
#pragma unroll
		for (int j = 0; j < 4; j++) a += v[j];
	}
	for (; s < nslots; s++) a += src[(size_t)s * slot_stride];
	if (blockIdx.y == 0) { if (ts) ts[n] = (float)a; }
	else if (ls) ls[n] = (float)a / mtr;
}

// xout[i] = sum of the obuf rows [a0, a0 + na) and [b0, b0 + nb), in that order (a scale sub-range: its octave items form
// one run per class); no rows at all give zeros.
__global__ void __launch_bounds__(256) k_inv_combine_ranges(const double *__restrict__ obuf, size_t slot_stride, unsigned a0, unsigned na, unsigned b0,
                                                            unsigned nb, size_t total, double *__restrict__ xout)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	double a = 0;
	for (unsigned s = 0; s < na; s++) a += obuf[(size_t)(a0 + s) * slot_stride + i];
	for (unsigned s = 0; s < nb; s++) a += obuf[(size_t)(b0 + s) * slot_stride + i];
	xout[i] = a;
}
