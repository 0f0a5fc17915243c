// fwd_tl.h -- "taps in LDS" polyphase forward frame CWT for gfx950 (included by tspws_hip.hip).
//
// Third arrangement of the same decomposition (fwd_poly.h: thread = 8 consecutive outputs of one phase x 2
// traces, sliding register window over the x rows, phase lanes combined at the end):
//   * the scale's taps (one 64-phase chunk of them when D >= 64) are staged ONCE per workgroup into LDS and
//     stay resident; afterwards the waves of the workgroup never synchronise again,
//   * every wave walks its own list of work items (output-group block x trace pair), reading the x rows it
//     needs straight from L2/L1 (coalesced 512-byte rows when D >= 64) four tap steps ahead of their use,
//   * registers are capped at 128 (launch bound) so that four waves share a SIMD: tools/fma64_peak.hip shows
//     the FP64 pipe of this chip needs >= 4 FMA-issuing waves per SIMD (23 / 42 / 49 TFLOP/s at 1 / 2 / 4),
//   * the 64-lane phase reduction goes through a wave-private LDS transpose (8 values per round).
// No barrier in the main loop, no x image in LDS: the waves only share read-only taps.
#pragma once

#define TL_R 8
#define TL_SCR 528 /* doubles of reduction scratch per wave: 8 rows x 65 (+8) */

template <typename TIn>
__global__ void __launch_bounds__(256, 4) k_fwd_tl(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N,
                                                   const ScaleDesc *__restrict__ sc, unsigned S, const double2 *__restrict__ w,
                                                   double2 *__restrict__ part, size_t npart, unsigned tap_rows)
{
	constexpr int R = TL_R;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	double2 *tL = (double2 *)smem;                                  // [tap_rows][64] (D >= 64) or [Q*D] (D < 64)
	double *scr_all = (double *)(smem + (size_t)tap_rows * 64 * 16);
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	unsigned lo = 0, hi = S;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (sc[mid].lds_off <= blockIdx.x) lo = mid; else hi = mid;
	}
	const ScaleDesc d = sc[lo];
	const unsigned wl = blockIdx.x - d.lds_off;
	const unsigned chunk = wl / d.lds_bps, blk = wl - chunk * d.lds_bps;   // lds_bps = workgroups per (scale, chunk)
	const unsigned D = d.D, DL = d.DL, logDL = d.logDL;
	const bool small = D < 64;
	const unsigned GW = 64u >> logDL;
	const unsigned lane_m = lane & (DL - 1), lane_g = lane >> logDL;
	const unsigned m0 = small ? 0u : chunk * 64u;
	const unsigned m = m0 + lane_m;
	const bool mvalid = m < D;
	const double2 *ws = w + d.tap_off;
	const unsigned XS = small ? D : 64u;                               // tap row stride in LDS

	// ---- stage the taps once ----
	if (small) {
		const unsigned ntap = d.Q * D; // <= tap_rows * 64
		for (unsigned e = tid; e < ntap; e += 256) tL[e] = e < d.L ? ws[e] : make_double2(0.0, 0.0);
	} else {
		double2 tv[6];
#pragma unroll
		for (int i = 0; i < 6; i++) {
			const unsigned q = wv + 4u * (unsigned)i, l = q * D + m;
			tv[i] = (q < d.Q && mvalid && l < d.L) ? ws[l] : make_double2(0.0, 0.0);
		}
#pragma unroll
		for (int i = 0; i < 6; i++) {
			const unsigned q = wv + 4u * (unsigned)i;
			if (q < d.Q) tL[q * 64 + lane] = tv[i];
		}
	}
	__syncthreads();

	const unsigned npairs = (ntr + 1) / 2;
	const unsigned nitems = d.ngw * npairs;
	const unsigned Dw = D % N;
	double *scr = scr_all + wv * TL_SCR;
	const double2 *tbase = tL + lane_m;

	for (unsigned it = blk * 4 + wv; it < nitems; it += d.lds_bps * 4) {
		const unsigned gb = it / npairs, pair = it - gb * npairs;
		const unsigned g = gb * GW + lane_g, k0 = g * R;
		const unsigned ta = 2 * pair, tb = (ta + 1 < ntr) ? ta + 1 : ta;
		const TIn *xa = x + (size_t)ta * ld, *xb = x + (size_t)tb * ld;
		unsigned row = wrap_index((long long)k0 * D + (mvalid ? m : 0) - d.c, N);
		double ar[2][R], ai[2][R];
#pragma unroll
		for (int b = 0; b < 2; b++)
#pragma unroll
			for (int r = 0; r < R; r++) { ar[b][r] = 0; ai[b][r] = 0; }
		double xw[2][R];
#pragma unroll
		for (int j = 0; j < R - 1; j++) {
			xw[0][j] = (double)xa[row]; xw[1][j] = (double)xb[row];
			row += Dw; if (row >= N) row -= N;
		}
		const double2 *tp = tbase;
		for (unsigned q = 0; q < d.Q; q += R) {
#pragma unroll
			for (int h = 0; h < 2; h++) { // two bursts of four steps: rows fetched four steps ahead of their FMAs
				if (q + (unsigned)h * 4u < d.Q) {
					double xn[2][4];
#pragma unroll
					for (int u = 0; u < 4; u++) {
						xn[0][u] = (double)xa[row]; xn[1][u] = (double)xb[row];
						row += Dw; if (row >= N) row -= N;
					}
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int sidx = h * 4 + u;
						if (q + (unsigned)sidx < d.Q) {
							const double2 t = tp[(unsigned)sidx * XS];
							xw[0][(sidx + R - 1) % R] = xn[0][u];
							xw[1][(sidx + R - 1) % R] = xn[1][u];
							const double tx = t.x, ty = t.y; // rows of idle phase lanes / past the filter end were staged as zeros
#pragma unroll
							for (int b = 0; b < 2; b++)
#pragma unroll
								for (int r = 0; r < R; r++) {
									ar[b][r] = fma(xw[b][(sidx + r) % R], tx, ar[b][r]);
									ai[b][r] = fma(xw[b][(sidx + r) % R], ty, ai[b][r]);
								}
						}
					}
				}
			}
			tp += (size_t)R * XS;
		}

		// ---- combine the DL phase lanes of every group, store the split partial (split == chunk) ----
		double *pa = (double *)(part + (size_t)ta * npart + d.part_off + (size_t)chunk * d.Ns);
		double *pb = (ta + 1 < ntr) ? (double *)(part + (size_t)(ta + 1) * npart + d.part_off + (size_t)chunk * d.Ns) : nullptr;
		if (DL == 64) {
			// wave-private LDS transpose, 8 values per round: row i (stride 65: conflict-free both ways) holds value i
			// of every lane; lane (o = lane&7, eighth = lane>>3) sums 8 entries of row o, 3 shuffle-adds finish.
#pragma unroll
			for (int b = 0; b < 2; b++)
#pragma unroll
				for (int hh = 0; hh < 2; hh++) {
#pragma unroll
					for (int r = 0; r < 4; r++) {
						scr[(2 * r) * 65 + lane] = ar[b][hh * 4 + r];
						scr[(2 * r + 1) * 65 + lane] = ai[b][hh * 4 + r];
					}
					const unsigned o = lane & 7, e8 = lane >> 3;
					const double *src = scr + o * 65 + e8 * 8;
					double sum = src[0];
#pragma unroll
					for (int tt = 1; tt < 8; tt++) sum += src[tt];
					sum += __shfl_xor(sum, 8, 64);
					sum += __shfl_xor(sum, 16, 64);
					sum += __shfl_xor(sum, 32, 64);
					const unsigned k = k0 + (unsigned)hh * 4u + (o >> 1);
					double *dst = b ? pb : pa;
					if (e8 == 0 && k < d.Ns && dst) dst[(size_t)k * 2 + (o & 1)] = (o & 1) ? -sum : sum; // conj
				}
		} else {
			constexpr int NV = 4 * R;
			double v[NV];
#pragma unroll
			for (int b = 0; b < 2; b++)
#pragma unroll
				for (int r = 0; r < R; r++) { v[(b * R + r) * 2] = ar[b][r]; v[(b * R + r) * 2 + 1] = ai[b][r]; }
			int n = NV;
			unsigned first = 0;
			ReduceScatter<NV, 0>::run(v, logDL, lane, n, first);
#pragma unroll
			for (int i = 0; i < NV; i++) {
				if (i < n) {
					const unsigned id = first + i, ri = id & 1, r = (id >> 1) % R, b = (id >> 1) / R;
					const unsigned k = k0 + r;
					double *dst = b ? pb : pa;
					if (k < d.Ns && dst) dst[(size_t)k * 2 + ri] = ri ? -v[i] : v[i]; // conj
				}
			}
		}
	}
}
