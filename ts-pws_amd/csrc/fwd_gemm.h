// fwd_gemm.h -- the coarsest scales of a many-trace batch as ONE dense contraction on the FP64 matrix pipe.
//
// Forward frame CWT (cdotx.c:44-70): Y_s[k] = conj( sum_l x[(k D_s - c_s + l) mod N] w_s[l] ).  For the scales at the coarse end of a frame
// whose N is not a power of two the filters are as long as the trace (L_s = N once 2 ceil(5 scale) + 1 > N, wavelet_def_v7.c:312-322) or too
// long for the spectral engine's transform window (spectral.hip), and they have few outputs (N_s = ceil(N / D_s) = 9 .. 33 at the shipped
// example's 16501 samples).  Written over the SAMPLES instead of the taps,
//     r_s[k D_s] = sum_{n < N} x[n] w_s[(n - k D_s + c_s) mod N]        (taps past L_s count as zero)
// every coefficient is a length-N dot product of the trace with a rotated copy of the filter: for a batch of traces that is the matrix product
//     C[trace][column] = X[trace][n] . B[n][column],   column = (scale, k),   B[n][(s, k)] = w_s[(n + o_{s,k}) mod N],  o = (c_s - k D_s) mod N,
// 4 N N_s flop per scale and trace -- the FIR count when L_s = N -- and DENSE: v_mfma_f64_16x16x4_f64 takes a 16 x 4 tile of X (lane = (trace,
// sample)) and a 4 x 16 tile of B (lane = (sample, column)), every lane loads one value of each, nothing is broadcast or reduced across lanes.
// The direct kernel (fwd_poly.h) spent 0.36-0.42 ms on the five clipped scales of the 499 x 16501 example (9 % of the frame's FIR work) as
// chains of dependent L2 loads; the same sums take ~30 us of matrix-pipe time.  north_star's "MFMA left unused" is about the skinny per-scale FIR
// convolutions; this one is a real contraction (K = N samples, 64 traces x 16 columns per wave).
//
//   wave      = (64-trace block, tile of 16 columns, run of KC samples): 4 trace tiles x (re, im) = 8 MFMAs per step of 4 samples
//   workgroup = the (up to) four column tiles of one (block, run): they read the same rows of the transposed batch
//   A operand = xT[n][trace] (the batch transposed once, fwd_tl.h): lane (i, k) loads the four traces 4 i .. 4 i + 3 of row n0 + k as ONE
//               16-byte (float) / 32-byte (double) load -- trace tile t holds the traces 4 i + t
//   B operand = the scale's own tap array, gathered: lane (k, j) keeps the running index l = (n + o_j) mod N of its column
//   output    = the runs' partial sums g[run][trace][column] (256-byte pieces per store); k_gemm_reduce adds the runs in run order and writes the
//               per-trace coefficient, conjugated like the direct kernel's, into the many-trace partial layout (part[trace][part_off_s + k], one
//               "split"): k_accumulate_parts phase-normalises per trace and adds the traces (ts_pws1f_lib.c:486-494) -- unchanged.  (The runs as
//               64 splits of that layout cost k_accumulate_parts 0.25 ms: a wave per coefficient walked 499 x 64 partials 1 KB apart.)
#pragma once

struct GemmCol {
	unsigned L;                  // taps of the column's scale (0: idle pad column)
	unsigned o;                  // tap index that multiplies sample 0: (c - k D) mod N
	unsigned Ns, pad;
	unsigned long long tap_off;  // first tap of the scale
	unsigned long long dst;      // part_off of the scale + k
};

typedef double gemm_v4d __attribute__((ext_vector_type(4)));
#define GEMM_PF 8 /* steps (of 4 samples) in flight per wave: 8 x 512 cycles of MFMAs cover a round trip to HBM */

template <typename TIn> struct GemmRow;
template <> struct GemmRow<float> { typedef float4 type; };
template <> struct GemmRow<double> { typedef double4 type; };

template <typename TIn>
__global__ void __launch_bounds__(256) k_fwd_gemm(const TIn *__restrict__ xT, unsigned TP, unsigned ntr, unsigned N, const GemmCol *__restrict__ cols,
                                                  unsigned ncoltiles, unsigned KC, const double2 *__restrict__ w, double2 *__restrict__ g)
{
	typedef typename GemmRow<TIn>::type Row;
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned ct = blockIdx.x * 4 + wv;
	if (ct >= ncoltiles) return;
	const unsigned ks = blockIdx.y, tb = blockIdx.z;
	const unsigned li = lane & 15, lk = lane >> 4;
	const GemmCol col = cols[(size_t)ct * 16 + li];
	const unsigned n_beg = ks * KC, n_end = min(N, n_beg + KC);
	if (n_beg >= n_end) return;
	const unsigned nsteps = (n_end - n_beg + 3) / 4;
	const TIn *__restrict__ ap = xT + (size_t)tb * 64 + 4 * li;
	const double2 *__restrict__ wp = w + col.tap_off;
	unsigned l = (unsigned)(((unsigned long long)n_beg + lk + col.o) % N); // tap index of this lane's sample of step 0
	gemm_v4d cre[4], cim[4];
#pragma unroll
	for (int t = 0; t < 4; t++) { cre[t] = (gemm_v4d){0, 0, 0, 0}; cim[t] = (gemm_v4d){0, 0, 0, 0}; }
	Row a[GEMM_PF];
	double2 b[GEMM_PF];
	unsigned ok[GEMM_PF]; // bit 0: the sample is inside the run, bit 1: the tap exists
	auto request = [&](const int u, const unsigned step) { // (requests are issued in step order: l advances by 4 samples each time)
		const unsigned n = n_beg + 4 * step + lk;
		const bool in = n < n_end, tap = l < col.L;
		a[u] = *(const Row *)(ap + (size_t)(in ? n : n_end - 1) * TP);
		b[u] = wp[tap ? l : 0u];
		ok[u] = (in ? 1u : 0u) | (tap ? 2u : 0u);
		l += 4; if (l >= N) l -= N;
	};
#pragma unroll
	for (int u = 0; u < GEMM_PF; u++) request(u, (unsigned)u);
	for (unsigned s0 = 0; s0 < nsteps; s0 += GEMM_PF) {
#pragma unroll
		for (int u = 0; u < GEMM_PF; u++) {
			const Row av = a[u];
			const double2 bv = b[u];
			const unsigned o = ok[u];
			request(u, s0 + GEMM_PF + (unsigned)u); // (past the run: clamped rows, counted as zeros)
			const bool in = o & 1u, tap = o & 2u;
			const double x0 = in ? (double)av.x : 0.0, x1 = in ? (double)av.y : 0.0, x2 = in ? (double)av.z : 0.0, x3 = in ? (double)av.w : 0.0;
			const double br = tap ? bv.x : 0.0, bi = tap ? bv.y : 0.0;
			cre[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, br, cre[0], 0, 0, 0);
			cim[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, bi, cim[0], 0, 0, 0);
			cre[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, br, cre[1], 0, 0, 0);
			cim[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, bi, cim[1], 0, 0, 0);
			cre[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, br, cre[2], 0, 0, 0);
			cim[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, bi, cim[2], 0, 0, 0);
			cre[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, br, cre[3], 0, 0, 0);
			cim[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, bi, cim[3], 0, 0, 0);
		}
	}
	// result register q of tile t: row 4 q + lk of the tile (tools/probes/mfma_f64_layout.hip) = trace 4 (4 q + lk) + t of the block, column li
	const size_t ncol = (size_t)ncoltiles * 16;
	double2 *__restrict__ dst = g + ((size_t)ks * TP + (size_t)tb * 64) * ncol + (size_t)ct * 16 + li;
#pragma unroll
	for (int t = 0; t < 4; t++)
#pragma unroll
		for (int q = 0; q < 4; q++) dst[(size_t)(4 * (4 * q + (int)lk) + t) * ncol] = make_double2(cre[t][q], cim[t][q]);
}

// part[trace][dst_column] = conj( sum over the runs g[run][trace][column] ), runs added in order; thread = (trace, column)
__global__ void __launch_bounds__(256) k_gemm_reduce(const double2 *__restrict__ g, unsigned TP, unsigned ntr, unsigned ncol, unsigned KS, const GemmCol *__restrict__ cols,
                                                     double2 *__restrict__ part, size_t npart)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const unsigned c = (unsigned)(i % ncol), tr = (unsigned)(i / ncol);
	if (tr >= ntr) return;
	const GemmCol col = cols[c];
	if (!col.L) return;
	const size_t stride = (size_t)TP * ncol;
	const double2 *__restrict__ p = g + i;
	double2 a = make_double2(0.0, 0.0);
	unsigned ks = 0;
	for (; ks + 8 <= KS; ks += 8) { // eight runs in flight
		double2 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = p[(size_t)(ks + (unsigned)j) * stride];
#pragma unroll
		for (int j = 0; j < 8; j++) { a.x += v[j].x; a.y += v[j].y; }
	}
	for (; ks < KS; ks++) { const double2 v = p[(size_t)ks * stride]; a.x += v.x; a.y += v.y; }
	part[(size_t)tr * npart + col.dst] = make_double2(a.x, -a.y);
}
