// spectral.hip -- the flop-reducing forward engine for MANY traces: the far-decimated octaves of the frame through the trace's
// spectrum instead of per-scale FIR sums.
//
// The reference's decimating FIR (cdotx.c:35-72, driver wavelet_v7.c:43-64) is a CIRCULAR correlation sampled every D_s-th lag:
//     Y_s[k] = conj( r_s[k D_s] ),   r_s[m] = sum_l x[(m - c_s + l) mod N] w_s[l].
// With X = DFT_N(x) and H_s[f] = (1/N) sum_l w_s[l] e^{+2 pi i f (l - c_s) / N}:   r_s[m] = sum_f X[f] H_s[f] e^{+2 pi i f m / N}, and for
// D_s | N (N_s = N / D_s) the samples m = k D_s only see the spectrum folded by N_s:
//     r_s[k D_s] = sum_{q < N_s} G_s[q] e^{+2 pi i q k / N_s},     G_s[q] = sum_{j < D_s} X[q + j N_s] H_s[q + j N_s].
// Exact (no band limit is assumed: the tables are the spectra of the clipped taps themselves), 4 FMAs per (frequency, scale) instead
// of 2 L_s / D_s per coefficient: ~8 N flop per scale against 40 .. 76 N for the FIR, plus the transforms.
//
// MI355X mapping: everything here runs with lane = TRACE (like fwd_tl.h): the 64 lanes of a wave are 64 traces of a block, so
//   * every operand that does not depend on the trace -- twiddles, the tables H_s -- is wave-uniform and comes through the SCALAR unit
//     (s_load into SGPRs; the FMAs take it as their SGPR source), and there is no cross-lane traffic before the very end,
//   * every load / store is one coalesced 1-KB row (a row = one frequency / sample of the block's 64 traces),
//   * transforms are Stockham radix-2..32 passes in registers (k_spec_*: one butterfly per wave-item, out of place through HBM / MALL),
//     the trace transform packs the real samples in pairs (N/2-point complex transform + split, half spectrum X[0 .. N/2]),
//   * the multiply-and-fold (k_spec_fold) walks a residue class f = r (mod R) in bit-reversed order so that ONE live accumulator per
//     scale suffices: a scale with D_s = 2^d completes a folded bin every 2^d steps; classes coarser than a scale's N_s leave partial
//     sums that the first inverse pass adds,
//   * the last inverse pass conjugates, phase-normalises (ts_pws1f_lib.c:489-492) and adds the 64 traces of the block on the VALU
//     (lane_reduce.h) straight into the block's ST / PS planes -- per-trace coefficients of these scales never exist in memory.
// The fine octaves (large N_s: per-coefficient FIR work is smallest there, per-trace spectra largest) stay on k_fwd_tl.
// Reference citations are relative to /root/reference/src.
#include "tspws_internal.h"
#include "lane_reduce.h"
#include "spectral.h"


// ------------------------------------------------------------------------------------------
// compile-time twiddles of the register transforms: cos / sin (2 pi k / 64)
// ------------------------------------------------------------------------------------------
namespace {
constexpr double quarter64(int k)
{ // cos(2 pi k / 64), 0 <= k <= 16
	return k == 0 ? 1.0 : k == 1 ? 0.9951847266721969 : k == 2 ? 0.9807852804032304 : k == 3 ? 0.9569403357322088 : k == 4 ? 0.9238795325112867
	     : k == 5 ? 0.881921264348355 : k == 6 ? 0.8314696123025452 : k == 7 ? 0.773010453362737 : k == 8 ? 0.7071067811865476
	     : k == 9 ? 0.6343932841636455 : k == 10 ? 0.5555702330196023 : k == 11 ? 0.4713967368259978 : k == 12 ? 0.38268343236508984
	     : k == 13 ? 0.29028467725446233 : k == 14 ? 0.19509032201612833 : k == 15 ? 0.09801714032956077 : 0.0;
}
constexpr double cos64(int k)
{
	k = ((k % 64) + 64) % 64;
	return k <= 16 ? quarter64(k) : k <= 32 ? -quarter64(32 - k) : k <= 48 ? -quarter64(k - 32) : quarter64(64 - k);
}
constexpr double sin64(int k) { return cos64(k - 16); }

__device__ __forceinline__ double2 cadd(const double2 a, const double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(const double2 a, const double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul(const double2 a, const double2 b) { return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x)); }

// R-point DFT in registers (decimation in time, all indices static): natural order in, natural order out.
// INV: kernel e^{+2 pi i nk/R}, else e^{-2 pi i nk/R}; unnormalised.
template <int R, bool INV, int K> struct SpecBfly {
	static __device__ __forceinline__ void run(double2 (&v)[R], const double2 (&e)[R / 2], const double2 (&o)[R / 2])
	{
		constexpr double cs = cos64(K * (64 / R)), sn = INV ? sin64(K * (64 / R)) : -sin64(K * (64 / R));
		double2 t;
		if constexpr (K == 0) t = o[0];
		else if constexpr (4 * K == R) t = INV ? make_double2(-o[K].y, o[K].x) : make_double2(o[K].y, -o[K].x);
		else t = make_double2(fma(o[K].x, cs, -o[K].y * sn), fma(o[K].x, sn, o[K].y * cs));
		v[K] = cadd(e[K], t);
		v[K + R / 2] = csub(e[K], t);
		if constexpr (K + 1 < R / 2) SpecBfly<R, INV, K + 1>::run(v, e, o);
	}
};
template <int R, bool INV> struct SpecDFT {
	static __device__ __forceinline__ void run(double2 (&v)[R])
	{
		constexpr int H = R / 2;
		double2 e[H], o[H];
#pragma unroll
		for (int i = 0; i < H; i++) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
		SpecDFT<H, INV>::run(e);
		SpecDFT<H, INV>::run(o);
		SpecBfly<R, INV, 0>::run(v, e, o);
	}
};
template <bool INV> struct SpecDFT<1, INV> { static __device__ __forceinline__ void run(double2 (&)[1]) {} };

__device__ __forceinline__ const SpecSeg *spec_find_seg(const SpecSeg *__restrict__ segs, unsigned nseg, unsigned item)
{
	unsigned lo = 0, hi = nseg;
	while (hi - lo > 1) {
		const unsigned mid = (lo + hi) >> 1;
		if (segs[mid].item0 <= item) lo = mid; else hi = mid;
	}
	return segs + lo;
}

// twiddle e^{-+ 2 pi i idx / N} from the table of the forward kernel (tw[i] = e^{-2 pi i i / N}); idx is wave-uniform
template <bool INV>
__device__ __forceinline__ double2 spec_tw(const double2 *__restrict__ tw, unsigned idx)
{
	const double2 t = tw[idx];
	return INV ? make_double2(t.x, -t.y) : t;
}
} // namespace

// ------------------------------------------------------------------------------------------
// plan-time kernels: twiddle table, tap norms, the tables H_s
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_spec_twiddle(double2 *__restrict__ tw, unsigned N)
{
	const unsigned i = blockIdx.x * 256 + threadIdx.x;
	if (i >= N) return;
	double s, c;
	sincospi(-2.0 * (double)i / (double)N, &s, &c);
	tw[i] = make_double2(c, s);
}

// ||w_s||_2 of every scale: one workgroup per scale
__global__ void __launch_bounds__(256) k_spec_wnorm(const ScaleDesc *__restrict__ sc, const double2 *__restrict__ w, double *__restrict__ out)
{
	__shared__ double red[4];
	const ScaleDesc d = sc[blockIdx.x];
	double a = 0;
	for (unsigned l = threadIdx.x; l < d.L; l += 256) { const double2 t = w[d.tap_off + l]; a = fma(t.x, t.x, fma(t.y, t.y, a)); }
	a = wave_sum(a);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = sqrt(red[0] + red[1] + red[2] + red[3]);
}

// h[blk][n][lane]: the taps of slot (blk, lane) placed circularly, tap l at n = (l - c) mod N (the buffer was zeroed)
__global__ void __launch_bounds__(256) k_spec_place(const ScaleDesc *__restrict__ sc, const double2 *__restrict__ w, const unsigned *__restrict__ slot_scale,
                                                    unsigned N, double2 *__restrict__ h)
{
	const unsigned slot = blockIdx.y, s = slot_scale[slot];
	if (s == ~0u) return;
	const ScaleDesc d = sc[s];
	const unsigned l = blockIdx.x * 256 + threadIdx.x;
	if (l >= d.L) return;
	long long n = (long long)l - d.c;
	n %= (long long)N; if (n < 0) n += N;
	h[((size_t)(slot >> 6) * N + (size_t)n) * 64 + (slot & 63)] = w[d.tap_off + l];
}

// Htab[((g R + r) nsteps + c) NS + s] = Hfull[slot = g NS + s][f = r + R bitrev(c)] / N
__global__ void __launch_bounds__(256) k_spec_permute(const double2 *__restrict__ hf, unsigned N, unsigned R, unsigned logsteps, unsigned NS, unsigned ngroups,
                                                      double2 *__restrict__ tab)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t total = (size_t)ngroups * N * NS;
	if (i >= total) return;
	const unsigned s = (unsigned)(i % NS);
	const size_t rc = i / NS;                 // (g R + r) nsteps + c
	const unsigned nsteps = 1u << logsteps;
	const unsigned c = (unsigned)(rc & (nsteps - 1));
	const size_t gr = rc >> logsteps;
	const unsigned r = (unsigned)(gr % R), g = (unsigned)(gr / R);
	const unsigned ib = logsteps ? (__brev(c) >> (32 - logsteps)) : 0u;
	const unsigned f = r + R * ib;
	const unsigned slot = g * NS + s;
	const double2 v = hf[((size_t)(slot >> 6) * N + f) * 64 + (slot & 63)];
	const double inv = 1.0 / (double)N;
	tab[i] = make_double2(v.x * inv, v.y * inv);
}

// ------------------------------------------------------------------------------------------
// per-call kernels
// ------------------------------------------------------------------------------------------
// ---- first pass of the trace transform: z[m] = y[2m] + i y[2m+1] from the transposed batch, L = 1 (no twiddles) -------------------
// y = the trace itself when its length Nx is the transform length NT (the circular correlation of cdotx.c:44-70 IS the transform's), else its
// PERIODIC EXTENSION over a window of NT >= Nx + L - 1 samples: y[n] = x[n mod Nx] for n < split = NT - cneg, y[n] = x[(n - NT) mod Nx] above
// (the cneg samples in front of the trace, where the filters' left halves reach at the first outputs).  A circular correlation of length Nx
// with L taps is a LINEAR correlation of the periodic signal; over that window the NT-circular one reproduces it at every lag m < Nx.
__device__ __forceinline__ unsigned spec_ext_row(unsigned n, const unsigned Nx, const unsigned split, const unsigned NT)
{
	if (n >= split) return n - NT + Nx; // (cneg <= Nx)
	while (n >= Nx) n -= Nx;             // (wave-uniform; NT < 4 Nx: at most three rounds)
	return n;
}

template <typename TIn, int R>
__device__ __forceinline__ void spec_fwd_first_body(const TIn *__restrict__ col, unsigned TP, double2 *__restrict__ dst, unsigned j, unsigned m, unsigned Nx,
                                                    unsigned split)
{
	double2 v[R];
	const unsigned NT = 2u * m * R;
#pragma unroll
	for (int n = 0; n < R; n++) {
		const unsigned idx = j + (unsigned)n * m;
		v[n] = make_double2((double)col[(size_t)spec_ext_row(2 * idx, Nx, split, NT) * TP], (double)col[(size_t)spec_ext_row(2 * idx + 1, Nx, split, NT) * TP]);
	}
	SpecDFT<R, false>::run(v);
#pragma unroll
	for (int n = 0; n < R; n++) dst[((size_t)j * R + n) * 64] = v[n];
}

// (R64: the 64-point instantiation on its own -- 256 VGPRs, one wave per SIMD; in one kernel with the smaller radices it would set THEIR register count too)
template <typename TIn, bool R64 = false>
__global__ void __launch_bounds__(256) k_spec_fwd_first(const TIn *__restrict__ xT, unsigned TP, double2 *__restrict__ dst, size_t dst_rows, unsigned M, unsigned radix,
                                                        unsigned Nx, unsigned split)
{
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned m = M / radix;
	const unsigned j = blockIdx.x * 4 + wv, tb = blockIdx.y;
	if (j >= m) return;
	const TIn *col = xT + (size_t)tb * 64 + lane;
	double2 *d = dst + (size_t)tb * dst_rows * 64 + lane;
	if constexpr (R64) { spec_fwd_first_body<TIn, 64>(col, TP, d, j, m, Nx, split); return; }
	switch (radix) {
	case 32: spec_fwd_first_body<TIn, 32>(col, TP, d, j, m, Nx, split); break;
	case 16: spec_fwd_first_body<TIn, 16>(col, TP, d, j, m, Nx, split); break;
	default: spec_fwd_first_body<TIn, 8>(col, TP, d, j, m, Nx, split); break;
	}
}

// ---- generic Stockham pass: out[j0 + n L] = DFT_R( tw_n in[j + n len/R] ), k = j mod L, j0 = (j - k) R + k ---------------------------
template <int R, bool INV>
__device__ __forceinline__ void spec_load_tw(double2 (&v)[R], const double2 *__restrict__ src, const SpecSeg *__restrict__ sg, unsigned j, unsigned k,
                                             const double2 *__restrict__ tw)
{
	const unsigned m = sg->len / R, nfold = sg->nfold;
#pragma unroll
	for (int n = 0; n < R; n++) v[n] = src[((size_t)j + (size_t)n * m) * 64];
	for (unsigned p = 1; p < nfold; p++) { // partial folds of a coarse scale (first inverse pass): entry q = sum of the rows q + p len -- R loads in flight per round
		double2 a[R];
#pragma unroll
		for (int n = 0; n < R; n++) a[n] = src[((size_t)j + (size_t)n * m + (size_t)p * sg->len) * 64];
#pragma unroll
		for (int n = 0; n < R; n++) v[n] = cadd(v[n], a[n]);
	}
	if (k) {
#pragma unroll
		for (int n = 1; n < R; n++) v[n] = cmul(v[n], spec_tw<INV>(tw, (unsigned)n * k * sg->tw_mul));
	}
}

template <int R, bool INV>
__device__ __forceinline__ void spec_mid_body(const double2 *__restrict__ src, double2 *__restrict__ dst, const SpecSeg *__restrict__ sg, unsigned j,
                                              const double2 *__restrict__ tw)
{
	const unsigned L = sg->L, k = j & (L - 1);
	double2 v[R];
	spec_load_tw<R, INV>(v, src, sg, j, k, tw);
	SpecDFT<R, INV>::run(v);
	const size_t j0 = (size_t)(j - k) * R + k;
#pragma unroll
	for (int n = 0; n < R; n++) dst[(j0 + (size_t)n * L) * 64] = v[n];
}

template <bool INV, bool R64 = false>
__global__ void __launch_bounds__(256) k_spec_mid(const SpecSeg *__restrict__ segs, unsigned nseg, unsigned nitems, const double2 *__restrict__ src, size_t src_rows,
                                                  double2 *__restrict__ dst, size_t dst_rows, const double2 *__restrict__ tw)
{
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned item = blockIdx.x * 4 + wv, tb = blockIdx.y;
	if (item >= nitems) return;
	const SpecSeg *sg = spec_find_seg(segs, nseg, item);
	const unsigned j = item - sg->item0;
	const double2 *s = src + ((size_t)tb * src_rows + sg->src) * 64 + lane;
	double2 *d = dst + ((size_t)tb * dst_rows + sg->dst) * 64 + lane;
	if constexpr (R64) { spec_mid_body<64, INV>(s, d, sg, j, tw); return; }
	switch (sg->radix) {
	case 32: spec_mid_body<32, INV>(s, d, sg, j, tw); break;
	case 16: spec_mid_body<16, INV>(s, d, sg, j, tw); break;
	case 8: spec_mid_body<8, INV>(s, d, sg, j, tw); break;
	case 4: spec_mid_body<4, INV>(s, d, sg, j, tw); break;
	default: spec_mid_body<2, INV>(s, d, sg, j, tw); break;
	}
}

// ---- last pass of the trace transform: butterflies k and L - k together, split of the packed pairs, half spectrum X[0 .. M] ---------
//   X[f] = 1/2 ( (Z[f] + conj Z[M - f]) - i e^{-2 pi i f / N} (Z[f] - conj Z[M - f]) )
template <int R>
__device__ __forceinline__ void spec_fwd_last_body(const double2 *__restrict__ src, double2 *__restrict__ dst, unsigned M, unsigned k, const double2 *__restrict__ tw)
{
	const unsigned L = M / R, kb = (L - k) & (L - 1);
	double2 za[R], zb[R];
#pragma unroll
	for (int n = 0; n < R; n++) { za[n] = src[((size_t)k + (size_t)n * L) * 64]; zb[n] = src[((size_t)kb + (size_t)n * L) * 64]; }
	if (k) {
#pragma unroll
		for (int n = 1; n < R; n++) { za[n] = cmul(za[n], tw[2u * (unsigned)n * k]); zb[n] = cmul(zb[n], tw[2u * (unsigned)n * kb]); } // e^{-2 pi i n k / M}
	}
	SpecDFT<R, false>::run(za);
	SpecDFT<R, false>::run(zb);
	auto split = [&](const double2 zf, const double2 zm, const unsigned f) { // zm = Z[M - f] (not conjugated yet)
		const double2 e = make_double2(zf.x + zm.x, zf.y - zm.y), o = make_double2(zf.x - zm.x, zf.y + zm.y); // Z[f] +- conj Z[M-f]
		const double2 w = tw[f];                                                                              // e^{-2 pi i f / N}
		// -i w o = (w.y o.x + w.x o.y) - i (w.x o.x - w.y o.y)
		const double2 t = make_double2(fma(w.y, o.x, w.x * o.y), -fma(w.x, o.x, -w.y * o.y));
		dst[(size_t)f * 64] = make_double2(0.5 * (e.x + t.x), 0.5 * (e.y + t.y));
	};
	if (k) {
#pragma unroll
		for (int n = 0; n < R; n++) {
			split(za[n], zb[R - 1 - n], k + (unsigned)n * L);
			split(zb[n], za[R - 1 - n], kb + (unsigned)n * L);
		}
	} else { // f = n L: the partner M - f = (R - n) L is in the same butterfly; f = M closes the half spectrum
#pragma unroll
		for (int n = 0; n < R; n++) split(za[n], za[(R - n) % R], (unsigned)n * L);
		dst[(size_t)M * 64] = make_double2(za[0].x - za[0].y, 0.0);
	}
}

__global__ void __launch_bounds__(256) k_spec_fwd_last(const double2 *__restrict__ src, size_t src_rows, double2 *__restrict__ dst, size_t dst_rows, unsigned M,
                                                       unsigned radix, const double2 *__restrict__ tw)
{
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned L = M / radix;
	const unsigned k = blockIdx.x * 4 + wv, tb = blockIdx.y;
	if (k > L / 2) return;
	const double2 *s = src + (size_t)tb * src_rows * 64 + lane;
	double2 *d = dst + (size_t)tb * dst_rows * 64 + lane;
	switch (radix) {
	case 16: spec_fwd_last_body<16>(s, d, M, k, tw); break;
	case 8: spec_fwd_last_body<8>(s, d, M, k, tw); break;
	case 4: spec_fwd_last_body<4>(s, d, M, k, tw); break;
	default: spec_fwd_last_body<2>(s, d, M, k, tw); break;
	}
}

// ---- multiply-and-fold ---------------------------------------------------------------------------------------------------------------
// workgroup = ONE class r: its WG scale groups x WT sets of NTB trace blocks (a wave = one group, one set).  Steps c = 0 .. nsteps-1 over
// f = r + R bitrev(c).  Slot s of the group (scales in the order of growing D, partial ones last) adds X[f] H_s[f]; it completes bin
// q = r + R bitrev(c >> d_s) whenever (c + 1) % 2^{d_s} == 0.
//
// The table values H_s[f] are the same for every lane.  Three ways to bring them to the FMAs were measured on 1024 x 32768 (cfg2):
//   * scalar loads (SGPR operands): two SGPR sets of 8 values + everything else do not fit 102 SGPRs -- the compiler spills the freshly
//     loaded set through v_writelane / v_readlane (two VALU instructions per FMA pair) and waits for every request on the spot: 380-430 us;
//   * vector loads of one address: the 64 lanes' copies come back over the 64 B/clk L1 return path like any 1-KB load: slower still;
//   * THIS: the class's table stream is staged into LDS by global_load_lds (1 KB per instruction, a ring of SP_RB blocks: requests run
//     thousands of cycles ahead, in order) and read as broadcast ds_read_b128 (4 LDS cycles per wave-instruction; with two trace blocks
//     per wave a value feeds 8 FMAs, so the LDS pipe is at ~50 %); no SGPR pressure, no VALU overhead.
#define SP_RB 4 /* ring blocks of 1 KB per scale group */
template <int NS, int NTB>
__global__ void __launch_bounds__(512) k_spec_fold(const double2 *__restrict__ Xh, size_t xrows, const double2 *__restrict__ tab, const SpecSlot *__restrict__ slots,
                                                   unsigned R, unsigned logsteps, unsigned N, unsigned nblk, unsigned WG, double2 *__restrict__ G, size_t grows, unsigned abl)
{
	static_assert(NS == 8 || NS == 16, "a 1-KB block of the table stream holds a whole number of steps");
	constexpr unsigned SPL = 64 / NS;               // steps per 1-KB block
	extern __shared__ __attribute__((aligned(16))) char smem[]; // [WG][SP_RB][64] double2
	typedef __attribute__((address_space(3))) void lds_void;
	typedef __attribute__((address_space(1))) const void glb_void;
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned r = blockIdx.x, g = wv % WG, ts = wv / WG, WT = (blockDim.x >> 6) / WG;
	const unsigned nsets = (nblk + NTB - 1) / NTB;
	unsigned tbs = blockIdx.y * WT + ts;
	if (tbs >= nsets) tbs = nsets - 1;              // (a spare wave repeats the last set: same values to the same places -- it must keep the barriers)
	const unsigned item = g * R + r;
	const unsigned nsteps = 1u << logsteps, M = N >> 1;
	const double2 *__restrict__ hp = tab + (size_t)item * nsteps * NS; // this group's table stream of the class: step c, slot s at hp[c NS + s]
	const SpecSlot *__restrict__ sl = slots + (size_t)g * NS;
	const double2 *xc[NTB];
	double2 *gc[NTB];
#pragma unroll
	for (int b = 0; b < NTB; b++) {
		const unsigned tb = (NTB * tbs + b < nblk) ? NTB * tbs + b : NTB * tbs; // (block count not a multiple of NTB: the last wave does a block twice)
		xc[b] = Xh + (size_t)tb * xrows * 64 + lane;
		gc[b] = G + (size_t)tb * grows * 64 + lane;
	}
	double2 acc[NTB][NS];
#pragma unroll
	for (int b = 0; b < NTB; b++)
#pragma unroll
		for (int s = 0; s < NS; s++) acc[b][s] = make_double2(0.0, 0.0);
	// the group's slot descriptors in LDS for the whole walk (a scalar load + wait per slot at every completion was a chain of dependent
	// round trips to L2 in the middle of the FMA stream; an LDS read is ~100 cycles)
	SpecSlot *lsl = (SpecSlot *)(smem + (size_t)WG * SP_RB * 1024) + (size_t)wv * NS; // (one copy per wave: no barrier needed before its use)
	if (lane < (unsigned)NS) lsl[lane] = sl[lane];
	const unsigned mask0 = (1u << sl[0].ld) - 1u;   // the finest scale of the group: nothing completes before it does
	auto xrow = [&](const unsigned c, bool &cj) -> size_t {
		const unsigned ib = logsteps ? (__brev(c) >> (32 - logsteps)) : 0u;
		const unsigned f = r + R * ib;
		cj = f > M;
		if (abl & 1u) return (size_t)((c & 7u) * 64); // ABLATION: eight hot rows instead of the class's rows
		return (size_t)(cj ? N - f : f) * 64;
	};
	// table ring of this group: block k (steps k SPL ..) in slot k % SP_RB; the set-0 wave of the group loads it
	char *ringb = smem + (size_t)g * SP_RB * 1024;
	const bool loader = ts == 0;
	const unsigned nblocks = nsteps / SPL;           // (nsteps >= 16 >= SPL)
	auto request = [&](const unsigned k) {           // block k -> its ring slot (one wave-instruction: 64 lanes x 16 B)
		__builtin_amdgcn_global_load_lds((glb_void *)(hp + (size_t)k * 64 + lane), (lds_void *)(ringb + (size_t)(k % SP_RB) * 1024), 16, 0, 0);
	};
	if (loader) {
#pragma unroll
		for (unsigned k = 0; k < SP_RB - 1; k++) if (k < nblocks) request(k);
	}
	constexpr int XU = 2;                            // rows of the spectra in flight: the next XU steps travel while these are computed
	double2 xa[XU][NTB], xb[XU][NTB];
	bool ca[XU], cb[XU];
#pragma unroll
	for (int u = 0; u < XU; u++) {
		const size_t o = xrow((unsigned)u, ca[u]);
#pragma unroll
		for (int b = 0; b < NTB; b++) xa[u][b] = xc[b][o];
	}
	for (unsigned k = 0; k < nblocks; k++) {
		// block k has landed (the loader waits for its own request -- older than every row it has asked for since) and everybody is done
		// with block k - 1, whose slot the next request overwrites
		if (loader && !(abl & 4u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (!(abl & 8u)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		if (loader && k + SP_RB - 1 < nblocks) request(k + SP_RB - 1);
		const double2 *__restrict__ hb = (const double2 *)(ringb + (size_t)(k % SP_RB) * 1024);
		// the block's 64 table values in chunks of 8: chunk i + 1 is requested from LDS BEFORE the FMAs of chunk i (two register sets;
		// the scheduling barriers keep the compiler from sinking the request to its use)
		constexpr int CPS = NS / 8, NCB = 8;          // chunks per step, chunks per block
		double2 hq[2][8];
#pragma unroll
		for (int i = 0; i < 8; i++) hq[0][i] = hb[i]; // same address in every lane: broadcast
#pragma unroll
		for (int ci = 0; ci < NCB; ci++) {
			const int c1 = ci / CPS, ch = ci % CPS, u = c1 % XU;
			const unsigned c = k * SPL + (unsigned)c1;
			if (ch == 0 && u == 0 && c + XU < nsteps) { // the rows of the next XU steps
#pragma unroll
				for (int uu = 0; uu < XU; uu++) {
					const size_t o = xrow(c + XU + (unsigned)uu, cb[uu]);
#pragma unroll
					for (int b = 0; b < NTB; b++) xb[uu][b] = xc[b][o];
				}
			}
			__builtin_amdgcn_sched_barrier(0);
			if (ci + 1 < NCB) {
#pragma unroll
				for (int i = 0; i < 8; i++) hq[(ci + 1) & 1][i] = hb[(ci + 1) * 8 + i];
			}
			__builtin_amdgcn_sched_barrier(0);
			double xr[NTB], xi[NTB];
#pragma unroll
			for (int b = 0; b < NTB; b++) { xr[b] = xa[u][b].x; xi[b] = ca[u] ? -xa[u][b].y : xa[u][b].y; }
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const double2 hv = hq[ci & 1][i];
				const int s = ch * 8 + i;
#pragma unroll
				for (int b = 0; b < NTB; b++) {
					acc[b][s].x = fma(xr[b], hv.x, acc[b][s].x);
					acc[b][s].y = fma(xr[b], hv.y, acc[b][s].y);
				}
#pragma unroll
				for (int b = 0; b < NTB; b++) {
					acc[b][s].x = fma(-xi[b], hv.y, acc[b][s].x);
					acc[b][s].y = fma(xi[b], hv.x, acc[b][s].y);
				}
			}
			if (ch == CPS - 1) { // end of step c
				if (((c + 1) & mask0) == 0) {
#pragma unroll
					for (int s = 0; s < NS; s++) {
						const SpecSlot d = lsl[s];              // same address in every lane: broadcast
						const unsigned ld = __builtin_amdgcn_readfirstlane(d.ld);
						if (((c + 1) & ((1u << ld) - 1u)) != 0) break; // (partial slots carry ld = 31)
						const unsigned lb = __builtin_amdgcn_readfirstlane(d.lb);
						const unsigned ilo = lb ? (__brev(c >> ld) >> (32 - lb)) : 0u;
						const size_t o = ((size_t)__builtin_amdgcn_readfirstlane((unsigned)d.goff) + r + (size_t)R * ilo) * 64;
#pragma unroll
						for (int b = 0; b < NTB; b++) { if (!(abl & 2u)) gc[b][o] = acc[b][s]; acc[b][s] = make_double2(0.0, 0.0); }
					}
				}
				if (u == XU - 1) {
#pragma unroll
					for (int uu = 0; uu < XU; uu++) {
						ca[uu] = cb[uu];
#pragma unroll
						for (int b = 0; b < NTB; b++) xa[uu][b] = xb[uu][b];
					}
				}
			}
		}
	}
#pragma unroll
	for (int s = 0; s < NS; s++)
		if (sl[s].ld == 31u && sl[s].lb != 31u) { // partial sums of the classes (lb == 31: an idle pad slot)
			const size_t o = (sl[s].goff + r) * 64;
#pragma unroll
			for (int b = 0; b < NTB; b++) gc[b][o] = acc[b][s];
		}
}

// ---- multiply-and-fold on the FP64 matrix pipe --------------------------------------------------------------------------------------------
// G_s[q] = sum_j X[q + j N_s] H_s[q + j N_s] for 64 traces and 16 scales at a time IS a complex matrix product with the aliases j as the
// contraction index: (traces x steps) . (steps x scales).  v_mfma_f64_16x16x4_f64 takes a 16 x 4 tile of X (lane = (trace i, step k)) and a
// 4 x 16 tile of H (lane = (step k, scale j)) -- every lane loads ONE value of each, nothing is broadcast, so there is no SGPR / LDS operand
// path to feed at all -- and leaves a 16 x 16 tile of sums in four registers per lane (register q: trace lane / 16 + 4 q of scale lane % 16).
// Four MFMAs per tile (re / im), 64 cycles each on gfx950: the FP64 matrix rate equals the vector rate, but ONE wave per SIMD sustains
// ~95 % of it (the vector form of this kernel: 45-50 %).  wave = (class r, group of 16 scales, trace block); steps in the same bit-reversed
// order as above, four at a time (every spectral scale has D >= 8: a bin can only complete with the last step of an aligned group of four).
typedef double spec_v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_spec_fold_mfma(const double2 *__restrict__ Xh, size_t xrows, const double2 *__restrict__ tab, const SpecSlot *__restrict__ slots,
                                                        unsigned R, unsigned logsteps, unsigned N, unsigned nblk, unsigned ngroups, double2 *__restrict__ G, size_t grows)
{
	constexpr int NS = 16;
	__shared__ double tile[4][4][2][64]; // per wave: finished bins on their way out, four scales at a time ([re | im][trace])
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// waves of a workgroup: four trace blocks of one scale group -- or, with one / two trace blocks (rows in columns), the blocks of four / two
	// scale groups: they walk the same rows of the spectra at the same time, so the groups' reads of a row are ONE read from memory (cfg4: 20
	// scales in two groups, the fold moved 0.74 GB and ran at the caches' rate)
	const unsigned bpw = nblk >= 3 ? 4u : nblk; // trace blocks per workgroup (gpw = 4 / bpw groups)
	const unsigned r = blockIdx.x, g = blockIdx.y * (4u / bpw) + wv / bpw, tb = blockIdx.z * 4 + wv % bpw;
	if (tb >= nblk || g >= ngroups) return;
	const unsigned nsteps = 1u << logsteps, M = N >> 1;
	const unsigned li = lane & 15, lk = lane >> 4;  // A: (trace li of a tile, step lk); B: (step lk, scale li); C: traces 4 lk .. + 3 of scale li
	const double2 *__restrict__ hp = tab + ((size_t)g * R + r) * nsteps * NS + li; // table stream: step c, slot s at [c NS + s]
	const SpecSlot my = slots[(size_t)g * NS + li]; // this lane's scale
	const unsigned mask0 = (1u << slots[(size_t)g * NS].ld) - 1u; // the finest scale of the group: nothing completes before it does
	const double2 *xc = Xh + (size_t)tb * xrows * 64 + li;
	double2 *gc = G + (size_t)tb * grows * 64 + lk; // (result register q of a lane: trace lk + 4 q of the tile -- tools/probes/mfma_f64_layout.hip)
	spec_v4d cre[4], cim[4];
#pragma unroll
	for (int t = 0; t < 4; t++) { cre[t] = (spec_v4d){0, 0, 0, 0}; cim[t] = (spec_v4d){0, 0, 0, 0}; }
	// operands of four steps: the rows are requested RAW; the conjugation of a mirrored row (sign bit of the imaginary part) is applied
	// when the values move into the operand registers, after the MFMAs of the block in front -- applied at the load it made the compiler
	// wait for the next block's rows before this block's MFMAs
	auto request = [&](const unsigned c0, double2 (&xr)[4], double2 &hr, int &flip) {
		const unsigned c = c0 + lk;
		const unsigned ib = logsteps ? (__brev(c) >> (32 - logsteps)) : 0u;
		const unsigned f = r + R * ib;
		const bool cj = f > M;
		const double2 *row = xc + (size_t)(cj ? N - f : f) * 64;
		flip = cj ? (int)0x80000000 : 0;
#pragma unroll
		for (int t = 0; t < 4; t++) xr[t] = row[16 * t];
		hr = hp[(size_t)c * NS];
	};
	// Three operand sets in rotation (the loop body three times, no register moves: a copy of a set in flight would wait for it): the rows of
	// the block after next are requested before this block's MFMAs.  With one set ahead a wave had ~500 cycles of MFMAs to cover a round trip of
	// several thousand; with few trace blocks (rows in columns: two) the kernel is one wave lifetime long and that round trip is all it waits for.
	struct Ops { double2 x[4]; double2 h; int f; };
	Ops oa, ob, oc;
	request(0, oa.x, oa.h, oa.f);
	request(4 < nsteps ? 4u : 0u, ob.x, ob.h, ob.f);
	auto block = [&](Ops &cur, Ops &far, const unsigned c0) {
		request(c0 + 8 < nsteps ? c0 + 8 : 0u, far.x, far.h, far.f); // (past the last block: a harmless re-request of the first -- no branch around the loads)
#pragma unroll
		for (int t = 0; t < 4; t++) cur.x[t].y = __hiloint2double(__double2hiint(cur.x[t].y) ^ cur.f, __double2loint(cur.x[t].y)); // conjugate of a mirrored row
		__builtin_amdgcn_sched_barrier(0);
		// (eight independent accumulators between two MFMAs on the same one)
#pragma unroll
		for (int t = 0; t < 4; t++) {
			cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.x[t].x, cur.h.x, cre[t], 0, 0, 0);
			cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.x[t].x, cur.h.y, cim[t], 0, 0, 0);
		}
#pragma unroll
		for (int t = 0; t < 4; t++) {
			cre[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur.x[t].y, cur.h.y, cre[t], 0, 0, 0);
			cim[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.x[t].y, cur.h.x, cim[t], 0, 0, 0);
		}
		__builtin_amdgcn_sched_barrier(0);
		const unsigned c = c0 + 3;
		if (((c + 1) & mask0) == 0) { // (wave-uniform) some scale of the group completes a bin: the lanes of those scales hand it over and start over
			// A finished bin of a scale = 64 traces x 16 bytes = one 1-KB row of G, but the matrix pipe leaves it in 4 lanes x 16 registers: stored from
			// there it is 16 store instructions of 64-byte pieces per completion, whatever the number of scales that complete -- a thousand store
			// instructions per wave at N = 131072, and the loads of the next steps wait behind them (one in-order counter for loads and stores).  Through
			// the wave's LDS tile the row goes out as ONE instruction per completed scale.
			const bool done = my.ld < 31u && ((c + 1) & ((1u << my.ld) - 1u)) == 0;
			unsigned m = (unsigned)(__ballot(done) & 0xffffull); // (lanes 0 .. 15: lk = 0, one per scale)
			const unsigned rank = (unsigned)__popc(m & ((1u << li) - 1u)); // of this lane's scale among the completing ones
#pragma unroll 1
			for (unsigned base = 0; m; base += 4) { // four completing scales at a time (4 KB of LDS per wave)
				if (done && rank - base < 4u) { // (real and imaginary parts apart: pairs would have to be gathered into register quads first -- 64 moves and registers)
					double *tr = tile[wv][rank - base][0], *ti = tile[wv][rank - base][1];
					const unsigned sw = (rank - base) << 3; // (swizzled: the four rows lie 1 KB apart, i.e. on the same banks)
#pragma unroll
					for (int t = 0; t < 4; t++)
#pragma unroll
						for (int q = 0; q < 4; q++) { tr[(unsigned)(16 * t + 4 * q + (int)lk) ^ sw] = cre[t][q]; ti[(unsigned)(16 * t + 4 * q + (int)lk) ^ sw] = cim[t][q]; }
				}
#pragma unroll 1
				for (unsigned k = 0; k < 4 && m; k++) {
					const unsigned sidx = (unsigned)__builtin_ctz(m);
					m &= m - 1u;
					const unsigned sld = (unsigned)__builtin_amdgcn_readlane((int)my.ld, (int)sidx), slb = (unsigned)__builtin_amdgcn_readlane((int)my.lb, (int)sidx);
					const unsigned long long sgo = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my.goff >> 32), (int)sidx) << 32) |
					                               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)my.goff, (int)sidx);
					const unsigned ilo = slb ? (__brev(c >> sld) >> (32 - slb)) : 0u;
					G[((size_t)tb * grows + sgo + r + (size_t)R * ilo) * 64 + lane] = make_double2(tile[wv][k][0][lane ^ (k << 3)], tile[wv][k][1][lane ^ (k << 3)]);
				}
			}
			if (done) {
#pragma unroll
				for (int t = 0; t < 4; t++) { cre[t] = (spec_v4d){0, 0, 0, 0}; cim[t] = (spec_v4d){0, 0, 0, 0}; }
			}
		}
	};
	for (unsigned c0 = 0;;) {
		block(oa, oc, c0); if ((c0 += 4) >= nsteps) break;
		block(ob, oa, c0); if ((c0 += 4) >= nsteps) break;
		block(oc, ob, c0); if ((c0 += 4) >= nsteps) break;
	}
	if (my.ld == 31u && my.lb != 31u) { // partial sums of the classes (lb == 31: an idle pad slot)
		double2 *dst = gc + (my.goff + r) * 64;
#pragma unroll
		for (int t = 0; t < 4; t++)
#pragma unroll
			for (int q = 0; q < 4; q++) dst[16 * t + 4 * q] = make_double2(cre[t][q], cim[t][q]);
	}
}

// ---- inverse transforms of the folded spectra + stacks -----------------------------------------------------------------------------
// A segment = one pass of one scale.  Passes before the last write the other ping-pong buffer; the last pass conjugates
// (Y = conj r), phase-normalises and adds the block's 64 traces into the block's ST / PS planes (COEF: writes the traces' own
// coefficients instead -- the per-trace API and the parity tests).
struct SpecEpi {
	double2 *ST, *PS;          // planes of trace block 0; block tb at + tb * stride
	size_t stride;
	const unsigned *amax;      // float bits of max |x| per trace (lane)
	double2 *Y;                // COEF: [trace][ncoef]
	size_t ncoef;
	unsigned ntr;
};

template <int R, bool COEF>
__device__ __forceinline__ void spec_inv_last_body(const double2 *__restrict__ src, const SpecSeg *__restrict__ sg, unsigned j, const double2 *__restrict__ tw,
                                                   const SpecEpi &ep, unsigned tb, unsigned lane)
{
	const unsigned L = sg->L, k = j & (L - 1);
	double2 v[R];
	spec_load_tw<R, true>(v, src, sg, j, k, tw);
	SpecDFT<R, true>::run(v);
	const size_t j0 = (size_t)(j - k) * R + k;
	const unsigned t = tb * 64 + lane;
	if constexpr (COEF) {
		if (t < ep.ntr) {
			double2 *y = ep.Y + (size_t)t * ep.ncoef + sg->coff;
#pragma unroll
			for (int n = 0; n < R; n++) if (j0 + (size_t)n * L < sg->nvalid) y[j0 + (size_t)n * L] = make_double2(v[n].x, -v[n].y);
		}
		return;
	}
	// noise floor of the transforms for this (trace, scale): a coefficient at or below it is an exact zero of the FIR form (all samples
	// under the filter are zero) and is skipped by the phase stack like the reference's 0 / 0 (ts_pws1f_lib.c:491-492)
	const double fl = sg->tau * (double)__uint_as_float(ep.amax[t]);
	const double fl2 = fl * fl;
	const unsigned o16 = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1); // element left by valu_reduce16
	double2 *pS = ep.ST + (size_t)tb * ep.stride + sg->coff, *pP = ep.PS + (size_t)tb * ep.stride + sg->coff;
	constexpr int NCH = (R + 7) / 8;
#pragma unroll
	for (int ch = 0; ch < NCH; ch++) {
		if (j0 + (size_t)(ch * 8) * L >= sg->nvalid) break; // (wave-uniform) a transform longer than the trace: the outputs from N_s on are not coefficients
		double st16[16], ps16[16];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const int n = ch * 8 + i;
			double2 y = make_double2(0.0, 0.0), u = make_double2(0.0, 0.0);
			if (n < R) {
				y = make_double2(v[n < R ? n : 0].x, -v[n < R ? n : 0].y);
				const double r2 = fma(y.x, y.x, y.y * y.y);
				if (r2 > fl2) add_unit_phasor(u, y);
			}
			st16[2 * i] = y.x; st16[2 * i + 1] = y.y;
			ps16[2 * i] = u.x; ps16[2 * i + 1] = u.y;
		}
		const double sm = valu_reduce16(st16, lane), q = valu_reduce16(ps16, lane);
		const int n = ch * 8 + (int)(o16 >> 1);
		if ((lane & 3) == 0 && n < R) {
			const size_t kk = j0 + (size_t)n * L;
			if (kk < sg->nvalid) {
				((double *)(pS + kk))[o16 & 1] = sm;
				((double *)(pP + kk))[o16 & 1] = q;
			}
		}
	}
}

template <bool COEF>
__global__ void __launch_bounds__(256) k_spec_inv(const SpecSeg *__restrict__ segs, unsigned nseg, unsigned nitems, const double2 *__restrict__ src, size_t src_rows,
                                                  double2 *__restrict__ dst, size_t dst_rows, const double2 *__restrict__ tw, const SpecEpi ep, const int rows_mode)
{
	const unsigned lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned item = blockIdx.x * 4 + wv, tb = blockIdx.y;
	if (item >= nitems) return;
	const SpecSeg *sg = spec_find_seg(segs, nseg, item);
	const unsigned j = item - sg->item0;
	const double2 *s = src + ((size_t)tb * src_rows + sg->src) * 64 + lane;
	if (sg->last && !rows_mode) { // (rows_mode: the last pass leaves its output rows r_s[k D] in the other buffer too: k_spec_stack_rows takes them from there)
		switch (sg->radix) {
		case 16: spec_inv_last_body<16, COEF>(s, sg, j, tw, ep, tb, lane); break;
		case 8: spec_inv_last_body<8, COEF>(s, sg, j, tw, ep, tb, lane); break;
		case 4: spec_inv_last_body<4, COEF>(s, sg, j, tw, ep, tb, lane); break;
		default: spec_inv_last_body<2, COEF>(s, sg, j, tw, ep, tb, lane); break;
		}
		return;
	}
	double2 *d = dst + ((size_t)tb * dst_rows + sg->dst) * 64 + lane;
	switch (sg->radix) {
	case 32: spec_mid_body<32, true>(s, d, sg, j, tw); break;
	case 16: spec_mid_body<16, true>(s, d, sg, j, tw); break;
	case 8: spec_mid_body<8, true>(s, d, sg, j, tw); break;
	case 4: spec_mid_body<4, true>(s, d, sg, j, tw); break;
	default: spec_mid_body<2, true>(s, d, sg, j, tw); break;
	}
}

// ---- stacks of FEW transformed rows grouped in columns (the K partial stacks of every jackknife replica: resample.hip) --------------------
// The inverse passes left r_s[k D] of every row (lane) in rows of 64 lanes.  A wave takes one coefficient (scale, k): conj, phase-normalise
// per lane, then column c = the lanes [c tps, (c + 1) tps) -- across the trace blocks -- is added IN ROW ORDER (the order of
// ts_pws1f_lib.c:885-906) by one lane per column from the wave's LDS slab; the weighted coefficient goes straight to the column's set
// (tspws_biased / tspws_unbiased, :909-984), or the linear / phase stacks to the column's planes.
struct SpecRowScale {
	unsigned long long roff;   // first coefficient of the scale in the flattened list of the spectral coefficients
	unsigned long long goff;   // first row of the scale in a trace block's region
	unsigned long long coff;   // first coefficient of the scale in a coefficient set
	double tau;
	unsigned Ns, buf;          // buf: which of the two buffers the scale's last pass wrote
};
struct SpecRowsOut {
	double2 *OUT; size_t out_stride; const double *Mv; double M, K, wu; int mode, keep; double2 *keepST; // weighted sets (OUT != NULL) ...
	double2 *accST, *accPS; size_t stride;                                                                 // ... or plane pairs per column
};

__global__ void __launch_bounds__(256) k_spec_stack_rows(const SpecRowScale *__restrict__ rs, unsigned nrs, unsigned long long ntot, const double2 *__restrict__ b0,
                                                         size_t rows0, const double2 *__restrict__ b1, size_t rows1, unsigned nblk, unsigned ntr, unsigned tps,
                                                         unsigned ncol, const unsigned *__restrict__ amax, const SpecRowsOut o)
{
	extern __shared__ double slab[]; // [4 waves][nblk 64 lanes][4]
	const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long i = (unsigned long long)blockIdx.x * 4 + wv;
	if (i >= ntot) return;
	// the scale of coefficient i: the last one with roff <= i (rs[0].roff = 0) -- counted by the lanes (a binary search is five DEPENDENT loads;
	// at most 128 spectral scales)
	const unsigned long long ro0 = lane < nrs ? rs[lane].roff : ~0ull, ro1 = lane + 64 < nrs ? rs[lane + 64].roff : ~0ull;
	const unsigned lo = (unsigned)__popcll(__ballot(ro0 <= i)) + (unsigned)__popcll(__ballot(ro1 <= i)) - 1u;
	const SpecRowScale d = rs[lo];
	const size_t k = (size_t)(i - d.roff);
	const double2 *__restrict__ src = d.buf ? b1 : b0;
	const size_t rows = d.buf ? rows1 : rows0;
	double *w = slab + (size_t)wv * nblk * 64 * 4;
	for (unsigned tb = 0; tb < nblk; tb++) {
		const unsigned t = tb * 64 + lane;
		const double2 r = src[((size_t)tb * rows + d.goff + k) * 64 + lane];
		const double2 y = t < ntr ? make_double2(r.x, -r.y) : make_double2(0.0, 0.0); // Y = conj r
		double2 u = make_double2(0.0, 0.0);
		const double fl = d.tau * (double)__uint_as_float(amax[t]);
		if (fma(y.x, y.x, y.y * y.y) > fl * fl) add_unit_phasor(u, y);               // (at or below the transforms' noise floor: an exact zero of the FIR form)
		double *e = w + (size_t)t * 4;
		e[0] = y.x; e[1] = y.y; e[2] = u.x; e[3] = u.y;
	}
	// (the slab is the wave's own: LDS operations of a wave complete in order)
	for (unsigned c = lane; c < ncol; c += 64) {
		double2 st = make_double2(0.0, 0.0), ps = make_double2(0.0, 0.0);
		const unsigned t0 = c * tps, t1 = min(ntr, t0 + tps);
		for (unsigned t = t0; t < t1; t++) {
			const double *e = w + (size_t)t * 4;
			st.x += e[0]; st.y += e[1]; ps.x += e[2]; ps.y += e[3];
		}
		const size_t ci = (size_t)d.coff + k;
		if (o.OUT) {
			o.OUT[(size_t)c * o.out_stride + ci] = weight_value(st, ps, o.mode, o.K, o.Mv ? o.Mv[c] : o.M, o.wu);
			if ((int)c == o.keep && o.keepST) o.keepST[ci] = st;
		} else {
			o.accST[(size_t)c * o.stride + ci] = st;
			o.accPS[(size_t)c * o.stride + ci] = ps;
		}
	}
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static unsigned ilog2u(unsigned v) { unsigned l = 0; while ((1u << l) < v) l++; return l; }

// radices (as bit counts) of a 2^m-point transform: as few passes as possible with <= 5 bits each, the LAST pass <= last_max bits
static std::vector<unsigned> radix_bits(unsigned m, unsigned last_max)
{
	std::vector<unsigned> b;
	if (!m) return b;
	for (unsigned np = (m + 4) / 5;; np++) {
		const unsigned base = m / np, extra = m % np;
		if (base > last_max || base + (extra ? 1u : 0u) > 5u) continue;
		for (unsigned i = 0; i < np; i++) b.push_back(base + (i < extra ? 1u : 0u));
		return b;
	}
}

struct SpecPlan {
	unsigned s_first = 0, s_end = 0;      // scales [s_first, s_end) are spectral
	unsigned N = 0;                       // transform length NT (a power of two; the trace length itself when that is one)
	unsigned Nx = 0, split = 0;           // trace length; samples [split, NT) of the transform window hold the samples in FRONT of the trace (spec_ext_row)
	unsigned M = 0, R = 0, logsteps = 0, NS = 0, ngroups = 0;
	size_t grows = 0;                     // rows of a trace block's folded spectra (all slots)
	std::vector<unsigned> fwd_bits;       // passes of the trace transform
	double2 *d_tw = nullptr, *d_tab = nullptr;
	SpecSlot *d_slots = nullptr;
	SpecSeg *d_fseg = nullptr;            // middle passes of the trace transform (one segment each)
	std::vector<SpecSeg *> d_iseg;        // inverse levels
	std::vector<unsigned> iseg_n, iseg_items;
	SpecRowScale *d_rows = nullptr;       // k_spec_stack_rows: the spectral scales in order
	unsigned long long nrowcoef = 0;      // ... and their coefficients altogether
};

static void spec_plan_free(SpecPlan *sp)
{
	if (!sp) return;
	if (sp->d_tw) (void)hipFree(sp->d_tw);
	if (sp->d_tab) (void)hipFree(sp->d_tab);
	if (sp->d_slots) (void)hipFree(sp->d_slots);
	if (sp->d_fseg) (void)hipFree(sp->d_fseg);
	for (SpecSeg *s : sp->d_iseg) if (s) (void)hipFree(s);
	if (sp->d_rows) (void)hipFree(sp->d_rows);
	delete sp;
}

static void spec_decomp_free(SpecDecomp *d)
{
	if (!d) return;
	spec_plan_free(d->sp);
	if (d->T.d_sc) (void)hipFree(d->T.d_sc);
	if (d->T.d_items) (void)hipFree(d->T.d_items);
	if (d->T.d_gcols) (void)hipFree(d->T.d_gcols);
	delete d;
}

void tspws_spectral_destroy(tspws_hip_plan *p)
{
	for (SpecDecomp *d : p->spec) spec_decomp_free(d);
	p->spec.clear();
}

// Transform geometry of this frame: the transform length NT, the end of the scales whose filters fit its window, the samples kept in front
// of the trace.  N a power of two: NT = N, the reference's circular correlation (cdotx.c:44-70) IS the transform's, every scale fits.  Any
// other N: the correlation is evaluated as a linear one over a window of the trace's periodic extension (k_spec_fwd_first), which takes
// NT >= N + L - 1 for a filter of L taps.  NT0 = the next power of two above N serves the scales with L <= NT0 - N + 1; the longer ones (the
// coarsest scales, few outputs each) stay on the direct FIR kernel -- unless they are more than a quarter of the far-decimated work, then the
// window is 2 NT0 >= 2 N and holds every filter (L <= N).  TSPWS_SPEC_NT = min / double (sweeps) pins the choice.
void tspws_spectral_geometry(const tspws_hip_plan *p, unsigned *NT, unsigned *s_end, unsigned *cneg)
{
	const unsigned N = p->N, S = p->S;
	*NT = N; *s_end = S; *cneg = 0;
	if ((N & (N - 1)) == 0) return;
	unsigned nt = 1;
	while (nt < N) nt <<= 1;
	unsigned e = 0;
	while (e < S && (unsigned long long)N + p->sc[e].L - 1 <= nt) e++; // (L grows with the scale)
	double in = 0, out = 0;
	for (unsigned s = 0; s < S; s++) {
		const unsigned D = p->sc[s].D;
		if (D < 8 || (D & (D - 1))) continue;
		(s < e ? in : out) += (double)p->sc[s].L * (double)p->sc[s].Ns;
	}
	bool dbl = e < S && out > 0.25 * (in + out);
	if (const char *v = sweep_env("TSPWS_SPEC_NT")) { if (!strcmp(v, "min")) dbl = false; else if (!strcmp(v, "double")) dbl = true; }
	if (dbl && nt <= (1u << 30)) { nt <<= 1; e = S; }
	int cm = 0;
	for (unsigned s = 0; s < e; s++) cm = std::max(cm, p->sc[s].c);
	*NT = nt; *s_end = e; *cneg = (unsigned)cm;
}

unsigned tspws_spectral_end_scale(const tspws_hip_plan *p)
{
	unsigned NT, e, c;
	tspws_spectral_geometry(p, &NT, &e, &c);
	return e;
}

// Can scale s go through the spectrum?  N >= 1024, D a power of two >= 8 with at least two folded bins, the filter inside the transform window
// (N a power of two: D | N and the window is the trace).
static bool spec_scale_ok(const tspws_hip_plan *p, unsigned s, unsigned NT, unsigned s_end)
{
	const unsigned N = p->N, D = p->sc[s].D;
	if (N < 1024 || s >= s_end || D < 8 || (D & (D - 1)) || NT / D < 2 || p->sc[s].L > N) return false;
	if (NT == N) return N % D == 0 && p->sc[s].Ns == N / D && p->sc[s].Ns >= 2;
	return (unsigned long long)N + p->sc[s].L - 1 <= NT && p->sc[s].Ns == (N + D - 1) / D;
}

// first scale of the spectral set when every octave with at most nsmax outputs is to go through the spectrum (none: the set's end).
// The set is a run of scales that ends at tspws_spectral_end_scale (S for most frames): whole octaves at its fine end, D doubling from octave to octave.
unsigned tspws_spectral_first_scale(const tspws_hip_plan *p, unsigned nsmax)
{
	unsigned NT, s_end, cneg;
	tspws_spectral_geometry(p, &NT, &s_end, &cneg);
	unsigned first = s_end;
	for (unsigned e = s_end; e > 0;) {
		unsigned s = e - 1;
		while (s > 0 && p->sc[s - 1].D == p->sc[e - 1].D && p->sc[s - 1].Ns == p->sc[e - 1].Ns) s--;
		bool ok = p->sc[s].Ns <= nsmax && s_end - s <= 128; // (at most 8 groups of 16 accumulators)
		for (unsigned v = s; v < e && ok; v++) ok = spec_scale_ok(p, v, NT, s_end);
		if (ok && e < s_end && p->sc[e].D != 2 * p->sc[s].D) ok = false; // consecutive octaves double the decimation
		if (!ok) break;
		first = s;
		e = s;
	}
	return first < s_end ? first : p->S; // (S: no set)
}

static int spec_build(tspws_hip_plan *p, unsigned s_first, unsigned nblk_hint, SpecPlan *sp)
{
	// (device temporaries of the plan-time transforms: freed on every exit)
	struct Tmp {
		void *v[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
		~Tmp() { for (void *q : v) if (q) (void)hipFree(q); }
	} tmp;
	unsigned NT, s_end, cneg;
	tspws_spectral_geometry(p, &NT, &s_end, &cneg);
	const unsigned N = NT, M = N / 2, S = p->S;
	if (s_first >= s_end) return fail(TSPWS_E_ARG, "spectral: empty set");
	const unsigned nsc = s_end - s_first;
	sp->s_first = s_first; sp->s_end = s_end; sp->N = N; sp->M = M; sp->Nx = p->N; sp->split = N - cneg;
	// scale groups: groups of at most 16 scales (a wave's accumulators for two trace blocks: 128 VGPRs), padded to 8 or 16 slots (a 1-KB
	// block of a group's table stream holds 8 or 4 steps); the groups of a class share a workgroup (k_spec_fold)
	if (nsc > 128) return fail(TSPWS_E_ARG, "spectral: more than 128 scales in the spectral set");
	// (8 slots per group only where that saves a quarter of the slots -- 18 scales: 3 x 8 = 24 against 2 x 16 = 32; 40 scales: 5 x 8 = 40
	// against 48, but five groups of one wave each lose more than the eight idle slots cost: 0.67 vs 0.42 ms on cfg2)
	unsigned nsw = 16; // (the matrix-pipe fold takes tiles of 16 scales; the vector-FMA form's 8-slot groups: sweeps only)
	if (const char *e = sweep_env("TSPWS_SPEC_NSW")) nsw = atoi(e) <= 8 ? 8u : 16u; // sweeps
	if (nsc > 8 * nsw) nsw = 16;
	const unsigned ngroups = (nsc + nsw - 1) / nsw;
	const unsigned per = (nsc + ngroups - 1) / ngroups, NS = per <= 8 ? 8u : 16u;
	sp->ngroups = ngroups; sp->NS = NS;
	// classes: enough waves for the chip (>= ~4096 items with the caller's trace blocks), at least 16 steps per wave, at most N / 4 .. and
	// not more classes than the finest spectral scale has outputs would be wasteful but is legal (partial sums)
	unsigned R = 64;
	while ((size_t)R * ngroups * std::max(1u, nblk_hint) < 4096 && N / (2 * R) >= 32) R *= 2;
	sp->R = R; sp->logsteps = ilog2u(N / R);
	// slots of a group in the order of the scales (D grows), partial ones (fewer folded bins than classes) last by construction; pads are idle.
	// A scale's folded spectrum has N / D bins (= its N_s outputs when the transform is the trace's length)
	std::vector<SpecSlot> slots((size_t)ngroups * NS);
	std::vector<unsigned> slot_scale((size_t)ngroups * NS, ~0u);
	std::vector<size_t> goff(S, 0);
	size_t rows = 0;
	for (unsigned g = 0; g < ngroups; g++) {
		for (unsigned i = 0; i < NS; i++) {
			SpecSlot &sl = slots[(size_t)g * NS + i];
			const unsigned s = s_first + g * per + i;
			if (i < per && s < s_end) {
				const unsigned D = p->sc[s].D, Nb = N / D;
				slot_scale[(size_t)g * NS + i] = s;
				goff[s] = rows;
				sl.goff = rows;
				if (Nb >= R) { sl.ld = ilog2u(D); sl.lb = ilog2u(Nb / R); rows += Nb; }
				else { sl.ld = 31; sl.lb = 0; rows += R; }
			} else { sl.ld = 31; sl.lb = 31; sl.goff = 0; }
		}
		// a group whose FIRST slot is partial never completes anything inside the loop: mask0 must not fire -- ld = 31 gives mask 2^31 - 1
	}
	if (rows >> 32) return fail(TSPWS_E_ARG, "spectral: folded spectra of a trace block exceed 2^32 rows");
	sp->grows = rows;
	HIP_TRY(hipMalloc(&sp->d_slots, slots.size() * sizeof(SpecSlot)));
	HIP_TRY(hipMemcpy(sp->d_slots, slots.data(), slots.size() * sizeof(SpecSlot), hipMemcpyHostToDevice));
	HIP_TRY(hipMalloc(&sp->d_tw, (size_t)N * sizeof(double2)));
	hipLaunchKernelGGL(k_spec_twiddle, dim3((N + 255) / 256), dim3(256), 0, 0, sp->d_tw, N);
	// ---- trace transform: M = N / 2 complex points, last pass <= 16 (it holds two butterflies) ----
	sp->fwd_bits = radix_bits(ilog2u(M), 4);
	{
		// One pass fewer where ONE 64-point pass makes it possible (every pass moves the whole batch through HBM / MALL once each way): M = 32 768
		// (N = 65 536) 16 x 16 x 16 x 8 -> 64 x 32 x 16, the 64-point pass first (no twiddles): 512 x 65 536 single-stage 1.76 -> 1.64 ms.  The
		// 64-point butterfly takes 256 VGPRs (one wave per SIMD, its own kernel instantiation): M = 65 536 as 64 x 64 x 16 -- TWO such passes -- lost
		// (256 x 131 072 1.85 -> 1.88 ms; cfg4's two trace blocks beside k_fwd_lds 1.92 -> 2.00: 512 one-wave workgroups take 0.54 ms for the first
		// pass), so only plans with one 64-point pass are taken (tools/experiments/r6_r64.sh).  TSPWS_SPEC_R64=0 (sweeps): never.
		static const bool r64_off = sweep_env("TSPWS_SPEC_R64") && !strcmp(sweep_env("TSPWS_SPEC_R64"), "0");
		const unsigned m = ilog2u(M);
		if (!r64_off && nblk_hint >= 4 && m == 15) { // (11 bits in front of the 16-point last pass: 6 + 5; batches of >= 4 trace blocks: enough one-wave workgroups)
			if (3 < sp->fwd_bits.size()) sp->fwd_bits = std::vector<unsigned>{6u, 5u, 4u};
		}
	}
	{
		std::vector<SpecSeg> fs;
		unsigned L = 1;
		for (size_t i = 0; i < sp->fwd_bits.size(); i++) {
			const unsigned Rr = 1u << sp->fwd_bits[i];
			SpecSeg g;
			memset(&g, 0, sizeof g);
			g.len = M; g.L = L; g.radix = Rr; g.nfold = 1; g.tw_mul = N / (L * Rr);
			fs.push_back(g);
			L *= Rr;
		}
		HIP_TRY(hipMalloc(&sp->d_fseg, fs.size() * sizeof(SpecSeg)));
		HIP_TRY(hipMemcpy(sp->d_fseg, fs.data(), fs.size() * sizeof(SpecSeg), hipMemcpyHostToDevice));
	}
	// ---- tap norms (noise floor of a scale) ----
	std::vector<double> wn(S, 0.0);
	{
		HIP_TRY(hipMalloc(&tmp.v[0], S * sizeof(double)));
		hipLaunchKernelGGL(k_spec_wnorm, dim3(S), dim3(256), 0, 0, (const ScaleDesc *)p->d_sc, (const double2 *)p->d_w, (double *)tmp.v[0]);
		HIP_TRY(hipMemcpy(wn.data(), tmp.v[0], S * sizeof(double), hipMemcpyDeviceToHost));
	}
	// |error| of an output of the transform-based correlation <~ eps log2(N) ||y||_2 ||w||_2 <= eps log2(N) sqrt(N) max|x| ||w||_2; x 8
	auto tau_of = [&](unsigned s) { return 8.0 * 1.1102230246251565e-16 * (double)ilog2u(N) * sqrt((double)N) * wn[s]; };
	// ---- inverse transforms: level l = pass l of every scale with more than l passes ----
	{
		std::vector<std::vector<SpecSeg>> lev;
		for (unsigned s = s_first; s < s_end; s++) {
			const unsigned Nb = N / p->sc[s].D;
			const std::vector<unsigned> bits = radix_bits(ilog2u(Nb), 4); // (last pass <= 16 points: its epilogue holds the normalised copies too)
			unsigned L = 1;
			for (size_t i = 0; i < bits.size(); i++) {
				if (lev.size() <= i) lev.emplace_back();
				const unsigned Rr = 1u << bits[i];
				SpecSeg g;
				memset(&g, 0, sizeof g);
				g.len = Nb; g.L = L; g.radix = Rr; g.tw_mul = N / (L * Rr);
				g.nfold = (i == 0 && Nb < R) ? R / Nb : 1u;
				g.src = g.dst = goff[s];
				g.last = (i + 1 == bits.size()) ? 1u : 0u;
				g.nvalid = p->sc[s].Ns;
				g.coff = p->sc[s].coef_off;
				g.tau = tau_of(s);
				lev[i].push_back(g);
				L *= Rr;
			}
		}
		{ // the scales' final rows for k_spec_stack_rows: the last of a scale's npass passes wrote buffer (npass & 1) (pass 0 reads buffer 0)
			std::vector<SpecRowScale> rv;
			unsigned long long ro = 0;
			for (unsigned s = s_first; s < s_end; s++) {
				const unsigned Ns = p->sc[s].Ns;
				SpecRowScale q;
				memset(&q, 0, sizeof q);
				q.roff = ro; q.goff = goff[s]; q.coff = p->sc[s].coef_off; q.Ns = Ns;
				q.buf = (unsigned)(radix_bits(ilog2u(N / p->sc[s].D), 4).size() & 1u);
				q.tau = tau_of(s);
				rv.push_back(q);
				ro += Ns;
			}
			sp->nrowcoef = ro;
			HIP_TRY(hipMalloc(&sp->d_rows, rv.size() * sizeof(SpecRowScale)));
			HIP_TRY(hipMemcpy(sp->d_rows, rv.data(), rv.size() * sizeof(SpecRowScale), hipMemcpyHostToDevice));
		}
		for (std::vector<SpecSeg> &v : lev) {
			unsigned items = 0;
			for (SpecSeg &g : v) { g.item0 = items; items += g.len / g.radix; }
			SpecSeg *d = nullptr;
			HIP_TRY(hipMalloc(&d, v.size() * sizeof(SpecSeg)));
			sp->d_iseg.push_back(d); sp->iseg_n.push_back((unsigned)v.size()); sp->iseg_items.push_back(items);
			HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(SpecSeg), hipMemcpyHostToDevice));
		}
	}
	// ---- tables H_s: taps placed circularly in the transform window -> N-point transform with lane = slot -> class / step order, 1 / N ----
	{
		const unsigned nslots = ngroups * NS, nsb = (nslots + 63) / 64;
		const size_t bytes = (size_t)nsb * N * 64 * sizeof(double2);
		HIP_TRY(hipMalloc(&tmp.v[1], slot_scale.size() * sizeof(unsigned)));
		unsigned *d_ss = (unsigned *)tmp.v[1];
		HIP_TRY(hipMemcpy(d_ss, slot_scale.data(), slot_scale.size() * sizeof(unsigned), hipMemcpyHostToDevice));
		HIP_TRY(hipMalloc(&tmp.v[2], bytes));
		HIP_TRY(hipMalloc(&tmp.v[3], bytes));
		double2 *a = (double2 *)tmp.v[2], *b = (double2 *)tmp.v[3];
		HIP_TRY(hipMemset(a, 0, bytes));
		hipLaunchKernelGGL(k_spec_place, dim3((p->N + 255) / 256, nslots), dim3(256), 0, 0, (const ScaleDesc *)p->d_sc, (const double2 *)p->d_w, (const unsigned *)d_ss, N, a);
		const std::vector<unsigned> bits = radix_bits(ilog2u(N), 5);
		std::vector<SpecSeg> ts;
		unsigned L = 1;
		for (unsigned bt : bits) {
			SpecSeg g;
			memset(&g, 0, sizeof g);
			g.len = N; g.L = L; g.radix = 1u << bt; g.nfold = 1; g.tw_mul = N / (L * g.radix);
			ts.push_back(g);
			L *= g.radix;
		}
		HIP_TRY(hipMalloc(&tmp.v[4], ts.size() * sizeof(SpecSeg)));
		SpecSeg *d_ts = (SpecSeg *)tmp.v[4];
		HIP_TRY(hipMemcpy(d_ts, ts.data(), ts.size() * sizeof(SpecSeg), hipMemcpyHostToDevice));
		for (size_t i = 0; i < ts.size(); i++) {
			const unsigned items = N / ts[i].radix;
			hipLaunchKernelGGL((k_spec_mid<true>), dim3((items + 3) / 4, nsb), dim3(256), 0, 0, (const SpecSeg *)(d_ts + i), 1u, items, (const double2 *)a, (size_t)N, b,
			                   (size_t)N, (const double2 *)sp->d_tw);
			std::swap(a, b);
		}
		const size_t total = (size_t)ngroups * N * NS;
		HIP_TRY(hipMalloc(&sp->d_tab, (total + 8) * sizeof(double2)));
		HIP_TRY(hipMemset(sp->d_tab + total, 0, 8 * sizeof(double2)));
		hipLaunchKernelGGL(k_spec_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (const double2 *)a, N, R, sp->logsteps, NS, ngroups, sp->d_tab);
		HIP_TRY(hipDeviceSynchronize());
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// The decomposition of a many-trace batch with the scales [s_first, end of the frame's spectral scales) on the spectral engine (built on
// first use, kept by the plan).  Built into a local object: a failed build (an allocation of the plan-time tables) leaves nothing behind
// that a later call could find under its key.
int tspws_spectral_decomp(tspws_hip_plan *p, unsigned s_first, unsigned nblk_hint, SpecDecomp **out, bool few)
{
	// (the class count of the fold is sized for the batch: a decomposition built for a few trace blocks is not the one for many)
	const bool small = nblk_hint < 8;
	for (SpecDecomp *d : p->spec) if (d->s_first == s_first && d->few == few && d->small == small) { *out = d; return 0; }
	SpecDecomp *d = new SpecDecomp;
	d->s_first = s_first; d->few = few; d->small = small;
	d->sp = new SpecPlan;
	int rc = spec_build(p, s_first, nblk_hint, d->sp);
	d->s_end = d->sp->s_end;
	if (!rc && !few) rc = tspws_build_tl_spectral(p, s_first, d->s_end, nblk_hint, d->T); // (few rows in columns: the FIR kernels of the few-trace path do the rest)
	if (rc) { spec_decomp_free(d); return rc; }
	p->spec.push_back(d);
	*out = d;
	return 0;
}

// Everything between the transposed batch and the planes of the spectral scales.  xT: [N][TP] (TP = nblk 64, pad lanes zero);
// planes of block tb at ST / PS + tb * stride; Y != NULL: per-trace coefficients [ntr][ncoef] of the spectral scales instead.
template <typename TIn>
static int spectral_run(tspws_hip_plan *p, SpecDecomp *dc, const TIn *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st,
                        const SpecRowsOut *ro = nullptr, unsigned tps = 0, unsigned ncol = 0)
{
	SpecPlan *sp = dc->sp;
	const unsigned N = sp->N, M = sp->M, nblk = TP / 64; // (N: the transform length; the traces have sp->Nx samples)
	const size_t xrows = (size_t)M + 1;
	void *v;
	int rc;
	if ((rc = scratch(p, SCR_SPA, (size_t)nblk * xrows * 64 * sizeof(double2), &v))) return rc;
	double2 *A = (double2 *)v;
	if ((rc = scratch(p, SCR_SPB, (size_t)nblk * xrows * 64 * sizeof(double2), &v))) return rc;
	double2 *B = (double2 *)v;
	if ((rc = scratch(p, SCR_SPG, (size_t)nblk * sp->grows * 64 * sizeof(double2), &v))) return rc;
	double2 *G = (double2 *)v;
	if ((rc = scratch(p, SCR_SPM, (size_t)TP * sizeof(unsigned), &v))) return rc;
	unsigned *amax = (unsigned *)v;
	// (amax: the largest |sample| of every trace / row, left there by the transposition -- tspws_spectral_transpose_*, spectral_rows)
	// trace transform
	const size_t np = sp->fwd_bits.size();
	{
		const unsigned R0 = 1u << sp->fwd_bits[0];
		if (R0 == 64) hipLaunchKernelGGL((k_spec_fwd_first<TIn, true>), dim3((M / R0 + 3) / 4, nblk), dim3(256), 0, st, xT, TP, A, xrows, M, R0, sp->Nx, sp->split);
		else hipLaunchKernelGGL((k_spec_fwd_first<TIn>), dim3((M / R0 + 3) / 4, nblk), dim3(256), 0, st, xT, TP, A, xrows, M, R0, sp->Nx, sp->split);
	}
	double2 *cur = A, *oth = B;
	for (size_t i = 1; i + 1 < np; i++) {
		const unsigned items = M >> sp->fwd_bits[i];
		if (sp->fwd_bits[i] == 6) hipLaunchKernelGGL((k_spec_mid<false, true>), dim3((items + 3) / 4, nblk), dim3(256), 0, st, (const SpecSeg *)(sp->d_fseg + i), 1u, items, (const double2 *)cur, xrows, oth,
		                                             xrows, (const double2 *)sp->d_tw);
		else hipLaunchKernelGGL((k_spec_mid<false>), dim3((items + 3) / 4, nblk), dim3(256), 0, st, (const SpecSeg *)(sp->d_fseg + i), 1u, items, (const double2 *)cur, xrows, oth,
		                        xrows, (const double2 *)sp->d_tw);
		std::swap(cur, oth);
	}
	{
		const unsigned Rl = 1u << sp->fwd_bits[np - 1], L = M / Rl;
		hipLaunchKernelGGL(k_spec_fwd_last, dim3((L / 2 + 1 + 3) / 4, nblk), dim3(256), 0, st, (const double2 *)cur, xrows, oth, xrows, M, Rl, (const double2 *)sp->d_tw);
		std::swap(cur, oth);
	}
	const double2 *Xh = cur;
	// multiply-and-fold
	static const bool fold_lds = sweep_env("TSPWS_SPEC_FOLD") && !strcmp(sweep_env("TSPWS_SPEC_FOLD"), "lds"); // A/B: the vector-FMA form
	if (sp->NS == 16 && !fold_lds) {
		const unsigned gpw = nblk >= 3 ? 1u : 4u / nblk; // scale groups per workgroup (k_spec_fold_mfma)
		hipLaunchKernelGGL(k_spec_fold_mfma, dim3(sp->R, (sp->ngroups + gpw - 1) / gpw, (nblk + 3) / 4), dim3(256), 0, st, Xh, xrows, (const double2 *)sp->d_tab,
		                   (const SpecSlot *)sp->d_slots, sp->R, sp->logsteps, N, nblk, sp->ngroups, G, sp->grows);
	} else {
		static int ntb = -1;
		if (ntb < 0) { const char *e = sweep_env("TSPWS_SPEC_NTB"); ntb = (e && atoi(e) == 1) ? 1 : 2; } // sweeps
		const unsigned NTB = (unsigned)ntb;
		static const unsigned abl = sweep_env("TSPWS_SPEC_ABL") ? (unsigned)atoi(sweep_env("TSPWS_SPEC_ABL")) : 0u; // timing ablations (results wrong)
		const unsigned nsets = (nblk + NTB - 1) / NTB;
		const unsigned WG = sp->ngroups, WT = std::max(1u, std::min(8u / WG, nsets)); // (spec_build: at most 8 groups; no more waves than sets of blocks)
		const dim3 grid(sp->R, (nsets + WT - 1) / WT), block(64 * WG * WT);
		const size_t lds = (size_t)WG * SP_RB * 1024 + (size_t)WG * WT * 16 * sizeof(SpecSlot);
#define SPEC_FOLD(NSV, NT) hipLaunchKernelGGL((k_spec_fold<NSV, NT>), grid, block, lds, st, Xh, xrows, (const double2 *)sp->d_tab, (const SpecSlot *)sp->d_slots, sp->R, sp->logsteps, N, nblk, WG, G, sp->grows, abl)
		if (NTB == 2) { if (sp->NS == 8) SPEC_FOLD(8, 2); else SPEC_FOLD(16, 2); }
		else { if (sp->NS == 8) SPEC_FOLD(8, 1); else SPEC_FOLD(16, 1); }
#undef SPEC_FOLD
	}
	// inverse transforms: ping-pong between the folded spectra and the (now free) trace-transform buffer that does not hold Xh
	SpecEpi ep;
	ep.ST = ST; ep.PS = PS; ep.stride = stride; ep.amax = amax; ep.Y = Y; ep.ncoef = p->ncoef; ep.ntr = ntr;
	double2 *g2 = oth; // rows: xrows >= grows?  not in general: own buffer when it is too small
	size_t g2rows = xrows;
	if ((ro || sp->d_iseg.size() > 1) && sp->grows > xrows) {
		if ((rc = scratch(p, SCR_SPH, (size_t)nblk * sp->grows * 64 * sizeof(double2), &v))) return rc;
		g2 = (double2 *)v; g2rows = sp->grows;
	}
	const double2 *src = G;
	size_t src_rows = sp->grows;
	double2 *dst = g2;
	size_t dst_rows = g2rows;
	for (size_t l = 0; l < sp->d_iseg.size(); l++) {
		const unsigned items = sp->iseg_items[l];
		if (Y) hipLaunchKernelGGL((k_spec_inv<true>), dim3((items + 3) / 4, nblk), dim3(256), 0, st, (const SpecSeg *)sp->d_iseg[l], sp->iseg_n[l], items, src, src_rows, dst, dst_rows,
		                          (const double2 *)sp->d_tw, ep, 0);
		else hipLaunchKernelGGL((k_spec_inv<false>), dim3((items + 3) / 4, nblk), dim3(256), 0, st, (const SpecSeg *)sp->d_iseg[l], sp->iseg_n[l], items, src, src_rows, dst, dst_rows,
		                        (const double2 *)sp->d_tw, ep, ro ? 1 : 0);
		// the next level reads what this one wrote; the level after that may overwrite this level's input
		const double2 *ns = dst;
		const size_t nr = dst_rows;
		dst = (double2 *)src; dst_rows = src_rows;
		src = ns; src_rows = nr;
	}
	if (ro) {
		if (nblk > 8) return fail(TSPWS_E_ARG, "spectral: at most 512 rows in columns");
		const size_t lds = (size_t)4 * nblk * 64 * 4 * sizeof(double);
		hipLaunchKernelGGL(k_spec_stack_rows, dim3((unsigned)((sp->nrowcoef + 3) / 4)), dim3(256), lds, st, (const SpecRowScale *)sp->d_rows, sp->s_end - sp->s_first, sp->nrowcoef,
		                   (const double2 *)G, sp->grows, (const double2 *)g2, g2rows, nblk, ntr, tps, ncol, (const unsigned *)amax, *ro);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// The spectral scales of FEW rows grouped in columns of tps consecutive rows (column c = rows [c tps, (c + 1) tps)): weighted coefficient
// sets or plane pairs per column (SpecRowsOut); d_x: the rows themselves ([ntr][ld]), transposed here.
#define SPEC_TR_TILES 4 /* 64-sample tiles per workgroup of the rows transposition */
template <typename TIn> __global__ void __launch_bounds__(256) k_spec_transpose_rows(const TIn *__restrict__ x, size_t ld, unsigned ntr, unsigned N, unsigned TP,
                                                                                      TIn *__restrict__ xT, unsigned *__restrict__ pmax)
{
	// a workgroup transposes SPEC_TR_TILES tiles of 64 rows x 64 samples and keeps the rows' largest |sample| on the way (the noise floor of the
	// transforms scales with it): written per workgroup to pmax[blockIdx.x][TP], reduced by k_spec_rowmax.  (Atomics on the rows' maxima: one per
	// row and tile was 2048 colliding atomics per row at N = 131072, 0.9 ms; one per row and 1024 samples still ~0.1 ms of a 0.16-ms kernel that
	// left three quarters of the CUs' wave slots empty -- 256 workgroups.)
	__shared__ TIn tile[64][65];
	const unsigned t0 = blockIdx.y * 64;
	const unsigned tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	float m[16];
#pragma unroll
	for (int i = 0; i < 16; i++) m[i] = 0.f;
	for (unsigned it = 0; it < SPEC_TR_TILES; it++) {
		const unsigned n0 = (blockIdx.x * SPEC_TR_TILES + it) * 64;
		if (n0 >= N) break;
		TIn v[16];
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const unsigned t = t0 + ty + 4u * (unsigned)i, n = n0 + tx;
			v[i] = (t < ntr && n < N) ? x[(size_t)t * ld + n] : (TIn)0;
		}
		if (it) __syncthreads(); // the previous tile has been read out
#pragma unroll
		for (int i = 0; i < 16; i++) {
			tile[ty + 4 * i][tx] = v[i];
			const float a = fabsf((float)v[i]);
			m[i] = (a == a) ? fmaxf(m[i], a) : __int_as_float(0x7f800000);
		}
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 16; i++) { const unsigned r = ty + 4u * (unsigned)i, n = n0 + r; if (n < N) xT[(size_t)n * TP + t0 + tx] = tile[tx][r]; }
	}
#pragma unroll
	for (int i = 0; i < 16; i++) {
		float a = m[i];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
		if (tx == 0) pmax[(size_t)blockIdx.x * TP + t0 + ty + 4u * (unsigned)i] = __float_as_uint(a * 1.0000002f); // (non-negative floats order like their bit patterns)
	}
}

// amax[t] = max over the np workgroup rows of pmax[.][t]; grid = TP / 64 blocks of 1024 threads (16 waves share the rows, lane = trace)
__global__ void __launch_bounds__(1024) k_spec_rowmax(const unsigned *__restrict__ pmax, unsigned np, unsigned TP, unsigned *__restrict__ amax)
{
	__shared__ unsigned sm[16][64];
	const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t = blockIdx.x * 64 + lane;
	unsigned a = 0;
	for (unsigned g = wv; g < np; g += 16) a = max(a, pmax[(size_t)g * TP + t]);
	sm[wv][lane] = a;
	__syncthreads();
	if (wv == 0) {
#pragma unroll
		for (int w = 1; w < 16; w++) a = max(a, sm[w][lane]);
		amax[t] = a;
	}
}

// xT[n][t] = x[t][n] for a batch of ntr traces / rows (TP = padded count, pad lanes zero) + the largest |sample| of each into the plan's
// SCR_SPM block, where spectral_run looks for them (the noise floor of the transforms scales with it).  One pass over the batch: a pass of its
// own over the transposed copy for the maxima was 26 us at the head of the chain of a 499 x 16501 call.
template <typename TIn>
static int spectral_transpose(tspws_hip_plan *p, const TIn *d_x, size_t ld, unsigned ntr, TIn *xT, unsigned TP, hipStream_t st)
{
	const unsigned nblk = TP / 64;
	void *v;
	const unsigned np = (p->N + 64 * SPEC_TR_TILES - 1) / (64 * SPEC_TR_TILES);
	if (int rc = scratch(p, SCR_SPM, (size_t)TP * (np + 1) * sizeof(unsigned), &v)) return rc;
	unsigned *amax = (unsigned *)v, *pmax = amax + TP;
	hipLaunchKernelGGL((k_spec_transpose_rows<TIn>), dim3(np, nblk), dim3(256), 0, st, d_x, ld, ntr, p->N, TP, xT, pmax);
	hipLaunchKernelGGL(k_spec_rowmax, dim3(nblk), dim3(1024), 0, st, (const unsigned *)pmax, np, TP, amax);
	HIP_TRY(hipGetLastError());
	return 0;
}
int tspws_spectral_transpose_f32(tspws_hip_plan *p, const float *d_x, size_t ld, unsigned ntr, float *xT, unsigned TP, hipStream_t st)
{ return spectral_transpose<float>(p, d_x, ld, ntr, xT, TP, st); }
int tspws_spectral_transpose_f64(tspws_hip_plan *p, const double *d_x, size_t ld, unsigned ntr, double *xT, unsigned TP, hipStream_t st)
{ return spectral_transpose<double>(p, d_x, ld, ntr, xT, TP, st); }

template <typename TIn>
static int spectral_rows(tspws_hip_plan *p, SpecDecomp *dc, const TIn *d_x, size_t ld, unsigned ntr, unsigned tps, const SpecRowsOut &ro, hipStream_t st,
                         hipEvent_t after_transposition)
{
	const unsigned nblk = (ntr + 63) / 64, TP = nblk * 64, ncol = (ntr + tps - 1) / tps;
	void *v;
	int rc;
	if ((rc = scratch(p, SCR_XT, (size_t)p->N * TP * sizeof(TIn), &v))) return rc;
	TIn *xT = (TIn *)v;
	if ((rc = spectral_transpose<TIn>(p, d_x, ld, ntr, xT, TP, st))) return rc;
	if (after_transposition) HIP_TRY(hipEventRecord(after_transposition, st)); // (the rows themselves are not read again by this chain)
	return spectral_run<TIn>(p, dc, (const TIn *)xT, TP, ntr, nullptr, nullptr, 0, nullptr, st, &ro, tps, ncol);
}

int tspws_spectral_rows_f64(tspws_hip_plan *p, SpecDecomp *dc, const double *d_x, size_t ld, unsigned ntr, unsigned tps, const FuseOut &fz, hipStream_t st,
                            hipEvent_t after_transposition)
{
	SpecRowsOut ro;
	memset(&ro, 0, sizeof ro);
	const FuseFinal &f = fz.fin;
	if (f.OUT) { ro.OUT = f.OUT; ro.out_stride = f.out_stride; ro.Mv = f.Mv; ro.M = f.M; ro.K = f.K; ro.wu = f.wu; ro.mode = f.mode; ro.keep = f.keep_slice; ro.keepST = f.keepST; }
	else { ro.accST = fz.accST; ro.accPS = fz.accPS; ro.stride = fz.stride; ro.keep = -1; }
	return spectral_rows<double>(p, dc, d_x, ld, ntr, tps, ro, st, after_transposition);
}

int tspws_spectral_run_f32(tspws_hip_plan *p, SpecDecomp *dc, const float *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st)
{
	return spectral_run<float>(p, dc, xT, TP, ntr, ST, PS, stride, Y, st);
}
int tspws_spectral_run_f64(tspws_hip_plan *p, SpecDecomp *dc, const double *xT, unsigned TP, unsigned ntr, double2 *ST, double2 *PS, size_t stride, double2 *Y, hipStream_t st)
{
	return spectral_run<double>(p, dc, xT, TP, ntr, ST, PS, stride, Y, st);
}
