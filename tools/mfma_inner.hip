// mfma_inner.hip -- how fast can the (phase, pair) products of fwd_mfma.h run when their operands come from LDS?
// One wave = the product loop of k_fwd_mfma on a private LDS image; no global memory in the loop.
//   variant 0: loads of a product, then its instructions (what fwd_mfma.h does)
//   variant 1: software pipelined -- the operands of the next product are read while the current one multiplies
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_inner.hip -o tools/mfma_inner.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TQ, int KQ>
__device__ __forceinline__ void load_ops(double (&A)[TQ + KQ - 1], double (&B)[KQ], const double *ap, const double *bp)
{
#pragma unroll
	for (int k = 0; k < KQ; k++) B[k] = bp[k * 16];
#pragma unroll
	for (int i = 0; i < TQ + KQ - 1; i++) A[i] = ap[4 * i];
}
template <int TQ, int KQ>
__device__ __forceinline__ void mult(double (&C)[TQ], const double (&A)[TQ + KQ - 1], const double (&B)[KQ])
{
#pragma unroll
	for (int k = 0; k < KQ; k++)
#pragma unroll
		for (int a = 0; a < TQ; a++) C[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[a + k], B[k], C[a], 0, 0, 0);
}

template <int VARIANT, int TQ, int KQ0, int KQ1>
__global__ void __launch_bounds__(256) k(double *out, int iters, unsigned P, unsigned Pu)
{
	extern __shared__ double smem[];
	const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const unsigned l_hi = lane >> 4, l_blk = (lane >> 2) & 3, l_lo = lane & 3;
	double *Bl = smem;                 // 4 phases x (KQ0 + KQ1) x 16
	double *img = smem + 2048 + wv * 4 * Pu;
	for (unsigned i = tid; i < 2048 + 16 * Pu + 64; i += 256) smem[i] = 1e-3 * (i % 97);
	__syncthreads();
	const double *ap0 = img + l_blk * Pu + l_lo + l_hi + 1, *ap1 = img + l_blk * Pu + l_lo + l_hi;
	const double *b0 = Bl + l_hi * 4 + l_lo, *b1 = b0 + 4 * KQ0 * 16;
	double C0[TQ], C1[TQ];
#pragma unroll
	for (int a = 0; a < TQ; a++) { C0[a] = 0; C1[a] = 0; }
	if (VARIANT == 0) {
		for (int it = 0; it < iters; it++) {
			for (unsigned ph = 0; ph < 4; ph++) {
				double A0[TQ + KQ0 - 1], B0[KQ0], A1[TQ + KQ1 - 1], B1[KQ1];
				const unsigned sh = (unsigned)it & 3; // the address changes every iteration: no hoisting of the reads
				load_ops<TQ, KQ0>(A0, B0, ap0 + ph * P + sh, b0 + ph * KQ0 * 16 + sh * 256);
				mult<TQ, KQ0>(C0, A0, B0);
				load_ops<TQ, KQ1>(A1, B1, ap1 + ph * P + sh, b1 + ph * KQ1 * 16 + sh * 256);
				mult<TQ, KQ1>(C1, A1, B1);
			}
		}
	} else {
		double A0[TQ + KQ0 - 1], B0[KQ0], A1[TQ + KQ1 - 1], B1[KQ1];
		load_ops<TQ, KQ0>(A0, B0, ap0, b0);
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (unsigned ph = 0; ph < 4; ph++) {
				const unsigned sh = (unsigned)it & 3;
				load_ops<TQ, KQ1>(A1, B1, ap1 + ph * P + sh, b1 + ph * KQ1 * 16 + sh * 256);
				mult<TQ, KQ0>(C0, A0, B0);
				const unsigned nph = (ph + 1) & 3, nsh = (unsigned)(it + (ph == 3)) & 3;
				load_ops<TQ, KQ0>(A0, B0, ap0 + nph * P + nsh, b0 + nph * KQ0 * 16 + nsh * 256);
				mult<TQ, KQ1>(C1, A1, B1);
			}
		}
	}
	double s = 0;
#pragma unroll
	for (int a = 0; a < TQ; a++) s += C0[a] + C1[a];
	out[blockIdx.x * 256 + tid] = s;
}

template <int V, int TQ, int KQ0, int KQ1> void run(double *out)
{
	const unsigned P = 80, Pu = 328;
	const size_t lds = (2048 + 16 * Pu + 64) * 8;
	(void)hipFuncSetAttribute((const void *)k<V, TQ, KQ0, KQ1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
	for (int blocks : {256, 512, 768, 1024}) {
		const int iters = 400;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL((k<V, TQ, KQ0, KQ1>), dim3(blocks), dim3(256), lds, 0, out, iters, P, Pu);
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL((k<V, TQ, KQ0, KQ1>), dim3(blocks), dim3(256), lds, 0, out, iters, P, Pu);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		const double mf = (double)blocks * 4 * iters * 4 * TQ * (KQ0 + KQ1);
		printf("variant %d TQ %d KQ %d+%d  waves/SIMD %.2f: %.3f ms  %.1f TFLOP/s  (%.1f%% of 78.6)\n", V, TQ, KQ0, KQ1, blocks / 256.0, ms, mf * 512 / ms / 1e9,
		       mf * 512 / ms / 1e9 / 78.6 * 100);
	}
}
int main()
{
	double *out; (void)hipMalloc(&out, 256 * 2048 * 8);
	run<0, 8, 4, 5>(out);
	run<1, 8, 4, 5>(out);
	run<0, 4, 4, 5>(out);
	run<1, 4, 4, 5>(out);
	run<1, 8, 3, 5>(out);
	return 0;
}
