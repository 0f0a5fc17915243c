// fma64_issue.hip -- how v_fma_f64 issues on gfx950: shader cycles per wave-instruction measured INSIDE the kernel
// (s_memtime), by waves per SIMD and by the shape of the stream, so that DVFS does not blur the picture.
//   shape 0: 16 independent accumulators, acc[i] = fma(x[i&7], t[i&1], acc[i])      (the FIR inner loop)
//   shape 1: 32 independent accumulators
//   shape 2: shape 0 with one independent integer VALU op after every 4 FMAs
//   shape 3: 16 accumulators, both multiplicands wave-uniform (SGPR)
//   shape 4: shape 0 with one ds_read_b64 (LDS) per 8 FMAs, result consumed 16 FMAs later
// build: hipcc --offload-arch=gfx950 -O3 -o fma64_issue fma64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SHAPE>
__global__ void __launch_bounds__(256) k(double *out, const double *in, double a, double b, int iters, unsigned long long *cyc)
{
	__shared__ double lds[2048];
	constexpr int NA = SHAPE == 1 ? 32 : 16;
	double acc[NA], x[8], t[2];
	for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = in[i];
	__syncthreads();
#pragma unroll
	for (int i = 0; i < NA; i++) acc[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
	for (int i = 0; i < 8; i++) x[i] = in[threadIdx.x + 256 * i];
	t[0] = in[threadIdx.x + 4096]; t[1] = in[threadIdx.x + 8192];
	unsigned iv = threadIdx.x;
	const unsigned long long c0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int rep = 0; rep < 4; rep++) {
			if (SHAPE == 3) {
#pragma unroll
				for (int i = 0; i < NA; i++) acc[i] = fma(acc[i], a, b);
			} else if (SHAPE == 4) {
				double nx0 = lds[(threadIdx.x + rep * 64) & 2047], nx1 = lds[(threadIdx.x + rep * 64 + 512) & 2047];
#pragma unroll
				for (int i = 0; i < NA; i++) acc[i] = fma(x[i & 7], t[i & 1], acc[i]);
				x[rep] = nx0; x[rep + 4] = nx1;
			} else {
#pragma unroll
				for (int i = 0; i < NA; i++) {
					acc[i] = fma(x[i & 7], t[i & 1], acc[i]);
					if (SHAPE == 2 && (i & 3) == 3) iv = iv * 3 + 1;
				}
			}
		}
	}
	const unsigned long long c1 = __builtin_readcyclecounter();
	double s = (double)iv;
#pragma unroll
	for (int i = 0; i < NA; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int SHAPE> void run(double *out, const double *in, unsigned long long *cyc, const char *name)
{
	constexpr int NA = SHAPE == 1 ? 32 : 16;
	for (int wps : {1, 2, 3, 4, 6, 8}) {
		const int blocks = 256 * wps, iters = 2048;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, in, 0.999, 1e-3, iters, cyc);
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, in, 0.999, 1e-3, iters, cyc);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h(blocks * 4);
		(void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
		double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
		const double nf = (double)iters * 4 * NA;
		printf("%-34s waves/SIMD %d: %7.3f ms  %5.1f TFLOP/s   %.2f cycles per wave-FMA (per wave), %.2f per SIMD; clock %.2f GHz\n", name, wps, ms,
		       2.0 * nf * 64 * 4 * blocks / ms / 1e9, mean / nf, mean / nf / wps, mean / (ms * 1e6));
	}
}
int main()
{
	double *out, *in; unsigned long long *cyc;
	(void)hipMalloc(&out, 256 * 8 * 256 * 8); (void)hipMalloc(&in, 16384 * 8); (void)hipMemset(in, 0, 16384 * 8); (void)hipMalloc(&cyc, 256 * 8 * 4 * 8);
	run<0>(out, in, cyc, "16 acc, VGPR operands");
	run<1>(out, in, cyc, "32 acc, VGPR operands");
	run<2>(out, in, cyc, "16 acc + int VALU every 4 FMAs");
	run<3>(out, in, cyc, "16 acc, SGPR multiplicands");
	run<4>(out, in, cyc, "16 acc + 2 LDS reads per 16 FMAs");
	return 0;
}
