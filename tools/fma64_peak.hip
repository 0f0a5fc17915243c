// fma64_peak.hip -- what the FP64 vector pipe of this GPU sustains, by occupancy and operand kind:
//   mode 0: acc = fma(acc, a, b) with a, b wave-uniform (SGPR operands)      -- 16 independent chains per lane
//   mode 1: acc[r] = fma(x[r], t, acc[r]) with x, t, acc all in VGPRs         -- the shape of the FIR inner loop
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, const double *in, double a, double b, int iters)
{
	double acc[16], x[8];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
	for (int i = 0; i < 8; i++) x[i] = in[threadIdx.x + 256 * i];
	double t0 = in[threadIdx.x + 4096], t1 = in[threadIdx.x + 8192];
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) {
#pragma unroll
			for (int i = 0; i < 16; i++) acc[i] = fma(acc[i], a, b);
		} else {
#pragma unroll
			for (int r = 0; r < 8; r++) { acc[2 * r] = fma(x[r], t0, acc[2 * r]); acc[2 * r + 1] = fma(x[r], t1, acc[2 * r + 1]); }
			t0 += a; t1 += b; // keep the operands live / changing (2 extra VALU per 16 FMAs)
		}
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(double *out, const double *in)
{
	for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
		const int iters = 4096;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 0.999, 1e-3, iters);
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 0.999, 1e-3, iters);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		printf("mode %d  waves/SIMD %4.1f: %.3f ms  %.1f TFLOP/s (fp64 fma)\n", MODE, blocks / 256.0, ms, 2.0 * 16 * iters * 256.0 * blocks / ms / 1e9);
	}
}
int main()
{
	double *out, *in; (void)hipMalloc(&out, 256 * 8192 * 8); (void)hipMalloc(&in, 16384 * 8); (void)hipMemset(in, 0, 16384 * 8);
	run<0>(out, in);
	run<1>(out, in);
	return 0;
}
