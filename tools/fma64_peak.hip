// fma64_peak.hip -- what the FP64 vector pipe of this GPU sustains: 16 independent v_fma_f64 chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(double *out, double a, double b, int iters)
{
	double acc[16];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = fma(acc[i], a, b);
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
	double *out; (void)hipMalloc(&out, 256 * 8192 * 8);
	for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
		const int iters = 4096;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 0.999, 1e-3, iters);
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 0.999, 1e-3, iters);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		printf("blocks %5d: %.3f ms  %.1f TFLOP/s (fp64 fma)\n", blocks, ms, 2.0 * 16 * iters * 256.0 * blocks / ms / 1e9);
	}
	return 0;
}
