// coissue.hip -- do an FP64 VALU stream and an FP64 MFMA stream from different waves share a SIMD without slowing each other?
// build: hipcc -O3 --offload-arch=gfx950 tools/coissue.hip -o tools/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_valu(double *out, const double *in, int iters)
{
	double acc[16], x[8];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
	for (int i = 0; i < 8; i++) x[i] = in[threadIdx.x + 256 * i];
	double t0 = in[threadIdx.x + 4096], t1 = in[threadIdx.x + 8192];
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 8; r++) { acc[2 * r] = fma(x[r], t0, acc[2 * r]); acc[2 * r + 1] = fma(x[r], t1, acc[2 * r + 1]); }
		t0 += 1e-9; t1 += 1e-9;
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_mfma(double *out, const double *in, int iters)
{
	double a = in[threadIdx.x], b = in[threadIdx.x + 256], acc[16];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = 0;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
	double *out, *in; (void)hipMalloc(&out, 256 * 8192 * 8 * 2); (void)hipMalloc(&in, 16384 * 8); (void)hipMemset(in, 0, 16384 * 8);
	hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int wv : {1, 2}) {
		const int blocks = 256 * wv, itv = 8192 * 4, itm = 8192 * 2;
		auto timeit = [&](bool A, bool B) {
			(void)hipDeviceSynchronize();
			(void)hipEventRecord(e0, 0);
			(void)hipStreamWaitEvent(s1, e0, 0); (void)hipStreamWaitEvent(s2, e0, 0);
			if (A) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, s1, out, in, itv);
			if (B) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s2, out + 256 * 8192, in, itm);
			(void)hipDeviceSynchronize();
			(void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
		};
		timeit(true, true);
		const float a = timeit(true, false), b = timeit(false, true), ab = timeit(true, true);
		const double fv = 2.0 * 16 * itv * 256.0 * blocks, fm = 512.0 * 16 * itm * 4.0 * blocks;
		printf("waves/SIMD each %d: VALU alone %.3f ms (%.1f TF)  MFMA alone %.3f ms (%.1f TF)  both %.3f ms (VALU %.1f + MFMA %.1f TF)\n", wv, a, fv / a / 1e9, b,
		       fm / b / 1e9, ab, fv / ab / 1e9, fm / ab / 1e9);
	}
	return 0;
}
