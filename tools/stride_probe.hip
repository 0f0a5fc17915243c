// stride_probe.hip -- cost of a workgroup's strided window load: 4 waves, wave w loads rows w, w+4, ... (22 rows per lane,
// 512 B per row and wave) at a row stride of S bytes, from 10 "traces" of 4 MiB each.  Reports shader cycles until the
// loads are ISSUED and until they have all RETURNED, per window, with 2 workgroups per CU doing the same thing.
// build: hipcc --offload-arch=gfx950 -O3 -o stride_probe stride_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(const double *x, size_t row_stride, size_t trace_stride, unsigned ntr, unsigned span_rows,
                                        unsigned long long *out, double *sink)
{
	const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	// window base: spread the workgroups over the trace like the chunk / block decomposition does
	const size_t base = ((size_t)blockIdx.x * 977u % span_rows) * row_stride / 8 + (blockIdx.x & 7) * 64;
	unsigned long long t_issue = 0, t_done = 0;
	double acc = 0;
	for (unsigned t = 0; t < ntr; t++) {
		const double *src = x + t * (trace_stride / 8) + base + (size_t)wv * (row_stride / 8) + lane;
		double v[22];
		const unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll
		for (int i = 0; i < 22; i++) v[i] = src[(size_t)i * 4 * (row_stride / 8)];
		const unsigned long long c1 = __builtin_readcyclecounter();
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const unsigned long long c2 = __builtin_readcyclecounter();
#pragma unroll
		for (int i = 0; i < 22; i++) acc += v[i];
		t_issue += c1 - c0; t_done += c2 - c0;
		__syncthreads();
	}
	if (acc == 12345.678) sink[0] = acc;
	if (lane == 0) { atomicAdd(&out[0], t_issue); atomicAdd(&out[1], t_done); atomicAdd(&out[2], (unsigned long long)ntr); }
}
int main()
{
	const size_t trace = 4u << 20, ntr = 10, bytes = trace * ntr + (64u << 20);
	double *x, *sink; unsigned long long *out;
	(void)hipMalloc(&x, bytes); (void)hipMemset(x, 0, bytes); (void)hipMalloc(&sink, 8); (void)hipMalloc(&out, 24);
	for (size_t stride : {4096ul, 4096ul, 4608ul, 6144ul, 8192ul, 8704ul, 12288ul, 16384ul, 17408ul, 16384ul + 4096ul}) {
		for (int rep = 0; rep < 2; rep++) {
			(void)hipMemset(out, 0, 24);
			const unsigned span_rows = (unsigned)((trace - 88 * stride > 0 ? trace - 88 * stride : stride) / stride);
			hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, x, stride, trace, (unsigned)ntr, span_rows ? span_rows : 1, out, sink);
			(void)hipDeviceSynchronize();
			unsigned long long h[3]; (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
			if (rep) printf("row stride %6zu B: issue %7.0f cycles, all returned %7.0f cycles per 45 KB window (2 workgroups per CU)\n", stride, (double)h[0] / h[2], (double)h[1] / h[2]);
		}
	}
	return 0;
}
