// corun_probe.hip -- what do a dense FP64 FMA kernel and an HBM-streaming kernel cost each other when they share the GPU?
// (round 4: the pipelined masked-replica call runs the forward transforms beside the streaming pass; each slowed ~1.8x)
//   A: k_fma   256 workgroups x 256 threads (one wave per SIMD), register-only v_fma_f64 chains, reports shader cycles per
//              wave (s_memtime domain = shader clock) AND wall time -> cycles / wall time = the clock the kernel ran at
//   B: k_read  256 workgroups x 256 threads, eight 16-byte non-temporal loads in flight per lane over a 4-GiB buffer
// timed: A alone, B alone, A beside B (two streams).
//   hipcc --offload-arch=gfx950 -O3 -o corun_probe corun_probe.hip && ./corun_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_fma(double *out, const double *in, int iters, unsigned long long *cyc)
{
	double acc[16], x[8], t[2];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
	for (int i = 0; i < 8; i++) x[i] = in[threadIdx.x + 256 * i];
	t[0] = in[threadIdx.x + 4096]; t[1] = in[threadIdx.x + 8192];
	const unsigned long long c0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int rep = 0; rep < 4; rep++)
#pragma unroll
			for (int i = 0; i < 16; i++) acc[i] = fma(x[i & 7], t[i & 1], acc[i]);
	}
	const unsigned long long c1 = __builtin_readcyclecounter();
	double s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_read(const float *__restrict__ x, size_t rows, size_t ld, double *out)
{
	// workgroup b owns columns [b * 1024, (b + 1) * 1024) of every row (ld = gridDim.x * 1024 floats)
	const float *src = x + (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	for (size_t t = 0; t + 8 <= rows; t += 8) {
		v4f v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (t + j) * ld));
#pragma unroll
		for (int j = 0; j < 8; j++) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main()
{
	const size_t ld = 256 * 1024, rows = 4096; // 4 GiB
	float *x; double *o1, *o2, *in; unsigned long long *cyc;
	CK(hipMalloc(&x, rows * ld * 4)); CK(hipMemset(x, 0, rows * ld * 4));
	CK(hipMalloc(&o1, 256 * 256 * 8)); CK(hipMalloc(&o2, 256 * 256 * 8)); CK(hipMalloc(&in, 16384 * 8)); CK(hipMemset(in, 0, 16384 * 8));
	CK(hipMalloc(&cyc, 1024 * 8));
	hipStream_t A, B;
	CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
	hipEvent_t a0, a1, b0, b1;
	CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
	const int iters = 12000; // ~64 FMAs x 12000 x 4.x cycles ~ 3.3 M cycles ~ 1.5 ms
	std::vector<unsigned long long> h(1024);
	for (int mode = 0; mode < 3; mode++) { // 0: A alone, 1: B alone, 2: both
		for (int rep = 0; rep < 4; rep++) {
			if (mode != 1) { CK(hipEventRecord(a0, A)); hipLaunchKernelGGL(k_fma, dim3(256), dim3(256), 0, A, o1, in, iters, cyc); CK(hipEventRecord(a1, A)); }
			if (mode != 0) { CK(hipEventRecord(b0, B)); hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, B, x, rows, ld, o2); CK(hipEventRecord(b1, B)); }
			CK(hipDeviceSynchronize());
			float ta = 0, tb = 0;
			if (mode != 1) CK(hipEventElapsedTime(&ta, a0, a1));
			if (mode != 0) CK(hipEventElapsedTime(&tb, b0, b1));
			double mc = 0;
			if (mode != 1) { CK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost)); for (auto v : h) mc += (double)v; mc /= 1024; }
			if (rep) printf("mode %d  fma %.3f ms (%.0f cycles/wave -> %.2f GHz, %.1f TFLOP/s)   read %.3f ms (%.2f TB/s)\n", mode, ta, mc, ta > 0 ? mc / (ta * 1e6) : 0.0,
			                ta > 0 ? 2.0 * 64 * iters * 64 * 1024 / (ta * 1e-3) / 1e12 : 0.0, tb, tb > 0 ? rows * ld * 4.0 / (tb * 1e-3) / 1e12 : 0.0);
		}
	}
	return 0;
}
