// mfma64_peak.hip -- what the FP64 matrix instructions of this GPU sustain, by occupancy:
//   mode 0: v_mfma_f64_16x16x4_f64  (1024 MACs / instr, 4 result doubles per lane), 8 independent accumulators
//   mode 1: v_mfma_f64_4x4x4_4b_f64 (4 blocks x 64 MACs / instr, 1 result double per lane), 16 independent accumulators
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma64_peak.hip -o /tmp/mfma64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, const double *in, int iters)
{
	double a = in[threadIdx.x], b = in[threadIdx.x + 256];
	double s = 0;
	if (MODE == 0) {
		v4d acc[8];
#pragma unroll
		for (int i = 0; i < 8; i++) acc[i] = (v4d){0, 0, 0, 0};
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		}
#pragma unroll
		for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
	} else {
		double acc[16];
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = 0;
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
		}
#pragma unroll
		for (int i = 0; i < 16; i++) s += acc[i];
	}
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(double *out, const double *in)
{
	for (int blocks : {256, 512, 1024, 2048}) {
		const int iters = 2048;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		const double macs_per_instr = MODE == 0 ? 1024.0 : 256.0, n_instr = (MODE == 0 ? 8.0 : 16.0) * iters;
		const double waves = blocks * 4.0;
		printf("mode %d  waves/SIMD %4.1f: %.3f ms  %.1f TFLOP/s  (%.1f cycles/instr/SIMD at 2.4 GHz)\n", MODE, blocks / 256.0, ms,
		       2.0 * macs_per_instr * n_instr * waves / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_instr * waves / 1024.0));
	}
}
int main()
{
	double *out, *in; (void)hipMalloc(&out, 256 * 8192 * 8); (void)hipMalloc(&in, 16384 * 8); (void)hipMemset(in, 0, 16384 * 8);
	run<0>(out, in);
	run<1>(out, in);
	return 0;
}
