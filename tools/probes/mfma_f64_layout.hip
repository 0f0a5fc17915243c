// mfma_f64_layout.hip -- which element of D = A B a lane's four result registers of v_mfma_f64_16x16x4_f64 hold (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout tools/probes/mfma_f64_layout.hip && ./mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out)
{
	const unsigned lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
	// A[i][k] = 100 i + k  (lane = i + 16 k assumed), B[k][j] = (k == 2) ? j + 1 : 0  ->  D[i][j] = (100 i + 2) (j + 1) if the A / B layouts hold
	const double a = 100.0 * li + lk, b = lk == 2 ? (double)(li + 1) : 0.0;
	v4d c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	for (int q = 0; q < 4; q++) out[lane * 4 + q] = c[q];
}
int main()
{
	double *d, h[256];
	hipMalloc(&d, sizeof h);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	for (int lane = 0; lane < 64; lane += 5) {
		printf("lane %2d (lane%%16 = %2d, lane/16 = %d):", lane, lane & 15, lane >> 4);
		for (int q = 0; q < 4; q++) { const double v = h[lane * 4 + q]; const int j = lane & 15; const double i = (v / (j + 1) - 2) / 100; printf("  reg %d = %8.0f -> row i = %4.1f", q, v, i); }
		printf("\n");
	}
	return 0;
}
