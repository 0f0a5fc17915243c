// stream_sweep.hip -- micro-benchmark behind the shape of k_partial (the HBM-streaming kernel):
// FP64 column sums of a float32 [mtr][N] matrix, sweeping loads in flight, cache policy, block count.
//   hipcc --offload-arch=gfx950 -O3 -o stream_sweep tools/stream_sweep.hip && ./stream_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, int COLS>
__global__ void __launch_bounds__(256) k(const float *__restrict__ x, size_t ld, unsigned count, double *__restrict__ out, size_t ldo)
{
	const size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4 * COLS;
	const float *src = x + (size_t)blockIdx.y * count * ld + col;
	double a[COLS][4] = {};
	for (unsigned t = 0; t + UNROLL <= count; t += UNROLL) {
		v4f v[UNROLL][COLS];
#pragma unroll
		for (int j = 0; j < UNROLL; j++)
#pragma unroll
			for (int c = 0; c < COLS; c++) {
				const v4f *p = (const v4f *)(src + (size_t)(t + j) * ld) + c;
				v[j][c] = NT ? __builtin_nontemporal_load(p) : *p;
			}
#pragma unroll
		for (int j = 0; j < UNROLL; j++)
#pragma unroll
			for (int c = 0; c < COLS; c++) { a[c][0] += (double)v[j][c].x; a[c][1] += (double)v[j][c].y; a[c][2] += (double)v[j][c].z; a[c][3] += (double)v[j][c].w; }
	}
	double *dst = out + (size_t)blockIdx.y * ldo + col;
#pragma unroll
	for (int c = 0; c < COLS; c++) { ((double2 *)dst)[2 * c] = make_double2(a[c][0], a[c][1]); ((double2 *)dst)[2 * c + 1] = make_double2(a[c][2], a[c][3]); }
}

template <int U, bool NT, int C>
static void run(const char *name, const float *x, size_t N, size_t mtr, unsigned chunks, double *out)
{
	const unsigned count = (unsigned)(mtr / chunks);
	dim3 grid((unsigned)(N / (1024 * C)), chunks);
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<U, NT, C>), grid, dim3(256), 0, 0, x, N, count, out, N);
	hipEventRecord(a);
	const int it = 10;
	for (int i = 0; i < it; i++) hipLaunchKernelGGL((k<U, NT, C>), grid, dim3(256), 0, 0, x, N, count, out, N);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
	printf("%-28s chunks=%4u blocks=%6u  %8.1f us  %7.1f GB/s\n", name, chunks, grid.x * grid.y, ms * 1e3, 4.0 * N * count * chunks / ms / 1e6);
}

int main()
{
	const size_t N = 131072, mtr = 10000;
	float *x; double *out;
	hipMalloc(&x, N * mtr * 4); hipMalloc(&out, N * 8 * 512);
	hipMemset(x, 0x3c, N * mtr * 4);
	for (unsigned chunks : {10u, 20u, 40u, 80u, 160u}) {
		run<8, true, 1>("unroll8 nt cols1", x, N, mtr, chunks, out);
		run<8, false, 1>("unroll8 plain cols1", x, N, mtr, chunks, out);
		run<16, true, 1>("unroll16 nt cols1", x, N, mtr, chunks, out);
		run<4, true, 1>("unroll4 nt cols1", x, N, mtr, chunks, out);
		run<8, true, 2>("unroll8 nt cols2", x, N, mtr, chunks, out);
		run<4, true, 2>("unroll4 nt cols2", x, N, mtr, chunks, out);
	}
	return 0;
}
