// read_ceiling.hip -- what a pure READ of the north-star trace matrix (10 000 x 131 072 float32 = 5.24 GB) can reach on this box:
// the anchor for k_partial's achieved rate.  No FP64 accumulate: a lane XORs the words it loads (one integer op per
// dword, so the loads cannot be elided) and the block writes 4 bytes per thread at the end.
//   hipcc --offload-arch=gfx950 -O3 -o read_ceiling tools/read_ceiling.hip && ./read_ceiling
// Forms: the launch geometry of k_partial (one workgroup per CU and launch, 5 launches of 128 column blocks x 2 groups, a thread walks
// 1000 rows of its 16-byte column), and flat grid-stride reads of the whole buffer with 4 .. 16 loads in flight, plain / non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT> // segment blockIdx.y starts seg_stride words further on
__global__ void __launch_bounds__(256) k_cols2(const unsigned *__restrict__ x, size_t ld, unsigned count, size_t seg_stride, unsigned *__restrict__ out)
{
	const size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
	const unsigned *src = x + (size_t)blockIdx.y * seg_stride + col;
	v4u a = {0, 0, 0, 0};
	for (unsigned t = 0; t + UNROLL <= count; t += UNROLL) {
		v4u v[UNROLL];
#pragma unroll
		for (int j = 0; j < UNROLL; j++) {
			const v4u *p = (const v4u *)(src + (size_t)(t + j) * ld);
			v[j] = NT ? __builtin_nontemporal_load(p) : *p;
		}
#pragma unroll
		for (int j = 0; j < UNROLL; j++) a ^= v[j];
	}
	out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = a.x ^ a.y ^ a.z ^ a.w;
}

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_cols(const unsigned *__restrict__ x, size_t ld, unsigned count, unsigned *__restrict__ out)
{
	const size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
	const unsigned *src = x + (size_t)blockIdx.y * count * ld + col;
	v4u a = {0, 0, 0, 0};
	for (unsigned t = 0; t + UNROLL <= count; t += UNROLL) {
		v4u v[UNROLL];
#pragma unroll
		for (int j = 0; j < UNROLL; j++) {
			const v4u *p = (const v4u *)(src + (size_t)(t + j) * ld);
			v[j] = NT ? __builtin_nontemporal_load(p) : *p;
		}
#pragma unroll
		for (int j = 0; j < UNROLL; j++) a ^= v[j];
	}
	out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = a.x ^ a.y ^ a.z ^ a.w;
}

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_flat(const v4u *__restrict__ x, size_t n16, unsigned *__restrict__ out)
{
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	v4u a = {0, 0, 0, 0};
	for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
		v4u v[UNROLL];
#pragma unroll
		for (int j = 0; j < UNROLL; j++) v[j] = NT ? __builtin_nontemporal_load(x + i + j * stride) : x[i + j * stride];
#pragma unroll
		for (int j = 0; j < UNROLL; j++) a ^= v[j];
	}
	out[(size_t)blockIdx.x * 256 + threadIdx.x] = a.x ^ a.y ^ a.z ^ a.w;
}

static float timed(void (*launch)(void), int it)
{
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 3; i++) launch();
	hipEventRecord(a);
	for (int i = 0; i < it; i++) launch();
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	return ms / it;
}

static unsigned *g_x, *g_out;
static const size_t N = 131072, MTR = 10000;
template <int U, bool NT> static void launch_cols5()
{ // k_partial's geometry: 5 launches x (128 column blocks x 2 groups of 1000 traces)
	for (int l = 0; l < 5; l++)
		hipLaunchKernelGGL((k_cols<U, NT>), dim3(128, 2), dim3(256), 0, 0, g_x + (size_t)l * 2000 * N, N, 1000u, g_out);
}
// the walk's arithmetic, step by step: MODE 1 = FP64 sums of the 4 columns of a lane (k_partial's loop), 2 = + a run end every RUN rows that adds the
// run's sums to 11 columns' running sums, 3 = + every 30th run end stores the 11 columns (k_rows_walk's flush)
template <int MODE, int RUN>
__global__ void __launch_bounds__(256) k_walkish(const float *__restrict__ x, size_t ld, unsigned count, size_t seg_stride, double *__restrict__ rows, size_t N, unsigned member)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	const size_t col = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
	const float *src = x + (size_t)blockIdx.y * seg_stride + col;
	double P[12][4];
#pragma unroll
	for (int c = 0; c < 12; c++)
#pragma unroll
		for (int k = 0; k < 4; k++) P[c][k] = 0;
	unsigned nrun = 0;
	for (unsigned r0 = 0; r0 < count; r0 += RUN) {
		double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
		const unsigned n = count - r0 < (unsigned)RUN ? count - r0 : (unsigned)RUN;
		for (unsigned t = 0; t + 8 <= n; t += 8) {
			v4f v[8];
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)(r0 + t + j) * ld));
#pragma unroll
			for (int j = 0; j < 8; j++) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
		}
		if (MODE >= 2) {
			const unsigned mb = member ^ (nrun & 1u); // (uniform, opaque to the compiler)
#pragma unroll
			for (int c = 0; c < 12; c++) {
				if ((mb >> c) & 1u) { P[c][0] += a0; P[c][1] += a1; P[c][2] += a2; P[c][3] += a3; }
				if (MODE >= 3 && nrun % 30 == 29 && c < 11) {
					double *rowp = rows + ((size_t)((MODE >= 7 ? 0 : nrun / 30 * 11) + c) + (size_t)blockIdx.y * 60) * N; // (7, 8: the same 11 rows every time)
					if (MODE == 3) { double *dst = rowp + col; *(double2 *)dst = make_double2(P[c][0], P[c][1]); *(double2 *)(dst + 2) = make_double2(P[c][2], P[c][3]); }
					else if (MODE == 4) { // the same bytes, every store instruction of a wave one contiguous KB (values in the wrong places: timing only)
						double *dst = rowp + (col & ~(size_t)255) + (threadIdx.x & 63) * 2;
						*(double2 *)dst = make_double2(P[c][0], P[c][1]); *(double2 *)(dst + 128) = make_double2(P[c][2], P[c][3]);
					} else if (MODE == 5) { // non-temporal stores
						double *dst = rowp + col;
						__builtin_nontemporal_store(P[c][0], dst); __builtin_nontemporal_store(P[c][1], dst + 1); __builtin_nontemporal_store(P[c][2], dst + 2); __builtin_nontemporal_store(P[c][3], dst + 3);
					} else if (MODE == 8) { double *dst = rowp + col; *(double2 *)dst = make_double2(P[c][0], P[c][1]); *(double2 *)(dst + 2) = make_double2(P[c][2], P[c][3]); }
					else { // 6, 7: contiguous KB per instruction, non-temporal
						typedef double v2d __attribute__((ext_vector_type(2)));
						double *dst = rowp + (col & ~(size_t)255) + (threadIdx.x & 63) * 2;
						__builtin_nontemporal_store((v2d){P[c][0], P[c][1]}, (v2d *)dst); __builtin_nontemporal_store((v2d){P[c][2], P[c][3]}, (v2d *)(dst + 128));
					}
#pragma unroll
					for (int k = 0; k < 4; k++) P[c][k] = 0;
				}
			}
		} else { P[0][0] += a0; P[0][1] += a1; P[0][2] += a2; P[0][3] += a3; }
		nrun++;
	}
	double s = 0;
#pragma unroll
	for (int c = 0; c < 12; c++)
#pragma unroll
		for (int k = 0; k < 4; k++) s += P[c][k];
	rows[(size_t)(120 + blockIdx.y) * N + col] = s;
}
static double *g_rows;
template <int MODE, int RUN> static void launch_walkish()
{
	hipLaunchKernelGGL((k_walkish<MODE, RUN>), dim3(128, 2), dim3(256), 0, 0, (const float *)g_x, N, 5000u, (size_t)5000 * N, g_rows, N, 0x7ffu);
}
template <int U, bool NT, int L> static void launch_colsL()
{ // k_rows_walk's geometry cut into L launches: 128 column blocks x 2 segments of 5000 / L rows per launch (L = 1: the walk as it is)
	for (int l = 0; l < L; l++)
		for (int dummy = 0; dummy < 1; dummy++)
			hipLaunchKernelGGL((k_cols2<U, NT>), dim3(128, 2), dim3(256), 0, 0, g_x + (size_t)l * (5000 / L) * N, N, (unsigned)(5000 / L), (size_t)5000 * N, g_out);
}
template <int U, bool NT> static void launch_cols1() { hipLaunchKernelGGL((k_cols<U, NT>), dim3(128, 10), dim3(256), 0, 0, g_x, N, 1000u, g_out); }
template <int U, bool NT, int BLK> static void launch_flat() { hipLaunchKernelGGL((k_flat<U, NT>), dim3(BLK), dim3(256), 0, 0, (const v4u *)g_x, N * MTR / 4, g_out); }

int main()
{
	hipMalloc(&g_x, N * MTR * 4); hipMalloc(&g_out, (size_t)1 << 24);
	hipMemset(g_x, 0x3c, N * MTR * 4);
	hipMalloc(&g_rows, (size_t)128 * N * 8);
	const double gb = 4.0 * N * MTR / 1e9;
#define RUN(name, fn) do { const float ms = timed(fn, 20); printf("%-52s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms * 1e3, gb / ms * 1e3, gb / ms / 8.0); } while (0)
	RUN("columns, 5 launches of 256 workgroups, 8 loads nt", (launch_cols5<8, true>));
	RUN("columns, 5 launches of 256 workgroups, 8 loads plain", (launch_cols5<8, false>));
	RUN("columns, 5 launches of 256 workgroups, 16 loads nt", (launch_cols5<16, true>));
	RUN("columns, 1 launch of 1280 workgroups, 8 loads nt", (launch_cols1<8, true>));
	RUN("walk: 1 launch, 2 segments of 5000 rows, 8 loads nt", (launch_colsL<8, true, 1>));
	RUN("walk + FP64 sums, one run of 5000 rows", (launch_walkish<1, 5000>));
	RUN("walk + FP64 sums, runs of 1000 rows", (launch_walkish<1, 1000>));
	RUN("walk + FP64 sums, runs of 32 rows", (launch_walkish<1, 32>));
	RUN("walk + FP64 sums, runs of 32 rows, 11 columns", (launch_walkish<2, 32>));
	RUN("walk + FP64 sums, runs of 32 rows, 11 columns, stores", (launch_walkish<3, 32>));
	RUN("  the same, a contiguous KB per store instruction", (launch_walkish<4, 32>));
	RUN("  the same, lane-strided, non-temporal stores", (launch_walkish<5, 32>));
	RUN("  the same, contiguous KB, non-temporal", (launch_walkish<6, 32>));
	RUN("  contiguous KB, non-temporal, the SAME 11 rows every time", (launch_walkish<7, 32>));
	RUN("  lane-strided plain stores, the SAME 11 rows every time", (launch_walkish<8, 32>));
	RUN("walk + FP64 sums, runs of 1000 rows, 11 columns", (launch_walkish<2, 1000>));
	RUN("walk: 2 launches, 2 segments of 2500 rows", (launch_colsL<8, true, 2>));
	RUN("walk: 5 launches, 2 segments of 1000 rows", (launch_colsL<8, true, 5>));
	RUN("walk: 10 launches, 2 segments of 500 rows", (launch_colsL<8, true, 10>));
	RUN("walk: 25 launches, 2 segments of 200 rows", (launch_colsL<8, true, 25>));
	RUN("walk: 1 launch, 2 segments of 5000 rows, 8 loads nt", (launch_colsL<8, true, 1>));
	RUN("flat grid-stride, 256 blocks, 8 loads nt", (launch_flat<8, true, 256>));
	RUN("flat grid-stride, 1024 blocks, 8 loads nt", (launch_flat<8, true, 1024>));
	RUN("flat grid-stride, 2048 blocks, 8 loads nt", (launch_flat<8, true, 2048>));
	RUN("flat grid-stride, 2048 blocks, 8 loads plain", (launch_flat<8, false, 2048>));
	RUN("flat grid-stride, 2048 blocks, 16 loads nt", (launch_flat<16, true, 2048>));
	RUN("flat grid-stride, 4096 blocks, 4 loads nt", (launch_flat<4, true, 4096>));
	RUN("flat grid-stride, 8192 blocks, 4 loads nt", (launch_flat<4, true, 8192>));
	return 0;
}
