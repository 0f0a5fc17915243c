// read_ceiling.hip -- what a pure READ of the north-star trace matrix (10 000 x 131 072 float32 = 5.24 GB) can reach on this box:
// the anchor for k_partial's achieved rate.  No FP64 accumulate: a lane XORs the words it loads (one integer op per
// dword, so the loads cannot be elided) and the block writes 4 bytes per thread at the end.
//   hipcc --offload-arch=gfx950 -O3 -o read_ceiling tools/read_ceiling.hip && ./read_ceiling
// Forms: the launch geometry of k_partial (one workgroup per CU and launch, 5 launches of 128 column blocks x 2 groups, a thread walks
// 1000 rows of its 16-byte column), and flat grid-stride reads of the whole buffer with 4 .. 16 loads in flight, plain / non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

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
template <int U, bool NT> static void launch_cols1() { hipLaunchKernelGGL((k_cols<U, NT>), dim3(128, 10), dim3(256), 0, 0, g_x, N, 1000u, g_out); }
template <int U, bool NT, int BLK> static void launch_flat() { hipLaunchKernelGGL((k_flat<U, NT>), dim3(BLK), dim3(256), 0, 0, (const v4u *)g_x, N * MTR / 4, g_out); }

int main()
{
	hipMalloc(&g_x, N * MTR * 4); hipMalloc(&g_out, (size_t)1 << 24);
	hipMemset(g_x, 0x3c, N * MTR * 4);
	const double gb = 4.0 * N * MTR / 1e9;
#define RUN(name, fn) do { const float ms = timed(fn, 20); printf("%-52s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms * 1e3, gb / ms * 1e3, gb / ms / 8.0); } while (0)
	RUN("columns, 5 launches of 256 workgroups, 8 loads nt", (launch_cols5<8, true>));
	RUN("columns, 5 launches of 256 workgroups, 8 loads plain", (launch_cols5<8, false>));
	RUN("columns, 5 launches of 256 workgroups, 16 loads nt", (launch_cols5<16, true>));
	RUN("columns, 1 launch of 1280 workgroups, 8 loads nt", (launch_cols1<8, true>));
	RUN("flat grid-stride, 256 blocks, 8 loads nt", (launch_flat<8, true, 256>));
	RUN("flat grid-stride, 1024 blocks, 8 loads nt", (launch_flat<8, true, 1024>));
	RUN("flat grid-stride, 2048 blocks, 8 loads nt", (launch_flat<8, true, 2048>));
	RUN("flat grid-stride, 2048 blocks, 8 loads plain", (launch_flat<8, false, 2048>));
	RUN("flat grid-stride, 2048 blocks, 16 loads nt", (launch_flat<16, true, 2048>));
	RUN("flat grid-stride, 4096 blocks, 4 loads nt", (launch_flat<4, true, 4096>));
	RUN("flat grid-stride, 8192 blocks, 4 loads nt", (launch_flat<4, true, 8192>));
	return 0;
}
