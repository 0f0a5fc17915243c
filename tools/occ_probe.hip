// occ_probe.hip -- how many workgroups per CU does the runtime grant as a function of dynamic LDS and block size?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k256(double *o) { extern __shared__ double s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
__global__ void __launch_bounds__(128) k128(double *o) { extern __shared__ double s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[127 - threadIdx.x]; }
int main()
{
	hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
	printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  regsPerMultiprocessor %d  maxThreadsPerMultiProcessor %d\n", p.sharedMemPerBlock,
	       p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor);
	(void)hipFuncSetAttribute((const void *)k256, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void *)k128, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	for (size_t kb : {8, 12, 16, 18, 20, 21, 24, 32, 40, 53, 56, 64, 70, 80, 81}) {
		int a = -1, b = -1;
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k256, 256, kb * 1024);
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k128, 128, kb * 1024);
		printf("LDS %3zu KB: %d blocks of 256, %d blocks of 128\n", kb, a, b);
	}
	return 0;
}
