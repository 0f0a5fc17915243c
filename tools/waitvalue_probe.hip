// waitvalue_probe.hip -- can a kernel that is still RUNNING release work on another stream?
//
// A long-running "producer" kernel finishes a piece of work, publishes it (device-scope fence) and then stores a value to a
// signal word; a second stream holds a hipStreamWaitValue32 on that word in front of a "consumer" kernel that checks the
// data.  Measured: whether the API works on this device for (a) hipMallocSignalMemory words, (b) plain device memory,
// whether the consumer sees the producer's data, and the delay between the producer's store and the consumer's start
// (both stamp the constant-rate clock).  Every wait is rescued by a host-side write after 3 s, so the probe cannot hang.
//
//   hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip && ./waitvalue_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(double *data, size_t n, unsigned *cnt, unsigned *sig, unsigned seq, unsigned long long *stamp, int spin_after)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) data[i] = (double)i * 0.5 + seq;
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned old = atomicAdd(cnt, 1u);
		if (old + 1 == gridDim.x) {
			stamp[0] = wall_clock64();
			__hip_atomic_store(sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	// keep running: the consumer must start while this kernel is still resident
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < (unsigned long long)spin_after) __builtin_amdgcn_s_sleep(32);
	if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2] = wall_clock64();
}

__global__ void consumer(const double *data, size_t n, unsigned seq, unsigned long long *stamp, unsigned *bad)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
	unsigned b = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		if (data[i] != (double)i * 0.5 + seq) b++;
	if (b) atomicAdd(bad, b);
}

static int run(const char *what, unsigned *sig, bool host_visible)
{
	const size_t n = 1 << 22;
	double *data; unsigned *cnt, *bad; unsigned long long *stamp;
	CK(hipMalloc(&data, n * sizeof(double)));
	CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&stamp, 32));
	hipStream_t A, B;
	CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
	for (unsigned seq = 1; seq <= 3; seq++) {
		CK(hipMemsetAsync(cnt, 0, 4, A)); CK(hipMemsetAsync(bad, 0, 4, A)); CK(hipMemsetAsync(stamp, 0, 32, A));
		CK(hipStreamSynchronize(A));
		hipError_t e = hipStreamWaitValue32(B, sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFu);
		if (e != hipSuccess) { printf("%-28s hipStreamWaitValue32 -> %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); return 0; }
		hipLaunchKernelGGL(consumer, dim3(256), dim3(256), 0, B, data, n, seq, stamp, bad);
		// producer spins 100 MHz x 2 ms = 200000 ticks after publishing
		hipLaunchKernelGGL(producer, dim3(256), dim3(256), 0, A, data, n, cnt, sig, seq, stamp, 200000);
		const auto t0 = std::chrono::steady_clock::now();
		bool rescued = false;
		while (hipStreamQuery(B) == hipErrorNotReady) {
			if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) {
				rescued = true;
				if (host_visible) *(volatile unsigned *)sig = seq; else (void)hipMemcpy(sig, &seq, 4, hipMemcpyHostToDevice);
				break;
			}
			std::this_thread::sleep_for(std::chrono::microseconds(200));
		}
		CK(hipStreamSynchronize(B)); CK(hipStreamSynchronize(A));
		unsigned long long st[4]; unsigned hb;
		CK(hipMemcpy(st, stamp, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
		printf("%-28s seq %u: %s, consumer saw %u bad values, store->consumer start %.2f us, consumer started %s the producer ended\n", what, seq,
		       rescued ? "RESCUED BY HOST (wait never fired)" : "wait fired", hb, (double)((long long)st[1] - (long long)st[0]) / 100.0,
		       st[1] < st[2] ? "BEFORE" : "after");
	}
	CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B));
	CK(hipFree(data)); CK(hipFree(cnt)); CK(hipFree(bad)); CK(hipFree(stamp));
	return 0;
}

int main()
{
	int can = 0;
	CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
	printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
	unsigned *sig = nullptr;
	hipError_t e = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
	printf("hipExtMallocWithFlags(8, hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
	if (e == hipSuccess) { *(volatile unsigned long long *)sig = 0; if (run("signal memory", sig, true)) return 1; (void)hipFree(sig); }
	else (void)hipGetLastError();
	e = hipExtMallocWithFlags((void **)&sig, 64, hipDeviceMallocUncached);
	if (e == hipSuccess) { CK(hipMemset(sig, 0, 64)); if (run("uncached device memory", sig, false)) return 1; (void)hipFree(sig); } else (void)hipGetLastError();
	CK(hipMalloc(&sig, 64)); CK(hipMemset(sig, 0, 64));
	if (run("plain device memory", sig, false)) return 1;
	(void)hipFree(sig);
	e = hipHostMalloc((void **)&sig, 64, hipHostMallocCoherent);
	if (e == hipSuccess) { *(volatile unsigned *)sig = 0; if (run("coherent host memory", sig, true)) return 1; (void)hipHostFree(sig); }
	return 0;
}
