// xstream_probe.hip -- what does a cross-stream dependency cost between two kernels?
//   A: K1 (spins ~100 us, stamps its end)                     B: [wait] K2 (stamps its start)
// mechanisms: same stream (reference), event record + hipStreamWaitEvent (timing disabled, with / without the system fence),
// hipStreamWriteValue32 + hipStreamWaitValue32 on plain device memory.  Also: what an event record in FRONT of the next kernel
// of the same stream costs that kernel (K1, record, K3 on A).
//   hipcc --offload-arch=gfx950 -O2 -o xstream_probe xstream_probe.hip && ./xstream_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(unsigned long long *stamp, int slot, unsigned long long ticks)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0 && blockIdx.x == 0) stamp[slot] = wall_clock64();
}
__global__ void k_stamp(unsigned long long *stamp, int slot)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) stamp[slot] = wall_clock64();
}

int main()
{
	unsigned long long *stamp; unsigned *flag;
	CK(hipMalloc(&stamp, 64)); CK(hipMalloc(&flag, 8)); CK(hipMemset(flag, 0, 8));
	hipStream_t A, B;
	CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
	hipEvent_t e1, e2;
	CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
	CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming | hipEventDisableSystemFence));
	const double tick_us = 0.01; // wall_clock64: 100 MHz
	unsigned seq = 0;
	hipEvent_t e3, e4;
	CK(hipEventCreate(&e3)); CK(hipEventCreate(&e4));
	for (int mode = 0; mode < 8; mode++) {
		double sum = 0, sum3 = 0; int n = 0;
		for (int rep = 0; rep < 12; rep++) {
			CK(hipMemsetAsync(stamp, 0, 64, A)); CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
			seq++;
			// pre-enqueue B's wait for the value mode so that it is parked before K1 ends
			if (mode == 3) { CK(hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu)); hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 1); }
			if (mode == 6) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, nullptr, e2, 0, stamp, 0, 10000ull); // the kernel's own completion signal is the event
			else if (mode == 7) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, e3, e4, 0, stamp, 0, 10000ull); // timing events on the kernel itself
			else
			hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, stamp, 0, 10000ull); // 100 us
			switch (mode) {
			case 0: hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 1); break;                       // same stream
			case 1: CK(hipEventRecord(e1, A)); CK(hipStreamWaitEvent(B, e1, 0)); hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 1);
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 3); break;                       // event (system fence) + K3 behind the record
			case 2: CK(hipEventRecord(e2, A)); CK(hipStreamWaitEvent(B, e2, 0)); hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 1);
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 3); break;                       // event (no system fence)
			case 3: CK(hipStreamWriteValue32(A, flag, seq, 0)); hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 3); break; // write value / parked wait
			case 4: CK(hipStreamWriteValue32(A, flag, seq, 0)); CK(hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 1);
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 3); break;                       // write value / wait enqueued afterwards
			case 6: CK(hipStreamWaitEvent(B, e2, 0)); hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 1);
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 3); break;
			case 7: CK(hipStreamWaitEvent(B, e4, 0)); hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 1);
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 3); break;
			case 5: // join direction: B's kernel ended long ago; A waits for its event in front of K2
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, B, stamp, 2); CK(hipEventRecord(e2, B)); CK(hipStreamWaitEvent(A, e2, 0));
			        hipLaunchKernelGGL(k_stamp, dim3(256), dim3(256), 0, A, stamp, 1); break;
			}
			CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
			unsigned long long h[8]; CK(hipMemcpy(h, stamp, 64, hipMemcpyDeviceToHost));
			if (rep >= 2) { sum += (double)(h[1] - h[0]) * tick_us; if (h[3]) sum3 += (double)(h[3] - h[0]) * tick_us; n++; }
		}
		const char *names[] = {"same stream", "event + wait (system fence)", "event + wait (no system fence)", "write value, wait parked early", "write value, wait enqueued after", "join: wait for an event long since done", "stop event of the kernel launch (no timing)", "start + stop timing events of the launch"};
		printf("%-42s K1 end -> dependent kernel start %6.2f us", names[mode], sum / n);
		if (sum3 > 0) printf("   next kernel of the SAME stream behind the record/write %6.2f us", sum3 / n);
		if (mode == 7) { float ms = 0; (void)hipEventElapsedTime(&ms, e3, e4); printf("   elapsed(start, stop) %.1f us", ms * 1e3); }
		printf("\n");
	}
	return 0;
}
