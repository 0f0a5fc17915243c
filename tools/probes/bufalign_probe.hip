// Does a raw buffer load of 8 / 16 bytes per lane return the right data when the address is only 4- (8-) byte aligned?
// hipcc --offload-arch=gfx950 -O2 -o bufalign_probe bufalign_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, unsigned n, unsigned base_sh, unsigned so_el, unsigned vo_el, float *out)
{
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + base_sh), 0, (n - base_sh) * 4u, 0x00020000);
	const unsigned vb = (2u * threadIdx.x + vo_el) * 4u;
	const auto q = __builtin_amdgcn_raw_buffer_load_b64(rs, vb, so_el * 4u, 0);
	struct F2 { float a, b; };
	const F2 f = __builtin_bit_cast(F2, q);
	out[2 * threadIdx.x] = f.a;
	out[2 * threadIdx.x + 1] = f.b;
}
__global__ void k4(const double *x, unsigned n, unsigned base_sh, unsigned so_el, unsigned vo_el, double *out)
{
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + base_sh), 0, (n - base_sh) * 8u, 0x00020000);
	const unsigned vb = (2u * threadIdx.x + vo_el) * 8u;
	const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, vb, so_el * 8u, 0);
	out[2 * threadIdx.x] = __builtin_bit_cast(double, ((unsigned long long)q[1] << 32) | q[0]);
	out[2 * threadIdx.x + 1] = __builtin_bit_cast(double, ((unsigned long long)q[3] << 32) | q[2]);
}
int main()
{
	const unsigned n = 4096;
	std::vector<float> h(n); std::vector<double> hd(n);
	for (unsigned i = 0; i < n; i++) { h[i] = (float)i; hd[i] = (double)i; }
	float *d, *o; double *dd, *od;
	hipMalloc(&d, n * 4); hipMalloc(&o, 128 * 4); hipMalloc(&dd, n * 8); hipMalloc(&od, 128 * 8);
	hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dd, hd.data(), n * 8, hipMemcpyHostToDevice);
	for (unsigned c = 0; c < 8; c++) {
		const unsigned b = c & 1, s = (c >> 1) & 1, v = (c >> 2) & 1;
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, b, s, v, o);
		float r[128]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
		unsigned bad = 0; for (unsigned i = 0; i < 128; i++) bad += r[i] != (float)(i + b + s + v);
		hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dd, n, b, s, v, od);
		double rd[128]; hipMemcpy(rd, od, sizeof rd, hipMemcpyDeviceToHost);
		unsigned badd = 0; for (unsigned i = 0; i < 128; i++) badd += rd[i] != (double)(i + b + s + v);
		printf("base+%u soffset+%u voffset+%u: float pairs wrong %u (first %g %g), double pairs wrong %u (first %g %g)\n", b, s, v, bad, r[0], r[1], badd, rd[0], rd[1]);
	}
	return 0;
}
