// lane_reduce.h -- sums over the 64 lanes of a wave on the VALU (permlane swaps + DPP, no LDS traffic); shared by the forward
// kernels (fwd_poly.h, fwd_lds.h, fwd_tl.h) and the spectral engine (spectral.hip).
#pragma once

// ---- 64-lane reduce-scatter of 16 doubles on the VALU (no LDS traffic) ------------------------------------------
// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / 16-lane rows in ONE VALU op: with A = the
// element the lower partner keeps and B = the element the upper partner keeps, swap(A, B) leaves {own A, partner's A}
// in the lower lanes and {partner's B, own B} in the upper lanes, so A + B is the reduced element on both sides -- no
// selects.  Lane bits 3 and 2 use DPP (row_ror:8 = xor 8, row_half_mirror pairs i <-> 7-i across bit 2), bits 1, 0 are
// plain quad butterflies.  On return every lane holds ONE finished sum: element  8*b5 + 4*b4 + 2*b3 + b2  of v
// (b_k = bit k of the lane id); the four lanes of a quad hold the same value.
__device__ __forceinline__ double dpp_mov_f64(double x, const int ctrl_sel)
{
	const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
	unsigned rl, rh;
	switch (ctrl_sel) {
	case 0: rl = __builtin_amdgcn_update_dpp(0u, lo, 0x128, 0xf, 0xf, true); rh = __builtin_amdgcn_update_dpp(0u, hi, 0x128, 0xf, 0xf, true); break; // row_ror:8
	case 1: rl = __builtin_amdgcn_update_dpp(0u, lo, 0x141, 0xf, 0xf, true); rh = __builtin_amdgcn_update_dpp(0u, hi, 0x141, 0xf, 0xf, true); break; // row_half_mirror
	case 2: rl = __builtin_amdgcn_update_dpp(0u, lo, 0x4E, 0xf, 0xf, true); rh = __builtin_amdgcn_update_dpp(0u, hi, 0x4E, 0xf, 0xf, true); break;   // quad xor 2
	case 4: rl = __builtin_amdgcn_update_dpp(0u, lo, 0x124, 0xf, 0xf, true); rh = __builtin_amdgcn_update_dpp(0u, hi, 0x124, 0xf, 0xf, true); break; // row_ror:4
	default: rl = __builtin_amdgcn_update_dpp(0u, lo, 0xB1, 0xf, 0xf, true); rh = __builtin_amdgcn_update_dpp(0u, hi, 0xB1, 0xf, 0xf, true); break;  // quad xor 1
	}
	return __hiloint2double((int)rh, (int)rl);
}

template <bool ROW16>
__device__ __forceinline__ double swap_add_f64(double a, double b)
{
	const unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
	const unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
	if (ROW16) {
		const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
		const auto h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
		return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
	}
	const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
	const auto h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
	return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

__device__ __forceinline__ double valu_reduce16(const double (&v)[16], const unsigned lane)
{
	double s8[8], s4[4], s2[2];
#pragma unroll
	for (int i = 0; i < 8; i++) s8[i] = swap_add_f64<false>(v[i], v[i + 8]);   // lane bit 5
#pragma unroll
	for (int i = 0; i < 4; i++) s4[i] = swap_add_f64<true>(s8[i], s8[i + 4]);  // lane bit 4
	const bool up3 = (lane & 8) != 0, up2 = (lane & 4) != 0;
#pragma unroll
	for (int i = 0; i < 2; i++) {                                              // lane bit 3
		const double send = up3 ? s4[i] : s4[i + 2], keep = up3 ? s4[i + 2] : s4[i];
		s2[i] = keep + dpp_mov_f64(send, 0);
	}
	const double send = up2 ? s2[0] : s2[1], keep = up2 ? s2[1] : s2[0];       // lane bit 2 (mirror pairing)
	double r = keep + dpp_mov_f64(send, 1);
	r += dpp_mov_f64(r, 2);
	r += dpp_mov_f64(r, 3);
	return r;
}
