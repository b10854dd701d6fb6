// Fixed-point digits of the integer Gram engine (nrm_gram_i8.hip), shared by K1 and the stand-alone quantiser.
#pragma once
#include "nrm_common.h"

// q = rint(v 2^-sh) for four consecutive cells, cut into NS balanced base-256 digits (d_s in [-128, 127], top digit the rest),
// returned as one 32-bit word per digit plane (cell i in byte i).  No 64-bit shifts or per-digit borrows:
//  * adding 1.5 2^52 leaves q, rounded to nearest even like rint(), as a two's complement integer in the low mantissa bits;
//  * adding 0x80 to each of the lower NS-1 bytes turns the balanced digits into the ordinary base-256 digits of the biased
//    number, and flipping those 0x80 bits turns each unsigned byte back into the signed digit -- the bytes of the result ARE
//    the digits (|q| < 2^(8 NS - 2), so byte NS-1 holds the whole top digit);
//  * three byte permutes per plane gather byte s of the four cells.
template <int NS>
__device__ __forceinline__ void nrm_digits4(const double (&v)[4], int sh, unsigned (&w)[NS]) {
	unsigned long long bias = 0;
#pragma unroll
	for (int s = 0; s < NS - 1; s++) bias |= 0x80ull << (8 * s);
	unsigned lo[4], hi[4];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const double t = ldexp(v[i], -sh) + 6755399441055744.0;
		const unsigned long long b = ((unsigned long long)__double_as_longlong(t) + bias) ^ bias;
		lo[i] = (unsigned)b;
		hi[i] = (unsigned)(b >> 32);
	}
#pragma unroll
	for (int s = 0; s < NS; s++) {
		const unsigned* src = s < 4 ? lo : hi;
		const unsigned sel = 0x0c0c0000u | ((4u + (s & 3)) << 8) | (unsigned)(s & 3);
		const unsigned t01 = __builtin_amdgcn_perm(src[1], src[0], sel), t23 = __builtin_amdgcn_perm(src[3], src[2], sel);
		w[s] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
	}
}
