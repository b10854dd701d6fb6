// Pieces shared by the K1 kernels (nrm_residualize.hip: k_residualize, k_residualize_v4; tools/experiments/nrm_residualize_res.hip: k_residualize_res).
#pragma once
#include "nrm_common.h"
#include "nrm_digits.h"
#include "nrm_fix.h"

#define RES_NC_MAX 2048  // OLS tables a = x C^T and b = a dci live in dynamic LDS: 2 * RES_R * nc doubles (128 KiB at 2048)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}

// Vectorised variant: every lane moves 4 consecutive cells per step (16-byte loads of fp32 input, 32-byte
// loads/stores of fp64 covariates and residuals), so one wave-instruction covers 1-2 KiB of a row.
// Needs 16-byte aligned rows: ldx % (16/sizeof(T)) == 0, ldc % 2 == 0, ldo % 4 == 0.
template <typename T>
struct Vec4Load;
template <>
struct Vec4Load<float> {
	static __device__ __forceinline__ void ld(const float* p, double (&v)[4]) {
		float4 t = *reinterpret_cast<const float4*>(p);
		v[0] = t.x;
		v[1] = t.y;
		v[2] = t.z;
		v[3] = t.w;
	}
};
template <>
struct Vec4Load<double> {
	static __device__ __forceinline__ void ld(const double* p, double (&v)[4]) {
		double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
		v[0] = a.x;
		v[1] = a.y;
		v[2] = b.x;
		v[3] = b.y;
	}
};

// Input rows are streamed (each line is used once per sweep and is gone from L2 long before the next sweep of a long row): their loads
// can be marked non-temporal (NT) so that they do not push the covariates -- re-read by every workgroup -- out of L2.  Measured: with 20
// covariates x 100 000 cells (16 MB of them) K1 20.3 -> 16.9 ms; with 5 x 50 000 (2 MB) 3.27 -> 3.20; but rows of 10 000 cells, whose second
// sweep is served from L2, 0.186 -> 0.225 ms, and fp64 rows of 500 000 cells 10.4 -> 10.9: the launcher asks for it only for fp32 rows
// with at least 8 MB of covariates.
typedef float k1_f4 __attribute__((ext_vector_type(4)));
typedef double k1_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ k1_f4 k1_stream4(const float* p) {
	if (NT) return __builtin_nontemporal_load(reinterpret_cast<const k1_f4*>(p));
	return *reinterpret_cast<const k1_f4*>(p);
}
template <bool NT>
__device__ __forceinline__ k1_d2 k1_stream2(const double* p) {
	if (NT) return __builtin_nontemporal_load(reinterpret_cast<const k1_d2*>(p));
	return *reinterpret_cast<const k1_d2*>(p);
}
// input rows: four consecutive cells as doubles
template <typename T, bool NT = false>
struct RowLoad;
template <bool NT>
struct RowLoad<float, NT> {
	static __device__ __forceinline__ void ld(const float* p, double (&v)[4]) {
		const k1_f4 t = k1_stream4<NT>(p);
		v[0] = t[0];
		v[1] = t[1];
		v[2] = t[2];
		v[3] = t[3];
	}
};
template <bool NT>
struct RowLoad<double, NT> {
	static __device__ __forceinline__ void ld(const double* p, double (&v)[4]) {
		const k1_d2 a = k1_stream2<NT>(p), b = k1_stream2<NT>(p + 2);
		v[0] = a[0];
		v[1] = a[1];
		v[2] = b[0];
		v[3] = b[1];
	}
};

// four consecutive cells in the input's own type
template <typename T, bool NT = false>
struct RawLoad;
template <bool NT>
struct RawLoad<float, NT> {
	static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
		const k1_f4 t = k1_stream4<NT>(p);
		v[0] = t[0];
		v[1] = t[1];
		v[2] = t[2];
		v[3] = t[3];
	}
};
template <bool NT>
struct RawLoad<double, NT> {
	static __device__ __forceinline__ void ld(const double* p, double (&v)[4]) {
		const k1_d2 a = k1_stream2<NT>(p), b = k1_stream2<NT>(p + 2);
		v[0] = a[0];
		v[1] = a[1];
		v[2] = b[0];
		v[3] = b[1];
	}
};
template <typename T>
__device__ __forceinline__ void k1_ld4raw(const T* p, T (&v)[4]);
template <>
__device__ __forceinline__ void k1_ld4raw<float>(const float* p, float (&v)[4]) {
	const k1_f4 t = k1_stream4<false>(p);
	v[0] = t[0];
	v[1] = t[1];
	v[2] = t[2];
	v[3] = t[3];
}
template <>
__device__ __forceinline__ void k1_ld4raw<double>(const double* p, double (&v)[4]) {
	const k1_d2 a = k1_stream2<false>(p), b = k1_stream2<false>(p + 2);
	v[0] = a[0];
	v[1] = a[1];
	v[2] = b[0];
	v[3] = b[1];
}

// Fixed-point output for the integer Gram engine (csrc/nrm_gram_i8.hip): digit planes in its tiled layout and one exponent per
// row, written straight from K1 so that the fp64 residuals never make the round trip through HBM (NS = 0: none).
struct QuantOut {
	char* q;              // NS planes of plane_bytes each
	int64_t plane_bytes;
	int64_t nks;          // k-steps of 32 cells
	int64_t cks;          // k-steps per cell chunk (== nks: one chunk); chunk c is an operand of its own at q + c * chunk_bytes
	int64_t chunk_bytes;
	int* exps;            // x = digits * 2^exps[row]
	const double* cmax;   // (nc) largest |C_c| of every covariate row: bounds the residuals without a sweep of their own
	double* fix;          // (rows_pad, NRM_FIX_STRIDE) row records for K3's correction and guard (nrm_fix.h), or nullptr
};

// Loosest fixed-point scale K1 accepts from the bound max|x| + sum_c |b_c| max|C_c| without looking: bound / rms of the residuals
// (the rms estimated from the first sweep as |x|^2 - a.b).  Beyond it -- rows whose mean dwarfs their spread, near-collinear
// covariates with large opposite coefficients -- the residuals are swept for their true maximum, so that no more than log2 of
// this / (true max / rms) of the 8 NS - 2 bits are lost to the overestimate.
#define RES_LOOSE 12.0
