// Row statistics of the integer Gram engine and what K3 does with them: the exact correction for the coherent part of the digit
// products the engine leaves out, and the data-dependent accuracy guard (the reference computes this contraction in fp64 end to
// end, association.py:224-235,248-249; tools/i8_error_model.py restates the arithmetic below on the CPU).
//
// The engine (nrm_gram_i8.hip) sums the digit pairs with s + t >= NS - 1 exactly and leaves out
//     E_ij = sum_{s+t <= NS-2} 256^(s+t) sum_k d_is[k] d_jt[k].
// Rows that take few distinct values (sparse or binary rows under an intercept or one-hot covariates) have low-order digits that
// are the same in most cells, so E adds up coherently -- 5 2^-46 (max/rms)_i (max/rms)_j in Pearson r, 1e-10 for 0.01 %-dense rows
// at 500 000 cells.  K1 therefore records, per row, the digit sums S_s and sums of squares of planes s <= NS - 2, and K3
//   1. adds the product of the digit MEANS back exactly:  sum_{s+t <= NS-2} 256^(s+t) S_is S_jt / n;
//   2. bounds what is still missing -- the digit covariances sum_k (d_is - mean)(d_jt - mean) -- by Cauchy-Schwarz with the digit
//      variances, plus the rounding of the fixed-point conversion itself:
//          |delta r_ij| <= K c_i c_j + g_i + g_j,     K = sum_{w <= NS-2} (w + 1) 256^w,
//      c_i = sqrt(max_s V_is / sum_k q_ik^2),  g_i = sqrt(n) 2^(sh_i - 1) / |x~_i|;
//   3. counts the pairs whose P-value that bound could move by more than the budget (relative; |d ln p| <= dof |r| |delta r| /
//      (1 - R^2)) unless the P-value is 0 on the whole interval.  The host reruns a call with such pairs on the fp64 Gram kernel.
//
// Per-row record, NRM_FIX_STRIDE doubles: [0..4] u_s = 2^sh 256^s S_s (0 beyond NS - 2), [5] c, [6] g, [7] 2^(sh + B) / rms (the
// effective max/rms of the quantisation, diagnostics only).
#pragma once
#include "nrm_common.h"

#ifndef NRM_FIX_STRIDE
#define NRM_FIX_STRIDE 8
#endif

struct FixArgs {
	const double* fx;  // (nx, NRM_FIX_STRIDE) records of the x rows, nullptr: no correction, no guard (fp64 Gram kernels)
	const double* fy;  // (ny, NRM_FIX_STRIDE)
	double inv_n;      // 1 / cells
	double kconst;     // K above
	double budget;     // largest tolerated relative change of a P-value (<= 0: count nothing)
	double dof;
	int top;           // NS - 2: highest digit plane with a dropped product
};

static inline FixArgs nrm_fix_args(const double* fx, const double* fy, int nslices, int64_t n_cells, double dof, double budget) {
	FixArgs f = {nullptr, nullptr, 0.0, 0.0, 0.0, dof, 0};
	if (!fx || !fy || nslices < 2) return f;
	f.fx = fx;
	f.fy = fy;
	f.inv_n = 1.0 / (double)n_cells;
	f.top = nslices - 2;
	double k = 0.0, w256 = 1.0;
	for (int w = 0; w <= f.top; w++, w256 *= 256.0) k += (w + 1) * w256;
	f.kconst = k;
	f.budget = budget;
	return f;
}

// record of a column (y) row as the per-thread constants of a sweep: prefix sums v[m] = sum_{t <= m} u[t], c and g
struct FixCol {
	double v[5], c, g;
};

__device__ __forceinline__ FixCol nrm_fix_col(const double* rec) {
	FixCol f;
	double acc = 0.0;
#pragma unroll
	for (int t = 0; t < 5; t++) {
		acc += rec[t];
		f.v[t] = acc;
	}
	f.c = rec[5];
	f.g = rec[6];
	return f;
}

// the mean-product correction of one pair, in the units of dot
__device__ __forceinline__ double nrm_fix_corr(const double* __restrict__ ux, const FixCol& y, int top, double inv_n) {
	double acc = 0.0;
#pragma unroll
	for (int s = 0; s < 5; s++)  // (constant register indices: top is 4 at six digit planes, 3 at five)
		acc = fma(ux[s], top == 4 ? y.v[4 - s] : (s <= 3 ? y.v[3 - s] : 0.0), acc);
	return acc * inv_n;
}

// The guard of one pair whose P-value p came from R^2 = r2: 1 when the bound on the engine's error could move p by more than the
// budget.  |d ln p / d r| <= (sqrt(dof) + dof |r|) / (1 - R^2)^2: the hazard rate of the normal limit (<= 1 + t, which the t
// distribution's stays under) times dt/dr = sqrt(dof) (1 - R^2)^-3/2.  A P-value that is 0 on the whole interval |r| +- bound is
// exempt.  `worst` collects the largest error estimate of the pairs that are not exempt (diagnostics: how close a call came to the
// budget).  K3 is bound by the fp64 vector ALU: the decision is two products on either side of one comparison -- |r| comes from a
// single-precision square root rounded up, the diagnostic quotient is taken in single precision.
struct FixAcc {
	int bad;         // pairs over the budget whose P-value is not 0
	float worst;     // largest error estimate met (among pairs within the budget, or over it with a non-zero P-value)
	double lo2_min;  // pairs over the budget with P = 0: the smallest (|r| - bound)^2 among them -- P is monotone in R^2, so if P is still
					 // 0 there, every one of them is exempt (one more P-value evaluation per thread, after its pairs: nrm_fix_close)
	double r2max, cmax, gmax;  // first level: the largest R^2, c and g among the thread's pairs (one column, several rows)
};
__device__ __forceinline__ FixAcc nrm_fix_acc() { return FixAcc{0, 0.f, INFINITY, 0.0, 0.0, 0.0}; }

// Two levels, because K3 is bound by the fp64 vector ALU.  Per pair only three maxima are kept (nrm_fix_note).  After its pairs a
// thread evaluates the bound ONCE for the worst combination of them (nrm_fix_screen): within the budget -- the normal case -- every
// pair of the thread is certified; otherwise the thread goes over its pairs again with the exact per-pair test (nrm_fix_guard).
__device__ __forceinline__ void nrm_fix_note(FixAcc& a, double cx, double gx, double r2) {
	a.r2max = fmax(a.r2max, r2);
	a.cmax = fmax(a.cmax, cx);
	a.gmax = fmax(a.gmax, gx);
}

// error estimate num / den of a pair with these c, g and R^2; |d ln p / d r| <= (sqrt(dof) + dof |r|) / (1 - R^2)^2: the hazard rate of the
// normal limit (<= 1 + t, which the t distribution's stays under) times dt/dr = sqrt(dof) (1 - R^2)^-3/2.  |r| comes from a
// single-precision square root rounded up (R^2 below the float range: |r| < 1e-19 counts as 0).
__device__ __forceinline__ void nrm_fix_bound(const FixArgs& f, double cx, double gx, const FixCol& y, double r2, double sqrt_dof, double& dr,
											  double& ar, double& num, double& den) {
	dr = fma(f.kconst * cx, y.c, gx + y.g);
	ar = (double)(sqrtf((float)r2) * 1.0000002f);
	const double om = fmax(1.0 - r2, 1e-150);
	num = dr * fma(f.dof, ar, sqrt_dof);
	den = om * om;
}

// true: the thread has to look at its pairs one by one
__device__ __forceinline__ bool nrm_fix_screen(const FixArgs& f, FixAcc& a, const FixCol& y, double sqrt_dof) {
	double dr, ar, num, den;
	nrm_fix_bound(f, a.cmax, a.gmax, y, a.r2max, sqrt_dof, dr, ar, num, den);
	if (num > f.budget * den || !(num == num)) return true;
	a.worst = fmaxf(a.worst, __fdividef((float)num, (float)den));
	return false;
}

// the exact test of one pair whose P-value is (p_nonzero) or is not 0 as stored -- a P-value that underflowed in the output type
// counts as 0 here and goes through the exemption test, which is evaluated in fp64: a P-value that is 0 on the whole interval
// |r| +- bound is exempt
__device__ __forceinline__ void nrm_fix_guard(const FixArgs& f, double cx, double gx, const FixCol& y, double r2, double sqrt_dof, bool p_nonzero, FixAcc& a) {
	double dr, ar, num, den;
	nrm_fix_bound(f, cx, gx, y, r2, sqrt_dof, dr, ar, num, den);
	const float err = __fdividef((float)num, (float)den);
	if (num > f.budget * den) {
		if (p_nonzero) {
			a.bad++;
			a.worst = fmaxf(a.worst, err);
		} else {
			const double lo = fmax(ar - dr, 0.0);
			a.lo2_min = fmin(a.lo2_min, lo * lo);
		}
	} else
		a.worst = fmaxf(a.worst, err);
}

template <typename Plan, typename PFn>
__device__ __forceinline__ void nrm_fix_close(FixAcc& a, const Plan& pl, PFn pvalue) {
	if (a.lo2_min < INFINITY && pvalue(a.lo2_min, pl) != 0.0) a.bad++;
}

// ---- K1 side: digit statistics of a row -> its record ---------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (every lane of the row gets it)
__device__ __forceinline__ int row16_sum(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
	v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
	v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
	v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
	return v;
}
__device__ __forceinline__ long long wave_sum_i32(int v) {
	v = row16_sum(v);
	return (long long)__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ long long wave_sum_u32(unsigned v) {
	v = (unsigned)row16_sum((int)v);  // (two's complement: the 16-lane sums stay below 2^32 for rows of up to 2^22 cells)
	return (long long)(unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
		   (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}


// S[s], Q[s]: sums and sums of squares of the digits of plane s <= NS - 2 over the row's n cells; sh: x~ = q 2^sh; ss = |x~|^2
template <int NS>
__device__ __forceinline__ void nrm_fix_record(double* __restrict__ rec, const double (&S)[5], const double (&Q)[5], int sh, double ss, double n) {
	double vmax = 0.0;
#pragma unroll
	for (int s = 0; s < 5; s++) {
		double u = 0.0;
		if (s <= NS - 2) {
			vmax = fmax(vmax, Q[s] - S[s] * S[s] / n);
			u = ldexp(S[s], sh + 8 * s);
		}
		rec[s] = u;
	}
	const bool ok = ss > 0.0 && ss < INFINITY;
	const double inv_q = ok ? ldexp(rsqrt(ss), sh) : 0.0;  // 1 / |q|, q = x~ 2^-sh
	rec[5] = sqrt(vmax) * inv_q;
	rec[6] = 0.5 * sqrt(n) * inv_q;
	rec[7] = ldexp(sqrt(n) * inv_q, 8 * NS - 2);
}
