// Device arithmetic for p = I_x(a, 1/2), x = 1 - R^2, a = dof/2 -- the value the reference gets from
// scipy.stats.beta.cdf(1 - R2, dof/2, 0.5) at association.py:249.
//
// Two evaluation routes, chosen per element (wave-divergent only for rare strongly-correlated pairs):
//  * fast path (a >= 8 and u = -ln x <= umax): with alpha = a - 1/4,
//        B(a,1/2) p = Int_u^inf exp(-alpha s) s^(-1/2) h(s) ds,   h(s) = sqrt((s/2)/sinh(s/2)) = sum h_k s^2k,
//    termwise integration gives incomplete gamma functions Gamma(2k+1/2, alpha u), which reduce by the
//    recurrence Gamma(s+1,z) = s Gamma(s,z) + z^s e^-z to erfc(sqrt z) plus e^-z sqrt(z) * polynomial.
//    After normalising with the same series at u = 0:
//        p = exp(-alpha u) * ( erfcx(sqrt(alpha u)) + sqrt(alpha u) * sum_j coef[j] u^j ).
//    coef[] depends only on dof and is built once per call on the host (nrm_pvalue_plan_init).
//    Relative error <= 3e-14 for a >= 16, u <= 1.5 (measured against 60-digit mpmath).
//  * general path: continued fraction DLMF 8.17.22 (modified Lentz) on the convergent side.
//
// K3 is bound by the fp64 vector ALU and this function is most of it, so the two library calls that dominated the fast path are
// replaced by forms that cost every lane the same few instructions (round 3; 500 -> 200 instructions per P-value):
//  * u = -ln(1 - w) for w = R^2 < 1/4 (every pair whose P-value is not 0 at a few thousand cells) as 2 atanh(w / (2 - w)): one
//    reciprocal and a polynomial of degree 8 in s^2 (s <= 1/7; truncation 3e-17) instead of log1p's double-double arithmetic;
//  * erfcx from a table (nrm_erfcx_tab.h: 64 pieces of degree 7 in q = 4 / (4 + y), 3e-16 against mpmath) instead of the
//    library's range-by-range rational functions, of which a wavefront executes every branch any of its lanes takes.
// nrm_pvalue_fast is the straight-line form of the fast path for the sweep kernels (no branch: the unrolled pairs of a thread
// interleave); pairs it cannot take (ok = false) go through nrm_pvalue afterwards, which returns the same bits wherever both apply.
#pragma once
#include "nrm_common.h"
#include "nrm_erfcx_tab.h"

struct PvalPlan {
	double a, alpha, ln_front, umax;
	double coef[NRM_PCOEF];
};

__device__ __forceinline__ double nrm_betacf(double a, double b, double x) {
	const double tiny = 1e-300, eps = 2e-16;
	double qab = a + b, qap = a + 1.0, qam = a - 1.0;
	double c = 1.0, d = 1.0 - qab * x / qap;
	if (fabs(d) < tiny) d = tiny;
	d = 1.0 / d;
	double h = d;
	for (int m = 1; m <= 20000; m++) {
		double m2 = 2.0 * m;
		double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
		d = 1.0 + aa * d;
		if (fabs(d) < tiny) d = tiny;
		c = 1.0 + aa / c;
		if (fabs(c) < tiny) c = tiny;
		d = 1.0 / d;
		h *= d * c;
		aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
		d = 1.0 + aa * d;
		if (fabs(d) < tiny) d = tiny;
		c = 1.0 + aa / c;
		if (fabs(c) < tiny) c = tiny;
		d = 1.0 / d;
		double del = d * c;
		h *= del;
		if (fabs(del - 1.0) < eps) break;
	}
	return h;
}

// 1 / d for finite d >= 1: the hardware estimate and two Newton steps (each squares the relative error: 2^-14 would be enough)
__device__ __forceinline__ double nrm_rcp(double d) {
	double r = __builtin_amdgcn_rcp(d);
	r = fma(fma(-d, r, 1.0), r, r);
	r = fma(fma(-d, r, 1.0), r, r);
	return r;
}

// u = -ln(1 - w) for 0 <= w < 1/4: u = 2 atanh(s), s = w / (2 - w) <= 1/7
__device__ __forceinline__ double nrm_neglog1m_small(double w) {
	const double s = w * nrm_rcp(2.0 - w);
	const double t = s * s;
	double q = 1.0 / 17.0;
	q = fma(q, t, 1.0 / 15.0);
	q = fma(q, t, 1.0 / 13.0);
	q = fma(q, t, 1.0 / 11.0);
	q = fma(q, t, 1.0 / 9.0);
	q = fma(q, t, 1.0 / 7.0);
	q = fma(q, t, 1.0 / 5.0);
	q = fma(q, t, 1.0 / 3.0);
	const double s2 = s + s;
	return fma(s2 * t, q, s2);
}

// erfcx(y) = exp(y^2) erfc(y) for finite y >= 0
__device__ __forceinline__ double nrm_erfcx_pos(double y) {
	const double q = 256.0 * nrm_rcp(4.0 + y);  // NRM_ERFCX_PIECES * 4 / (4 + y) in (0, 64]
	int i = (int)q;
	i = i > NRM_ERFCX_PIECES - 1 ? NRM_ERFCX_PIECES - 1 : i;  // y = 0: the right end of the last piece
	const double t = fma(2.0, q - (double)i, -1.0);
	const double* __restrict__ c = kNrmErfcx + NRM_ERFCX_TERMS * i;
	double e = c[NRM_ERFCX_TERMS - 1];
#pragma unroll
	for (int j = NRM_ERFCX_TERMS - 2; j >= 0; j--) e = fma(e, t, c[j]);
	return e;
}

// the fast path's formula for u = -ln(1 - R^2) <= umax.  NT: coefficients used -- all NRM_PCOEF up to u = 1.5; NRM_PCOEF_SMALL for
// u <= -ln(3/4) (R^2 < 1/4), where the terms left out, h_k u^2k with k >= 6, are below (u / 2 pi)^12 = 1e-16 of the sum
#define NRM_PCOEF_SMALL 12
template <int NT>
__device__ __forceinline__ double nrm_pvalue_series(double u, const PvalPlan& pl) {
	const double z = pl.alpha * u;
	const double sz = sqrt(z);
	double poly = pl.coef[NT - 1];
#pragma unroll
	for (int j = NT - 2; j >= 0; j--) poly = fma(poly, u, pl.coef[j]);
	return exp(-z) * fma(sz, poly, nrm_erfcx_pos(sz));
}

// From a few thousand cells on every R^2 >= 1/4 has P = 0 in double precision: exp(-alpha u) underflows to 0 for alpha u > 745.2,
// and u >= -ln(3/4).  (What both routes of nrm_pvalue return there anyway; said once, up front, it spares strongly correlated
// data the library logarithm.)
__device__ __forceinline__ bool nrm_zero_above_quarter(const PvalPlan& pl) { return pl.alpha * 0.2876820724517809 > 746.0; }

// r2: the R^2 statistic as computed (may exceed 1 by rounding).  Mirrors the reference's order of
// operations: x = fl(1 - r2) is formed first (association.py:249), then w = 1 - x is exact.
__device__ __forceinline__ double nrm_pvalue(double r2, const PvalPlan& pl) {
	double x = 1.0 - r2;
	if (!(x > 0.0)) return (x != x) ? x : 0.0;  // beta.cdf clips x <= 0 to 0 (Q15); NaN propagates
	if (x >= 1.0) return 1.0;
	double w = 1.0 - x;
	if (w >= 0.25 && nrm_zero_above_quarter(pl)) return 0.0;
	if (w < 0.25 && pl.umax >= 0.3) return nrm_pvalue_series<NRM_PCOEF_SMALL>(nrm_neglog1m_small(w), pl);
	double u = -log1p(-w);
	if (u <= pl.umax) return nrm_pvalue_series<NRM_PCOEF>(u, pl);
	double a = pl.a;
	double lnf = -a * u + 0.5 * log(w) + pl.ln_front;  // ln[ x^a (1-x)^(1/2) / B(a,1/2) ]
	if (x < (a + 1.0) / (a + 2.5)) return exp(lnf) * nrm_betacf(a, 0.5, x) / a;
	return 1.0 - 2.0 * exp(lnf) * nrm_betacf(0.5, a, w);
}

// Straight-line form: the value of nrm_pvalue(r2, pl) whenever ok comes back true (R^2 in [0, 1/4) and a plan with a fast path, or
// an R^2 >= 1/4 whose P-value is 0 anyway);
// otherwise the result is meaningless and the caller evaluates nrm_pvalue.
__device__ __forceinline__ double nrm_pvalue_fast(double r2, const PvalPlan& pl, bool& ok) {
	const double x = 1.0 - r2, w = 1.0 - x;
	const bool small = w >= 0.0 && w < 0.25 && pl.umax >= 0.3;  // u <= -ln(3/4) = 0.2877
	const bool zero = w >= 0.25 && nrm_zero_above_quarter(pl);  // (also R^2 >= 1: x <= 0)
	ok = small || zero;
	const double p = nrm_pvalue_series<NRM_PCOEF_SMALL>(nrm_neglog1m_small(small ? w : 0.0), pl);
	return x >= 1.0 ? 1.0 : (zero ? 0.0 : p);
}
