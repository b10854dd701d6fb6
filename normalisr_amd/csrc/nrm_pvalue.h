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
#pragma once
#include "nrm_common.h"

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

// r2: the R^2 statistic as computed (may exceed 1 by rounding).  Mirrors the reference's order of
// operations: x = fl(1 - r2) is formed first (association.py:249), then w = 1 - x is exact.
__device__ __forceinline__ double nrm_pvalue(double r2, const PvalPlan& pl) {
	double x = 1.0 - r2;
	if (!(x > 0.0)) return (x != x) ? x : 0.0;  // beta.cdf clips x <= 0 to 0 (Q15); NaN propagates
	if (x >= 1.0) return 1.0;
	double w = 1.0 - x;
	double u = -log1p(-w);
	if (u <= pl.umax) {
		double z = pl.alpha * u;
		double sz = sqrt(z);
		double poly = pl.coef[NRM_PCOEF - 1];
#pragma unroll
		for (int j = NRM_PCOEF - 2; j >= 0; j--) poly = fma(poly, u, pl.coef[j]);
		return exp(-z) * (erfcx(sz) + sz * poly);
	}
	double a = pl.a;
	double lnf = -a * u + 0.5 * log(w) + pl.ln_front;  // ln[ x^a (1-x)^(1/2) / B(a,1/2) ]
	if (x < (a + 1.0) / (a + 2.5)) return exp(lnf) * nrm_betacf(a, 0.5, x) / a;
	return 1.0 - 2.0 * exp(lnf) * nrm_betacf(0.5, a, w);
}
