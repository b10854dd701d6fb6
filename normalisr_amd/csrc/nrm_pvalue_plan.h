// The P-value plan of nrm_pvalue.h -- the constants of p = I_{1-R^2}(dof/2, 1/2) that depend on dof alone -- in plain double
// arithmetic, for host AND device.  The host's nrm_pvalue_plan_init (nrm_host_logic.h) builds the same numbers in long double; a
// single=1 screen needs one plan per grouping (dof_i = ns_i - 1 - rank_i - dimreduce, association.py:372-374), and those are built
// where the groupings' statistics are: on the device (k_s1_group_info).  Every sum below runs from its smallest term up; against the
// long-double plan the polynomial sum_j coef_j u^j moves by <= 5e-16 of its terms over dof 16 ... 2e6, u <= 1.5
// (tests/test_cabi_cpu.py::test_device_pvalue_plans_match_the_host_plans holds the two together).
#pragma once
#include <cmath>
#include "../../include/normalisr_hip.h"

#ifndef NRM_HD
#if defined(__HIPCC__)
#define NRM_HD __host__ __device__
#else
#define NRM_HD
#endif
#endif

// Taylor coefficients h_k of sqrt((s/2)/sinh(s/2)) = sum_k h_k s^(2k) (the table of nrm_host_logic.h; a function, so that device code can index it)
NRM_HD inline double nrm_plan_h(int k) {
	switch (k) {
		case 0: return 1.0;
		case 1: return -0.02083333333333333333333;
		case 2: return 0.000390625;
		case 3: return -0.000007879670965608465608466;
		case 4: return 1.696766579172178130511e-7;
		case 5: return -3.805064191721906565657e-9;
		case 6: return 8.748377596315407304061e-11;
		case 7: return -2.044523359411973817584e-12;
		case 8: return 4.833351797967704408319e-14;
		case 9: return -1.152434101767385923873e-15;
		default: return 2.766052043599370042286e-17;
	}
}

// ln( Gamma(a+1/2)/Gamma(a) ): recurrence up to a >= 24, then the asymptotic series (DLMF 5.11.13)
NRM_HD inline double nrm_plan_ln_gamma_ratio_half(double a) {
	double shift = 0.0;
	while (a < 24.0) {
		shift += log(a / (a + 0.5));
		a += 1.0;
	}
	const double i = 1.0 / a, i2 = i * i;
	const double s = i * (-1.0 / 8 + i2 * (1.0 / 192 + i2 * (-1.0 / 640 + i2 * (17.0 / 14336 + i2 * (-31.0 / 18432 + i2 * (691.0 / 180224))))));
	return 0.5 * log(a) + s + shift;
}

// out[0 .. 3 + NRM_PCOEF] = a, alpha, ln_front, umax, coef[NRM_PCOEF] (the layout of struct nrm_pvalue_plan); dof > 0
NRM_HD inline void nrm_pvalue_plan_fill(double dof, double* out) {
	constexpr int K = NRM_PCOEF / 2;  // series terms k = 0..K
	const double a = 0.5 * dof, alpha = a - 0.25;
	out[0] = a;
	out[1] = alpha;
	out[2] = nrm_plan_ln_gamma_ratio_half(a) - 0.57236494292470008707;  // - ln(pi)/2
	for (int j = 0; j < NRM_PCOEF; j++) out[4 + j] = 0.0;
	if (a < 8.0) {
		out[3] = 0.0;  // asymptotic series in 1/alpha not accurate enough: continued fraction only
		return;
	}
	out[3] = 1.5;
	const double ia = 1.0 / alpha;
	double ipow[NRM_PCOEF + 1];  // alpha^-m
	ipow[0] = 1.0;
	for (int m = 1; m <= NRM_PCOEF; m++) ipow[m] = ipow[m - 1] * ia;
	// S = sum_k h_k alpha^-2k c'_2k,  c'_m = prod_{i<m} (i + 1/2)
	double cp[K + 1];
	cp[0] = 1.0;
	for (int k = 1; k <= K; k++) cp[k] = cp[k - 1] * (2 * k - 1.5) * (2 * k - 0.5);
	double S = 0.0;
	for (int k = K; k >= 0; k--) S += nrm_plan_h(k) * ipow[2 * k] * cp[k];
	const double front = 1.0 / (S * 1.7724538509055160272981674833411);
	// coef_j = (1/(S sqrt(pi))) sum_{k: 2k > j} h_k alpha^(j-2k) prod_{i=j+1}^{2k-1} (i + 1/2)
	for (int j = 0; j < NRM_PCOEF; j++) {
		double term[K + 1];
		const int k0 = j / 2 + 1;
		double pr = 1.0;
		int hi = j + 1;
		for (int k = k0; k <= K; k++) {
			for (; hi <= 2 * k - 1; hi++) pr *= hi + 0.5;
			term[k] = nrm_plan_h(k) * ipow[2 * k - j] * pr;
		}
		double c = 0.0;
		for (int k = K; k >= k0; k--) c += term[k];
		out[4 + j] = c * front;
	}
}
