/*
 * normalisr_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into, imported by or
 * called from the product path (normalisr_amd/); only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * Restates, in plain C, the third-party arithmetic the reference's hot path calls at
 *   /root/reference/src/normalisr/association.py:249
 *       ansp = beta.cdf(1 - ansp, (n - 1 - dcr - dimreduce) / 2, 0.5)
 * i.e. scipy.stats.beta.cdf -> scipy.special.betainc(a, 1/2, x) (scipy 1.15.3 here; unpinned in
 * the reference's setup.py:31).  scipy is not part of /root/reference, so the published definition
 * is restated: the regularised incomplete beta function I_x(a, b) evaluated with the continued
 * fraction DLMF 8.17.22 (modified Lentz), on whichever side of x = (a+1)/(a+b+2) converges, with
 * the b = 1/2 normalisation Gamma(a+1/2)/Gamma(a) from its asymptotic series (DLMF 5.11.13).
 * Pinned against golden vectors generated from scipy in this container: tests/golden/G3_ptable.npz
 * (tests/test_oracle.py).  The plain-loop block routine below restates association.py:224-249.
 */
#include <math.h>
#include <stddef.h>

/* ln( Gamma(a+1/2) / Gamma(a) ), a > 0.  Shift a up to >= 24 with the recurrence, then DLMF 5.11.13. */
double nrm_oracle_lngamma_ratio_half(double a) {
	double shift = 0.0;
	while (a < 24.0) {
		/* G(a+1/2)/G(a) = G(a+3/2)/G(a+1) * a/(a+1/2) */
		shift += log(a / (a + 0.5));
		a += 1.0;
	}
	double i = 1.0 / a, i2 = i * i;
	double s = i * (-1.0 / 8 + i2 * (1.0 / 192 + i2 * (-1.0 / 640 + i2 * (17.0 / 14336 + i2 * (-31.0 / 18432 + i2 * (691.0 / 180224))))));
	return 0.5 * log(a) + s + shift;
}

/* Continued fraction of DLMF 8.17.22 by modified Lentz; returns the value f with
 * I_x(a,b) = x^a (1-x)^b / (a B(a,b)) * f. */
static double betacf(double a_, double b_, double x_) {
	/* 80-bit long double on the x86 host: the Lentz product runs O(sqrt(a)) .. O(a x) factors and
	 * would otherwise accumulate ~1e-16 per factor (2e-11 at dof = 5e5). */
	const long double tiny = 1e-300L, eps = 1e-18L;
	long double a = a_, b = b_, x = x_;
	long double qab = a + b, qap = a + 1.0L, qam = a - 1.0L;
	long double c = 1.0L, d = 1.0L - qab * x / qap;
	if (fabsl(d) < tiny) d = tiny;
	d = 1.0L / d;
	long double h = d;
	for (long m = 1; m <= 4000000; m++) {
		long double m2 = 2.0L * m;
		long double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
		d = 1.0L + aa * d;
		if (fabsl(d) < tiny) d = tiny;
		c = 1.0L + aa / c;
		if (fabsl(c) < tiny) c = tiny;
		d = 1.0L / d;
		h *= d * c;
		aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
		d = 1.0L + aa * d;
		if (fabsl(d) < tiny) d = tiny;
		c = 1.0L + aa / c;
		if (fabsl(c) < tiny) c = tiny;
		d = 1.0L / d;
		long double del = d * c;
		h *= del;
		if (fabsl(del - 1.0L) < eps) break;
	}
	return (double)h;
}

/* I_x(a, 1/2) for x given as the pair (x, omx = 1 - x) so that neither end loses digits. */
static double ibeta_half(double a, double x, double omx) {
	if (!(x > 0.0)) return 0.0;
	if (!(omx > 0.0)) return 1.0;
	/* ln of x^a (1-x)^(1/2) / B(a,1/2);  B(a,1/2) = sqrt(pi) Gamma(a)/Gamma(a+1/2) */
	double lnx = (omx < 0.5) ? log1p(-omx) : log(x);
	double lnfront = a * lnx + 0.5 * log(omx) + nrm_oracle_lngamma_ratio_half(a) - 0.57236494292470008707 /* ln(pi)/2 */;
	if (x < (a + 1.0) / (a + 2.5))
		return exp(lnfront) * betacf(a, 0.5, x) / a;
	return 1.0 - exp(lnfront) * betacf(0.5, a, omx) / 0.5;
}

/* scipy.stats.beta.cdf(x, a, 0.5) semantics: clipped to the support (association.py:249, Q15). */
double nrm_oracle_beta_cdf_half(double x, double a) {
	if (x != x) return x;
	if (x <= 0.0) return 0.0;
	if (x >= 1.0) return 1.0;
	return ibeta_half(a, x, 1.0 - x);
}

/* p[i] = beta.cdf(1 - r2[i], dof/2, 0.5), forming 1 - r2 in fp64 first exactly as association.py:249 does. */
void nrm_oracle_pvalues(const double* r2, size_t cnt, double dof, double* p) {
	for (size_t i = 0; i < cnt; i++) {
		volatile double x = 1.0 - r2[i];
		p[i] = nrm_oracle_beta_cdf_half(x, 0.5 * dof);
	}
}

/*
 * Plain-loop restatement of one block of association_test_1 (association.py:224-235) for small
 * cases: residualise dx (nx,n) and dy (ny,n) against dc (nc,n) with the pseudo-inverse dci (nc,nc),
 * row variances with 0 -> 1, gamma = x~.y~ / (n vx), R2 = gamma^2 vx / vy, p as above.
 * All arrays row-major fp64.  work must hold (nx+ny)*n + (nx+ny)*nc doubles.
 */
void nrm_oracle_block(const double* dx, const double* dy, const double* dc, const double* dci,
					  long nx, long ny, long nc, long n, long dcr, long dimreduce,
					  double* p, double* gamma, double* vx, double* vy, double* work) {
	double* rx = work;
	double* ry = rx + nx * n;
	double* cc = ry + ny * n;
	const double* src[2] = {dx, dy};
	double* dst[2] = {rx, ry};
	long rows[2] = {nx, ny};
	double* var[2] = {vx, vy};
	for (int s = 0; s < 2; s++) {
		for (long i = 0; i < rows[s]; i++) {
			const double* row = src[s] + i * n;
			double* out = dst[s] + i * n;
			double* b = cc + (s ? nx * nc : 0) + i * nc;
			for (long k = 0; k < n; k++) out[k] = row[k];
			if (dcr > 0) {
				/* ccx = dci @ (dc @ dx.T)  (association.py:226-227) */
				double t[64];
				for (long c = 0; c < nc; c++) {
					double acc = 0;
					for (long k = 0; k < n; k++) acc += dc[c * n + k] * row[k];
					t[c] = acc;
				}
				for (long c = 0; c < nc; c++) {
					double acc = 0;
					for (long e = 0; e < nc; e++) acc += dci[c * nc + e] * t[e];
					b[c] = acc;
				}
				/* dx1 = dx - ccx @ dc  (association.py:228-229) */
				for (long k = 0; k < n; k++) {
					double acc = 0;
					for (long c = 0; c < nc; c++) acc += b[c] * dc[c * n + k];
					out[k] = row[k] - acc;
				}
			}
			double ss = 0;
			for (long k = 0; k < n; k++) ss += out[k] * out[k];
			double v = ss / (double)n;
			var[s][i] = (v == 0.0) ? 1.0 : v; /* association.py:230-233 */
		}
	}
	double dof = (double)(n - 1 - dcr - dimreduce);
	for (long i = 0; i < nx; i++)
		for (long j = 0; j < ny; j++) {
			double acc = 0;
			for (long k = 0; k < n; k++) acc += ry[j * n + k] * rx[i * n + k];
			double c = acc / ((double)n * vx[i]);       /* association.py:234 */
			double r2 = c * c * vx[i] / vy[j];          /* association.py:235 */
			gamma[i * ny + j] = c;
			volatile double x = 1.0 - r2;
			p[i * ny + j] = nrm_oracle_beta_cdf_half(x, 0.5 * dof);
		}
}
