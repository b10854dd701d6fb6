// Pseudo-inverse and rank of ONE small symmetric matrix by the reference's rule (association.py:77-80: singular values -- |eigenvalues| of a
// symmetric matrix -- below tol x the largest count as zero; M^+ = V_kept diag(1 / s_kept) V_kept^T): the cyclic Jacobi eigenvalue iteration, which
// for matrices this small converges in a handful of sweeps and delivers every eigenvalue to full relative accuracy.  Host and device: the
// host's threaded stack (csrc/nrm_small_pinv.hip: single=1's groupings, normvar1's genes) and the per-gene solve of normvar on the device
// (csrc/nrm_normvar.hip) run the SAME code, so their integer ranks agree.
#pragma once
#include <cmath>
#include <cstdint>

#ifndef NRM_HD
#if defined(__HIPCC__)
#define NRM_HD __host__ __device__
#else
#define NRM_HD
#endif
#endif

// a (n x n, row-major, symmetric; destroyed) -> eigenvalues w, eigenvectors as the COLUMNS of v
NRM_HD inline void nrm_jacobi(double* a, double* v, double* w, int n) {
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) v[i * n + j] = i == j ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 60; sweep++) {
		double off = 0.0, diag = 0.0;
		for (int i = 0; i < n; i++) {
			diag += a[i * n + i] * a[i * n + i];
			for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
		}
		if (off == 0.0 || off <= 1e-34 * diag) break;  // (relative to the diagonal: the rotations below keep shrinking it quadratically)
		for (int p = 0; p < n - 1; p++)
			for (int q = p + 1; q < n; q++) {
				const double apq = a[p * n + q];
				if (apq == 0.0) continue;
				const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
				const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
				for (int k = 0; k < n; k++) {  // columns p, q
					const double akp = a[k * n + p], akq = a[k * n + q];
					a[k * n + p] = c * akp - s * akq;
					a[k * n + q] = s * akp + c * akq;
				}
				for (int k = 0; k < n; k++) {  // rows p, q
					const double apk = a[p * n + k], aqk = a[q * n + k];
					a[p * n + k] = c * apk - s * aqk;
					a[q * n + k] = s * apk + c * aqk;
				}
				a[p * n + q] = a[q * n + p] = 0.0;
				for (int k = 0; k < n; k++) {
					const double vkp = v[k * n + p], vkq = v[k * n + q];
					v[k * n + p] = c * vkp - s * vkq;
					v[k * n + q] = s * vkp + c * vkq;
				}
			}
	}
	for (int i = 0; i < n; i++) w[i] = a[i * n + i];
}

// m (n x n) -> inv = its pseudo-inverse, *rank; NMAX >= n bounds the scratch
template <int NMAX>
NRM_HD inline void nrm_small_pinv_one(const double* m, int n, double tol, double* inv, int64_t* rank) {
	double a[NMAX * NMAX], v[NMAX * NMAX], w[NMAX];
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) a[i * n + j] = 0.5 * (m[i * n + j] + m[j * n + i]);
	nrm_jacobi(a, v, w, n);
	double smax = 0.0;
	for (int i = 0; i < n; i++) smax = fmax(smax, fabs(w[i]));
	int r = 0;
	double iw[NMAX];
	for (int i = 0; i < n; i++) {
		const bool keep = fabs(w[i]) >= tol * smax;  // (a zero matrix keeps everything and divides by zero, as the reference does)
		r += keep;
		iw[i] = keep ? 1.0 / w[i] : 0.0;
	}
	for (int i = 0; i < n; i++)
		for (int j = 0; j <= i; j++) {
			double t = 0.0;
			for (int k = 0; k < n; k++)
				if (iw[k] != 0.0) t += v[i * n + k] * iw[k] * v[j * n + k];
			inv[i * n + j] = inv[j * n + i] = t;
		}
	*rank = r;
}
