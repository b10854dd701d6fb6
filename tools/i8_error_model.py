"""Host model (numpy, exact integers) of the integer Gram engine's arithmetic and of its error guard.

The engine (csrc/nrm_gram_i8.hip) rounds every residual row once to B = 8 NS - 2 bit fixed point (scale 2^sh per row), cuts it
into NS balanced base-256 digits and sums the digit-pair products with s + t >= NS - 1 exactly.  What it leaves out is
    E_ij = sum_{s+t <= NS-2} 256^(s+t) sum_k d_is[k] d_jt[k].
K1 records the digit sums S_is of every row, so that K3 adds the coherent part of E back exactly,
    sum_{s+t <= NS-2} 256^(s+t) S_is S_jt / n           (the product of the digit MEANS),
and what then remains is the digit covariance sum_k (d_is - mean)(d_jt - mean), bounded by Cauchy-Schwarz with the digit
variances V_is that K1 records too:
    |delta r_ij| <= K c_i c_j + g_i + g_j,   c_i = sqrt(max_s V_is / sum_k q_ik^2),   K = sum_{s+t<=NS-2} 256^(s+t),
    g_i = sqrt(n) 2^(sh_i - 1) / |x~_i|  (the rounding of the fixed-point conversion itself).
This file evaluates all of that on the CPU; tests/test_i8_guard_cpu.py holds the bound to the actual error.
"""
import numpy as np


def quantise(x, ns=6, sh=None):
	"""q = rint(x 2^-sh), sh = e - B with 2^e > max|x| (frexp), as K1 does with the row's true maximum."""
	b = 8 * ns - 2
	x = np.asarray(x, dtype=np.float64)
	if sh is None:
		m = np.abs(x).max(axis=-1)
		e = np.where(m > 0, np.frexp(m)[1], 0)
		sh = e - b
	sh = np.asarray(sh, dtype=np.int64)
	q = np.rint(np.ldexp(x, -sh[..., None])).astype(np.int64)
	return q, sh


def digits(q, ns=6):
	"""Balanced base-256 digits of int64 q: list of NS int64 arrays, d_s in [-128, 127], the top digit takes the rest."""
	out = []
	q = q.copy()
	for s in range(ns):
		d = q.copy() if s == ns - 1 else ((q & 0xff) ^ 0x80) - 0x80
		q = (q - d) >> 8
		out.append(d)
	return out


def kept_and_dropped(da, db, ns=6):
	"""Exact Python-integer sums over the kept (s + t >= NS - 1) and the dropped digit pairs of one pair of rows."""
	kept = dropped = 0
	for s in range(ns):
		for t in range(ns):
			v = int(np.dot(da[s], db[t])) << (8 * (s + t))
			if s + t >= ns - 1:
				kept += v
			else:
				dropped += v
	return kept, dropped


def row_stats(d, q, sh, n, ns=6):
	"""What K1 records for one row: u_s = 2^sh 256^s S_s (s <= NS-2), c, g."""
	m = ns - 2
	S = [int(d[s].sum()) for s in range(m + 1)]
	Q = [int((d[s] * d[s]).sum()) for s in range(m + 1)]
	V = [max(0.0, Q[s] - S[s] * S[s] / n) for s in range(m + 1)]
	ssq = float((q.astype(np.float64)**2).sum())
	u = [np.ldexp(float(S[s]), int(sh) + 8 * s) for s in range(m + 1)]
	c = np.sqrt(max(V) / ssq) if ssq > 0 else 0.0
	g = 0.5 * np.sqrt(n / ssq) if ssq > 0 else 0.0  # sqrt(n) 2^(sh-1) / (|q| 2^sh)
	return dict(S=S, V=V, u=u, c=c, g=g, ssq=ssq)


def k_const(ns=6):
	return float(sum((w + 1) * 256**w for w in range(ns - 1)))


def mean_correction(sa, sb, n, ns=6):
	"""sum_{s+t <= NS-2} u_a[s] u_b[t] / n, in the units of x_a . x_b."""
	m = ns - 2
	tot = 0.0
	for s in range(m + 1):
		tot += sa['u'][s] * sum(sb['u'][t] for t in range(m - s + 1))
	return tot / n


def analyse_pair(xa, xb, ns=6):
	"""Everything about one pair of (already residualised) rows: exact r, engine r without / with the mean correction, the bound."""
	n = xa.shape[0]
	(qa, qb), (sha, shb) = zip(quantise(xa, ns), quantise(xb, ns))
	da, db = digits(qa, ns), digits(qb, ns)
	kept, dropped = kept_and_dropped(da, db, ns)
	sa, sb = row_stats(da, qa, sha, n, ns), row_stats(db, qb, shb, n, ns)
	scale = np.ldexp(1.0, int(sha + shb))
	nrm = np.sqrt(float((xa**2).sum()) * float((xb**2).sum()))
	exact_q = kept + dropped
	import fractions
	true_dot = float(sum(fractions.Fraction(float(a)) * fractions.Fraction(float(b)) for a, b in zip(xa, xb))) if n <= 4096 else float(
		np.dot(xa.astype(np.longdouble), xb.astype(np.longdouble)))
	r_true = true_dot / nrm
	r_kept = float(kept) * scale / nrm
	r_fixed = (float(kept) * scale + mean_correction(sa, sb, n, ns)) / nrm
	r_allq = float(exact_q) * scale / nrm
	bound = k_const(ns) * sa['c'] * sb['c'] + sa['g'] + sb['g']
	return dict(r_true=r_true, r_kept=r_kept, r_fixed=r_fixed, r_allq=r_allq, bound=bound, c=(sa['c'], sb['c']), g=(sa['g'], sb['g']))


if __name__ == '__main__':
	rng = np.random.default_rng(0)
	for name, n, make in (
		('gaussian 10k', 10000, lambda: rng.normal(size=10000)),
		('binary 0.01% 500k', 500000, lambda: (rng.random(500000) < 1e-4).astype(float)),
		('binary 1% 100k', 100000, lambda: (rng.random(100000) < 1e-2).astype(float)),
		('sparse continuous 0.01% 500k', 500000, lambda: np.where(rng.random(500000) < 1e-4, rng.normal(size=500000) + 3, 0.0)),
		('log1p poisson 100k', 100000, lambda: np.log1p(rng.poisson(2, 100000).astype(float))),
	):
		a, b = make(), make()
		a, b = a - a.mean(), b - b.mean()
		res = analyse_pair(a, b)
		dof = n - 2
		print('%-30s r=%+.3e  err kept %.2e  err fixed %.2e  bound %.2e  -> dof*r*bound %.2e   c=%.2e,%.2e g=%.1e' % (
			name, res['r_true'], abs(res['r_kept'] - res['r_true']), abs(res['r_fixed'] - res['r_true']), res['bound'],
			dof * abs(res['r_true']) * res['bound'], res['c'][0], res['c'][1], res['g'][0]))
