#!/usr/bin/env python3
"""G12: P-values of the device's p = I_x(dof/2, 1/2), x = fl(1 - R^2), against mpmath at 60 digits (tests/golden/G12_pvalue_mp.npz).

The reference's own value is scipy.stats.beta.cdf(1 - R2, dof/2, 0.5) (association.py:249; G3 holds a table of it); this grid pins the
device function itself two orders of magnitude tighter than scipy is accurate, over everything the sweeps can meet: dof from 16 to
500 000 cells, P from 1 down to 1e-320, both sides of R^2 = 1/4 (where the device's logarithm changes form) and of u = 1.5 (where
it changes from the series to the continued fraction).  Needs mpmath; a few minutes.

    python3 tests/golden/make_pvalue_grid.py
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))


def pval(r2, dof):
	x = np.float64(1.0) - np.float64(r2)  # the rounding the reference and the device both apply first
	if x <= 0:
		return mp.mpf(0)
	if x >= 1:
		return mp.mpf(1)
	return mp.betainc(mp.mpf(dof) / 2, mp.mpf(1) / 2, 0, mp.mpf(float(x)), regularized=True)


def main():
	dofs = [16.0, 17.0, 30.0, 100.0, 996.0, 9996.0, 49990.0, 99980.0, 499996.0]
	r2s, ds, ps, lps = [], [], [], []
	for dof in dofs:
		alpha = dof / 2 - 0.25
		zs = np.concatenate([np.logspace(-12, 0, 13), np.linspace(1.5, 30, 20), np.linspace(35, 740, 25)])
		grid = [float(-mp.expm1(-mp.mpf(z) / alpha)) for z in zs]  # R^2 with alpha u = z
		grid += [0.2, 0.2499999, 0.25, 0.2500001, 0.3, 0.5, 0.77, 0.7768, 0.78, 0.9, 0.99, 0.999999, 1e-300, 5e-324]
		for r2 in grid:
			if not 0 < r2 < 1:
				continue
			p = pval(r2, dof)
			r2s.append(r2)
			ds.append(dof)
			ps.append(float(p))
			lps.append(float(mp.log10(p)) if p > 0 else -np.inf)
	out = os.path.join(HERE, 'G12_pvalue_mp.npz')
	np.savez_compressed(out, r2=np.array(r2s), dof=np.array(ds), p=np.array(ps), log10p=np.array(lps))
	print(out, len(r2s), 'points; smallest non-zero p', min(p for p in ps if p > 0))


if __name__ == '__main__':
	main()
