"""CPU: the arithmetic of the integer Gram engine's correction and guard (csrc/nrm_fix.h), restated with exact integers in
tools/i8_error_model.py.  What is checked here is the MATHS the kernels implement -- that adding the product of the digit means
removes the coherent part of the dropped digit products, and that the Cauchy-Schwarz bound K c_i c_j + g_i + g_j really bounds
what is left -- on the row types the reference meets (association.py:224-235 computes the same contraction in fp64)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import i8_error_model as model  # noqa: E402


def rows(kind, n, rng):
	if kind == 'gaussian':
		return rng.normal(size=n)
	if kind == 'binary':
		return (rng.random(n) < 3e-3).astype(float)
	if kind == 'sparse_continuous':
		return np.where(rng.random(n) < 3e-3, rng.normal(size=n) + 3, 0.0)
	if kind == 'counts':
		return np.log1p(rng.poisson(0.3, n).astype(float))
	if kind == 'offset':
		return 1e4 + rng.normal(size=n)
	if kind == 'spike':
		x = rng.normal(size=n) * 1e-3
		x[n // 3] = 40.
		return x
	raise ValueError(kind)


@pytest.mark.parametrize('ns', [6, 5])
def test_mean_correction_and_bound(ns):
	rng = np.random.default_rng(5)
	n = 20000
	kinds = ['gaussian', 'binary', 'sparse_continuous', 'counts', 'offset', 'spike']
	few = ('binary', 'sparse_continuous', 'counts')  # rows that take few distinct values after the intercept is removed
	worst_kept, worst_fixed = 0.0, 0.0
	for ka in kinds:
		for kb in kinds:
			a, b = rows(ka, n, rng), rows(kb, n, rng)
			if ka == kb and ka == 'binary':
				b = np.maximum(a, b)  # overlapping sparse rows: a true positive
			a, b = a - a.mean(), b - b.mean()
			res = model.analyse_pair(a, b, ns)
			err_fixed = abs(res['r_fixed'] - res['r_true'])
			# the bound holds (1e-17: the fp64 rounding of r_true itself)
			assert err_fixed <= res['bound'] * (1 + 1e-9) + 1e-16, (ka, kb, err_fixed, res['bound'])
			if ka in few and kb in few:
				worst_kept = max(worst_kept, abs(res['r_kept'] - res['r_true']))
				worst_fixed = max(worst_fixed, err_fixed)
	# the coherent error of few-valued rows is what the correction removes: orders of magnitude at once
	assert worst_kept > 1000 * worst_fixed, (worst_kept, worst_fixed)
	if ns == 6:
		assert worst_kept > 1e-12 and worst_fixed < 2e-15, (worst_kept, worst_fixed)


def test_digits_reassemble():
	rng = np.random.default_rng(6)
	x = rng.normal(size=(3, 500)) * np.array([[1.0], [1e-20], [1e12]])
	q, sh = model.quantise(x)
	d = model.digits(q)
	assert np.array_equal(sum(d[s] << (8 * s) for s in range(6)), q)
	assert all(np.abs(d[s]).max() <= 128 for s in range(5)) and np.abs(d[5]).max() <= 64
	assert np.allclose(np.ldexp(q.astype(float), sh[:, None]), x, rtol=0, atol=np.ldexp(0.5, sh).max())
