"""Seeded random shapes through the device paths built in round 5, against the oracle: de on sparse designs (the library's list kernels and the
one-pass gather kernel), single=1 (the library's cell selection), single=4 on sparse designs, normvar on the device.  Each case is small enough
for the oracle's loops; the shapes wander over what the fixed parity tests pin at a few points -- cell counts off every grid, one to several
passes of design rows, 0 .. 9 covariates with and without an intercept, 0/1 and valued entries, fp32 and fp64 rows, empty design rows."""
import os

import numpy as np
import pytest

import oracle
from conftest import relerr
from test_gpu_parity import close, p_close

pytestmark = pytest.mark.gpu
_MORE = int(os.environ.get('NRM_TEST_SEEDS', '0'))  # a longer one-off sweep: NRM_TEST_SEEDS=300 python -m pytest tests/test_gpu_random_shapes.py


def _case(seed, single=False):
	"""single: a design the reference's single=1 / single=4 take (entries 0 / 1, single=1 asserts it at association.py:914; every grouping with cells; enough cells)."""
	rng = np.random.default_rng(seed)
	nx = int(rng.choice([1, 2, 7, 33, 64, 65, 130]))
	ny = int(rng.integers(1, 70))
	n = int(rng.choice([rng.integers(60, 300), rng.integers(2040, 2060), rng.integers(2100, 7000)]))
	nc = int(rng.choice([0, 1, 2, 3, 5, 9]))
	f32 = bool(rng.integers(0, 2))
	valued = bool(rng.integers(0, 2)) and not single
	if single:
		n = max(n, 6 * (nx + nc))
	dens = float(rng.choice([0.002, 0.01, 0.05]))
	dx = (rng.random((nx, n)) < dens).astype(np.float64)
	if valued:
		dx *= rng.uniform(0.5, 2.0, dx.shape)
	if nx > 2 and not single:
		dx[int(rng.integers(0, nx))] = 0  # a design row without entries
	if single:
		for i in range(nx):  # every grouping has cells
			dx[i, rng.integers(0, n, 4)] = 1
	dx[0, :max(3, n // 50)] = 1.5 if valued else 1.0  # every case has entries; one list much longer than its neighbours'
	dc = rng.normal(size=(nc, n))
	if nc and rng.integers(0, 2):
		dc[int(rng.integers(0, nc))] = rng.uniform(0.5, 2.0)  # an intercept (any constant)
	dy = rng.normal(size=(ny, n)) + rng.uniform(0, 9)
	dy[: min(ny, 3)] += 0.5 * dx[0]
	if f32:
		dx, dy = dx.astype(np.float32), dy.astype(np.float32)
	return dx, dy, dc, f32


def _tols(f32):
	"""(P-values, statistics and variances: relative; floor of the statistic IN UNITS OF PEARSON r).  The north star's bar, 1e-6, for fp32 inputs too: the
	arithmetic is fp64 whatever the input dtype and the outputs are rounded ONCE to the input dtype (6e-8 relative); the oracle runs on the same fp32 values
	upcast.  (Round 5 held fp32 cases to 3e-4 / 3e-5 with an absolute floor of 1e-3 on gamma -- 300x the bar; round-5 verdict, weak item 3.)"""
	return (1e-6, 1e-6, 1e-12) if f32 else (1e-6, 1e-7, 1e-12)


def _r(stat, vx, vy, return_dot):
	"""The statistic as a Pearson r: gamma sqrt(var_x / var_y) (association.py:234-235) or covariance / sqrt(var_x var_y); vy (ny,) or (nx, ny)."""
	stat, vx, vy = np.asarray(stat, dtype=np.float64), np.asarray(vx, dtype=np.float64)[:, None], np.asarray(vy, dtype=np.float64)
	vy = vy[None, :] if vy.ndim == 1 else vy
	return stat / np.sqrt(vx * vy) if return_dot else stat * np.sqrt(vx / vy)


def _stat_close(got, ref, return_dot, rtol, floor):
	"""|r - r_ref| <= rtol |r_ref| + floor, r formed on either side from its own statistic and variances."""
	a, b = _r(got[1], got[3], got[4], return_dot), _r(ref[1], ref[3], ref[4], return_dot)
	return bool((np.abs(a - b) <= rtol * np.abs(b) + floor).all())


@pytest.mark.parametrize('seed', range(24 + _MORE))
def test_de_on_random_sparse_designs(seed, monkeypatch):
	from normalisr_amd.association import association_tests
	dx, dy, dc, f32 = _case(seed)
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')
	lowmem = bool(seed % 2)
	got = association_tests(dx, dy, dc, return_dot=bool(seed % 3), lowmem=lowmem)
	ref = oracle.association_tests(dx.astype(np.float64), dy.astype(np.float64), dc, return_dot=bool(seed % 3), lowmem=lowmem)
	ptol, stol, floor = _tols(f32)
	ok = ref[0] > (1e-30 if f32 else 1e-290)
	assert got[0].shape == ref[0].shape and got[0].dtype == (np.float32 if f32 else np.float64)
	assert relerr(got[0][ok], ref[0][ok]) < ptol, (seed, dx.shape, dy.shape, dc.shape)
	assert _stat_close(got, ref, bool(seed % 3), stol, floor) and close(got[3], ref[3], stol, 1e-15) and close(got[4], ref[4], stol, 1e-15), (seed, dx.shape, dy.shape, dc.shape, f32)
	if not lowmem and dc.shape[0]:
		# alpha = b_y - gamma b_x (association.py:238-243) is a DIFFERENCE: its error is that of its parts, not of itself.  For fp32 inputs the outputs are fp32 and
		# alpha is formed from the rounded gamma, as the reference's fp32 path forms it: 6e-8 of |b_y| + |gamma| |b_x| per rounding.  Held to 1e-6 (1e-9 for fp64
		# inputs) of that scale, with b_x = x C^T (C C^T)^+ taken here in numpy -- under a bound relative to alpha itself (round 5: 3e-4) 32 of 1024 sweep cases
		# with |alpha| << |b_y| fail at 1e-5 with errors of 2-3e-7 |b_y|: the rounding of the output, not a difference between the device and the oracle.
		c64, x64 = dc.astype(np.float64), dx.astype(np.float64)
		bx = (x64 @ c64.T) @ np.linalg.pinv(c64 @ c64.T)
		gam = np.asarray(ref[1], dtype=np.float64) / (np.asarray(ref[3], dtype=np.float64)[:, None] if seed % 3 else 1.0)
		scale = np.abs(ref[2]) + np.abs(gam)[:, :, None] * np.abs(bx)[:, None, :]
		assert (np.abs(got[2].astype(np.float64) - ref[2]) <= (1e-6 if f32 else 1e-9) * scale + 1e-12).all(), (seed, dx.shape, dy.shape, dc.shape, f32)


@pytest.mark.parametrize('seed', range(1000, 1016 + _MORE))
def test_single1_and_single4_on_random_sparse_designs(seed, monkeypatch):
	from normalisr_amd.association import association_tests
	dx, dy, dc, f32 = _case(seed, single=True)
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')
	ptol, stol, floor = _tols(f32)
	for single in (1, 4):
		try:
			ref = oracle.association_tests(dx.astype(np.float64), dy.astype(np.float64), dc, single=single, return_dot=False)
		except Exception as e:  # the reference's own assertions on this draw (e.g. R^2 of a degenerate grouping): the device must raise as well
			with pytest.raises(type(e)):
				association_tests(dx, dy, dc, single=single, return_dot=False)
			continue
		got = association_tests(dx, dy, dc, single=single, return_dot=False)
		ok = ref[0] > (1e-30 if f32 else 1e-290)
		assert relerr(got[0][ok], ref[0][ok]) < ptol, (seed, single, dx.shape, dy.shape, dc.shape)
		assert _stat_close(got, ref, False, stol, floor) and close(got[3], ref[3], stol, 1e-15) and close(got[4], ref[4], stol, 1e-15), (seed, single, dx.shape, dy.shape, dc.shape, f32)


@pytest.mark.parametrize('seed', range(5000, 5012 + _MORE))
def test_normvar_on_random_shapes(seed):
	import normalisr_amd.normalisr as norm
	rng = np.random.default_rng(seed)
	ng, n, nc = int(rng.integers(1, 90)), int(rng.choice([rng.integers(30, 200), rng.integers(1000, 5000)])), int(rng.integers(1, 9))
	f32 = bool(rng.integers(0, 2))
	dt = rng.normal(size=(ng, n)) - 9
	dc = rng.normal(size=(nc, n))
	if rng.integers(0, 2):
		dc[-1] = 1.0
	if nc > 2 and rng.integers(0, 2):
		dc[1] = 2.0 * dc[0]  # rank-deficient covariates: the integer ranks must follow inv_rank's rule
	w, wt = np.exp(0.3 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)
	wt[rng.integers(0, ng)] = 0.0
	if f32:
		dt = dt.astype(np.float32)
	got = norm.normvar(dt, dc, w, wt)
	ref = oracle.normvar(dt.astype(np.float64), dc, w, wt)
	scale = np.abs(ref[0]).max()
	assert np.abs(got[0] - ref[0]).max() < (1e-6 if f32 else 1e-9) * scale and close(got[1], ref[1], 1e-12, 1e-15), (seed, ng, n, nc, f32)


@pytest.mark.parametrize('seed', range(9000, 9016 + _MORE))
def test_coex_and_de_on_random_dense_shapes(seed):
	"""The paths of the earlier rounds under this round's refactors (K1 without its workspace arguments, the streaming sweep without the int8 variant, the
	degenerate-row rule of nrm_residualize_wide): coex and de on dense rows -- fp64 Gram kernel below 2048 cells, integer engine above, streaming de when
	nx + nc <= 32, K1 + K2 otherwise -- against the oracle."""
	import normalisr_amd.normalisr as norm
	rng = np.random.default_rng(seed)
	ng = int(rng.integers(2, 150))
	n = int(rng.choice([rng.integers(40, 400), rng.integers(2048, 2100), rng.integers(2200, 6000)]))
	nc = int(rng.choice([0, 1, 3, 6, 9]))
	nx = int(rng.choice([1, 3, 20, 40]))
	f32 = bool(rng.integers(0, 2))
	dt = np.log1p(rng.poisson(rng.uniform(0.2, 3.0), (ng, n))).astype(np.float64) + 1e-3 * rng.normal(size=(ng, n))
	dc = rng.normal(size=(nc, n))
	if nc and rng.integers(0, 2):
		dc[-1] = 1.0
	dg = (rng.random((nx, n)) < rng.uniform(0.1, 0.6)).astype(np.float64)
	dg[:, :3] = [1, 0, 1]  # (no constant grouping)
	if n <= nc + 3:
		pytest.skip('too few cells')
	if f32:
		dt = dt.astype(np.float32)
	ptol, stol, _ = _tols(f32)
	d64 = dt.astype(np.float64)
	p, dot, var = norm.coex(dt, dc)
	po, do, vo = oracle.coex(d64, dc)
	off = ~np.eye(ng, dtype=bool)
	ok = off & (po > (1e-30 if f32 else 1e-290))
	assert relerr(p[ok], po[ok]) < ptol, (seed, ng, n, nc, f32)
	# (covariances near zero: the integer engine is exact to ~1e-13 of sqrt(var_i var_j) at 2048 cells, DESIGN 4 -- i.e. |delta r| < 1e-12; fp32 outputs round at 6e-8 of the value)
	assert (np.abs(dot - do)[off] <= stol * np.abs(do[off]) + 1e-12 * np.sqrt(np.outer(vo, vo))[off]).all() and close(var, vo, stol, 1e-15), (seed, ng, n, nc, f32)
	assert (p == p.T).all() and (np.diag(p) == 0).all()
	got = norm.de(dg, dt, dc)
	ref = oracle.de(dg, d64, dc)
	ok = ref[0] > (1e-30 if f32 else 1e-290)
	assert relerr(got[0][ok], ref[0][ok]) < ptol, (seed, 'de', nx, ng, n, nc, f32)
	assert _stat_close(got, ref, False, stol, 1e-12), (seed, 'de', nx, ng, n, nc, f32)
	assert close(got[3], ref[3], stol, 1e-15) and close(got[4], ref[4], stol, 1e-15)

