"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors generated from the reference.  Tolerances (BASELINE.json north_star): integers/zeros/shapes
bit-exact; Pearson r, t and p within 1e-6 relative.  A pure relative bound is meaningless where the
true value is ~0 (r of an uncorrelated pair) or subnormal (p < 2.3e-308), so each comparison carries
an explicit absolute floor, stated at the call site."""
import ctypes

import numpy as np
import pytest

import oracle
from conftest import relerr

pytestmark = pytest.mark.gpu

RTOL = 1e-6  # north-star tolerance on r, t, p
R_FLOOR = 1e-12  # |r| below this is rounding noise of an fp64 dot product over <=1e5 cells
# Problems with >= 2048 cells run K2 on the integer engine (46-bit fixed point, csrc/nrm_gram_i8.hip).  Its error in Pearson r
# -- the low-order digit products it drops -- was measured at <= 3e-14 over the shapes tested here (2e-15 typical at 10 000
# Gaussian cells, more for sparse binary designs), so r, t, gamma and the covariance meet 1e-6 relative for |r| >= 1e-7 and
# 1e-13 absolute (in units of r) below; P-values are unaffected (d ln p / d r = -dof r: 4e-10 measured).  NRM_GRAM=f64 selects
# the fp64 matrix-core kernel, which meets R_FLOOR (test_gram_engines_on_config1_shape runs all engines).
I8_FLOOR = 1e-7


def gamma_close(gam, vx, vy, ref, vxr, vyr, floor):
	"""gamma (association.py:234) compared as Pearson r = gamma sqrt(var_x / var_y), the scale the floors above refer to."""
	r = np.asarray(gam, dtype=np.float64) * np.sqrt(np.asarray(vx, dtype=np.float64)[:, None] / np.asarray(vy, dtype=np.float64)[None, :])
	rr = np.asarray(ref, dtype=np.float64) * np.sqrt(np.asarray(vxr, dtype=np.float64)[:, None] / np.asarray(vyr, dtype=np.float64)[None, :])
	return close(r, rr, floor=floor)
P_TINY = 2.3e-308  # below: subnormal, compared absolutely


def close(a, b, rtol=RTOL, floor=0.):
	return relerr(a, b, floor) < rtol


def p_close(p, ref, rtol=RTOL):
	p, ref = np.asarray(p, dtype=np.float64), np.asarray(ref, dtype=np.float64)
	normal = ref >= P_TINY
	ok = relerr(p[normal], ref[normal]) < rtol if normal.any() else True
	return ok and np.all(np.abs(p[~normal] - ref[~normal]) <= 1e-307)


@pytest.fixture(scope='module')
def norm():
	import normalisr_amd.normalisr as norm
	return norm


@pytest.fixture(params=['auto', 'general'])
def de_path(request, monkeypatch):
	"""de runs either through the streaming kernel (K2s, picked automatically for nx + nc <= 32) or the general
	K1 -> K2 -> K3 path; both must meet the same parity bar."""
	monkeypatch.setenv('NRM_DE_PATH', request.param)
	return request.param


@pytest.fixture(scope='module')
def eng():
	from normalisr_amd.engine import get_engine
	return get_engine()


def test_pvalue_kernel_table(golden, eng):
	"""K3 arithmetic alone against scipy's table (G3) and the oracle, all dof regimes incl. dof < 16."""
	import torch
	from normalisr_amd import _lib
	g = golden('G3_ptable')
	dof, r2, ref = g['dof'], g['r2'], g['p']
	d_r2 = torch.from_numpy(r2).cuda()
	d_p = torch.empty_like(d_r2)
	worst = 0.
	for i, d in enumerate(dof):
		_lib.check(eng.lib.nrm_pvalues_from_r2(d_r2.data_ptr(), r2.size, float(d), d_p.data_ptr(), 0))
		torch.cuda.synchronize()
		p = d_p.cpu().numpy()
		normal = ref[i] >= P_TINY
		worst = max(worst, relerr(p[normal], ref[i][normal]))
		assert np.all(np.abs(p[~normal] - ref[i][~normal]) <= 1e-307)
		assert p[r2 == 0][0] == 1.0 and p[r2 == 1][0] == 0.0
		assert relerr(p[normal], oracle.pvalues(r2, d)[normal]) < 1e-9
	assert worst < 1e-9, worst  # far inside the 1e-6 north-star bound
	# R^2 a hair above 1 (rounding) clips to p = 0; NaN propagates
	x = torch.tensor([1 + 5e-9, float('nan'), 0.5], dtype=torch.float64).cuda()
	o = torch.empty_like(x)
	_lib.check(eng.lib.nrm_pvalues_from_r2(x.data_ptr(), 3, 100., o.data_ptr(), 0))
	o = o.cpu().numpy()
	assert o[0] == 0. and np.isnan(o[1]) and 0 < o[2] < 1


def test_gram_kernel_layout(eng):
	"""K2 alone with exact small-integer data and asymmetric operands (catches row/col swaps of the MFMA maps)."""
	import torch
	from normalisr_amd.engine import Residualized
	rng = np.random.default_rng(11)
	a = rng.integers(-3, 4, (256, 48)).astype(np.float64)
	b = rng.integers(-3, 4, (128, 48)).astype(np.float64)
	A = Residualized(256, 48, torch.from_numpy(a).cuda(), None, None)
	B = Residualized(128, 48, torch.from_numpy(b).cuda(), None, None)
	dot = eng.gram(A, B, False).cpu().numpy()
	assert np.array_equal(dot, a @ b.T)
	dot = eng.gram(A, A, True).cpu().numpy()
	ref = a @ a.T
	iu = np.triu_indices(256)  # symmetric launches guarantee the upper triangle only (16x16 sub-block granularity)
	assert np.array_equal(dot[iu], ref[iu])
	# valid-row counts: padding sub-blocks are skipped, everything inside the valid range is exact
	A2 = Residualized(200, 48, A.data, None, None)
	B2 = Residualized(70, 48, B.data, None, None)
	dot = eng.gram(A2, B2, False).cpu().numpy()
	assert np.array_equal(dot[:200, :70], (a @ b.T)[:200, :70])


def test_block_g7(golden, de_path):
	from normalisr_amd.association import association_test_1
	g = golden('G7_block')
	r = association_test_1(3, 5, g['dx'], g['dy'], g['dc'], g['dci'], int(g['dcr']), lowmem=False, return_stats=True)
	assert r[0] == 3 and r[1] == 5
	assert p_close(r[2], g['p']) and close(r[3], g['gamma'], floor=1e-12) and close(r[4], g['alpha'], floor=1e-10)
	assert close(r[5], g['vx'], 1e-12) and close(r[6], g['vy'], 1e-12)
	dof = 1000 - 1 - int(g['dcr'])
	rr, tt = oracle.pearson_r_t((g['gamma'].T * g['vx']).T, g['vx'], g['vy'], dof)
	assert close(r[7], rr, floor=R_FLOOR) and close(r[8], tt, floor=1e-9)


def test_c1_de_coex_golden(golden, norm, de_path):
	g = golden('G1_c1')
	dt, dc, dg = g['dt'], g['dc'], g['dg']
	for lm in (1, 0):
		p, gam, a, vg, vt = norm.de(dg, dt, dc, lowmem=bool(lm))
		k = 'de_lm{}_'.format(lm)
		assert p.shape == (4, 500) and p.dtype == np.float64
		assert p_close(p, g[k + 'p']) and close(gam, g[k + 'gamma'], floor=1e-12)
		assert close(vg, g[k + 'varg'], 1e-12, 1e-300) and close(vt, g[k + 'vart'], 1e-12, 1e-300)
		assert (a is None) if lm else close(a, g[k + 'alpha'], floor=1e-10)
		# constant grouping row re-inflated exactly (de.py:107-122)
		assert (p[2] == 1).all() and (gam[2] == 0).all() and vg[2] == 0 and (vt[2] == 0).all()
	ns = int(g['coex_n'])
	p, d, v = norm.coex(dt[:ns], dc)
	assert p_close(p, g['coex_p']) and close(d, g['coex_dot'], floor=1e-13) and close(v, g['coex_var'], 1e-12)
	# bit-exact structure: zero diagonals and exact symmetry (association.py:1050-1057)
	assert (np.diag(p) == 0).all() and (np.diag(d) == 0).all() and (p == p.T).all() and (d == d.T).all()
	from normalisr_amd.association import association_tests
	p, d, a, vx, vy = association_tests(dg[[0, 1, 3]], dt[:64], dc, return_dot=True)
	assert p_close(p, g['at_p']) and close(d, g['at_dot'], floor=1e-13) and close(vx, g['at_vx'], 1e-12) and a is None


def test_edge_cases_golden(golden, norm, de_path):
	from normalisr_amd.association import association_tests
	g = golden('G2_edge')
	dt, dc, dg = g['dt'], g['dc'], g['dg']
	n = dt.shape[1]
	p, gam, a, vg, vt = norm.de(dg, dt, np.zeros((0, n)))
	assert p_close(p, g['nc0_de_p']) and close(gam, g['nc0_de_gamma'], floor=1e-12) and close(vt, g['nc0_de_vart'], 1e-12)
	p, d, v = norm.coex(dt[:40], np.zeros((0, n)))
	assert p_close(p, g['nc0_coex_p']) and close(d, g['nc0_coex_dot'], floor=1e-13)
	p, gam, a, vg, vt = norm.de(dg, dt, g['rd_dc'], lowmem=False)  # rank 3 of 5
	assert p_close(p, g['rd_de_p']) and close(gam, g['rd_de_gamma'], floor=1e-12) and close(vt, g['rd_de_vart'], 1e-9)
	p, d, v = norm.coex(dt[:40], g['rd_dc'])
	assert p_close(p, g['rd_coex_p']) and close(v, g['rd_coex_var'], 1e-9)
	p, gam, a, vg, vt = norm.de(dg, dt, dc, dimreduce=2)
	assert p_close(p, g['dr2_de_p'])
	p, d, v = norm.coex(dt[:40], dc, dimreduce=2)
	assert p_close(p, g['dr2_coex_p'])
	p, gam, a, vg, vt = norm.de(dg.astype(np.int64), dt, dc)
	assert p_close(p, g['int_de_p']) and p.dtype == np.float64 and close(vg, g['int_de_varg'], 1e-12)
	p, d, a, vx, vy = association_tests(dg, dt, dc, bsx=2, bsy=13)
	assert p_close(p, g['tile_at_p']) and close(d, g['tile_at_dot'], floor=1e-13)
	p, d, a, vx, vy = association_tests(dt[:45], None, dc, bsx=7)
	assert p_close(p, g['tile_coex_p']) and vx is None and close(vy, g['tile_coex_vy'], 1e-12)
	p, d, v = norm.coex(dt[:45], dc, bs=7)  # `bs` alias accepted (the reference crashes here)
	assert p_close(p, g['tile_coex_p'])
	# zero-variance gene: variance reported as 1, p = 1, dot = 0 (association.py:231-233)
	dz = g['zc_dt']
	p, d, v = norm.coex(dz, dc)
	assert v[5] == 1 and (p[5] == np.where(np.arange(30) == 5, 0, 1)).all() and (d[5] == 0).all()
	assert p[6, 7] == 0. and p[7, 6] == 0.  # collinear pair: R^2 -> 1 => p = 0
	ok = np.ones_like(p, dtype=bool)
	ok[6, 7] = ok[7, 6] = False
	assert p_close(p[ok], g['zc_coex_p'][ok])
	p, gam, a, vg, vt = norm.de(dg, dz, dc)
	assert p_close(p, g['zc_de_p']) and close(vt, g['zc_de_vart'], 1e-12, 1e-300)
	# fp32 inputs: oracle = reference run in fp64 on the same fp32-representable values
	dt32, dc32, dg32 = dt.astype(np.float32), dc.astype(np.float32), dg.astype(np.float32)
	p, gam, a, vg, vt = norm.de(dg32, dt32, dc32)
	assert p.dtype == np.float32 and gam.dtype == np.float32 and vt.dtype == np.float32
	assert close(p, g['f32_de_p'], 1e-6, 1e-38) and close(gam, g['f32_de_gamma'], 1e-6, 1e-7)
	p, d, v = norm.coex(dt32[:40], dc32)
	assert p.dtype == np.float32 and close(p, g['f32_coex_p'], 1e-6, 1e-38) and close(v, g['f32_coex_var'], 1e-6)
	# strong effects: tiny p-values keep their relative accuracy
	p, gam, a, vg, vt = norm.de(dg, g['se_dt'], dc)
	assert p_close(p, g['se_de_p']) and g['se_de_p'].min() < 1e-20


def _synthetic(seed, nx, ny, n, nc, dtype=np.float64):
	rng = np.random.default_rng(seed)
	lat = rng.normal(size=(2, n))
	dy = rng.normal(size=(ny, n)) + 0.3 * rng.normal(size=(ny, 2)) @ lat + rng.normal(size=(ny, 1)) * 3
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	dx = None
	if nx:
		dx = (rng.random((nx, n)) < 0.2).astype(np.float64) + 0.1 * lat[:1]
	cast = lambda a: None if a is None else a.astype(dtype)
	return cast(dx), cast(dy), cast(dc)


@pytest.mark.parametrize('ng,n,nc', [(300, 1000, 3), (517, 2049, 4), (129, 333, 1), (64, 5000, 0)])
def test_coex_vs_oracle_seeded(norm, ng, n, nc):
	from normalisr_amd.association import association_tests
	_, dt, dc = _synthetic(100 + ng, 0, ng, n, nc)
	p, d, a, vx, v, st = association_tests(dt, None, dc, return_stats=True)
	po, do, vo = oracle.coex(dt, dc)
	assert p_close(p, po) and close(d, do, floor=1e-13) and close(v, vo, 1e-12)
	assert (np.diag(p) == 0).all() and (p == p.T).all() and (d == d.T).all()
	dof = st['dof']
	assert dof == n - 1 - min(nc, np.linalg.matrix_rank(dc) if nc else 0)
	rr, tt = oracle.pearson_r_t(do, vo, vo, dof)
	off = ~np.eye(ng, dtype=bool)
	assert close(st['r'][off], rr[off], floor=R_FLOOR) and close(st['t'][off], tt[off], floor=1e-9)


@pytest.mark.parametrize('nx,ny,n,nc', [(5, 700, 1500, 3), (130, 260, 800, 2), (1, 1000, 4097, 20), (2, 777, 4100, 20), (12, 300, 10000, 0)])
def test_de_vs_oracle_seeded(norm, de_path, nx, ny, n, nc):
	dg, dt, dc = _synthetic(200 + nx, nx, ny, n, nc)
	p, gam, al, vg, vt = norm.de(dg, dt, dc, lowmem=False)
	po, go, ao, vgo, vto = oracle.de(dg, dt, dc, lowmem=False)
	assert p_close(p, po) and close(gam, go, floor=1e-12) and close(al, ao, floor=1e-9)
	assert close(vg, vgo, 1e-12) and close(vt, vto, 1e-12)


def test_fp32_inputs_vs_fp64_oracle(norm, de_path):
	dg, dt, dc = _synthetic(7, 3, 400, 3000, 3, np.float32)
	p, gam, al, vg, vt = norm.de(dg, dt, dc)
	po, go, ao, vgo, vto = oracle.de(dg.astype(np.float64), dt.astype(np.float64), dc.astype(np.float64))
	assert p.dtype == np.float32
	# fp64 arithmetic inside, one final rounding to fp32 (6e-8): bound 1e-6 with fp32 floors
	assert close(p, po, 1e-6, 1e-38) and close(gam, go, 1e-6, 1e-7) and close(vt, vto, 1e-6)


def test_host_entry_numpy_in_out(eng):
	"""nrm_association_tests_host: the whole-problem C entry without any torch plumbing."""
	from normalisr_amd import _lib
	dg, dt, dc = _synthetic(9, 6, 150, 700, 3)
	dci, rank = oracle.inv_rank(dc @ dc.T)
	nx, ny, n, nc = 6, 150, 700, 3
	p = np.empty((nx, ny))
	gam = np.empty((nx, ny))
	al = np.empty((nx, ny, nc))
	vx = np.empty(nx)
	vy = np.empty(ny)
	r = np.empty((nx, ny))
	t = np.empty((nx, ny))
	vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
	_lib.check(eng.lib.nrm_association_tests_host(vp(dg), 1, nx, vp(dt), 1, ny, vp(dc), 1, nc, n, vp(dci), rank, 0, 0,
												  vp(p), vp(gam), vp(al), vp(vx), vp(vy), vp(r), vp(t), 1))
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dt, dc, lowmem=False, return_dot=False)
	assert p_close(p, po) and close(gam, go, floor=1e-12) and close(al, ao, floor=1e-9)
	assert close(vx, vxo, 1e-12) and close(vy, vyo, 1e-12)
	# coex through the same entry (dy = NULL), fp32 in/out
	dt32 = dt[:90].astype(np.float32)
	p32 = np.empty((90, 90), dtype=np.float32)
	d32 = np.empty((90, 90), dtype=np.float32)
	v32 = np.empty(90, dtype=np.float32)
	_lib.check(eng.lib.nrm_association_tests_host(vp(dt32), 0, 90, None, 0, 0, vp(dc), 1, nc, n, vp(dci), rank, 0, 1,
												  vp(p32), vp(d32), None, None, vp(v32), None, None, 0))
	po, do, vo = oracle.coex(dt32.astype(np.float64), dc)
	assert close(p32, po, 1e-6, 1e-38) and close(d32, do, 1e-6, 1e-7) and (np.diag(p32) == 0).all()
	# several bands of output rows, page-locked result arrays (coex 2500 genes, fp32 in/out) against the Python engine path
	rng = np.random.default_rng(31)
	big = rng.standard_normal((2500, n), dtype=np.float32) + 0.4 * rng.standard_normal((2500, 1), dtype=np.float32) * rng.standard_normal((1, n), dtype=np.float32)
	pb, db, vb = np.empty((2500, 2500), np.float32), np.empty((2500, 2500), np.float32), np.empty(2500, np.float32)
	_lib.check(eng.lib.nrm_association_tests_host(vp(big), 0, 2500, None, 0, 0, vp(dc), 1, nc, n, vp(dci), rank, 0, 1,
												  vp(pb), vp(db), None, None, vp(vb), None, None, 0))
	res = eng.association_single0(big, None, dc, dci, rank, 0, True, False, np.float32)
	assert np.array_equal(pb, res['p']) and np.array_equal(db, res['stat']) and np.array_equal(vb, res['vary'])
	assert (pb == pb.T).all() and (np.diag(pb) == 0).all()
	# error mapping: too few cells -> ValueError like association.py:213-216
	with pytest.raises(ValueError):
		_lib.check(eng.lib.nrm_association_tests_host(vp(dg), 1, nx, vp(dt), 1, ny, vp(dc), 1, nc, 4, vp(dci), rank, 0, 0,
													  vp(p), vp(gam), None, vp(vx), vp(vy), None, None, 1))


def test_nonfinite_input_raises_assertion(norm):
	_, dt, dc = _synthetic(5, 0, 70, 300, 2)
	dt[3, 7] = np.nan
	with pytest.raises(AssertionError):  # association.py:248,252
		norm.coex(dt, dc)


def test_full_size_c2_properties(norm):
	"""BASELINE config 2 (coex 5k genes x 10k cells, fp32 in): size-independent properties plus a
	sampled comparison against the oracle on the sampled genes."""
	rng = np.random.default_rng(2)
	ng, n = 5000, 10000
	lat = rng.normal(size=(1, n)).astype(np.float32)
	dt = rng.normal(size=(ng, n)).astype(np.float32) + 0.3 * rng.normal(size=(ng, 1)).astype(np.float32) * lat
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))]).astype(np.float32)
	p, d, v = norm.coex(dt, dc)
	assert p.shape == (ng, ng) and p.dtype == np.float32 and v.shape == (ng, )
	assert (np.diag(p) == 0).all() and (np.diag(d) == 0).all()
	assert (p == p.T).all() and (d == d.T).all()
	assert (p >= 0).all() and (p <= 1).all() and np.isfinite(d).all() and (v > 0).all()
	idx = np.sort(rng.choice(ng, 48, replace=False))
	po, do, vo = oracle.coex(dt[idx].astype(np.float64), dc.astype(np.float64))
	sub = np.ix_(idx, idx)
	assert close(p[sub], po, 1e-6, 1e-38) and close(d[sub], do, 1e-6, 1e-7) and close(v[idx], vo, 1e-6)
	# linearity: scaling a gene leaves p unchanged and scales dot
	dt2 = dt[idx].copy()
	dt2[0] *= 4
	p2, d2, v2 = norm.coex(dt2, dc)
	assert close(p2, p[sub], 1e-5, 1e-38) and close(d2[0, 1:], 4 * d[sub][0, 1:], 1e-5, 1e-7)


def test_single4_golden_and_oracle(golden, norm):
	"""single=4 (other groupings as covariates): closed-form device path vs the reference's per-grouping SVD loop."""
	g = golden('G5_single')
	p, gam, a, vg, vt = norm.de(g['dg'], g['dt'], g['dc'], single=4, lowmem=False)
	assert p.shape == (12, 40) and vt.shape == (12, 40) and a.shape == (12, 40, 3)
	assert p_close(p, g['s4_p']) and close(gam, g['s4_gamma'], floor=1e-12) and close(a, g['s4_alpha'], floor=1e-10)
	assert close(vg, g['s4_varg'], 1e-9) and close(vt, g['s4_vart'], 1e-9)
	# larger seeded case with a constant grouping row (dropped by de) and return_dot through association_tests
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(31)
	nx, ny, n = 70, 300, 2500
	dg = (rng.random((nx, n)) < 0.05).astype(np.float64)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dt = rng.normal(size=(ny, n)) + (rng.normal(size=(ny, 8)) @ dg[:8]) * 0.7
	p, d, a, vx, vy = association_tests(dg, dt, dc, single=4, return_dot=True)
	po, do, ao, vxo, vyo = oracle.association_tests(dg, dt, dc, single=4, return_dot=True)
	assert p_close(p, po) and close(d, do, floor=1e-12) and close(vx, vxo, 1e-9) and close(vy, vyo, 1e-9) and a is None
	assert po.min() < 1e-30
	# rank-deficient A A^T (duplicated covariate row): truncated pseudo-inverses, rank m-2 -> host fallback on device Gram matrices
	dc2 = np.vstack([dc, dc[0]])
	p, gam, a, vg, vt = norm.de(dg[:7], dt[:50], dc2, single=4, lowmem=False)
	po, go, ao, vgo, vto = oracle.de(dg[:7], dt[:50], dc2, single=4, lowmem=False)
	assert p_close(p, po) and close(gam, go, floor=1e-10) and close(vt, vto, 1e-8) and close(a, ao, 1e-5, 1e-8)


def test_cli_round_trip_golden(golden, tmp_path):
	"""`normalisr de|coex` on the TSV fixtures of G6: text outputs equal to the reference CLI's up to the
	'%.8G' print precision."""
	import gzip
	from normalisr_amd.__main__ import main
	g = golden('G6_cli')
	files = {k: bytes(g[k]) for k in g.files}
	for name in ('g_tsv', 'e_tsv_gz', 'c_tsv'):
		raw = files[name]
		fn = tmp_path / name.replace('_tsv', '.tsv').replace('_gz', '.gz')
		fn.write_bytes(gzip.compress(raw) if name.endswith('gz') else raw)
	cwd = str(tmp_path)
	import os
	old = os.getcwd()
	os.chdir(cwd)
	try:
		assert main(['de', 'g.tsv', 'e.tsv.gz', 'c.tsv', 'pv.tsv', 'lfc.tsv', '--vard_out', 'vard.tsv', '--vart_out', 'vart.tsv', '-n', '1']) == 0
		assert main(['de', '-m', 'covariate', 'g.tsv', 'e.tsv.gz', 'c.tsv', 'pv4.tsv', 'lfc4.tsv', '-n', '1']) == 0
		assert main(['coex', 'e.tsv.gz', 'c.tsv', 'cpv.tsv.gz', '--var_out', 'cvar.tsv', '--dot_out', 'cdot.tsv', '-n', '1', '-d', '1', '-b', '5']) == 0
		assert main(['de', 'g.tsv', 'e.tsv.gz', 'c.tsv', 'pv2.tsv', 'lfc2.tsv', '--clfc_out', 'clfc.tsv']) == 0  # crashes in the reference (Q9)
	finally:
		os.chdir(old)
	load = lambda raw: np.loadtxt(raw.decode().splitlines(), delimiter='\t', ndmin=2)
	for mine, ref in (('pv.tsv', 'pv_tsv'), ('lfc.tsv', 'lfc_tsv'), ('vard.tsv', 'vard_tsv'), ('vart.tsv', 'vart_tsv'),
					  ('pv4.tsv', 'pv4_tsv'), ('lfc4.tsv', 'lfc4_tsv'), ('cpv.tsv.gz', 'cpv_tsv_gz'), ('cvar.tsv', 'cvar_tsv'),
					  ('cdot.tsv', 'cdot_tsv')):
		got = np.loadtxt(str(tmp_path / mine), delimiter='\t', ndmin=2)
		exp = load(files[ref])
		assert got.shape == exp.shape, mine
		assert relerr(got, exp, 1e-12) < 2e-7, mine  # '%.8G' keeps 8 significant digits
	assert np.loadtxt(str(tmp_path / 'clfc.tsv'), delimiter='\t', ndmin=2).shape == (3, 14 * 2)
	# coex -> binnet through files (run.py:313-321 writes '%i')
	os.chdir(cwd)
	try:
		assert main(['binnet', 'cpv.tsv.gz', 'net.tsv', '0.5']) == 0
	finally:
		os.chdir(old)
	net = np.loadtxt(str(tmp_path / 'net.tsv'), delimiter='\t', ndmin=2)
	assert np.array_equal(net.astype(bool), oracle.binnet(load(files['cpv_tsv_gz']), 0.5))
	assert main([]) == 1


def _sharded_worker(rank, world, port, q, dtype='float32'):
	import os
	import sys
	import torch
	import torch.distributed as dist
	from conftest import ROOT
	sys.path.insert(0, ROOT)
	from normalisr_amd.distributed import CoexPlan
	dist.init_process_group('gloo', init_method='tcp://127.0.0.1:{}'.format(port), rank=rank, world_size=world)
	torch.cuda.set_device(0)
	rng = np.random.default_rng(77)
	ng, n = 150 * world, 700
	dt = (rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))).astype(dtype)
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))]).astype(dtype)
	R = ng // world
	if world == 2:
		plan = CoexPlan(torch.from_numpy(dt[rank * R:(rank + 1) * R]).cuda(), torch.from_numpy(dc).cuda(), rank=rank, world=world,
						group=dist.group.WORLD)
		assert plan.exchange_raw == (dtype == 'float32')  # fp64 rows: the fp64 residual blocks travel, per-pair launches
		plan.step()
		res = plan.assemble()
	else:  # the public wrappers, numpy rows in
		from normalisr_amd.distributed import coex as coex_sharded, de as de_sharded, coex_binnet
		res = coex_sharded(dt[rank * R:(rank + 1) * R], dc)
		if world == 3:
			dg = (np.random.default_rng(78).random((5, n)) < 0.3).astype(dtype)
			dg[2] = 1  # constant grouping: not tested, re-inflated (de.py:107-122)
			# ragged gene blocks: no collective on the de data path, so the ranks need not own equal shares
			cuts = [0, 100, 290, ng]
			rde = de_sharded(dg, dt[cuts[rank]:cuts[rank + 1]], dc)
			if rank == 0:
				res = res + (rde, )
		if world == 4:  # coex -> binnet with the P-values staying in HBM on every rank
			net = coex_binnet(dt[rank * R:(rank + 1) * R], dc, 0.3)
			if rank == 0:
				res = res + (net, )
	if rank == 0:
		q.put(tuple(None if a is None else (tuple(None if b is None else np.array(b) for b in a) if isinstance(a, tuple) else np.array(a)) for a in res))
	dist.barrier()
	dist.destroy_process_group()


@pytest.mark.parametrize('world,dtype', [(2, 'float32'), (3, 'float32'), (5, 'float32'), (8, 'float32'), (2, 'float64'), (4, 'float64'),
										 (8, 'float64'), (4, 'float32')])
def test_sharded_coex_hip_backend_two_ranks_one_gpu(world, dtype):
	"""The N>1 path with the real HIP backend: `world` processes share this box's single GPU and exchange
	blocks over gloo (RCCL needs one GPU per rank); result must equal the single-process oracle.  fp32 rows travel raw
	(partners residualised again, one merged launch), fp64 rows travel as fp64 residual blocks with per-pair launches
	and the half-split pair of an even world -- the BASELINE configs[4] code path."""
	import socket
	import torch.multiprocessing as mp
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, dtype)) for r in range(world)]
	for p in procs:
		p.start()
	from conftest import queue_get
	got = queue_get(q, procs)
	P, D, V = got[:3]
	for p in procs:
		p.join(timeout=120)
		assert p.exitcode == 0
	rng = np.random.default_rng(77)
	ng, n = 150 * world, 700
	dt = (rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))).astype(dtype)
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))]).astype(dtype)
	po, do, vo = oracle.coex(dt.astype(np.float64), dc.astype(np.float64))
	assert P.dtype == np.dtype(dtype)
	if dtype == 'float32':
		assert close(P, po, 1e-6, 1e-38) and close(D, do, 1e-6, 1e-7) and close(V, vo, 1e-6)
	else:
		assert p_close(P, po) and close(D, do, floor=1e-12) and close(V, vo, 1e-12)
	assert (np.diag(P) == 0).all() and (P == P.T).all() and (D == D.T).all()
	if world == 4:  # sharded coex -> binnet: bit-exact booleans (the network is not symmetric: BH runs per row)
		assert got[3].dtype == np.bool_ and np.array_equal(got[3], oracle.binnet(P, 0.3))
	if world == 3:  # sharded de through the public wrapper against the single-process oracle
		dg = (np.random.default_rng(78).random((5, n)) < 0.3).astype(dtype)
		dg[2] = 1
		pd, gd, ad, vgd, vtd = got[3]
		po, go, ao, vgo, vto = oracle.de(dg.astype(np.float64), dt.astype(np.float64), dc.astype(np.float64))
		assert ad is None and close(pd, po, 1e-6, 1e-38) and close(gd, go, 1e-6, 1e-7) and close(vgd, vgo, 1e-6, 1e-12) and close(vtd, vto, 1e-6, 1e-12)
		assert (pd[2] == 1).all() and (gd[2] == 0).all() and vgd[2] == 0


def test_randomised_shapes_vs_oracle(norm):
	"""Seeded sweep over awkward shapes (cells not a multiple of the K tile, single rows, many covariates, rank-deficient
	covariates, mixed dtypes) through de and coex, both de paths; every case against the oracle."""
	import os
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(2026)
	cases = []
	for _ in range(28):
		nx = int(rng.choice([1, 2, 3, 17, 40]))
		ny = int(rng.choice([1, 5, 63, 129, 300]))
		n = int(rng.choice([40, 97, 256, 1001, 2048, 3333]))
		nc = int(rng.choice([0, 1, 2, 7, 19, 33]))
		if n <= nc + 3:
			continue
		cases.append((nx, ny, n, nc, bool(rng.integers(2)), bool(rng.integers(2)), int(rng.integers(1 << 30))))
	assert len(cases) > 20
	for nx, ny, n, nc, f32, dup, seed in cases:
		r = np.random.default_rng(seed)
		lat = r.normal(size=(1, n))
		dy = r.normal(size=(ny, n)) * r.uniform(0.5, 3, (ny, 1)) + 0.4 * r.normal(size=(ny, 1)) * lat + r.normal(size=(ny, 1)) * 4
		dx = (r.random((nx, n)) < 0.3).astype(np.float64) + 0.05 * lat
		dc = np.vstack([r.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
		if dup and nc >= 3:
			dc[1] = dc[0] * 2  # rank deficient
		if f32:
			dy, dx, dc = dy.astype(np.float32), dx.astype(np.float32), dc.astype(np.float32)
		up = lambda a: a.astype(np.float64)
		po, go, ao, vxo, vyo = oracle.association_tests(up(dx), up(dy), up(dc), return_dot=False, lowmem=False)
		tol = dict(rtol=1e-6, floor=1e-38) if f32 else dict(rtol=RTOL, floor=0.)
		for path in ('auto', 'general'):
			os.environ['NRM_DE_PATH'] = path
			try:
				p, g, a, vx, vy = association_tests(dx, dy, dc, return_dot=False, lowmem=False)
			finally:
				os.environ.pop('NRM_DE_PATH', None)
			tag = (nx, ny, n, nc, f32, dup, path)
			if f32:
				assert close(p, po, 1e-6, 1e-38), tag
				assert close(g, go, 1e-6, 1e-6) and close(vy, vyo, 1e-6) and close(vx, vxo, 1e-6), tag
			else:
				assert p_close(p, po), tag
				assert (gamma_close(g, vx, vy, go, vxo, vyo, I8_FLOOR) if n >= 2048 else close(g, go, floor=1e-11)), tag
				assert close(vy, vyo, 1e-9) and close(vx, vxo, 1e-9) and close(a, ao, 1e-6, 1e-8), tag
		if ny > 1:
			p, d, v = norm.coex(dy, dc)
			pc, dcov, vc = oracle.coex(up(dy), up(dc))
			if f32:
				assert close(p, pc, 1e-6, 1e-38) and close(d, dcov, 1e-6, 1e-6), (ny, n, nc, f32, dup)
			else:
				sc = np.sqrt(np.outer(vc, vc)) if n >= 2048 else 1.0
				assert p_close(p, pc) and close(d / sc, dcov / sc, floor=I8_FLOOR if n >= 2048 else 1e-12), (ny, n, nc, f32, dup)
			assert (np.diag(p) == 0).all() and (p == p.T).all()


def test_single1_golden_and_oracle(golden, norm):
	"""single=1 (each grouping on the cells that carry no other grouping): two Gram contractions + sweep vs the
	reference's per-grouping loop (association.py:263-390)."""
	g = golden('G5_single')
	p, gam, a, vg, vt = norm.de(g['s1_dg'], g['dt'], g['dc'], single=1, lowmem=False)
	assert p.shape == (6, 40) and vt.shape == (6, 40) and a.shape == (6, 40, 3)
	assert p_close(p, g['s1_p']) and close(gam, g['s1_gamma'], floor=1e-12) and close(a, g['s1_alpha'], floor=1e-10)
	assert close(vg, g['s1_varg'], 1e-9) and close(vt, g['s1_vart'], 1e-9)
	# larger seeded low-MOI design, fp32 expression, chunked groupings, return_dot
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(41)
	nx, ny, n = 300, 200, 6000
	lab = rng.integers(0, nx + 60, n)  # labels >= nx: cells without any grouping
	dg = np.zeros((nx, n))
	dg[lab[lab < nx], np.nonzero(lab < nx)[0]] = 1
	extra = rng.choice(n, 400, replace=False)  # some doubly-infected cells: excluded for both groupings
	dg[rng.integers(0, nx, 400), extra] = 1
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dt = (rng.normal(size=(ny, n)) + (rng.normal(size=(ny, 10)) @ dg[:10]) * 1.5).astype(np.float32)
	p, d, a, vx, vy = association_tests(dg, dt, dc, single=1, return_dot=True)
	po, do, ao, vxo, vyo = oracle.association_tests(dg, dt.astype(np.float64), dc, single=1, return_dot=True)
	assert p.dtype == np.float32 and p.shape == (nx, ny) and a is None
	assert close(p, po, 1e-6, 1e-38) and close(d, do, 1e-6, 1e-7) and close(vx, vxo, 1e-6) and close(vy, vyo, 1e-6)
	with pytest.raises(AssertionError):  # a grouping that is constant on its own cells (association.py:917-918)
		association_tests(np.vstack([dg[:3], np.zeros((1, n))]), dt[:5], dc, single=1)


def test_binnet_golden_and_oracle(golden, norm):
	"""binnet on the device (per-row BH threshold by counting passes, no sort): booleans bit-exact vs the reference."""
	g = golden('G8_binnet')
	for k, q in (('net_q5', 0.05), ('net_q20', 0.2), ('net_q50', 0.5)):
		assert np.array_equal(norm.binnet(g['p'], q), g[k])
	for k, q in (('net32_q5', 0.05), ('net32_q30', 0.3)):
		assert np.array_equal(norm.binnet(g['p32'], q), g[k])
	for k, q in (('nett_q10', 0.1), ('nett_q25', 0.25)):
		assert np.array_equal(norm.binnet(g['pt'], q), g[k])  # tie-heavy, values on exact boundaries
	# coex -> binnet pipeline on seeded data (p-matrix from the device), fp64 and fp32, several cutoffs
	rng = np.random.default_rng(88)
	n, ng = 800, 700
	lat = rng.normal(size=(4, n))
	dt = rng.normal(size=(ng, n)) + (rng.normal(size=(ng, 4)) * (rng.random((ng, 4)) < 0.2)) @ lat
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	for dtype in (np.float64, np.float32):
		p = norm.coex(dt.astype(dtype), dc.astype(dtype))[0]
		for q in (0.01, 0.1, 0.37):
			net = norm.binnet(p, q)
			assert net.dtype == bool and not net.diagonal().any()
			assert np.array_equal(net, oracle.binnet(p, q)), (dtype, q)
	# device pipeline: the p-matrix never leaves HBM
	pd, dd, vd = norm.coex(dt, dc, device_out=True)
	assert pd.is_cuda and np.array_equal(pd.cpu().numpy(), norm.coex(dt, dc)[0])
	netd = norm.binnet(pd, 0.1)
	assert netd.is_cuda and np.array_equal(netd.cpu().numpy(), oracle.binnet(pd.cpu().numpy(), 0.1))
	# heavy ties + zeros + ones; boundary-hugging cutoffs
	pt = np.round(rng.random((300, 300))**4, 3)
	pt = np.triu(pt, 1) + np.triu(pt, 1).T
	for q in (0.003, 0.05, 0.25, 0.9):
		assert np.array_equal(norm.binnet(pt, q), oracle.binnet(pt, q)), q
	# wide rows: > 8192 genes uses the 96-value register path (fp32) / the L2 re-read path (fp64)
	ngw = 8300
	pw = rng.random((ngw, ngw))**3
	pw = np.triu(pw, 1) + np.triu(pw, 1).T
	rows = rng.choice(ngw, 40, replace=False)
	for dtype in (np.float32, np.float64):
		got = norm.binnet(pw.astype(dtype), 0.07)
		pwd = pw.astype(dtype)
		for i in rows:
			off = np.arange(ngw) != i
			exp = np.zeros(ngw, dtype=bool)
			exp[off] = oracle.bh(pwd[i, off]) <= (np.float32(0.07) if dtype == np.float32 else 0.07)
			assert np.array_equal(got[i], exp), (dtype, i)
	with pytest.raises(RuntimeError):
		norm.binnet(np.ones((20, 20)), 0.01)  # nothing survives: "Empty binary network."
	with pytest.raises(ValueError):
		norm.binnet(np.ones((20, 20)), 1.5)
	with pytest.raises(AssertionError):
		norm.binnet(np.full((20, 20), 1.5), 0.1)


def test_normvar_golden_and_oracle(golden, norm):
	"""normvar on the device (per-gene weighted OLS as two Gram contractions) vs the reference's per-gene loop."""
	g = golden('G9_normvar')
	dt, dc, w, wt = g['dt'], g['dc'], g['w'], g['wt']
	r = norm.normvar(dt, dc, w, wt)
	assert len(r) == 2 and r[0].dtype == np.float64
	assert close(r[0], g['a_dtn'], 1e-6, 1e-9) and np.array_equal(r[1], g['a_dcn'])
	r = norm.normvar(dt, dc, w, wt, dextra=g['dextra'], cat=2, keepvar=False, normmean=True)
	assert close(r[0], g['b_dtn'], 1e-6, 1e-9) and np.array_equal(r[1], g['b_dcn']) and np.array_equal(r[2], g['b_dex'])
	r = norm.normvar(dt, dc, w, wt, cat=0, bs=7, nth=3)
	assert close(r[0], g['c_dtn'], 1e-6, 1e-9) and np.array_equal(r[1], g['c_dcn'])
	# larger seeded case, fp32 expression, 12 covariates; normalised output feeds de exactly like the reference pipeline
	rng = np.random.default_rng(99)
	ng, n, nc = 700, 3000, 12
	dt = (rng.normal(size=(ng, n)) * rng.uniform(0.5, 2, (ng, 1)) - 9).astype(np.float32)
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
	w = np.exp(0.25 * rng.normal(size=n))
	wt = rng.uniform(0, 1.5, ng)
	wt[::50] = 0
	got = norm.normvar(dt, dc, w, wt)
	ref = oracle.normvar(dt.astype(np.float64), dc, w, wt)
	assert got[0].dtype == np.float64 and close(got[0], ref[0], 1e-6, 1e-8) and np.array_equal(got[1], ref[1])
	with pytest.raises(ValueError):
		norm.normvar(dt, dc, -w, wt)
	with pytest.raises(ValueError):
		norm.normvar(dt, dc[:0], w, wt)


def test_bitwise_reproducible(norm, de_path):
	"""No atomics on the result path: repeated calls return bit-identical p-values (stream-K partial tiles are summed
	in a fixed order).  Matters downstream: binnet thresholds compare p-values exactly."""
	rng = np.random.default_rng(123)
	dt = rng.normal(size=(900, 2000)) + 0.4 * rng.normal(size=(900, 1)) * rng.normal(size=(1, 2000))
	dc = np.vstack([rng.normal(size=(2, 2000)), np.ones((1, 2000))])
	dg = (rng.random((6, 2000)) < 0.3).astype(float)
	a = norm.coex(dt, dc)
	b = norm.coex(dt, dc)
	assert all(np.array_equal(x, y) for x, y in zip(a, b))
	a = norm.de(dg, dt, dc)
	b = norm.de(dg, dt, dc)
	assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[4], b[4])


def test_large_cell_counts_vs_oracle(norm):
	"""BASELINE configs[2]/[3] regimes at reduced gene counts but full cell counts (p is most sensitive to the dot product
	at large dof): streaming path at 100k cells / 20 covariates, general path at 50k cells with 256 groupings."""
	rng = np.random.default_rng(303)
	n, nc, ny = 100000, 20, 1500
	dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))]).astype(np.float32)
	dg = (rng.random((1, n)) < 0.5).astype(np.float32)
	dt = rng.standard_normal((ny, n), dtype=np.float32) * 1.3 - 7
	dt[:40] += np.linspace(0.002, 0.08, 40, dtype=np.float32)[:, None] * dg[0]
	p, gam, a, vg, vt = norm.de(dg, dt, dc)
	po, go, ao, vgo, vto = oracle.de(dg.astype(np.float64), dt.astype(np.float64), dc.astype(np.float64))
	assert po.min() < 1e-20 and po.max() > 0.5
	assert close(p, po, 1e-6, 1e-38) and close(gam, go, 1e-6, 1e-7) and close(vt, vto, 1e-6)
	# same data in fp64: the north-star tolerance without the final fp32 rounding
	p, gam, a, vg, vt = norm.de(dg.astype(np.float64), dt.astype(np.float64), dc.astype(np.float64))
	assert p_close(p, po) and gamma_close(gam, vg, vt[0], go, vgo, vto[0], I8_FLOOR)
	n, nx, ny, nc = 50000, 256, 900, 5
	dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))])
	dg = (rng.random((nx, n)) < 0.01).astype(np.float64)
	dt = rng.standard_normal((ny, n)) + 0.5 * (rng.standard_normal((ny, 6)) @ dg[:6])
	p, gam, a, vg, vt = norm.de(dg, dt, dc)
	po, go, ao, vgo, vto = oracle.de(dg, dt, dc)
	assert p_close(p, po) and gamma_close(gam, vg, vt[0], go, vgo, vto[0], I8_FLOOR) and close(vg, vgo, 1e-10) and close(vt, vto, 1e-10)


def test_schedule_regimes_many_tiles(norm, monkeypatch):
	"""K2's schedule phases at other tile counts: two full waves + remainder (1326 symmetric tiles), a rectangular grid
	with 626 tiles, and a single-tile problem; few cells keep the oracle cheap."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(515)
	n = 160
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	dt = rng.normal(size=(6500, n)) + 0.6 * rng.normal(size=(6500, 1)) * rng.normal(size=(1, n))
	p, d, v = norm.coex(dt, dc)
	po, do, vo = oracle.coex(dt, dc)
	assert p_close(p, po) and close(d, do, floor=1e-12) and close(v, vo, 1e-12)
	assert (np.diag(p) == 0).all() and (p == p.T).all() and (d == d.T).all()
	monkeypatch.setenv('NRM_DE_PATH', 'general')
	dg = (rng.random((130, n)) < 0.3).astype(float)
	dy = rng.normal(size=(40000, n))
	p, g, a, vx, vy = association_tests(dg, dy, dc, return_dot=False)
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dy, dc, return_dot=False)
	assert p_close(p, po) and close(g, go, floor=1e-12) and close(vy, vyo, 1e-12)
	p, g, a, vx, vy = association_tests(dg[:3], dy[:7], dc, return_dot=False)
	assert p_close(p, po[:3, :7]) and close(g, go[:3, :7], floor=1e-12)


def test_beyond_2g_elements(eng):
	"""Maximum sizes: expression matrices with more than 2^31 elements, resident in HBM (generated on the device), through
	coex (K1+K2+K3), the streaming de path and the general de path; sampled blocks are checked against the oracle run
	on the same rows copied to the host.  Guards the 64-bit index arithmetic of every kernel on the path."""
	import torch
	from normalisr_amd.association import inv_rank
	ng, n, nc = 22016, 100000, 4
	assert ng * n > 2**31
	gen = torch.Generator(device='cuda').manual_seed(77)
	dt = torch.randn((ng, n), dtype=torch.float32, device='cuda', generator=gen)
	lat = torch.randn((1, n), dtype=torch.float32, device='cuda', generator=gen)
	dt += 0.2 * torch.randn((ng, 1), dtype=torch.float32, device='cuda', generator=gen) * lat
	rng = np.random.default_rng(78)
	dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))])
	dci, rank = inv_rank(dc @ dc.T)
	blocks = [slice(0, 40), slice(11000, 11040), slice(ng - 40, ng)]
	rows = np.concatenate([np.arange(ng)[b] for b in blocks])
	sub = dt[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64)
	# coex
	res = eng.association_single0(dt, None, dc, dci, rank, 0, True, False, np.float32, device_out=True)
	po, do, vo = oracle.coex(sub, dc)
	idx = torch.from_numpy(rows).cuda()
	p = res['p'][idx][:, idx].cpu().numpy()
	d = res['stat'][idx][:, idx].cpu().numpy()
	assert close(p, po, 2e-6, 1e-38) and close(d, do, 1e-6, 1e-7) and close(res['vary'][rows], vo, 1e-6)
	assert bool((torch.diagonal(res['p']) == 0).all()) and bool((res['p'][-200:, :200] == res['p'][:200, -200:].T).all())
	del res
	# de: streaming (2 design rows) and general (40 design rows) over all 2.2e9 expression values
	for nx in (2, 40):
		dg = (rng.random((nx, n)) < 0.3).astype(np.float64)
		r = eng.association_single0(dg, dt, dc, dci, rank, 0, False, False, np.float32)
		po, go, ao, vxo, vyo = oracle.association_tests(dg, sub, dc, return_dot=False)
		assert close(r['p'][:, rows], po, 2e-6, 1e-38) and close(r['stat'][:, rows], go, 1e-6, 1e-7)
		assert close(r['vary'][rows], vyo, 1e-6)


def test_banded_pipeline_matches_one_launch(norm, eng, monkeypatch):
	"""Large results leave the device band by band (K2 -> K3 -> copy-out overlapped, engine.association_banded).  Same
	answers as the one-launch path (dot to fp64 round-off: the split of the cells between workgroups differs), exact
	symmetry / zero diagonal, bitwise reproducible, and the reference's assertion still fires (association.py:252)."""
	rng = np.random.default_rng(808)
	ng, n = 2600, 900
	dt = rng.normal(size=(ng, n)) + 0.5 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dg = (rng.random((1500, n)) < 0.2).astype(float)
	assert eng.banded_ok(ng, ng, np.float64) and eng.banded_ok(1500, ng, np.float64)
	monkeypatch.setenv('NRM_PIPELINE', '0')
	c0, d0 = norm.coex(dt, dc), norm.de(dg, dt, dc)
	monkeypatch.setenv('NRM_PIPELINE', '1')
	c1, d1 = norm.coex(dt, dc), norm.de(dg, dt, dc)
	c2 = norm.coex(dt, dc)
	assert all(np.array_equal(x, y) for x, y in zip(c1, c2))
	assert p_close(c1[0], c0[0], 1e-9) and close(c1[1], c0[1], 1e-9, 1e-6) and np.array_equal(c1[2], c0[2])
	assert (np.diag(c1[0]) == 0).all() and (c1[0] == c1[0].T).all() and (c1[1] == c1[1].T).all()
	assert p_close(d1[0], d0[0], 1e-9) and close(d1[1], d0[1], 1e-9, 1e-6) and d1[2] is None
	assert np.array_equal(d1[3], d0[3]) and np.array_equal(d1[4], d0[4])
	po, do, vo = oracle.coex(dt, dc)
	assert p_close(c1[0], po) and close(c1[1], do, floor=1e-12)
	bad = dt.copy()
	bad[1234, 5] = np.inf
	with pytest.raises(AssertionError):
		norm.coex(bad, dc)
	c3 = norm.coex(dt, dc)  # the page locks of the failed call were released
	assert np.array_equal(c3[0], c1[0])


def test_chunked_upload_de_matches_one_shot(norm, eng, monkeypatch):
	"""de with a large host-resident expression matrix streams it to the device in row chunks on a second stream
	(engine.association_de_chunked); here the chunk size is forced down so that a small problem takes that path with a
	ragged last chunk.  Same answers as the one-shot path and as the oracle; bitwise reproducible."""
	rng = np.random.default_rng(909)
	nx, ny, n = 40, 1900, 600
	dt = (rng.normal(size=(ny, n)) * rng.uniform(0.5, 2, (ny, 1)) - 5).astype(np.float32)
	dc = np.vstack([rng.normal(size=(3, n)), np.ones((1, n))]).astype(np.float32)
	dg = (rng.random((nx, n)) < 0.25).astype(np.float32)
	dt[:30] += 0.4 * dg[0]
	monkeypatch.setenv('NRM_DE_PATH', 'general')
	monkeypatch.setenv('NRM_PIPELINE', '0')
	one = norm.de(dg, dt, dc)
	monkeypatch.setenv('NRM_PIPELINE', '1')
	monkeypatch.setattr(eng, 'CHUNK_BYTES', 512 * n * 4, raising=False)
	assert eng.chunked_ok(dt)
	a = norm.de(dg, dt, dc)
	b = norm.de(dg, dt, dc)
	assert all(np.array_equal(x, y) for x, y in zip(a, b) if x is not None)
	assert close(a[0], one[0], 1e-6, 1e-38) and close(a[1], one[1], 1e-6, 1e-7) and np.array_equal(a[3], one[3]) and np.array_equal(a[4], one[4])
	po, go, ao, vgo, vto = oracle.de(dg.astype(np.float64), dt.astype(np.float64), dc.astype(np.float64))
	assert close(a[0], po, 1e-6, 1e-38) and close(a[1], go, 1e-6, 1e-7) and close(a[4], vto, 1e-6) and a[2] is None
	bad = dt.copy()
	bad[1500, 3] = np.nan
	with pytest.raises(AssertionError):
		norm.de(dg, bad, dc)


def test_de_plan_shards_equal_whole(norm, de_path):
	"""DePlan (the sharded de of bench.py --workload de_c3 / de_c4): every rank owns a block of genes and there is no
	collective, so the ranks' results side by side must equal the single-call result -- both de paths, resident steps."""
	import torch
	from normalisr_amd.distributed import DePlan
	rng = np.random.default_rng(616)
	streaming = de_path == 'auto'  # 'auto' picks the streaming kernel (K2s) for nx + nc <= 32; 'general' forces K1 -> K2 -> K3
	nx, ny, n, nc = (2, 900, 1500, 6)
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]).astype(np.float32)
	dg = (rng.random((nx, n)) < 0.3).astype(np.float32)
	dt = (rng.normal(size=(ny, n)) - 4).astype(np.float32)
	dt[:25] += 0.3 * dg[0]
	whole = norm.de(dg, dt, dc)
	world = 3
	R = ny // world
	parts = []
	for rank in range(world):
		plan = DePlan(torch.from_numpy(dg).cuda(), torch.from_numpy(dt[rank * R:(rank + 1) * R]).cuda(), dc, rank=rank, world=world)
		assert plan.streaming() == streaming
		plan.step()
		plan.step(timed=True)
		assert plan.step_ms() > 0
		parts.append(plan.results())
	p = np.concatenate([q[0] for q in parts], axis=1)
	g = np.concatenate([q[1] for q in parts], axis=1)
	vt = np.concatenate([q[3] for q in parts])
	assert close(p, whole[0], 1e-6, 1e-38) and close(g, whole[1], 1e-6, 1e-7) and close(vt, whole[4][0], 1e-6)
	assert all(close(q[2], whole[3], 1e-6) for q in parts)
	po, go, ao, vgo, vto = oracle.de(dg.astype(np.float64), dt.astype(np.float64), dc.astype(np.float64))
	assert close(p, po, 1e-6, 1e-38) and close(g, go, 1e-6, 1e-7) and close(vt, vto[0], 1e-6)
