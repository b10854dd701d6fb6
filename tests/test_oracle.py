"""CPU: pin the oracle (oracle/) against the golden vectors generated from the reference."""
import os

import numpy as np
import pytest

import oracle
from conftest import relerr

# Tolerances: the oracle restates the same fp64 algorithm; differences come from BLAS/SVD rounding
# only (reference itself varies ~4e-14 with tile size).  P-values: d ln p/d R2 = -dof/2
# amplifies those by up to ~1e3 here.
RT = 1e-9


def test_pvalue_table_matches_scipy(golden):
	g = golden('G3_ptable')
	dof, r2, ref = g['dof'], g['r2'], g['p']
	worst = 0.
	for i, d in enumerate(dof):
		p = oracle.pvalues(r2, d)
		normal = ref[i] >= 2.3e-308
		worst = max(worst, relerr(p[normal], ref[i][normal]))
		# subnormal/underflow region: absolute agreement
		assert np.all(np.abs(p[~normal] - ref[i][~normal]) <= 1e-310)
		assert p[r2 == 0][0] == 1.0 and p[r2 == 1][0] == 0.0
	# scipy itself is 3.5e-11 off the exact value (mpmath) at dof=1, R2=1e-12; elsewhere both agree to 1e-13
	assert worst < 1e-10, worst


def test_pvalue_clipping():
	# R2 slightly above 1 -> x < 0 -> cdf clipped to 0 (association.py:248-249, Q15)
	assert oracle.pvalues(np.array([1 + 5e-9]), 100.)[0] == 0.
	assert oracle.pvalues(np.array([0.]), 100.)[0] == 1.


def test_inv_rank(golden):
	g = golden('G4_invrank')
	for i in range(int(g['ncase'])):
		mi, r = oracle.inv_rank(g['m{}'.format(i)])
		assert r == int(g['r{}'.format(i)])
		ref = g['mi{}'.format(i)]
		assert np.abs(mi - ref).max() <= 1e-9 * np.abs(ref).max()
	with pytest.raises(ValueError):
		oracle.inv_rank(np.zeros((2, 3)))


def test_block(golden):
	g = golden('G7_block')
	r = oracle.association_test_1(0, 0, g['dx'], g['dy'], g['dc'], g['dci'], int(g['dcr']), lowmem=False)
	assert relerr(r[2], g['p']) < RT
	assert relerr(r[3], g['gamma'], 1e-14) < RT
	assert relerr(r[4], g['alpha'], 1e-12) < RT
	assert relerr(r[5], g['vx']) < 1e-12 and relerr(r[6], g['vy']) < 1e-12
	# plain C loops agree too (sub-block to keep it quick)
	p, gam, vx, vy = oracle.block_plain_c(g['dx'][:8], g['dy'][:6], g['dc'], g['dci'], int(g['dcr']))
	assert relerr(p, g['p'][:8, :6]) < RT and relerr(gam, g['gamma'][:8, :6], 1e-14) < RT


def test_c1_de_coex(golden):
	g = golden('G1_c1')
	dt, dc, dg = g['dt'], g['dc'], g['dg']
	for lm in (1, 0):
		p, gam, a, vg, vt = oracle.de(dg, dt, dc, lowmem=bool(lm))
		k = 'de_lm{}_'.format(lm)
		assert relerr(p, g[k + 'p']) < RT
		assert relerr(gam, g[k + 'gamma'], 1e-14) < RT
		assert relerr(vg, g[k + 'varg'], 1e-300) < 1e-12 and relerr(vt, g[k + 'vart'], 1e-300) < 1e-12
		if lm == 0:
			assert relerr(a, g[k + 'alpha'], 1e-12) < RT
		else:
			assert a is None
		# constant grouping row re-inflated exactly (de.py:107-122)
		assert (p[2] == 1).all() and (gam[2] == 0).all() and vg[2] == 0 and (vt[2] == 0).all()
	ns = int(g['coex_n'])
	p, d, v = oracle.coex(dt[:ns], dc)
	assert relerr(p, g['coex_p'], 1e-300) < RT and relerr(d, g['coex_dot'], 1e-14) < RT and relerr(v, g['coex_var']) < 1e-12
	assert (np.diag(p) == 0).all() and (np.diag(d) == 0).all() and (p == p.T).all() and (d == d.T).all()
	p, d, a, vx, vy = oracle.association_tests(dg[[0, 1, 3]], dt[:64], dc, return_dot=True)
	assert relerr(p, g['at_p']) < RT and relerr(d, g['at_dot'], 1e-14) < RT


def test_edge_cases(golden):
	g = golden('G2_edge')
	dt, dc, dg = g['dt'], g['dc'], g['dg']
	n = dt.shape[1]
	p, gam, a, vg, vt = oracle.de(dg, dt, np.zeros((0, n)))
	assert relerr(p, g['nc0_de_p']) < RT and relerr(gam, g['nc0_de_gamma'], 1e-14) < RT
	p, d, v = oracle.coex(dt[:40], np.zeros((0, n)))
	assert relerr(p, g['nc0_coex_p'], 1e-300) < RT and relerr(d, g['nc0_coex_dot'], 1e-14) < RT
	# rank-deficient covariates: integer rank bit-exact
	mi, r = oracle.inv_rank(g['rd_dc'] @ g['rd_dc'].T)
	assert r == int(g['rd_rank']) == 3
	p, gam, a, vg, vt = oracle.de(dg, dt, g['rd_dc'], lowmem=False)
	assert relerr(p, g['rd_de_p']) < 1e-7 and relerr(gam, g['rd_de_gamma'], 1e-12) < 1e-7
	p, d, v = oracle.coex(dt[:40], g['rd_dc'])
	assert relerr(p, g['rd_coex_p'], 1e-300) < 1e-7
	p, gam, a, vg, vt = oracle.de(dg, dt, dc, dimreduce=2)
	assert relerr(p, g['dr2_de_p']) < RT
	p, d, v = oracle.coex(dt[:40], dc, dimreduce=2)
	assert relerr(p, g['dr2_coex_p'], 1e-300) < RT
	p, gam, a, vg, vt = oracle.de(dg.astype(np.int64), dt, dc)
	assert relerr(p, g['int_de_p']) < RT and p.dtype == np.float64
	p, d, a, vx, vy = oracle.association_tests(dg, dt, dc, bsx=2, bsy=13)
	assert relerr(p, g['tile_at_p']) < RT and relerr(d, g['tile_at_dot'], 1e-14) < RT
	p, d, a, vx, vy = oracle.association_tests(dt[:45], None, dc, bsx=7)
	assert relerr(p, g['tile_coex_p'], 1e-300) < RT and vx is None
	dz = g['zc_dt']
	p, d, v = oracle.coex(dz, dc)
	assert v[5] == 1 and (p[5] == np.where(np.arange(30) == 5, 0, 1)).all() and (d[5] == 0).all()
	assert p[6, 7] == g['zc_coex_p'][6, 7] == 0.
	ok = np.ones_like(p, dtype=bool)
	ok[6, 7] = ok[7, 6] = False
	assert relerr(p[ok], g['zc_coex_p'][ok], 1e-300) < RT
	p, gam, a, vg, vt = oracle.de(dg, g['se_dt'], dc)
	assert relerr(p, g['se_de_p'], 1e-300) < RT
	assert g['se_de_p'].min() < 1e-20


def test_single4(golden):
	g = golden('G5_single')
	p, gam, a, vg, vt = oracle.de(g['dg'], g['dt'], g['dc'], single=4, lowmem=False)
	assert relerr(p, g['s4_p']) < 1e-8 and relerr(gam, g['s4_gamma'], 1e-12) < 1e-8
	assert relerr(vg, g['s4_varg']) < 1e-10 and relerr(vt, g['s4_vart']) < 1e-10
	assert relerr(a, g['s4_alpha'], 1e-10) < 1e-8


def test_single1(golden):
	g = golden('G5_single')
	p, gam, a, vg, vt = oracle.de(g['s1_dg'], g['dt'], g['dc'], single=1, lowmem=False)
	assert relerr(p, g['s1_p']) < 1e-8 and relerr(gam, g['s1_gamma'], 1e-12) < 1e-8
	assert relerr(vg, g['s1_varg']) < 1e-10 and relerr(vt, g['s1_vart']) < 1e-10
	assert relerr(a, g['s1_alpha'], 1e-10) < 1e-8


def test_binnet_bh(golden):
	g = golden('G8_binnet')
	for k, q in (('net_q5', 0.05), ('net_q20', 0.2), ('net_q50', 0.5)):
		assert np.array_equal(oracle.binnet(g['p'], q), g[k])  # booleans: bit-exact
	for k, q in (('net32_q5', 0.05), ('net32_q30', 0.3)):
		assert np.array_equal(oracle.binnet(g['p32'], q), g[k])
	for k, q in (('nett_q10', 0.1), ('nett_q25', 0.25)):
		assert np.array_equal(oracle.binnet(g['pt'], q), g[k])  # tie-heavy
	assert np.array_equal(oracle.bh(g['bh_in']), g['bh_out'])
	assert relerr(oracle.bh(g['bh_in'], weight=g['bh_w']), g['bh_wout'], 1e-300) < 1e-13


def test_normvar(golden):
	g = golden('G9_normvar')
	dt, dc, w, wt = g['dt'], g['dc'], g['w'], g['wt']
	r = oracle.normvar(dt, dc, w, wt)
	assert relerr(r[0], g['a_dtn'], 1e-12) < 1e-9 and np.array_equal(r[1], g['a_dcn'])
	r = oracle.normvar(dt, dc, w, wt, dextra=g['dextra'], cat=2, keepvar=False, normmean=True)
	assert relerr(r[0], g['b_dtn'], 1e-12) < 1e-9 and np.array_equal(r[1], g['b_dcn']) and np.array_equal(r[2], g['b_dex'])
	r = oracle.normvar(dt, dc, w, wt, cat=0)
	assert relerr(r[0], g['c_dtn'], 1e-12) < 1e-9 and np.array_equal(r[1], g['c_dcn'])


def test_single4_variants_golden(golden):
	"""G10: single=4 with dy=None (pairs given all other rows), one dimreduce per gene, a pseudo-inverse truncated to mpc
	components (scikit-learn randomized SVD, as the reference), rank-deficient covariates; single=1 with 40 covariates."""
	g = golden('G10_single4')
	dg, dc, dt = g['dg'], g['dc'], g['dt']
	for rd in (1, 0):
		p, d, a, vx, vy = oracle.association_tests(dt[:14], None, dc, single=4, return_dot=bool(rd))
		assert a is None and vx is None
		assert relerr(p, g['sx_p_rd%d' % rd], 1e-300) < 1e-8 and relerr(d, g['sx_dot_rd%d' % rd], 1e-12) < 1e-8
		assert relerr(vy, g['sx_vy_rd%d' % rd]) < 1e-10 and (np.diag(p) == 0).all() and (p == p.T).all()
	p, gam, a, vg, vt = oracle.de(dg, dt, dc, single=4, dimreduce=g['dr'])
	assert relerr(p, g['dr_p']) < 1e-8 and relerr(gam, g['dr_gamma'], 1e-12) < 1e-8 and relerr(vt, g['dr_vart']) < 1e-10
	p, gam, a, vg, vt = oracle.de(dg, dt, dc, single=4, mpc=5, lowmem=False)
	assert relerr(p, g['mpc_p']) < 1e-8 and relerr(gam, g['mpc_gamma'], 1e-12) < 1e-8 and relerr(a, g['mpc_alpha'], 1e-10) < 1e-8
	assert relerr(vg, g['mpc_varg']) < 1e-10 and relerr(vt, g['mpc_vart']) < 1e-10
	p, gam, a, vg, vt = oracle.de(dg, dt, g['rdd_dc'], single=4, dimreduce=g['dr'])
	assert relerr(p, g['rdd_p']) < 1e-8 and relerr(gam, g['rdd_gamma'], 1e-12) < 1e-8 and relerr(vt, g['rdd_vart']) < 1e-10
	p, gam, a, vg, vt = oracle.de(g['s1c_dg'], dt[:16], g['s1c_dc'], single=1, lowmem=False)
	assert relerr(p, g['s1c_p']) < 1e-8 and relerr(gam, g['s1c_gamma'], 1e-12) < 1e-8 and relerr(a, g['s1c_alpha'], 1e-10) < 1e-8


G15_CASES = (('a', dict(lowmem=False)), ('b', dict(return_dot=False)), ('c', dict(bsx=5, bsy=4, lowmem=False)), ('d', dict(dimreduce=2)))


def test_single5_masked_golden(golden):
	"""G15: single=5 with a mask (association.py:579-728,969-980) -- the oracle's restatement on the reference's own outputs: default and small
	tiles (the per-block variance of x), alpha, both forms of the statistic, a repeated covariate row (truncated pseudo-inverses)."""
	g = golden('G15_single5')
	for name, kw in G15_CASES:
		p, d, a, vx, vy = oracle.association_tests(g['dx'], None, g['dc'], single=5, mask=g['mask'], **kw)
		assert relerr(p, g[name + '_p'], 1e-300) < 1e-8 and relerr(d, g[name + '_dot'], 1e-12) < 1e-8 and relerr(vx, g[name + '_vx'], 1e-12) < 1e-10
		assert relerr(vy, g[name + '_vy'], 1e-12) < 1e-10 and (p[~g['mask']] == 1).all() and (d[~g['mask']] == 0).all()
		if name + '_alpha' in g.files:
			assert relerr(a, g[name + '_alpha'], 1e-10) < 1e-8
		else:
			assert a is None


def test_single1_with_one_dimreduce_per_gene_golden(golden):
	"""G16: single=1 with a (n_y, 1) column of dimreduce values through the reference -- dof per (grouping, gene) pair (association.py:372-377).  The oracle
	(whose restatement took the first gene's dof for a whole row until round 6) on the reference's outputs, and each gene's column equal to the scalar call
	with that gene's value."""
	g = golden('G16_single1_dimreduce')
	dr = g['dimreduce']
	ny = dr.size
	p, gam, a, vx, vy = oracle.association_tests(g['dx'], g['dy'], g['dc'], single=1, dimreduce=dr.reshape(ny, 1), return_dot=False, bsx=g['dx'].shape[0], bsy=ny, lowmem=False)
	assert relerr(p, g['p'], 1e-300) < 1e-10 and relerr(gam, g['gamma'], 1e-13) < 1e-10 and relerr(a, g['alpha'], 1e-12) < 1e-9 and relerr(vx, g['vx']) < 1e-12 and relerr(vy, g['vy']) < 1e-12
	for v in range(3):
		assert np.array_equal(g['p'][:, dr == v], g['p_scalar%d' % v][:, dr == v])
		assert not np.array_equal(g['p'][:, dr != v], g['p_scalar%d' % v][:, dr != v])


def test_g11_rows_that_are_hard_for_fixed_point(golden):
	"""G11 (4096 cells, reference-generated): sparse log1p-count rows, 0/1 rows, rows with a huge mean, heavy tails and single
	spikes under intercept-only, one-hot-batch, near-collinear and no covariates.  The oracle is fp64 like the reference, so it
	must sit on these vectors; the GPU tests then hold the integer Gram engine (with its guard) to the same vectors."""
	g = golden('G11_i8hard')
	dt, dg = g['dt'], g['dg']
	off = ~np.eye(dt.shape[0], dtype=bool)
	for name in ('intercept', 'onehot', 'collinear', 'none'):
		dc = g['dc_' + name]
		p, d, v = oracle.coex(dt, dc)
		pr = g['coex_%s_p' % name]
		ok = pr >= 2.3e-308
		# near-collinear covariates amplify the BLAS-order rounding of the residualisation itself (cond ~ 4e4): the reference's own
		# noise floor is higher there
		rt = 2e-7 if name == 'collinear' else 1e-8
		assert relerr(p[ok & off], pr[ok & off]) < rt, name
		assert relerr(v, g['coex_%s_var' % name]) < 1e-9, name
		scale = np.sqrt(np.outer(v, v))
		assert np.max(np.abs(d - g['coex_%s_dot' % name]) / scale) < 1e-11, name
		if name == 'onehot':
			continue
		pd_, gam, a, vg, vt = oracle.de(dg, dt, dc)
		pr = g['de_%s_p' % name]
		ok = pr >= 2.3e-308
		assert relerr(pd_[ok], pr[ok]) < rt, name


def g13_problem(golden):
	"""G13's inputs, rebuilt from the seed (tests/golden/g13_inputs.py), checked against the sums the reference run recorded."""
	import importlib.util
	spec = importlib.util.spec_from_file_location('g13_inputs', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g13_inputs.py'))
	mod = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(mod)
	g = golden('G13_100k')
	dt, dc, dg = mod.g13_inputs(int(g['seed']), int(g['n']), int(g['ng']))
	assert np.array_equal(np.array([dt.sum(), dc.sum(), dg.sum()]), g['check'])  # the same numbers the reference saw
	return g, dt, dc, dg


def test_g13_reference_outputs_at_100k_cells(golden):
	"""G13: the reference's coex and de outputs for 40 genes x 100 000 cells (the cell count of BASELINE configs[2]); only the outputs
	are stored, the inputs come from the seed.  The oracle must sit on them; the GPU tests hold the device paths to the same vectors."""
	g, dt, dc, dg = g13_problem(golden)
	p, d, v = oracle.coex(dt, dc)
	off = ~np.eye(dt.shape[0], dtype=bool)
	pr = g['coex_p']
	ok = off & (pr >= 2.3e-308)
	assert relerr(p[ok], pr[ok]) < 1e-7 and relerr(v, g['coex_var']) < 1e-10
	assert np.max(np.abs(d - g['coex_dot']) / np.sqrt(np.outer(v, v))) < 1e-11
	pd_, gam, a, vg, vt = oracle.de(dg, dt, dc)
	ok = g['de_p'] >= 2.3e-308
	assert relerr(pd_[ok], g['de_p'][ok]) < 1e-7 and relerr(gam, g['de_gamma'], 1e-12) < 1e-7
	assert relerr(vg, g['de_varg']) < 1e-10 and relerr(vt, g['de_vart']) < 1e-10


def g14_problem(golden):
	"""G14's inputs (BASELINE configs[1] at full size), rebuilt from the seed (tests/golden/g14_inputs.py)."""
	import importlib.util
	spec = importlib.util.spec_from_file_location('g14_inputs', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g14_inputs.py'))
	mod = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(mod)
	g = golden('G14_c2')
	dt, dc, _ = mod.g14_inputs(int(g['seed']))
	assert np.array_equal(np.array([float(dt.astype(np.float64).sum()), dc.sum()]), g['check'])  # the same numbers the reference saw
	return g, dt, dc


def test_g14_config1_rows_against_the_reference(golden):
	"""G14: the reference's outputs for 12 gene rows of BASELINE configs[1] at full size (5000 genes x 10 000 cells; the fp32 values
	upcast to fp64).  Here the oracle computes those rows against all genes (a rectangle of the same problem)."""
	g, dt, dc = g14_problem(golden)
	rows = g['rows']
	d64 = dt.astype(np.float64)
	p, gam, a, vx, vy = oracle.association_tests(d64[rows], d64, dc, return_dot=True)
	pr = g['p']
	ok = pr >= 2.3e-308
	ok[np.arange(len(rows)), rows] = False  # (the coex diagonal is 0 by definition)
	assert relerr(p[ok], pr[ok]) < 1e-7
	assert relerr(vy, g['var']) < 1e-10 and np.max(np.abs(gam - g['dot'])[ok]) < 1e-11 * np.sqrt(g['var'].max() * g['var'][rows].max())
