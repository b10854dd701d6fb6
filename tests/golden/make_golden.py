#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the reference.

Run in the dev container only (the reference lives at /root/reference and never travels):

    PYTHONPATH=/root/reference/src python3 tests/golden/make_golden.py

Each fixture is data only: seeded synthetic inputs plus the outputs the reference
(normalisr v1.0.0, numpy/scipy versions recorded in meta.json) produced for them.
Fixture names follow SURVEY.md section 8(c): G1..G7; G8..G11, G13 and G14 were added with the components they pin (G12: make_pvalue_grid.py).
"""
import gzip
import io
import json
import os
import subprocess
import sys
import tempfile
import warnings

import numpy as np

warnings.simplefilter('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'
sys.path.insert(0, REF)

import scipy  # noqa: E402
from scipy.stats import beta  # noqa: E402
import normalisr.normalisr as norm  # noqa: E402
from normalisr.association import (association_tests, association_test_1, inv_rank)  # noqa: E402


def save(name, **ka):
	path = os.path.join(HERE, name + '.npz')
	np.savez_compressed(path, **ka)
	print('{:28s} {:9.1f} KB'.format(name + '.npz', os.path.getsize(path) / 1024))


def c1_inputs(seed=1, ngene=500, n=300, ngroup=4):
	"""Poisson(2) counts -> log1p; dc = [N(0,1); ones]; dg Bernoulli(0.3)."""
	rng = np.random.default_rng(seed)
	counts = rng.poisson(2, (ngene, n)).astype(np.uint8)
	dt = np.log1p(counts.astype(np.float64))
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	dg = (rng.random((ngroup, n)) < 0.3).astype(np.float64)
	return counts, dt, dc, dg


def g1():
	counts, dt, dc, dg = c1_inputs()
	dg[2] = 1.  # constant grouping row -> dropped and re-inflated by de (de.py:92-122)
	out = dict(dt=dt, dc=dc, dg=dg)
	for lowmem in (True, False):
		p, g, a, vg, vt = norm.de(dg, dt, dc, lowmem=lowmem)
		k = 'de_lm{}_'.format(int(lowmem))
		out.update({k + 'p': p, k + 'gamma': g, k + 'varg': vg, k + 'vart': vt})
		if a is not None:
			out[k + 'alpha'] = a
	ns = 160
	p, d, v = norm.coex(dt[:ns], dc)
	out.update(coex_n=ns, coex_p=p, coex_dot=d, coex_var=v)
	# association_tests raw tuple with return_dot True/False for x!=y
	p, d, a, vx, vy = association_tests(dg[[0, 1, 3]], dt[:64], dc, return_dot=True)
	out.update(at_p=p, at_dot=d, at_vx=vx, at_vy=vy)
	save('G1_c1', **out)


def g2():
	rng = np.random.default_rng(2)
	n = 240
	dt = np.log1p(rng.poisson(3, (90, n)).astype(np.float64))
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dg = (rng.random((5, n)) < 0.25).astype(np.float64)
	out = dict(dt=dt, dc=dc, dg=dg)

	# (a) no covariates
	dc0 = np.zeros((0, n))
	p, g, a, vg, vt = norm.de(dg, dt, dc0)
	out.update(nc0_de_p=p, nc0_de_gamma=g, nc0_de_varg=vg, nc0_de_vart=vt)
	p, d, v = norm.coex(dt[:40], dc0)
	out.update(nc0_coex_p=p, nc0_coex_dot=d, nc0_coex_var=v)

	# (b) rank-deficient covariates (duplicate + scaled row) -> rank 3 of 5
	dcr = np.vstack([dc, dc[0], 2 * dc[1] - dc[0]])
	mi, r = inv_rank(np.matmul(dcr, dcr.T))
	p, g, a, vg, vt = norm.de(dg, dt, dcr, lowmem=False)
	out.update(rd_dc=dcr, rd_rank=r, rd_de_p=p, rd_de_gamma=g, rd_de_alpha=a, rd_de_varg=vg, rd_de_vart=vt)
	p, d, v = norm.coex(dt[:40], dcr)
	out.update(rd_coex_p=p, rd_coex_dot=d, rd_coex_var=v)

	# (c) dimreduce=2
	p, g, a, vg, vt = norm.de(dg, dt, dc, dimreduce=2)
	out.update(dr2_de_p=p, dr2_de_gamma=g)
	p, d, v = norm.coex(dt[:40], dc, dimreduce=2)
	out.update(dr2_coex_p=p)

	# (d) integer grouping matrix
	dgi = dg.astype(np.int64)
	p, g, a, vg, vt = norm.de(dgi, dt, dc)
	out.update(int_de_p=p, int_de_gamma=g, int_de_varg=vg, int_de_vart=vt)

	# (e) uneven tiles
	p, d, a, vx, vy = association_tests(dg, dt, dc, bsx=2, bsy=13)
	out.update(tile_at_p=p, tile_at_dot=d, tile_at_vx=vx, tile_at_vy=vy)
	p, d, a, vx, vy = association_tests(dt[:45], None, dc, bsx=7)
	out.update(tile_coex_p=p, tile_coex_dot=d, tile_coex_vy=vy)

	# (f) zero-variance gene (all zeros) and a perfectly collinear pair (R^2 -> 1 => p = 0)
	dz = dt[:30].copy()
	dz[5] = 0.
	dz[7] = 3. * dz[6]
	p, d, v = norm.coex(dz, dc)
	out.update(zc_dt=dz, zc_coex_p=p, zc_coex_dot=d, zc_coex_var=v)
	p, g, a, vg, vt = norm.de(dg, dz, dc)
	out.update(zc_de_p=p, zc_de_gamma=g, zc_de_vart=vt)

	# (g) fp32 inputs for de: reference fp32 result and its fp64 run on the same (fp32-representable) values
	dt32 = dt.astype(np.float32)
	dc32 = dc.astype(np.float32)
	dg32 = dg.astype(np.float32)
	p, g, a, vg, vt = norm.de(dg32, dt32, dc32)
	out.update(f32_de_p_ref32=p, f32_de_gamma_ref32=g)
	p, g, a, vg, vt = norm.de(dg32.astype(np.float64), dt32.astype(np.float64), dc32.astype(np.float64))
	out.update(f32_de_p=p, f32_de_gamma=g, f32_de_varg=vg, f32_de_vart=vt)
	p, d, v = norm.coex(dt32[:40].astype(np.float64), dc32.astype(np.float64))
	out.update(f32_coex_p=p, f32_coex_dot=d, f32_coex_var=v)

	# (h) strong effects: p-values that underflow towards 0 (notebook_de cell 25 shows p==0 in real data)
	eff = rng.normal(size=(12, 1)) * 2.
	ds = rng.normal(size=(12, n)) + eff * dg[0]
	p, g, a, vg, vt = norm.de(dg, ds, dc)
	out.update(se_dt=ds, se_de_p=p, se_de_gamma=g)
	save('G2_edge', **out)


def g3():
	"""p-function table: scipy.stats.beta.cdf(1-R2, dof/2, 0.5) (association.py:249)."""
	dofs = np.array([1, 2, 3, 4, 5, 7, 10, 20, 37, 64, 100, 297, 1000, 9996, 49994, 99979, 499996], dtype=np.float64)
	r2 = np.concatenate([
		[0.],
		10.**np.arange(-300, -12, 24.),
		10.**np.linspace(-12, -0.001, 190),
		1 - 10.**np.linspace(-3, -15, 25),
		np.linspace(0.05, 0.95, 19),
		[1.],
	])
	r2 = np.unique(r2)
	tab = np.empty((len(dofs), len(r2)))
	for i, d in enumerate(dofs):
		tab[i] = beta.cdf(1 - r2, d / 2, 0.5)
	# independent high-precision cross-check on a subset (mpmath), recorded in meta
	import mpmath
	mpmath.mp.dps = 60
	rng = np.random.default_rng(3)
	worst = 0.
	for _ in range(300):
		i = rng.integers(len(dofs))
		j = rng.integers(len(r2))
		if tab[i, j] < 1e-300 or r2[j] == 0:
			continue
		x = 1 - mpmath.mpf(float(r2[j]))
		ref = mpmath.betainc(dofs[i] / 2, 0.5, 0, x, regularized=True)
		worst = max(worst, abs(float((mpmath.mpf(tab[i, j]) - ref) / ref)))
	save('G3_ptable', dof=dofs, r2=r2, p=tab, scipy_vs_mpmath_maxrel=worst)
	return worst


def g4():
	rng = np.random.default_rng(4)
	out = {}
	cases = []
	a = rng.normal(size=(6, 50))
	cases.append(a @ a.T)  # SPD
	b = np.vstack([a[:4], a[0] + a[1], 3 * a[2]])
	cases.append(b @ b.T)  # rank 4 of 6
	cases.append(np.array([[2.5]]))  # 1x1
	q, _ = np.linalg.qr(rng.normal(size=(5, 5)))
	cases.append((q * np.array([1., 1e-3, 2e-8, 0.5e-8, 1e-12])) @ q.T)  # tol-boundary singular values
	cases.append(np.ones((3, 3)))  # rank 1
	a21 = rng.normal(size=(21, 400))
	a21[-1] = 1.
	cases.append(a21 @ a21.T)
	for i, m in enumerate(cases):
		mi, r = inv_rank(m)
		out['m{}'.format(i)] = m
		out['mi{}'.format(i)] = mi
		out['r{}'.format(i)] = r
	out['ncase'] = len(cases)
	save('G4_invrank', **out)


def g5():
	rng = np.random.default_rng(5)
	nx, ny, n = 12, 40, 400
	dg = (rng.random((nx, n)) < 0.15).astype(np.float64)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dt = rng.normal(size=(ny, n)) + (rng.normal(size=(ny, 3)) @ dg[:3]) * 0.5
	out = dict(dg=dg, dc=dc, dt=dt)
	p, g, a, vg, vt = norm.de(dg, dt, dc, single=4, lowmem=False)
	out.update(s4_p=p, s4_gamma=g, s4_alpha=a, s4_varg=vg, s4_vart=vt)
	# low-MOI style design for single=1: each cell has at most one grouping in most cells
	dg1 = np.zeros((6, n))
	lab = rng.integers(0, 9, n)
	for i in range(6):
		dg1[i, lab == i] = 1
	p, g, a, vg, vt = norm.de(dg1, dt, dc, single=1, lowmem=False)
	out.update(s1_dg=dg1, s1_p=p, s1_gamma=g, s1_alpha=a, s1_varg=vg, s1_vart=vt)
	save('G5_single', **out)


def g6():
	"""CLI round trip: tiny TSV inputs and the reference CLI's text outputs."""
	rng = np.random.default_rng(6)
	n = 60
	dt = np.log1p(rng.poisson(2, (14, n)).astype(np.float64))
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	dg = (rng.random((3, n)) < 0.4).astype(np.float64)
	d = tempfile.mkdtemp()
	np.savetxt(os.path.join(d, 'g.tsv'), dg, delimiter='\t', fmt='%i')
	np.savetxt(os.path.join(d, 'e.tsv.gz'), dt, delimiter='\t', fmt='%.8G')
	np.savetxt(os.path.join(d, 'c.tsv'), dc, delimiter='\t', fmt='%.8G')
	env = dict(os.environ, PYTHONPATH=REF, OPENBLAS_NUM_THREADS='1')
	run = lambda *a: subprocess.run([sys.executable, '-W', 'ignore', '-m', 'normalisr'] + list(a), cwd=d, env=env, check=True)
	run('de', 'g.tsv', 'e.tsv.gz', 'c.tsv', 'pv.tsv', 'lfc.tsv', '--vard_out', 'vard.tsv', '--vart_out', 'vart.tsv', '-n', '1')
	run('de', '-m', 'covariate', 'g.tsv', 'e.tsv.gz', 'c.tsv', 'pv4.tsv', 'lfc4.tsv', '-n', '1')
	run('coex', 'e.tsv.gz', 'c.tsv', 'cpv.tsv.gz', '--var_out', 'cvar.tsv', '--dot_out', 'cdot.tsv', '-n', '1', '-d', '1')
	out = {}
	for f in sorted(os.listdir(d)):
		raw = open(os.path.join(d, f), 'rb').read()
		if f.endswith('.gz'):
			raw = gzip.decompress(raw)
		out[f.replace('.', '_')] = np.frombuffer(raw, dtype=np.uint8)
	save('G6_cli', **out)


def g7():
	"""Block-level association_test_1 on one 64x48x1000 tile (association.py:137-260)."""
	rng = np.random.default_rng(7)
	n = 1000
	lat = rng.normal(size=(1, n))
	dx = rng.normal(size=(64, n)) + 0.4 * rng.normal(size=(64, 1)) * lat - 3.
	dy = rng.normal(size=(48, n)) + 0.4 * rng.normal(size=(48, 1)) * lat + 5.
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dci, dcr = inv_rank(np.matmul(dc, dc.T))
	r = association_test_1(0, 0, dx, dy, dc, dci, dcr, dimreduce=0, lowmem=False)
	save('G7_block', dx=dx, dy=dy, dc=dc, dci=dci, dcr=dcr, p=r[2], gamma=r[3], alpha=r[4], vx=r[5], vy=r[6])


def g8():
	"""binnet / bh (binnet.py:77-173): per-row Benjamini-Hochberg q-values of a coex p-matrix, thresholded."""
	from normalisr.binnet import binnet, bh
	rng = np.random.default_rng(8)
	n = 500
	lat = rng.normal(size=(3, n))
	dt = rng.normal(size=(150, n)) + (rng.normal(size=(150, 3)) * (rng.random((150, 3)) < 0.3)) @ lat
	dc = np.vstack([rng.normal(size=(1, n)), np.ones((1, n))])
	p, d, v = norm.coex(dt, dc)
	out = dict(p=p)
	for q in (0.05, 0.2, 0.5):
		out['net_q{}'.format(int(q * 100))] = binnet(p, q)
	p32 = p.astype(np.float32)
	out['p32'] = p32
	for q in (0.05, 0.3):
		out['net32_q{}'.format(int(q * 100))] = binnet(p32, q)
	# tie-heavy p-values (quantised) and exact boundary cases
	pt = np.round(rng.random((60, 60))**3, 2)
	pt = np.triu(pt, 1) + np.triu(pt, 1).T
	out['pt'] = pt
	out['nett_q10'] = binnet(pt, 0.1)
	out['nett_q25'] = binnet(pt, 0.25)
	vec = np.concatenate([rng.random(200)**4, [0., 0., 1., 1., 0.5, 0.5]])
	out['bh_in'] = vec
	out['bh_out'] = bh(vec)
	w = rng.random(vec.size) + 0.1
	out['bh_w'] = w
	out['bh_wout'] = bh(vec, weight=w)
	save('G8_binnet', **out)


def g9():
	"""normvar (norm.py:166-289): per-gene weighted covariate removal, the step right before de/coex."""
	from normalisr.norm import normvar
	rng = np.random.default_rng(9)
	ng, n = 60, 350
	dt = rng.normal(size=(ng, n)) * rng.uniform(0.5, 2, (ng, 1)) - 8 + 0.8 * rng.normal(size=(ng, 1)) * rng.normal(size=(1, n))
	dc = np.vstack([rng.normal(size=(2, n)), (rng.random((1, n)) < 0.4).astype(float), np.ones((1, n))])
	w = np.exp(0.3 * rng.normal(size=n))
	wt = rng.uniform(0, 1.2, ng)
	wt[[3, 17]] = 0.
	dextra = rng.normal(size=(2, n))
	out = dict(dt=dt, dc=dc, w=w, wt=wt, dextra=dextra)
	r = normvar(dt, dc, w, wt)
	out.update(a_dtn=r[0], a_dcn=r[1])
	r = normvar(dt, dc, w, wt, dextra=dextra, cat=2, keepvar=False, normmean=True)
	out.update(b_dtn=r[0], b_dcn=r[1], b_dex=r[2])
	r = normvar(dt, dc, w, wt, cat=0, bs=7)
	out.update(c_dtn=r[0], c_dcn=r[1])
	save('G9_normvar', **out)


def g10():
	"""single=4 variants (association.py:421-576,926-980): dy=None (every pair given all other rows), one dimreduce value per
	gene, a pseudo-inverse truncated to mpc principal components (scikit-learn randomized SVD, random_state=0); single=1 with
	more covariates than the first device kernel accepted."""
	import sklearn
	rng = np.random.default_rng(10)
	nx, ny, n = 12, 40, 400
	dg = (rng.random((nx, n)) < 0.15).astype(np.float64)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dt = rng.normal(size=(ny, n)) + (rng.normal(size=(ny, 3)) @ dg[:3]) * 0.5
	out = dict(dg=dg, dc=dc, dt=dt)
	for rd in (True, False):
		p, d, a, vx, vy = association_tests(dt[:14], None, dc, single=4, return_dot=rd)
		assert a is None and vx is None
		out.update({'sx_p_rd%d' % rd: p, 'sx_dot_rd%d' % rd: d, 'sx_vy_rd%d' % rd: vy})
	dr = rng.integers(0, 3, ny)
	p, g, a, vg, vt = norm.de(dg, dt, dc, single=4, dimreduce=dr)
	out.update(dr=dr, dr_p=p, dr_gamma=g, dr_varg=vg, dr_vart=vt)
	p, g, a, vg, vt = norm.de(dg, dt, dc, single=4, mpc=5, lowmem=False)
	out.update(mpc_p=p, mpc_gamma=g, mpc_alpha=a, mpc_varg=vg, mpc_vart=vt)
	dcd = np.vstack([dc, dc[0] - 2 * dc[1]])  # rank-deficient covariates: the reference's per-grouping pseudo-inverses
	p, g, a, vg, vt = norm.de(dg, dt, dcd, single=4, dimreduce=dr)
	out.update(rdd_dc=dcd, rdd_p=p, rdd_gamma=g, rdd_varg=vg, rdd_vart=vt)
	dg1 = np.zeros((5, n))
	lab = rng.integers(0, 8, n)
	for i in range(5):
		dg1[i, lab == i] = 1
	dcm = np.vstack([rng.normal(size=(39, n)), np.ones((1, n))])
	p, g, a, vg, vt = norm.de(dg1, dt[:16], dcm, single=1, lowmem=False)
	out.update(s1c_dg=dg1, s1c_dc=dcm, s1c_p=p, s1c_gamma=g, s1c_alpha=a, s1c_varg=vg, s1c_vart=vt)
	save('G10_single4', **out)
	return sklearn.__version__


def g11_inputs(seed=11, n=4096):
	"""Rows that are hard for a fixed-point Gram engine (the build's integer engine takes problems from 2048 cells on; the
	reference is fp64 throughout, association.py:224-235): few distinct values per row -- log1p(Poisson) counts from very sparse
	to dense, 0/1 rows from 0.2 % to 30 % dense --, rows whose mean dwarfs their spread, and a heavy-tailed row."""
	rng = np.random.default_rng(seed)
	lam = np.r_[np.geomspace(0.01, 5, 28)]
	lat = rng.normal(size=n)
	counts = rng.poisson(lam[:, None] * np.exp(0.4 * lat - 0.08)[None, :])
	dens = np.geomspace(0.002, 0.3, 8)
	binary = (rng.random((8, n)) < dens[:, None]).astype(np.float64)
	binary[1] = np.maximum(binary[0], binary[1])  # two overlapping sparse rows: a true positive among them
	offset = 1e4 + rng.normal(size=(6, n)) + 0.2 * lat
	heavy = rng.standard_t(2.5, size=(2, n))
	spike = rng.normal(size=(2, n)) * 1e-3
	spike[0, 17] = 50.
	spike[1, 999] = -20.
	dt = np.vstack([np.log1p(counts.astype(np.float64)), binary, offset, heavy, spike])
	batch = rng.integers(0, 4, n)
	onehot = (batch[None, :] == np.arange(4)[:, None]).astype(np.float64)
	z = rng.normal(size=n)
	cov = dict(
		intercept=np.ones((1, n)),
		onehot=np.vstack([onehot, np.ones((1, n))]),  # one-hot batches plus an intercept: rank 4 of 5
		collinear=np.vstack([z, z + 1e-2 * rng.normal(size=n), np.ones(n)]),  # cond(C C^T) ~ 4e4
		none=np.zeros((0, n)),
	)
	dg = np.vstack([(rng.random((2, n)) < 0.03).astype(np.float64), (batch == 0).astype(np.float64)[None, :]])
	return dt, cov, dg


def g11():
	"""Reference outputs for g11_inputs at 4096 cells: coex under every covariate set, de for three groupings."""
	dt, cov, dg = g11_inputs()
	out = dict(dt=dt, dg=dg)
	for name, dc in cov.items():
		out['dc_' + name] = dc
		p, d, v = norm.coex(dt, dc)
		out.update({'coex_%s_p' % name: p, 'coex_%s_dot' % name: d, 'coex_%s_var' % name: v})
		if name == 'onehot':
			continue  # (batch == 0) is in the span of these covariates: nothing to test
		pd_, g, a, vg, vt = norm.de(dg, dt, dc)
		out.update({'de_%s_p' % name: pd_, 'de_%s_gamma' % name: g, 'de_%s_varg' % name: vg, 'de_%s_vart' % name: vt})
	save('G11_i8hard', **out)


def g13():
	"""Reference outputs at 100 000 cells (BASELINE configs[2]'s cell count), fp64 and fp32 inputs; the inputs are rebuilt from the
	seed by g13_inputs on either side, only the outputs are stored."""
	from g13_inputs import g13_inputs
	dt, dc, dg = g13_inputs()
	out = dict(seed=13, n=dt.shape[1], ng=dt.shape[0], check=np.array([dt.sum(), dc.sum(), dg.sum()]))
	p, d, v = norm.coex(dt, dc)
	out.update(coex_p=p, coex_dot=d, coex_var=v)
	pd_, g, a, vg, vt = norm.de(dg, dt, dc, lowmem=False)  # (lowmem=False: alpha is returned)
	out.update(de_p=pd_, de_gamma=g, de_alpha=a, de_varg=vg, de_vart=vt)
	save('G13_100k', **out)


def g14():
	"""BASELINE configs[1] at full size through the reference (nth=8; ~1 minute): outputs of 12 gene rows against all 5000."""
	from g14_inputs import g14_inputs
	dt, dc, rows = g14_inputs()
	p, d, v = norm.coex(dt.astype(np.float64), dc, nth=8)
	rows = rows[:12]  # (1 MB of outputs)
	save('G14_c2', seed=14, rows=rows, check=np.array([float(dt.astype(np.float64).sum()), dc.sum()]), p=p[rows], dot=d[rows], var=v)


def g15():
	"""single=5 (association.py:579-728,969-980; "under development" upstream): every row of dx as a target, only the pairs a mask allows are tested,
	the other allowed rows of the same target join the covariates.  Default tiles (targets in blocks of 10) and small ones (the variance of x is
	written per block: a quirk that makes it depend on the tiling), lowmem on / off, return_dot on / off, a repeated covariate row (rank-deficient sets)."""
	rng = np.random.default_rng(15)
	nx, n = 23, 400
	dx = rng.normal(size=(nx, n))
	dx[2] += 0.5 * dx[0]
	dx[5] += 0.7 * dx[2] - 0.3 * dx[1]
	r1 = rng.normal(size=(1, n))
	dc = np.vstack([r1, r1, np.ones((1, n))])  # a repeated covariate row: every covariate set is rank deficient (the truncated pseudo-inverse, :677-678)
	mask = rng.random((nx, nx)) < 0.3
	np.fill_diagonal(mask, False)
	out = dict(dx=dx, dc=dc, mask=mask)
	for name, kw in (('a', dict(lowmem=False)), ('b', dict(return_dot=False)), ('c', dict(bsx=5, bsy=4, lowmem=False)), ('d', dict(dimreduce=2))):
		p, d, a, vx, vy = association_tests(dx, None, dc, single=5, mask=mask, **kw)
		out.update({name + '_p': p, name + '_dot': d, name + '_vx': vx, name + '_vy': vy})
		if a is not None:
			out[name + '_alpha'] = a
	save('G15_single5', **out)


def g16():
	"""single=1 with one dimreduce per gene (association.py:372-377: dof = n_selected - 1 - rank - dimreduce, elementwise; the reference's own broadcast takes
	a (n_y, 1) column and one tile): a low-MOI design, 3 covariates, dimreduce values 0 / 1 / 2 -- and the scalar calls beside it."""
	rng = np.random.default_rng(670)
	nx, ny, n, nc = 9, 23, 3000, 3
	dx = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
	dy = rng.normal(size=(ny, n))
	dy[:5] += 0.5 * dx[0]
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
	dr = rng.integers(0, 3, ny)
	dr[:3] = [0, 1, 2]
	out = dict(dx=dx, dy=dy, dc=dc, dimreduce=dr)
	p, g, a, vx, vy = association_tests(dx, dy, dc, single=1, dimreduce=dr.reshape(ny, 1), return_dot=False, bsx=nx, bsy=ny, lowmem=False)
	out.update(p=p, gamma=g, alpha=a, vx=vx, vy=vy)
	for v in range(3):
		out['p_scalar%d' % v] = association_tests(dx, dy, dc, single=1, dimreduce=v, return_dot=False)[0]
	save('G16_single1_dimreduce', **out)


def main():
	if len(sys.argv) > 1:  # selected fixtures only, e.g. `make_golden.py g11`
		for name in sys.argv[1:]:
			globals()[name]()
		return
	g1()
	g2()
	worst = g3()
	g4()
	g5()
	g6()
	g7()
	g8()
	g9()
	skl = g10()
	g11()
	g13()
	g14()
	g15()
	g16()
	meta = dict(sklearn=skl, reference='lingfeiwang/normalisr v1.0.0 (/root/reference)', python=sys.version.split()[0],
				numpy=np.__version__, scipy=scipy.__version__, g3_scipy_vs_mpmath_maxrel=worst)
	with open(os.path.join(HERE, 'meta.json'), 'w') as f:
		json.dump(meta, f, indent=1)
	print(meta)


if __name__ == '__main__':
	main()
