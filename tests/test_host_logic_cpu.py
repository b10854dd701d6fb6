"""CPU tests of the host-side mirror (no GPU): de's row filtering / re-inflation, coex's return selection and the
CLI runners' file IO, with the device call (association_tests) replaced by the oracle.  The arithmetic itself is
tested on the GPU; here the Python around it is held to the golden vectors."""
import gzip
import os

import numpy as np
import pytest

import oracle
from conftest import relerr


@pytest.fixture
def oracle_backend(monkeypatch):
	import normalisr_amd.association as assoc
	import normalisr_amd.coex as coex_mod
	import normalisr_amd.de as de_mod

	def fake(dx, dy, dc, **ka):
		ka.pop('nth', None)
		bs = ka.pop('bs', None)
		if bs is not None:
			ka.setdefault('bsx', bs)
			ka.setdefault('bsy', bs)
		return oracle.association_tests(np.asarray(dx, dtype=np.float64), None if dy is None else np.asarray(dy), np.asarray(dc), **ka)
	monkeypatch.setattr(de_mod, 'association_tests', fake)
	monkeypatch.setattr(coex_mod, 'association_tests', fake)
	monkeypatch.setattr(assoc, 'association_tests', fake)
	return fake


def test_de_filter_and_reinflate(golden, oracle_backend):
	from normalisr_amd.de import de, _varying_rows
	g = golden('G1_c1')
	p, gam, a, vg, vt = de(g['dg'], g['dt'], g['dc'], lowmem=False)
	assert relerr(p, g['de_lm0_p']) < 1e-9 and relerr(gam, g['de_lm0_gamma'], 1e-14) < 1e-9 and relerr(a, g['de_lm0_alpha'], 1e-12) < 1e-9
	assert (p[2] == 1).all() and (gam[2] == 0).all() and (a[2] == 0).all() and vg[2] == 0 and (vt[2] == 0).all()  # de.py:107-122
	assert vt.shape == (4, 500) and (vt[0] == vt[1]).all()  # per-gene variance broadcast over the tested rows (Q5)
	x = np.array([[0., 0, 0], [1, 0, 1], [np.nan, np.nan, np.nan], [2, 2, 2.]])
	assert _varying_rows(x).tolist() == [False, True, False, False]  # np.unique treats NaNs as one value (de.py:93)
	assert _varying_rows(np.array([[1, 1, 2], [3, 3, 3]])).tolist() == [True, False]
	with pytest.raises(AssertionError):
		de(np.ones((2, 300)), g['dt'], g['dc'])  # nothing left to test: assert len(ans0) > 0 (association.py:998)


def test_coex_returns(golden, oracle_backend):
	from normalisr_amd.coex import coex
	g = golden('G1_c1')
	ns = int(g['coex_n'])
	p, d, v = coex(g['dt'][:ns], g['dc'], nth=4)
	assert relerr(p, g['coex_p'], 1e-300) < 1e-9 and relerr(d, g['coex_dot'], 1e-14) < 1e-9 and relerr(v, g['coex_var']) < 1e-12
	p2 = coex(g['dt'][:ns], g['dc'], bs=50)[0]  # `bs` accepted
	assert relerr(p2, g['coex_p'], 1e-300) < 1e-9


def test_cli_runners_text_io(golden, oracle_backend, tmp_path):
	"""`normalisr de|coex` on the G6 TSV inputs: the written text equals the reference CLI's up to '%.8G' precision."""
	from normalisr_amd.__main__ import main
	g = golden('G6_cli')
	files = {k: bytes(g[k]) for k in g.files}
	(tmp_path / 'g.tsv').write_bytes(files['g_tsv'])
	(tmp_path / 'e.tsv.gz').write_bytes(gzip.compress(files['e_tsv_gz']))
	(tmp_path / 'c.tsv').write_bytes(files['c_tsv'])
	old = os.getcwd()
	os.chdir(str(tmp_path))
	try:
		assert main(['de', 'g.tsv', 'e.tsv.gz', 'c.tsv', 'pv.tsv', 'lfc.tsv', '--vard_out', 'vard.tsv', '--vart_out', 'vart.tsv', '-n', '1']) == 0
		assert main(['-v', 'coex', 'e.tsv.gz', 'c.tsv', 'cpv.tsv.gz', '--var_out', 'cvar.tsv', '--dot_out', 'cdot.tsv', '-d', '1']) == 0
	finally:
		os.chdir(old)
	load = lambda raw: np.loadtxt(raw.decode().splitlines(), delimiter='\t', ndmin=2)
	for mine, ref in (('pv.tsv', 'pv_tsv'), ('lfc.tsv', 'lfc_tsv'), ('vard.tsv', 'vard_tsv'), ('vart.tsv', 'vart_tsv'),
					  ('cpv.tsv.gz', 'cpv_tsv_gz'), ('cvar.tsv', 'cvar_tsv'), ('cdot.tsv', 'cdot_tsv')):
		got = np.loadtxt(str(tmp_path / mine), delimiter='\t', ndmin=2)
		exp = load(files[ref])
		assert got.shape == exp.shape and relerr(got, exp, 1e-12) < 2e-7, mine
	# identical text for the files that hold exactly representable numbers
	assert (tmp_path / 'vard.tsv').read_bytes().split() == files['vard_tsv'].split()


def test_varying_rows_matches_unique_rule():
	"""de.py:93 keeps groupings with more than one distinct value (np.unique collapses NaNs); the early-exit scan must
	agree on every dtype, on rows that differ only in their last cell, on all-NaN rows and on single-column input."""
	from normalisr_amd.de import _varying_rows
	rng = np.random.default_rng(0)

	def rule(dg):
		return np.array([len(np.unique(x)) > 1 for x in dg], dtype=bool)
	for dt in (np.float32, np.float64, np.int64, np.uint8, bool):
		for n in (1, 2, 5, 257, 258, 1300, 5000):
			a = (rng.random((40, n)) < 0.002).astype(dt)
			a[3] = a[3, 0]
			a[4] = 0
			a[4, n - 1] = 1  # differs in the last cell only
			if np.dtype(dt).kind == 'f':
				a[5] = np.nan
				a[6] = 1
				a[6, n - 1] = np.nan
				a[7, 0] = np.nan
			assert np.array_equal(_varying_rows(a), rule(a)), (dt, n)
	assert _varying_rows(np.zeros((0, 10))).shape == (0, )


def test_spd_inverse_and_partner_runs():
	"""single=4's Cholesky inverse against the eigendecomposition it replaced; and the contiguous-run bookkeeping of the
	merged partner launch (distributed.CoexPlan): partners rank+1..rank+K of every rank are K consecutive blocks of the
	gathered buffer once the first blocks are appended behind the last."""
	from normalisr_amd.single4 import _spd_inverse
	rng = np.random.default_rng(1)
	a = rng.normal(size=(60, 400))
	m = a @ a.T
	w, v = np.linalg.eigh(m)
	assert relerr(_spd_inverse(m), (v / w) @ v.T) < 1e-9
	for world in range(2, 10):
		K = (world - 1) // 2
		for rank in range(world):
			buf = list(range(world)) + [None] * K
			wrap = rank + K - (world - 1)
			for i in range(max(0, wrap)):
				buf[world + i] = buf[i]
			assert buf[rank + 1:rank + 1 + K] == [(rank + 1 + j) % world for j in range(K)]


def test_inv_rank_retries_with_gesvd(monkeypatch, caplog):
	"""inv_rank falls back to LAPACK's gesvd when the default SVD driver does not converge, with one warning, as the reference does
	(association.py:70-76 for one matrix, :111-119 for a stack)."""
	import logging
	from normalisr_amd.association import inv_rank
	rng = np.random.default_rng(12)
	a = rng.normal(size=(5, 40))
	m = a @ a.T
	want, rank = inv_rank(m)
	real = np.linalg.svd
	calls = []

	def failing(x, *args, **ka):
		calls.append(1)
		raise np.linalg.LinAlgError('SVD did not converge')
	monkeypatch.setattr(np.linalg, 'svd', failing)
	with caplog.at_level(logging.WARNING):
		got, r2 = inv_rank(m)
		stack, ranks = inv_rank(np.stack([m, 2 * m, m + np.eye(5)]))
	monkeypatch.setattr(np.linalg, 'svd', real)
	assert r2 == rank and np.allclose(got, want, rtol=1e-10, atol=1e-14)
	assert list(ranks) == [rank] * 3 and np.allclose(stack[1], want / 2, rtol=1e-10, atol=1e-14)
	assert len(calls) == 4 and sum('gesvd' in r.getMessage() for r in caplog.records) == 2  # one warning per call, not per matrix
