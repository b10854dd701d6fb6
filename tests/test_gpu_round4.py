"""GPU tests added in round 4 (K1 with the rows resident on chip left the library in round 5: tools/experiments/k1_res_parity.py): the engine called from several
threads, single=4 / single=1 on the integer engine, one-pass binnet.  Same tolerances as test_gpu_parity.py (BASELINE.json
north_star: 1e-6 relative on Pearson r, t and p; integers, shapes and zeros bit-exact)."""
import os
import sys

import numpy as np
import pytest

import oracle
from conftest import relerr  # noqa: F401
from test_gpu_parity import close, p_close, RTOL, I8_FLOOR  # noqa: F401

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')


@pytest.fixture(scope='module')
def norm():
	import normalisr_amd.normalisr as norm
	return norm


@pytest.fixture(scope='module')
def eng():
	from normalisr_amd.engine import get_engine
	return get_engine()


def test_four_threads_share_one_device(norm, eng):
	"""SURVEY section 8b, Threading: the reference calls its block kernel from a thread pool (association.py:895,997,
	parallel.py:47-52).  Four Python threads x 20 mixed norm.coex / norm.de calls on ONE device (>= 2048 cells, so the integer
	engine with its stream-K slabs, the shared Gram scratch and the fix-up launch are in play; several shapes, so the scratch is
	reallocated in between): every result bit-equal to the serial answer."""
	import threading
	rng = np.random.default_rng(403)
	cases = []
	for i in range(10):
		n = int(rng.choice([2304, 4096, 6000]))
		ng = int(rng.integers(130, 700))
		dt = np.log1p(rng.poisson(1.5, (ng, n)).astype(np.float32 if i % 2 else np.float64))
		dc = np.vstack([rng.normal(size=(int(rng.integers(1, 4)), n)), np.ones((1, n))])
		dg = (rng.random((int(rng.integers(1, 40)), n)) < 0.2).astype(np.float64)
		cases.append(('coex', (dt, dc)))
		cases.append(('de', (dg, dt, dc)))

	def run(kind, a):
		return norm.coex(*a) if kind == 'coex' else norm.de(*a)

	serial = [run(k, a) for k, a in cases]
	results = [[None] * len(cases) for _ in range(4)]
	errors = []

	def worker(t):
		try:
			order = np.random.default_rng(t).permutation(len(cases))
			for j in order:
				results[t][j] = run(*cases[j])
		except Exception as e:  # noqa: BLE001
			errors.append(e)

	threads = [threading.Thread(target=worker, args=(t, )) for t in range(4)]
	for t in threads:
		t.start()
	for t in threads:
		t.join()
	assert not errors, errors
	for t in range(4):
		for j, ref in enumerate(serial):
			for got, want in zip(results[t][j], ref):
				if want is None:
					assert got is None
				else:
					assert np.array_equal(got, want, equal_nan=True), (t, j, cases[j][0])


def test_reference_thread_pool_over_block_kernel(norm):
	"""The reference's own pattern: autopooler(nth=4, dummy=True) over (association_test_1, args, kwargs) tiles
	(association.py:905-909,997) -- here over this build's association_test_1 through normalisr_amd.parallel: the tiles' results
	equal the serial ones bit for bit and the oracle's to tolerance."""
	from normalisr_amd import parallel
	from normalisr_amd.association import association_test_1, inv_rank
	rng = np.random.default_rng(404)
	n, nx, ny, bs = 3000, 300, 260, 64
	dx = rng.normal(size=(nx, n))
	dy = rng.normal(size=(ny, n)) + 0.2 * dx[:1]
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dci, dcr = inv_rank(dc @ dc.T)
	tiles = [(association_test_1, (x0, y0, dx[x0:x0 + bs], dy[y0:y0 + bs], dc, dci, dcr), dict(lowmem=True))
			 for x0 in range(0, nx, bs) for y0 in range(0, ny, bs)]
	serial = parallel.autopooler(1, tiles, dummy=True)
	pooled = parallel.autopooler(4, tiles, dummy=True)
	for a, b in zip(serial, pooled):
		assert a[0] == b[0] and a[1] == b[1]
		for u, v in zip(a[2:], b[2:]):
			assert (u is None and v is None) or np.array_equal(u, v)
	ref = oracle.association_test_1(*tiles[3][1], lowmem=True)
	got = pooled[3]
	p_close(got[2], ref[2])
	close(got[3], ref[3], floor=1e-12)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_single4_on_the_integer_engine(norm, eng, dtype):
	"""single=4 (association.py:421-576,926-980) at cell counts where the large contraction Y~ X~^T runs on the integer Gram engine
	(>= 2048 cells): sparse 0/1 design rows (GSE120861-like gRNA incidence), covariates with an intercept and one-hot batches,
	lowmem=False (alpha), per-gene dimreduce -- against the oracle's per-grouping SVD loop; every pair certified by the guard."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(405)
	nx, ny, n = 40, 260, 3072
	dg = (rng.random((nx, n)) < 0.02).astype(np.float64)
	batch = rng.integers(0, 3, n)
	dc = np.vstack([(batch == 1).astype(float), (batch == 2).astype(float), rng.normal(size=(1, n)), np.ones((1, n))])
	dt = (np.log1p(rng.poisson(2.0, (ny, n))) + (rng.normal(size=(ny, 6)) @ dg[:6]) * 0.8).astype(dtype)
	rtol = 1e-6 if dtype == np.float64 else 2e-5  # (fp32 outputs)
	pc = (lambda p, ref, rt: p_close(p, ref, rt)) if dtype == np.float64 else (
		lambda p, ref, rt: p_close(np.where(ref < 1e-36, ref, p), ref, rt) and (p[ref < 1e-46] == 0).all())  # (fp32 P-values underflow at 1e-45)
	p, gam, a, vx, vy = association_tests(dg, dt, dc, single=4, return_dot=False, lowmem=False)
	assert eng.last_guard['hits'] == 0 and not eng.last_guard['fallback'] and eng.last_guard['worst'] > 0, eng.last_guard
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dt.astype(np.float64), dc, single=4, return_dot=False, lowmem=False)
	assert p.dtype == dtype and p.shape == (nx, ny) and a.shape == (nx, ny, 4)
	assert pc(p, po, rtol) and close(gam, go, rtol, 1e-10) and close(vx, vxo, rtol) and close(vy, vyo, rtol) and close(a, ao, 10 * rtol, 1e-8)
	assert po.min() < 1e-20
	# per-gene dimreduce and covariance output
	dr = rng.integers(0, 3, ny)
	p, d, a, vx, vy = association_tests(dg, dt, dc, single=4, return_dot=True, dimreduce=dr)
	po, do, ao, vxo, vyo = oracle.association_tests(dg, dt.astype(np.float64), dc, single=4, return_dot=True, dimreduce=dr)
	assert pc(p, po, rtol) and close(d, do, rtol, 1e-12) and a is None
	# no covariates at all (the reference warns and goes on, association.py:205-206): the design rows are their own residuals
	p, gam, a, vx, vy = association_tests(dg, dt, np.zeros((0, n)), single=4, return_dot=False)
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dt.astype(np.float64), np.zeros((0, n)), single=4, return_dot=False)
	assert pc(p, po, rtol) and close(gam, go, rtol, 1e-10) and close(vy, vyo, rtol)


def test_single4_guard_reroutes_to_fp64(norm, eng):
	"""Genes the integer engine cannot certify (single spikes over tiny noise: the fixed-point grid is coarse against the bulk of the
	row) make single=4 redo its contraction on the fp64 Gram kernel; the results equal the oracle's either way."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(406)
	nx, ny, n = 12, 64, 65536
	lat = rng.normal(size=n)
	dg = 1e-3 * (rng.normal(size=(nx, n)) + 0.05 * rng.normal(size=(nx, 1)) * lat)
	dg[np.arange(nx), rng.choice(n, nx, replace=False)] = 50.0  # one spike per design row carries all of its variance
	dc = np.ones((1, n))
	dt = 1e-3 * (rng.normal(size=(ny, n)) + 0.05 * rng.normal(size=(ny, 1)) * lat)
	dt[np.arange(ny), rng.choice(n, ny, replace=False)] = 50.0
	p, gam, a, vx, vy = association_tests(dg, dt, dc, single=4, return_dot=False)
	assert eng.last_guard['fallback'] and eng.last_guard['hits'] > 0, eng.last_guard
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dt, dc, single=4, return_dot=False)
	assert p_close(p, po) and close(gam, go, floor=1e-12) and close(vy, vyo, 1e-9)


def test_engine_path_and_throughput_are_logged(norm, caplog):
	"""SURVEY section 5 (metrics / logging; the reference logs its batch decisions, association.py:745-757): under -v a call says
	which engine ran, how fast, and what the accuracy guard decided."""
	import logging
	rng = np.random.default_rng(407)
	dt = rng.normal(size=(150, 2304))
	dc = np.ones((1, 2304))
	with caplog.at_level(logging.INFO):
		norm.coex(dt, dc)
		norm.coex(dt[:, :300], dc[:, :300])
		norm.de((rng.random((2, 2304)) < 0.3).astype(float), dt, dc)
	msgs = [r.getMessage() for r in caplog.records if r.getMessage().startswith('normalisr_amd:')]
	assert len(msgs) == 3, msgs
	assert 'coex, 11175 tests over 2304 cells on the integer Gram engine (46-bit' in msgs[0] and 'every P-value certified' in msgs[0] and 'tests/s' in msgs[0]
	assert 'the fp64 Gram kernel' in msgs[1] and 'no guard needed' in msgs[1]
	assert msgs[2].startswith('normalisr_amd: de, 300 tests over 2304 cells')


def test_host_mirrored_coex_results_equal_shipped_ones(norm, eng, monkeypatch):
	"""NRM_HOST_MIRROR=1: the numpy-out coex path ships only the rows a..b up to column b and mirrors them on the host
	(nrm_host_mirror_rows) -- bit for bit the arrays of the default path, which ships both halves (association.py:1049-1057)."""
	rng = np.random.default_rng(409)
	dt = rng.normal(size=(2600, 3328)).astype(np.float32)  # three row chunks of the pipelined path (>= 32 MB, > 1024 rows)
	dc = np.vstack([rng.normal(size=(1, 3328)), np.ones((1, 3328))])
	assert eng.coex_pipelined_ok(dt, dc, 3328)
	monkeypatch.setenv('NRM_HOST_MIRROR', '0')
	p0, d0, v0 = norm.coex(dt, dc)
	p0, d0 = p0.copy(), d0.copy()
	monkeypatch.setenv('NRM_HOST_MIRROR', '1')
	p1, d1, v1 = norm.coex(dt, dc)
	assert np.array_equal(p0, p1) and np.array_equal(d0, d1) and np.array_equal(v0, v1)
	assert (p1 == p1.T).all() and (d1 == d1.T).all() and (np.diag(p1) == 0).all()


def test_single4_redoes_only_the_genes_the_guard_flags(norm, eng):
	"""A screen with mutually exclusive gRNAs under an intercept (strongly correlated design rows: large kappa) and a few strongly
	associated genes: the guard cannot certify those genes' P-values from the integer engine's products, and only THEY are redone from
	fp64 residuals on the fp64 Gram kernel -- every result equal to the oracle's."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(410)
	nx, ny, n = 60, 400, 16384
	lab = rng.integers(0, nx + 12, n)
	dg = np.zeros((nx, n))
	dg[lab[lab < nx], np.nonzero(lab < nx)[0]] = 1
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dt = rng.normal(size=(ny, n))
	dt[:10] += 0.8 * dg[:10]
	p, gam, a, vx, vy = association_tests(dg, dt, dc, single=4, return_dot=False, lowmem=False)
	g = dict(eng.last_guard)
	po, go, ao, vxo, vyo = oracle.association_tests(dg, dt, dc, single=4, return_dot=False, lowmem=False)
	assert p_close(p, po) and close(gam, go, floor=1e-10) and close(vy, vyo, 1e-9) and close(a, ao, 1e-5, 1e-8)
	if g['fallback']:  # (what the guard decides depends on the bound's constants; when it fires here it must be for few genes)
		assert 0 < g.get('genes_redone', ny) <= 64, g


@pytest.mark.parametrize('sparse', ['1', '0'])
def test_config3_exact_shape_as_de_covariate(monkeypatch, sparse):
	"""(sparse = 1: the products Y~ X~^T from the raw expression rows at the design's entries, csrc/nrm_de_sparse.hip -- what a call takes
	for a gRNA design; 0: K1 + the integer Gram engine with its guard.)
	BASELINE configs[3] at its exact shape as the reference's example runs it (`de -m covariate`, cmd_highmoi.sh:19-22): 1 000 gRNAs x
	15 000 genes x 50 000 cells fp32 through single=4 on the device, resident; sampled gRNAs x sampled genes against the oracle.  The
	oracle's per-grouping loop needs one 1004 x 1004 pseudo-inverse per tested gRNA, so it is run on 5 of them with the other 995 joined to
	the covariates -- the same model for those 5 (single=4 IS "every other grouping a covariate", association.py:421-576)."""
	import torch
	from normalisr_amd.engine import get_engine
	from normalisr_amd.single4 import association_tests_single4
	monkeypatch.setenv('NRM_DE_SPARSE', sparse)
	eng = get_engine()
	nx, ny, n, nc = 1000, 15000, 50000, 5
	gen = torch.Generator(device='cuda').manual_seed(43)
	dc = torch.cat([torch.randn((nc - 1, n), generator=gen, device='cuda'), torch.ones((1, n), device='cuda')]).cpu().numpy().astype(np.float64)
	dx = (torch.rand((nx, n), generator=gen, device='cuda') < 0.01).float()
	dy = torch.randn((ny, n), generator=gen, device='cuda')
	dy[:15] += 0.5 * dx[:15]
	p, gam, a, vx, vy = association_tests_single4(dx, dy, dc, return_dot=False, device_out=True)
	g = dict(eng.last_guard)
	assert tuple(p.shape) == (nx, ny) and (g['worst'] > 0) == (sparse == '0') and (not g['fallback'] or g.get('genes_redone', ny) <= 64), g
	xs = np.array([0, 3, 14, 500, 999])
	ys = np.concatenate([np.arange(16), np.arange(7490, 7500), np.arange(ny - 10, ny)])
	dxh = dx.cpu().numpy().astype(np.float64)
	others = np.delete(np.arange(nx), xs)
	sub = dy[torch.from_numpy(ys).cuda()].cpu().numpy().astype(np.float64)
	po, go, ao, vxo, vyo = oracle.association_tests(dxh[xs], sub, np.vstack([dxh[others], dc]), single=4, return_dot=False)
	take = lambda t: t[torch.from_numpy(xs).cuda()][:, torch.from_numpy(ys).cuda()].cpu().numpy()
	assert po.min() < 1e-20
	assert close(take(p), po, 2e-5, 1e-38) and close(take(gam), go, 2e-5, 1e-7) and close(take(vy), vyo, 2e-5)  # (fp32 outputs)
	assert close(vx[xs], vxo, 1e-6)


@pytest.mark.parametrize('dtype,nc,n,ny', [(np.float32, 5, 6003, 301), (np.float64, 12, 6001, 13), (np.float32, 8, 6000, 64), (np.float64, 9, 6002, 100),
										   (np.float64, 20, 5999, 9)])
def test_single1_streams_the_expression_matrix_once(dtype, nc, n, ny):
	"""single=1 without a transposed copy of the expression matrix (csrc/nrm_single1.hip k_s1_stream + k_s1_cells): rows that are not
	16-byte aligned (n % 4 != 0: the element-load instantiation), gene counts that are not a multiple of the 8 (4) rows a workgroup takes,
	5 / 8 / 9 / 12 / 20 covariates (one pass, one full pass, several passes of which the last reaches back), device-resident inputs and
	outputs; against the oracle (association.py:263-390)."""
	import torch
	from normalisr_amd.association import association_tests
	from normalisr_amd.single1 import association_tests_single1
	rng = np.random.default_rng(500 + nc)
	nx = 23
	lab = rng.integers(0, nx + 20, n)
	dg = np.zeros((nx, n))
	has = lab < nx
	dg[lab[has], np.nonzero(has)[0]] = 1.0
	dbl = rng.choice(np.nonzero(has)[0], 150, replace=False)
	dg[rng.integers(0, nx, 150), dbl] = 1.0
	dg[2, np.nonzero(lab == 2)[0][::3]] = 0.25
	dt = (rng.normal(size=(ny, n)) + 0.5 * dg[rng.integers(0, nx, ny)] * rng.normal(size=(ny, 1))).astype(dtype)
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))])
	ref = oracle.association_tests(dg, dt.astype(np.float64), dc, single=1, lowmem=False, return_dot=False)
	f32 = dtype == np.float32
	out = association_tests(dg, dt, dc, single=1, lowmem=False, return_dot=False)
	dev = association_tests_single1(torch.as_tensor(dg.astype(dtype)).cuda(), torch.as_tensor(dt).cuda(), dc, lowmem=False, return_dot=False, device_out=True)
	dev = tuple(v.cpu().numpy() if hasattr(v, 'is_cuda') else v for v in dev)
	for p, gam, a, vx, vy in (out, dev):
		ok = ref[0] > (1e-30 if f32 else 1e-290)
		assert relerr(p[ok], ref[0][ok]) < (2e-4 if f32 else 1e-8)
		assert close(gam, ref[1], 2e-5 if f32 else 1e-9, 1e-6 if f32 else 1e-12) and close(vy, ref[4], 2e-6 if f32 else 1e-10)
		assert close(vx, ref[3], 2e-6 if f32 else 1e-10)
		assert close(a, ref[2], 2e-4 if f32 else 1e-8, 1e-5 if f32 else 1e-10)
	assert np.array_equal(out[0], dev[0])  # the same kernels on the same values, wherever the inputs lay


@pytest.mark.parametrize('dtype,nc,n,nx,ny,binary', [(np.float32, 5, 9000, 70, 203, True), (np.float64, 3, 8191, 40, 101, False), (np.float32, 0, 4100, 33, 64, True),
												   (np.float32, 8, 6002, 64, 130, False), (np.float64, 2, 5000, 1100, 66, True)])
def test_de_with_a_sparse_design_reads_the_expression_rows_once(monkeypatch, caplog, dtype, nc, n, nx, ny, binary):
	"""single=0 de with a sparse design matrix (csrc/nrm_de_sparse.hip: x~ . y~ = x . y - (y C^T) . b_x, the raw expression rows read once)
	against the oracle (association.py:137-260) and against the dense path it replaces (K1 + K2): 0 / 1 and valued entries, fp32 and fp64
	rows, 0 - 8 covariates, cell counts off the 16-byte grid and off the chunk size, more than 1024 design rows (two passes), gene
	counts off the row blocks, alpha, Pearson r and t, an all-zero design row (constant: the reference's varx 0 -> 1 rule)."""
	import logging
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(900 + nx)
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	if not binary:
		dx *= rng.uniform(0.5, 2.0, dx.shape)
	dx[5] = 0  # a design row without entries
	dx = dx.astype(dtype)
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	dy = (rng.normal(size=(ny, n)) + 3.0 + 0.8 * dx[rng.integers(0, nx, ny)].astype(np.float64) * rng.normal(size=(ny, 1))).astype(dtype)
	ref = oracle.association_tests(dx.astype(np.float64), dy.astype(np.float64), dc, lowmem=False, return_dot=False)
	f32 = dtype == np.float32
	outs = {}
	for mode in ('force', '0'):
		monkeypatch.setenv('NRM_DE_SPARSE', mode)
		with caplog.at_level(logging.DEBUG):
			caplog.clear()
			outs[mode] = association_tests(dx, dy, dc, lowmem=False, return_dot=False, return_stats=True)
		assert ('sparse-design kernel' in caplog.text) == (mode == 'force'), caplog.text[-400:]
	from test_gpu_parity import gamma_close
	for mode, got in outs.items():
		p, gam, a, vx, vy = got[:5]
		ok = ref[0] > (1e-30 if f32 else 1e-290)
		assert relerr(p[ok], ref[0][ok]) < (2e-4 if f32 else 1e-8), mode
		if mode == '0' and n >= 2048 and not f32:  # the dense path runs on the integer engine: its own tolerance (test_gpu_parity.py)
			assert gamma_close(gam, vx, vy[0] if vy.ndim > 1 else vy, ref[1], ref[3], ref[4][0] if ref[4].ndim > 1 else ref[4], I8_FLOOR), mode
		else:
			assert close(gam, ref[1], 2e-5 if f32 else 1e-9, 1e-6 if f32 else 1e-12), mode
		assert close(vy, ref[4], 2e-6 if f32 else 1e-10) and close(vx, ref[3], 2e-6 if f32 else 1e-10), mode
		if nc:
			assert close(a, ref[2], 2e-4 if f32 else (1e-8 if mode != '0' else 1e-6), 1e-5 if f32 else (1e-10 if mode != '0' else 1e-8)), mode
		assert (p[5] == 1).all() and (gam[5] == 0).all()
	s, d = outs['force'], outs['0']
	ok = d[0] > (1e-30 if f32 else 1e-290)
	assert relerr(s[0][ok], d[0][ok]) < (1e-5 if f32 else 1e-7)
	for key in ('r', 't'):
		assert close(s[5][key], d[5][key], 1e-5 if f32 else 1e-6, 1e-6 if f32 else 1e-7)


@pytest.mark.parametrize('dtype,nc,valued', [(np.float32, 4, False), (np.float64, 0, False), (np.float64, 3, False), (np.float64, 2, True)])
def test_single4_with_a_sparse_design(monkeypatch, dtype, nc, valued):
	"""single=4 (association.py:421-576) with the products Y~ X~^T taken from the raw expression rows at the design's entries
	(de_sparse.products, by gene) against the oracle's per-grouping SVD loop and against the integer-engine path; alpha included."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(77 + nc)
	nx, ny, n = 48, 150, 6004
	dx = (rng.random((nx, n)) < 0.02).astype(dtype)
	if valued:
		dx = (dx * rng.uniform(0.5, 2.0, dx.shape)).astype(dtype)
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	dy = (rng.normal(size=(ny, n)) + 2.0 + 0.7 * dx[rng.integers(0, nx, ny)].astype(np.float64) * rng.normal(size=(ny, 1))).astype(dtype)
	ref = oracle.association_tests(dx.astype(np.float64), dy.astype(np.float64), dc, single=4, lowmem=False, return_dot=False)
	f32 = dtype == np.float32
	outs = {}
	for mode in ('force', '0'):
		monkeypatch.setenv('NRM_DE_SPARSE', mode)
		outs[mode] = association_tests(dx, dy, dc, single=4, lowmem=False, return_dot=False)
	for mode, (p, gam, a, vx, vy) in outs.items():
		ok = ref[0] > (1e-30 if f32 else 1e-290)
		assert relerr(p[ok], ref[0][ok]) < (3e-4 if f32 else 1e-7), mode
		assert close(gam, ref[1], 3e-5 if f32 else 1e-7, 1e-6 if f32 else 1e-9) and close(vy, ref[4], 3e-6 if f32 else 1e-9) and close(vx, ref[3], 3e-6 if f32 else 1e-9), mode
		if nc:
			assert close(a, ref[2], 3e-4 if f32 else 1e-6, 1e-5 if f32 else 1e-8), mode
	ok = outs['0'][0] > (1e-30 if f32 else 1e-290)
	assert relerr(outs['force'][0][ok], outs['0'][0][ok]) < (1e-5 if f32 else 1e-7)


def test_sparse_design_path_hands_rows_near_the_covariate_span_back(monkeypatch, caplog):
	"""Expression rows that are a large constant plus a small signal: |y~|^2 = |y|^2 - a . b_y loses its digits, the sparse-design kernel
	counts such rows like pairs the integer engine cannot certify, and the call is redone on K1's two sweeps and the fp64 Gram kernel --
	results as the oracle's either way."""
	import logging
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(31)
	nx, ny, n = 40, 70, 5000
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dy = rng.normal(size=(ny, n)) * 1e-3 + 1e3  # residual^2 / row^2 = 1e-12
	dy[3] += 0.0004 * dx[1]
	ref = oracle.association_tests(dx, dy, dc, return_dot=False)
	assert 1e-200 < ref[0].min() < 1e-4
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')
	with caplog.at_level(logging.INFO):
		p, gam, a, vx, vy = association_tests(dx, dy, dc, return_dot=False)
	assert 'redoing the call on the fp64 matrix cores' in caplog.text
	assert relerr(p, ref[0]) < 1e-6 and close(vy, ref[4], 1e-8)
	dy2 = rng.normal(size=(ny, n)) + 9.0  # an ordinary log-expression row: mean 9, spread 1 -- no hand-back
	caplog.clear()
	with caplog.at_level(logging.INFO):
		p2 = association_tests(dx, dy2, dc, return_dot=False)[0]
	assert 'sparse-design kernel' in caplog.text and 'redoing' not in caplog.text
	assert relerr(p2, oracle.association_tests(dx, dy2, dc, return_dot=False)[0]) < 1e-8


def test_staged_upload_moves_every_byte(eng):
	"""Engine.upload of half a GB and more goes through the library's ring of page-locked staging blocks (csrc/nrm_upload.hip): every
	byte arrives, whatever the size is modulo the 32 MB blocks; the source may be overwritten the moment the call returns (what is
	still in flight comes from the ring); nrm_upload itself also for sizes below its own threshold."""
	import torch
	from normalisr_amd import _lib
	rng = np.random.default_rng(8)
	a = rng.integers(0, 255, size=(512 << 20) + 12345, dtype=np.uint8)  # 16 blocks and a bit
	keep = a.copy()
	d = eng.upload(a)
	a[:] = 0  # the caller's array is its own again
	torch.cuda.synchronize()
	assert d.dtype == torch.uint8 and d.shape == keep.shape
	assert bool((d.cpu() == torch.from_numpy(keep)).all())
	b = rng.standard_normal((3000, 50000), dtype=np.float32)  # 600 MB, a 2-D float array as the expression matrices are
	db = eng.upload(b)
	assert db.dtype == torch.float32 and torch.equal(db.cpu(), torch.from_numpy(b))
	for nbytes in (1, 4096, (32 << 20) - 1, (32 << 20) + 7, (96 << 20)):
		src = rng.integers(0, 255, size=nbytes, dtype=np.uint8)
		dst = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
		_lib.check(eng.lib.nrm_upload(src.ctypes.data, dst.data_ptr(), nbytes, 0, eng._stream()))
		torch.cuda.synchronize()
		assert bool((dst.cpu() == torch.from_numpy(src)).all()), nbytes


def test_single4_inverse_restarts_when_the_diagonal_start_diverges(monkeypatch, caplog):
	"""The Newton-Schulz inverse of M~ = X~ X~^T starts from diag(1 / M_ii), which converges iff 2 diag(M) - M is positive definite -- true of
	the nearly orthogonal rows of a gRNA screen, false for three design rows that share most of their cells; then the iteration must
	notice after two steps and start over from I / ||M||_1.  Results against the oracle either way, and the same from either start."""
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(5)
	nx, ny, n = 40, 90, 6000
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	base = (rng.random(n) < 0.03).astype(np.float64)
	for i in (3, 4, 5):  # three rows that are one row with a handful of cells changed: pairwise correlation ~0.9
		dx[i] = base
		flip = rng.choice(n, 12, replace=False)
		dx[i, flip] = 1 - dx[i, flip]
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dy = rng.normal(size=(ny, n)) + 2.0 + 0.5 * dx[rng.integers(0, nx, ny)] * rng.normal(size=(ny, 1))
	ref = oracle.association_tests(dx, dy, dc, single=4, return_dot=False)
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')
	outs = {}
	for start in ('diagonal', 'norm'):
		monkeypatch.setenv('NRM_S4_START', start)
		outs[start] = association_tests(dx, dy, dc, single=4, return_dot=False)
		ok = ref[0] > 1e-290
		assert relerr(outs[start][0][ok], ref[0][ok]) < 1e-7 and close(outs[start][1], ref[1], 1e-7, 1e-9) and close(outs[start][3], ref[3], 1e-9), start
	assert relerr(outs['diagonal'][0], outs['norm'][0]) < 1e-9
