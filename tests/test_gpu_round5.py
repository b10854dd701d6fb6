"""GPU tests added in round 5: the design-matrix lists and single=1's cell selection built by kernels of the library
(csrc/nrm_design_lists.hip) instead of torch / rocPRIM passes.  Integer work: bit-exact against numpy and against the torch builder kept
as test infrastructure (tests/tools/lists_reference.py)."""
import os
import sys

import numpy as np
import pytest

import oracle
from conftest import relerr  # noqa: F401
from test_gpu_parity import close, p_close, RTOL  # noqa: F401

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))


@pytest.fixture(scope='module')
def norm():
	import normalisr_amd.normalisr as norm
	return norm


@pytest.fixture(scope='module')
def eng():
	from normalisr_amd.engine import get_engine
	return get_engine()


def _design(rng, nx, n, density, binary, dtype):
	dx = (rng.random((nx, n)) < density).astype(np.float64)
	if not binary:
		dx *= rng.uniform(0.5, 2.0, dx.shape)
	dx[min(3, nx - 1)] = 0   # a design row without entries
	dx[0, :40] = 1 if binary else 1.5  # a dense stretch: a list much longer than its neighbours'
	return dx.astype(dtype)


@pytest.mark.parametrize('binary,nx,n,dtype', [
	(True, 70, 9000, np.float32), (False, 33, 4097, np.float64), (True, 1100, 5000, np.float32), (False, 130, 6151, np.float32),
	(True, 64, 2048, np.float64), (True, 1, 2050, np.float32), (True, 1000, 50000, np.float32)])
def test_design_lists_built_by_the_library(eng, binary, nx, n, dtype):
	"""nrm_design_count / _plan / _fill against numpy (CSR), the reader of tests/tools/lists_reference.py (ELL: every entry once, on its slot,
	in its chunk) and the torch builder (the dealing of every chunk, widths, offsets: equal arrays); built twice: the same bits."""
	import torch
	import lists_reference as lr
	from normalisr_amd import de_sparse
	rng = np.random.default_rng(nx + n)
	dx = _design(rng, nx, n, 0.01 if nx >= 1000 else 0.02, binary, dtype)
	d_x = torch.from_numpy(dx).cuda()
	if n % 4:  # rows that are not 16-byte aligned: the element-load instantiation
		assert (d_x.stride(0) * d_x.element_size()) % 16 != 0
	lst = de_sparse.Lists(eng, d_x)
	assert lst.ok and lst.nnz == np.count_nonzero(dx) and lst.binary == binary and (lst.vals is None) == binary and (lst.row_vals is None) == binary
	# CSR: numpy's own listing, row by row, cells ascending
	ii, kk = np.nonzero(dx)
	rp = np.zeros(nx + 1, dtype=np.int64)
	np.add.at(rp, ii + 1, 1)
	assert np.array_equal(lst.row_ptr.cpu().numpy(), np.cumsum(rp))
	assert np.array_equal(lst.cells.cpu().numpy()[:lst.nnz], kk.astype(np.int32))
	if not binary:
		assert np.array_equal(lst.row_vals.cpu().numpy()[:lst.nnz], dx[ii, kk].astype(np.float64))
	# ELL: read back the way k_de_sparse reads it
	got = dict(ell=lst.ell.cpu().numpy(), vals=None if binary else lst.vals.cpu().numpy(), base=lst.base.cpu().numpy(), w=lst.w.cpu().numpy(),
			   slot2x=lst.slot2x.cpu().numpy(), sig=lst.sig.cpu().numpy())
	if nx * n <= 6_000_000:  # (the reader is a Python loop)
		back, padded = lr.decode(nx, n, binary, got['ell'], got['vals'], got['base'], got['w'], got['slot2x'], got['sig'], lst.ngroups)
		assert np.array_equal(back, dx.astype(np.float64)) and padded == lst.padded
	# the torch builder, cells ascending inside a list: the same arrays
	ref = lr.ReferenceLists(d_x, order='cells')
	assert ref.padded == lst.padded and ref.nnz == lst.nnz
	assert np.array_equal(ref.sig.cpu().numpy(), got['sig']) and np.array_equal(ref.w.cpu().numpy(), got['w']) and np.array_equal(ref.base.cpu().numpy(), got['base'])
	assert np.array_equal(ref.ell.cpu().numpy()[:lst.padded], got['ell'][:lst.padded])
	if not binary:
		assert np.array_equal(ref.vals.cpu().numpy()[:lst.padded], got['vals'][:lst.padded])
	again = de_sparse.Lists(eng, d_x)
	assert torch.equal(again.ell, lst.ell) and torch.equal(again.cells, lst.cells) and torch.equal(again.sig, lst.sig)


def test_design_lists_refuse_dense_and_empty_designs_and_describe_the_entries(eng):
	import torch
	from normalisr_amd import de_sparse, _lib
	lst = de_sparse.Lists(eng, torch.ones((40, 3000), dtype=torch.float64, device='cuda'))
	assert not lst.ok and lst.nnz == 40 * 3000 and not hasattr(lst, 'ell') and lst.bits == _lib.DESIGN_HAS1
	lst = de_sparse.Lists(eng, torch.zeros((40, 3000), device='cuda'))
	assert not lst.ok and lst.nnz == 0 and lst.bits == 0
	x = torch.zeros((5, 4100), device='cuda')
	x[1, 7], x[2, 4099], x[4, 2048] = 1.0, -2.0, float('nan')
	lst = de_sparse.Lists(eng, x, max_density=1.0)
	assert lst.ok and lst.nnz == 3 and lst.bits == (_lib.DESIGN_HAS1 | _lib.DESIGN_NOTONE | _lib.DESIGN_NEG | _lib.DESIGN_NAN) and not lst.binary
	x[2, 4099] = float('inf')
	assert de_sparse.Lists(eng, x, max_density=1.0).bits & _lib.DESIGN_GT1
	# kept for the next call on the same unmodified tensor; an in-place write, or another tensor with the same content: listed again
	dx = torch.as_tensor((np.random.default_rng(2).random((40, 5000)) < 0.02).astype(np.float32)).cuda()
	a = de_sparse.lists_for(eng, dx)
	assert de_sparse.lists_for(eng, dx) is a
	dx[3, 7] = 1.0
	b = de_sparse.lists_for(eng, dx)
	assert b is not a and b.nnz in (a.nnz, a.nnz + 1)
	c = de_sparse.lists_for(eng, dx.clone())
	assert c is not b and c.nnz == b.nnz


@pytest.mark.parametrize('nx,n,nc,valued,dtype', [(60, 5000, 5, False, np.float32), (200, 9001, 0, True, np.float64), (33, 4100, 12, False, np.float64), (1000, 50000, 5, False, np.float32)])
def test_single1_selection_built_by_the_library(eng, nx, n, nc, valued, dtype):
	"""nrm_single1_select against numpy's statement of association.py:914-918 for entries >= 0: the codes of the cells, every grouping's own
	cells (ascending), its values and the covariates there, the count of shared cells and the covariate Gram matrix over them."""
	import torch
	from normalisr_amd import de_sparse, _lib
	rng = np.random.default_rng(nx)
	dx = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
	if valued:
		dx *= rng.uniform(0.5, 1.0, dx.shape)
		dx[0, 0] = 1.0
	dx = dx.astype(dtype)
	c64 = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	d_x = torch.from_numpy(dx).cuda()
	lists = de_sparse.Lists(eng, d_x, ell=False, max_density=0.25)
	assert lists.ok
	dev = d_x.device
	nnz = lists.nnz
	mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
	cnt, code, seg, idx, xe = mk(n, torch.int32), mk(n, torch.int32), mk(nx + 1, torch.int64), mk(nnz, torch.int64), mk(nnz, torch.float64)
	ce = mk((nnz, max(nc, 1)), torch.float64)
	rowinfo, info = mk((nx, 3), torch.float64), mk(8, torch.int64)
	gb, nb = int(eng.lib.nrm_single1_select_gram_blocks()), (nc + 7) // 8
	gpart = mk((max(1, nb * (nb + 1) // 2), gb, 64), torch.float64)
	d_c = torch.from_numpy(c64).cuda() if nc else None
	_lib.check(eng.lib.nrm_single1_select(lists.row_ptr.data_ptr(), lists.cells.data_ptr(), 0 if lists.row_vals is None else lists.row_vals.data_ptr(), nx, n, nnz,
										  0 if d_c is None else d_c.data_ptr(), n, nc, cnt.data_ptr(), code.data_ptr(), seg.data_ptr(), idx.data_ptr(), xe.data_ptr(),
										  ce.data_ptr() if nc else 0, rowinfo.data_ptr(), gpart.data_ptr() if nc else 0, info.data_ptr(), 0))
	torch.cuda.synchronize()
	per_cell = (dx != 0).sum(axis=0)
	common = per_cell == 0
	h_info = info.cpu().numpy()
	assert np.array_equal(cnt.cpu().numpy(), per_cell.astype(np.int32)) and h_info[3] == common.sum()
	want_code = np.where(common, _lib.NRM_S1_COMMON, _lib.NRM_S1_SKIP).astype(np.int32)
	w_idx, w_xe, w_seg = [], [], [0]
	for i in range(nx):
		own = np.nonzero((dx[i] != 0) & (per_cell == 1))[0]
		want_code[own] = len(w_idx) + np.arange(own.size)
		w_idx += own.tolist()
		w_xe += dx[i, own].astype(np.float64).tolist()
		w_seg.append(len(w_idx))
	n_e = len(w_idx)
	assert h_info[4] == n_e and np.array_equal(seg.cpu().numpy(), np.array(w_seg)) and np.array_equal(code.cpu().numpy(), want_code)
	assert np.array_equal(idx.cpu().numpy()[:n_e], np.array(w_idx, dtype=np.int64)) and np.array_equal(xe.cpu().numpy()[:n_e], np.array(w_xe))
	ri = rowinfo.cpu().numpy()
	assert np.array_equal(ri[:, 0], np.diff(w_seg).astype(np.float64))
	for i in (0, nx // 2, nx - 1):
		own = np.array(w_xe[w_seg[i]:w_seg[i + 1]])
		assert (ri[i, 1], ri[i, 2]) == ((own.min(), own.max()) if own.size else (np.inf, -np.inf))
	if nc:
		assert np.array_equal(ce.cpu().numpy()[:n_e, :nc], c64[:, w_idx].T)
		hp = gpart.cpu().numpy().sum(axis=1)
		cm = c64 * common
		want = cm @ c64.T
		q = 0
		for bi in range(nb):
			for bj in range(bi, nb):
				blk = hp[q].reshape(8, 8)[:min(8, nc - bi * 8), :min(8, nc - bj * 8)]
				assert np.allclose(blk, want[bi * 8:bi * 8 + 8, bj * 8:bj * 8 + 8], rtol=1e-12, atol=1e-9)
				q += 1


@pytest.mark.parametrize('nc', [0, 5, 12])
def test_single1_through_the_library_selection(norm, nc):
	"""norm.de(single=1) end to end on a low-MOI design against the oracle's per-grouping loop (association.py:263-390,911-925), 0/1 and
	valued entries, device-resident inputs included."""
	import torch
	rng = np.random.default_rng(40 + nc)
	nx, ny, n = 50, 70, 6000
	dx = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
	dx[1] *= rng.uniform(0.5, 1.0, n)
	dx[1, np.nonzero(dx[1])[0][0]] = 1.0
	dy = rng.normal(size=(ny, n))
	dy[:5] += 0.5 * dx[0]
	dc = np.vstack([rng.normal(size=(nc - 1, n)), np.ones((1, n))]) if nc else np.zeros((0, n))
	want = oracle.association_tests(dx, dy, dc, single=1, return_dot=False)
	from normalisr_amd.single1 import association_tests_single1
	for got in (association_tests_single1(dx, dy, dc, return_dot=False),
				association_tests_single1(torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), dc, return_dot=False)):
		assert p_close(got[0], want[0]) and close(got[1], want[1], 1e-9, 1e-12) and close(got[3], want[3]) and close(got[4], want[4])
	with pytest.raises(AssertionError):  # no entry equal to 1 (association.py:914)
		association_tests_single1(0.5 * dx, dy, dc)


def test_sparse_design_path_hands_design_rows_near_the_covariate_span_back(monkeypatch, caplog):
	"""The design side of the same difference (round-4 verdict, weak item 8): a valued design row that all but coincides with a covariate
	(a gRNA that marks one batch) has |x~|^2 = |x|^2 - a . b at 1e-12 of |x|^2; k_design_stats counts it like the expression-side rows,
	the call is redone on K1 and the fp64 Gram kernel, which residualise first -- results as the oracle's; a resident plan remembers the
	verdict and stops trying the sparse kernels; single=4 (M~ from the same kernels) hands back too."""
	import logging
	import torch
	from normalisr_amd.association import association_tests
	from normalisr_amd import distributed as nd
	rng = np.random.default_rng(77)
	nx, ny, n = 40, 70, 6000
	batch = (rng.random(n) < 0.03).astype(np.float64)
	dc = np.vstack([batch, rng.normal(size=(1, n)), np.ones((1, n))])
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	dx[5] = batch * (1.0 + 1e-6 * rng.normal(size=n))
	dy = rng.normal(size=(ny, n))
	dy[3] += 3e6 * (dx[5] - batch)  # an effect carried by the part of the row the covariates do not explain
	ref = oracle.association_tests(dx, dy, dc, return_dot=False)
	assert ref[0][5].min() < 1e-6
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')
	with caplog.at_level(logging.INFO):
		p, gam, a, vx, vy = association_tests(dx, dy, dc, return_dot=False)
	assert 'redoing the call on the fp64 matrix cores' in caplog.text
	assert p_close(p, ref[0]) and close(gam, ref[1], 1e-6, 1e-9) and close(vx, ref[3], 1e-6)
	# a resident plan: the first results() hands back and remembers; later steps run on K1 + the Gram engines, no sparse attempt
	plan = nd.DePlan(torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), dc)
	caplog.clear()
	with caplog.at_level(logging.WARNING):
		for _ in range(3):
			plan.step()
			pp = plan.results()[0]
			assert p_close(pp, ref[0])
	assert caplog.text.count('too close to the span of the covariates') == 1
	# single=4 on the same design: a nearly dependent row given the covariates -- the sparse M~ hands back, the fp64 path decides
	dx2 = dx.copy()
	dx2[5] = batch * (1.0 + 1e-3 * rng.normal(size=n))
	ref4 = oracle.association_tests(dx2, dy, dc, single=4, return_dot=False)
	caplog.clear()
	with caplog.at_level(logging.INFO):
		p4 = association_tests(dx2, dy, dc, single=4, return_dot=False)[0]
	assert 'too close to the span of the covariates' in caplog.text and p_close(p4, ref4[0])


@pytest.mark.parametrize('resident_rows', [False, True])
def test_sparse_design_path_redoes_only_the_rows_near_the_covariate_span(monkeypatch, caplog, resident_rows):
	"""One high-mean, low-variance gene among hundreds (|y~|^2 < 1e-4 |y|^2: the differences the sparse-design kernel takes lose their digits) used to send
	the WHOLE call to K1 + the fp64 Gram kernel (round-4 advice).  Only the flagged rows are redone now and their columns replaced: results as the
	oracle's for every row, P-values, statistic, alpha, variances, r and t; the log says how many rows."""
	import logging
	import torch
	from normalisr_amd.association import association_tests
	rng = np.random.default_rng(77)
	nx, ny, n = 48, 300, 6000
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dy = rng.normal(size=(ny, n)) + 9.0
	bad = [17, 205]
	dy[bad] = 1e3 + 1e-3 * rng.normal(size=(2, n))
	dy[17] += 4e-4 * dx[3]
	dy[40] += 0.6 * dx[5]
	ref = oracle.association_tests(dx, dy, dc, return_dot=False, lowmem=False)
	assert ref[0][3, 17] < 1e-4 and ref[0][5, 40] < 1e-6
	monkeypatch.setenv('NRM_DE_SPARSE', 'force')
	arg = torch.from_numpy(dy).cuda() if resident_rows else dy
	with caplog.at_level(logging.INFO):
		got = association_tests(dx, arg, dc, return_dot=False, lowmem=False, return_stats=True)
	assert '2 of 300 expression rows too close to the span of the covariates' in caplog.text and 'redoing the call' not in caplog.text
	assert p_close(got[0], ref[0]) and close(got[1], ref[1], 1e-6, 1e-12) and close(got[2], ref[2], 1e-6, 1e-9) and close(got[3], ref[3], 1e-9) and close(got[4], ref[4], 1e-7)
	ro, to = oracle.pearson_r_t((ref[1].T * ref[3]).T, ref[3], ref[4], got[5]['dof'])  # dot = gamma varx
	assert close(got[5]['r'], ro, 1e-6, 1e-9) and close(got[5]['t'], to, 1e-6, 1e-6)
	# without the two rows: no redo at all
	keep = np.setdiff1d(np.arange(ny), bad)
	caplog.clear()
	with caplog.at_level(logging.INFO):
		p2 = association_tests(dx, dy[keep], dc, return_dot=False)[0]
	assert 'too close to the span' not in caplog.text and p_close(p2, ref[0][:, keep])


_TORCH_FREE_ENTRIES = r'''

import ctypes, sys
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
lib.nrm_last_error.restype = ctypes.c_char_p
d = np.load(sys.argv[2])
vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
i64, dbl = ctypes.c_int64, ctypes.c_double
code = lambda a: 1 if a.dtype == np.float64 else 0
out = {}

def single(which, dx, dy, dc, dci=None, rank=0, lowmem=False, return_dot=0, out_dtype=np.float64):
	nx, n = dx.shape
	ny, nc = dy.shape[0], dc.shape[0]
	p, st, vy = (np.empty((nx, ny), dtype=out_dtype) for _ in range(3))
	vx = np.empty(nx, dtype=out_dtype)
	al = None if lowmem else np.empty((nx, ny, nc), dtype=out_dtype)
	if which == 1:
		rc = lib.nrm_association_tests_single1_host(vp(dx), code(dx), i64(nx), vp(dy), code(dy), i64(ny), vp(dc), 1, i64(nc), i64(n), 0, return_dot,
			vp(p), vp(st), vp(al), vp(vx), vp(vy), code(p))
	else:
		rc = lib.nrm_association_tests_single4_host(vp(dx), code(dx), i64(nx), vp(dy), code(dy), i64(ny), vp(dc), 1, i64(nc), i64(n), vp(dci), int(rank), 0, return_dot,
			dbl(1e-8), vp(p), vp(st), vp(al), vp(vx), vp(vy), code(p))
	return rc, p, st, al, vx, vy

for name in [k[3:] for k in d.files if k.startswith('dx_')]:
	which = int(d['which_' + name])
	rc, p, st, al, vx, vy = single(which, d['dx_' + name], d['dy_' + name], d['dc_' + name], d['dci_' + name] if which == 4 else None,
		int(d['rank_' + name]) if which == 4 else 0, lowmem=bool(d['lowmem_' + name]), out_dtype=np.float32 if d['dy_' + name].dtype == np.float32 else np.float64)
	out['rc_' + name] = rc
	out['err_' + name] = lib.nrm_last_error().decode() if rc else ''
	if rc == 0:
		out['p_' + name], out['st_' + name], out['vx_' + name], out['vy_' + name] = p, st, vx, vy
		if al is not None:
			out['al_' + name] = al
# single=0 de on a sparse design through the whole-problem entry (takes the sparse-design kernels by itself)
dx, dy, dc, dci = d['s0_dx'], d['s0_dy'], d['s0_dc'], d['s0_dci']
nx, n = dx.shape
ny, nc = dy.shape[0], dc.shape[0]
p, gam, vy = np.empty((nx, ny)), np.empty((nx, ny)), np.empty(ny)
vx, al = np.empty(nx), np.empty((nx, ny, nc))
rc = lib.nrm_association_tests_host(vp(dx), 1, i64(nx), vp(dy), 1, i64(ny), vp(dc), 1, i64(nc), i64(n), vp(dci), int(d['s0_rank']), 0, 0,
	vp(p), vp(gam), vp(al), vp(vx), vp(vy), None, None, 1)
assert rc == 0, lib.nrm_last_error()
out.update(s0_p=p, s0_gam=gam, s0_vx=vx, s0_vy=vy, s0_al=al)
# binnet
pm = d['bn_p']
net, tot = np.empty(pm.shape, dtype=np.uint8), i64(-1)
rc = lib.nrm_binnet_host(vp(pm), 1, i64(pm.shape[0]), dbl(0.05), vp(net), ctypes.byref(tot))
assert rc == 0, lib.nrm_last_error()
out.update(bn_net=net, bn_tot=tot.value)
assert 'torch' not in [m for m, v in sys.modules.items() if v is not None]
np.savez(sys.argv[3], **out)
'''


def test_c_entries_for_the_crispr_methods_from_a_process_without_torch(tmp_path, golden, monkeypatch):
	"""The association_tests seam in C for the calls of BASELINE configs[3] (round-4 verdict, missing item 2): a process that cannot import
	torch binds nrm_association_tests_single1_host / _single4_host / nrm_association_tests_host (sparse-design path) / nrm_binnet_host with
	ctypes alone -- golden G5 (the reference's own single=1 and single=4 outputs), G8 (binnet), a configs[3]-shaped sample (sparse 0/1
	design, fp32 rows, alpha) against the oracle, and what the entries do not cover answered with NRM_E_UNSUPPORTED."""
	import subprocess
	from normalisr_amd import _lib
	g5, g8 = golden('G5_single'), golden('G8_binnet')
	rng = np.random.default_rng(55)
	cases = {}

	def add(name, which, dx, dy, dc, lowmem=False):
		dci, rank = oracle.inv_rank(dc @ dc.T) if dc.shape[0] else (np.zeros((0, 0)), 0)
		cases.update({'dx_' + name: dx, 'dy_' + name: dy, 'dc_' + name: dc, 'dci_' + name: dci, 'rank_' + name: rank, 'which_' + name: which, 'lowmem_' + name: lowmem})
	add('g5s4', 4, g5['dg'], g5['dt'], g5['dc'])
	add('g5s1', 1, g5['s1_dg'], g5['dt'], g5['dc'])
	# configs[3]-shaped samples: gRNA incidence (1 % / 0.1 %), expression rows fp32, 5 covariates with an intercept
	nx, ny, n = 96, 130, 45056
	dc = np.vstack([rng.normal(size=(4, n)), np.ones((1, n))])
	dx4 = (rng.random((nx, n)) < 0.01).astype(np.float32)
	dx1 = (rng.random((nx, n)) < 1.0 / nx).astype(np.float32)
	dy = rng.normal(size=(ny, n)).astype(np.float32)
	dy[:6] += (0.3 * dx4[0] + 0.3 * dx1[1]).astype(np.float32)
	add('c4s4', 4, dx4, dy, dc)
	add('c4s1', 1, dx1, dy, dc, lowmem=True)
	add('neg', 1, np.where(rng.random((6, 3000)) < 0.1, -1.0, 0.0) + np.eye(6, 3000), rng.normal(size=(7, 3000)), np.ones((1, 3000)), lowmem=True)  # entries < 0: not covered
	add('rdef', 4, g5['dg'], g5['dt'], np.vstack([g5['dc'], g5['dc'][:1]]))  # rank-deficient covariates: not covered
	s0_dx = (rng.random((64, 8192)) < 0.02).astype(np.float64)
	s0_dy = rng.normal(size=(100, 8192)) + 0.4 * s0_dx[3]
	s0_dc = np.vstack([rng.normal(size=(2, 8192)), np.ones((1, 8192))])
	s0_dci, s0_rank = oracle.inv_rank(s0_dc @ s0_dc.T)
	np.savez(tmp_path / 'in.npz', s0_dx=s0_dx, s0_dy=s0_dy, s0_dc=s0_dc, s0_dci=s0_dci, s0_rank=s0_rank, bn_p=g8['p'], **cases)
	env = dict(os.environ, NRM_DE_SPARSE='force')  # (the size rule would leave the small cases to the dense kernels)
	r = subprocess.run([sys.executable, '-c', _TORCH_FREE_ENTRIES, _lib.LIB_PATH, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], capture_output=True, text=True,
					   timeout=900, env=env)
	assert r.returncode == 0, r.stderr[-3000:]
	o = np.load(tmp_path / 'out.npz')
	for name in ('g5s4', 'g5s1', 'c4s4', 'c4s1'):
		assert int(o['rc_' + name]) == 0, str(o['err_' + name])
	# golden G5: the reference's own outputs
	assert p_close(o['p_g5s4'], g5['s4_p']) and close(o['st_g5s4'], g5['s4_gamma'], floor=1e-12) and close(o['al_g5s4'], g5['s4_alpha'], floor=1e-9)
	assert close(o['vx_g5s4'], g5['s4_varg'], 1e-9) and close(o['vy_g5s4'], g5['s4_vart'], 1e-9)
	assert p_close(o['p_g5s1'], g5['s1_p']) and close(o['st_g5s1'], g5['s1_gamma'], floor=1e-12) and close(o['al_g5s1'], g5['s1_alpha'], floor=1e-9)
	assert close(o['vx_g5s1'], g5['s1_varg'], 1e-9) and close(o['vy_g5s1'], g5['s1_vart'], 1e-9)
	# the configs[3]-shaped samples against the oracle's loops (fp32 rows: fp32 outputs)
	for name, single, dx in (('c4s4', 4, dx4), ('c4s1', 1, dx1)):
		ref = oracle.association_tests(dx.astype(np.float64), dy.astype(np.float64), dc, single=single, lowmem=name == 'c4s1', return_dot=False)
		ok = ref[0] > 1e-30
		assert relerr(o['p_' + name][ok], ref[0][ok]) < 2e-4 and close(o['st_' + name], ref[1], 2e-5, 1e-6) and close(o['vy_' + name], ref[4], 2e-6) and close(o['vx_' + name], ref[3], 2e-6)
		if name == 'c4s4':
			assert close(o['al_' + name], ref[2], 2e-4, 1e-5)
	assert int(o['rc_neg']) == _lib.NRM_E_UNSUPPORTED and 'entries >= 0' in str(o['err_neg'])
	assert int(o['rc_rdef']) == _lib.NRM_E_UNSUPPORTED and 'full-rank' in str(o['err_rdef'])
	ref = oracle.association_tests(s0_dx, s0_dy, s0_dc, lowmem=False, return_dot=False)
	assert p_close(o['s0_p'], ref[0]) and close(o['s0_gam'], ref[1], floor=1e-12) and close(o['s0_al'], ref[2], floor=1e-9) and close(o['s0_vx'], ref[3]) and close(o['s0_vy'], ref[4])
	assert np.array_equal(o['bn_net'].astype(bool), g8['net_q5']) and int(o['bn_tot']) == int(g8['net_q5'].sum())


_NO_TORCH_CRISPR = r'''
import sys
sys.modules['torch'] = None  # `import torch` raises ImportError from here on
sys.path.insert(0, sys.argv[1])
import numpy as np
import normalisr.normalisr as norm            # the drop-in import name
d = np.load(sys.argv[2])
dg1, dg4, dt, dc = d['dg1'], d['dg4'], d['dt'], d['dc']
out = {}
for name, dg, single in (('s1', dg1, 1), ('s4', dg4, 4)):
	p, lfc, a, vg, vt = norm.de(dg, dt, dc, single=single, lowmem=False)
	out.update({'p_' + name: p, 'lfc_' + name: lfc, 'a_' + name: a, 'vg_' + name: vg, 'vt_' + name: vt})
out['net'] = norm.binnet(d['pc'], 0.05)
nv = norm.normvar(d['lcpm'], dc, d['w'], d['wt'])
out['nv_t'], out['nv_c'] = nv[0], nv[1]
nv12 = norm.normvar(d['lcpm'], np.vstack([dc] * 4), d['w'], d['wt'])  # 12 covariates, each given four times: the entry's Gram-launch form (round 6), ranks by inv_rank's rule
out['nv12_t'] = nv12[0]
try:
	norm.normvar(d['lcpm'], np.vstack([dc] * 11), d['w'], d['wt'])  # 33 covariates: beyond the entry, needs the package's form with numpy's stacked SVD, i.e. torch
	out['nv_unsupported'] = 0
except NotImplementedError:
	out['nv_unsupported'] = 1
try:
	from normalisr_amd.association import association_tests
	association_tests(dg1, None, dc, single=1)  # outside what the library's entries cover (and the reference itself has no such path)
	out['unsupported'] = 0
except NotImplementedError:
	out['unsupported'] = 1
assert not any(m == 'torch' or m.startswith('torch.') for m, v in sys.modules.items() if v is not None)
np.savez(sys.argv[3], **out)
'''


def test_de_methods_and_binnet_of_the_package_without_torch(tmp_path, golden):
	"""`norm.de(..., single=1 | 4)` and `norm.binnet` in a process that cannot import torch: the package routes them to the library's whole-problem
	entries (nrm_association_tests_single1_host / _single4_host / nrm_binnet_host) -- every method of `normalisr de` and the binarisation need numpy and
	libnormalisr_hip.so only.  Against the oracle; a call the entries do not cover says so (NotImplementedError) instead of answering something else."""
	import subprocess
	rng = np.random.default_rng(515)
	nx, ny, n = 40, 90, 6000
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dg4 = (rng.random((nx, n)) < 0.03).astype(np.float64)
	dg1 = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
	dg1[7] = 0  # a grouping without cells: de drops and re-inflates it
	dt = rng.normal(size=(ny, n)) + 0.5 * dg4[0] + 0.5 * dg1[1]
	g8 = golden('G8_binnet')
	lcpm = (rng.normal(size=(ny, n)) - 9).astype(np.float32)
	w, wt = np.exp(0.3 * rng.normal(size=n)), rng.uniform(0, 1.5, ny)
	np.savez(tmp_path / 'in.npz', dg1=dg1, dg4=dg4, dt=dt, dc=dc, pc=g8['p'], lcpm=lcpm, w=w, wt=wt)
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	r = subprocess.run([sys.executable, '-c', _NO_TORCH_CRISPR, root, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], capture_output=True, text=True, timeout=900)
	assert r.returncode == 0, r.stderr[-3000:]
	o = np.load(tmp_path / 'out.npz')
	for name, dg, single in (('s1', dg1, 1), ('s4', dg4, 4)):
		ref = oracle.de(dg, dt, dc, single=single, lowmem=False)
		assert p_close(o['p_' + name], ref[0]) and close(o['lfc_' + name], ref[1], floor=1e-12) and close(o['a_' + name], ref[2], 1e-8, 1e-9)
		assert close(o['vg_' + name], ref[3], 1e-9, 1e-15) and close(o['vt_' + name], ref[4], 1e-9, 1e-15)  # (the grouping without cells: variance 0 on both sides)
	assert (o['p_s1'][7] == 1).all() and (o['lfc_s1'][7] == 0).all()
	assert np.array_equal(o['net'], g8['net_q5']) and int(o['unsupported']) == 1
	refn = oracle.normvar(lcpm.astype(np.float64), dc, w, wt)
	assert np.abs(o['nv_t'] - refn[0]).max() < 1e-6 * np.abs(refn[0]).max() and close(o['nv_c'], refn[1], 1e-12, 1e-15) and int(o['nv_unsupported']) == 1
	ref12 = oracle.normvar(lcpm.astype(np.float64), np.vstack([dc] * 4), w, wt)  # (the repeated covariates change nothing but the ranks' bookkeeping: the same projection)
	assert np.abs(o['nv12_t'] - ref12[0]).max() < 1e-6 * np.abs(ref12[0]).max()


_CLI_NO_TORCH = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
tmp = sys.argv[2]
from normalisr_amd.__main__ import main
f = lambda name: os.path.join(tmp, name)
main(['coex', f('exp.npy'), f('cov.npy'), f('pv.npy'), '--dot_out', f('dot.npy'), '--var_out', f('var.npy')])
main(['binnet', f('pv.npy'), f('net.npy'), '0.05'])
for m in ('ignore', 'single', 'covariate'):
	main(['de', f('dg1.npy' if m == 'single' else 'dg.npy'), f('exp.npy'), f('cov.npy'), f('de_pv_%s.npy' % m), f('de_lfc_%s.npy' % m), '-m', m, '--vart_out', f('de_vt_%s.npy' % m)])
main(['normvar', f('lcpm.npy'), f('w.npy'), f('cov.npy'), f('wt.npy'), f('nv_exp.npy'), f('nv_cov.npy')])
assert not any(m == 'torch' or m.startswith('torch.') for m in sys.modules), 'the command line imported torch'
import torch  # the runtime torch bundles was loaded first: torch still finds its GPU in this process
assert torch.cuda.is_available() and float(torch.ones(3, device='cuda').sum()) == 3.0
'''


def test_command_line_runs_on_the_library_entries_without_importing_torch(tmp_path, norm):
	"""`normalisr coex | binnet | de -m ignore|single|covariate` from files on one GPU: through the library's whole-problem entries -- torch is not imported
	(1.0 of the 1.3 s of a small call; tools/cli_startup.sh) -- with the results of the torch engine; the HIP runtime torch bundles is loaded by itself
	first, so a later `import torch` in the same process still works (one runtime per process)."""
	import subprocess
	rng = np.random.default_rng(12)
	ng, nx, n = 150, 12, 3000
	dt = np.log1p(rng.poisson(2.0, (ng, n))).astype(np.float64)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones(n)])
	dg = (rng.random((nx, n)) < 0.05).astype(np.float64)
	dg1 = (rng.random((nx, n)) < 1.0 / nx).astype(np.float64)
	dt[:5] += 0.5 * dg[0] + 0.5 * dg1[1]
	lcpm, w, wt = rng.normal(size=(ng, n)) - 9, np.exp(0.3 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)
	for name, a in (('exp', dt), ('cov', dc), ('dg', dg), ('dg1', dg1), ('lcpm', lcpm), ('w', w), ('wt', wt)):
		np.save(tmp_path / (name + '.npy'), a)
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = {k: v for k, v in os.environ.items() if k != 'NRM_HOST_ENTRY'}
	r = subprocess.run([sys.executable, '-c', _CLI_NO_TORCH, root, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
	assert r.returncode == 0, r.stderr[-3000:]
	ld = lambda name: np.load(tmp_path / name)
	p, dot, var = norm.coex(dt, dc)
	assert p_close(ld('pv.npy'), p) and close(ld('dot.npy'), dot, 1e-9, 1e-12) and close(ld('var.npy'), var, 1e-12)
	assert np.array_equal(ld('net.npy').astype(bool), norm.binnet(p, 0.05))
	refn = oracle.normvar(lcpm, dc, w, wt)
	assert np.abs(ld('nv_exp.npy') - refn[0]).max() < 1e-9 * np.abs(refn[0]).max() and close(ld('nv_cov.npy'), refn[1], 1e-12, 1e-15)
	for m, single, d in (('ignore', 0, dg), ('single', 1, dg1), ('covariate', 4, dg)):
		ref = oracle.de(d, dt, dc, single=single)
		assert p_close(ld('de_pv_%s.npy' % m), ref[0]) and close(ld('de_lfc_%s.npy' % m), ref[1], 1e-8, 1e-12) and close(ld('de_vt_%s.npy' % m), ref[4], 1e-9, 1e-15)


_NO_TORCH_STREAMING = r'''
import sys
sys.modules['torch'] = None  # `import torch` raises ImportError from here on
sys.path.insert(0, sys.argv[1])
import numpy as np
from normalisr_amd.association import association_tests
d = np.load(sys.argv[2])
out = {}
for k in [k[3:] for k in d.files if k.startswith('dx_')]:
	r = association_tests(d['dx_' + k], d['dy_' + k], d['dc_' + k], return_dot=bool(d['rd_' + k]), lowmem=False, return_stats=True)
	out.update({'p_' + k: r[0], 'st_' + k: r[1], 'al_' + k: r[2], 'vx_' + k: r[3], 'vy_' + k: r[4], 'r_' + k: r[5]['r'], 't_' + k: r[5]['t']})
np.savez(sys.argv[3], **out)
'''


def test_c_entry_streams_case_control_de(tmp_path):
	"""de with nx + nc <= 32 (BASELINE configs[2]: one grouping, 20 covariates) through nrm_association_tests_host from a process without torch: the
	entry takes the streaming kernel as the Python engine does (csrc/nrm_host_entries.hip: nrm_host_de_streaming) -- an intercept anywhere among the
	covariates (it moves to the end of Z; alpha comes back in the caller's order), none, rank-deficient covariates, cell counts off the 16- and
	128-cell grids, fp32 and fp64 rows, 32 rows in Z exactly; P, statistic, alpha, variances, r and t against the oracle."""
	import subprocess
	rng = np.random.default_rng(303)
	cases = {}
	shapes = [(1, 300, 4096, 20, 5, True, False), (3, 77, 1003, 4, 0, False, True), (12, 130, 2500, 20, -1, True, True), (2, 40, 333, 0, -1, False, False),
			  (5, 64, 640, 6, 2, True, False), (1, 500, 100000, 20, 19, True, False)]
	for i, (nx, ny, n, nc, ci, f32, rd) in enumerate(shapes):
		dc = rng.normal(size=(nc, n))
		if nc and ci >= 0:
			dc[ci] = 1.7
		if i == 4:
			dc[4] = dc[0] - 2.0 * dc[1]  # rank-deficient covariates
		dx = (rng.random((nx, n)) < 0.4).astype(np.float64)
		dy = rng.normal(size=(ny, n)) + 5 + 0.2 * dx[0]
		if f32:
			dx, dy = dx.astype(np.float32), dy.astype(np.float32)
		cases.update({'dx_%d' % i: dx, 'dy_%d' % i: dy, 'dc_%d' % i: dc, 'rd_%d' % i: rd})
	# a constant design row beside an intercept: the covariates explain it exactly; its residual is rounding noise that the streaming formula must not
	# take for a direction (R^2 of millions, the reference's assertion) -- nrm_residualize_wide clears it: variance 0 -> 1, P = 1
	dgc = (rng.random((4, 3000)) < 0.2).astype(np.float64)
	dgc[2] = 1.0
	dcc = np.vstack([rng.normal(size=(2, 3000)), np.ones(3000)])
	dtc = np.log1p(rng.poisson(2.0, (260, 3000))).astype(np.float32)
	cases.update(dx_c=dgc, dy_c=dtc, dc_c=dcc, rd_c=False)
	np.savez(tmp_path / 'in.npz', **cases)
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	r = subprocess.run([sys.executable, '-c', _NO_TORCH_STREAMING, root, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], capture_output=True, text=True, timeout=900)
	assert r.returncode == 0, r.stderr[-3000:]
	o = np.load(tmp_path / 'out.npz')
	for i, (nx, ny, n, nc, ci, f32, rd) in enumerate(shapes):
		dx, dy, dc = cases['dx_%d' % i].astype(np.float64), cases['dy_%d' % i].astype(np.float64), cases['dc_%d' % i]
		ref = oracle.association_tests(dx, dy, dc, return_dot=rd, lowmem=False)
		ptol, stol = (3e-4, 3e-5) if f32 else (1e-6, 1e-8)
		ok = ref[0] > (1e-30 if f32 else 1e-290)
		k = str(i)
		assert relerr(o['p_' + k][ok], ref[0][ok]) < ptol, (i, relerr(o['p_' + k][ok], ref[0][ok]))
		assert close(o['st_' + k], ref[1], stol, 1e-9) and close(o['vx_' + k], ref[3], stol, 1e-15) and close(o['vy_' + k], ref[4], stol, 1e-15), i
		if nc:
			assert close(o['al_' + k], ref[2], 30 * stol, 1e-5 if f32 else 1e-9), i
		dof = n - 1 - np.linalg.matrix_rank(dc) - 0 if nc else n - 1
		ro, to = oracle.pearson_r_t((ref[1].T * ref[3]).T if not rd else ref[1], ref[3], ref[4], dof)
		assert close(o['r_' + k], ro, stol * 10, 1e-7) and close(o['t_' + k], to, stol * 100, 1e-3), i
	from normalisr_amd.association import association_tests
	keep = np.array([0, 1, 3])
	refc = oracle.association_tests(dgc[keep], dtc.astype(np.float64), dcc, return_dot=False)
	for got in ((o['p_c'], o['st_c']), association_tests(dgc, dtc, dcc, return_dot=False)[:2]):  # the C entry (child) and the torch engine (here)
		assert (got[0][2] == 1).all() and (got[1][2] == 0).all() and relerr(got[0][keep], refc[0]) < 3e-4 and close(got[1][keep], refc[1], 3e-5, 1e-9)


def test_normvar_on_the_device_and_the_resident_chain(golden, norm, eng, monkeypatch):
	"""normvar without the host (round-4 verdict, missing item 3 / weak item 7): per-gene moments in one pass, a thread per gene solves its small OLS
	with the host's Jacobi code (norm.py:131-163 per gene), one pass writes the result.  Golden G9 and a larger case against the oracle's
	per-gene loop; the device's integer ranks equal to inv_rank's, on rank-deficient covariate sets too; the same numbers as the Gram-launch
	form with the host's pseudo-inverses; and the pipeline of examples/GSE123139/code/cmd_coex.sh:38-46 -- normvar -> coex -> binnet -- with
	the expression matrix, the P-values and the network never leaving HBM between the three, against the host-side chain."""
	import torch
	from normalisr_amd.association import inv_rank
	from normalisr_amd.binnet import binnet
	g = golden('G9_normvar')
	dt, dc, w, wt = g['dt'], g['dc'], g['w'], g['wt']
	r = norm.normvar(dt, dc, w, wt)
	assert close(r[0], g['a_dtn'], 1e-6, 1e-9) and np.array_equal(r[1], g['a_dcn'])

	def ranks_by_inv_rank(dt, dc, w, wt):
		out = []
		for x in range(dt.shape[0]):
			c = dc * (w**wt[x] if wt[x] != 0 else 1.0)
			out.append(inv_rank(c @ c.T)[1])
		return np.array(out)
	assert np.array_equal(eng._normvar_ranks.cpu().numpy(), ranks_by_inv_rank(dt, dc, w, wt))
	# rank-deficient covariates (a repeated row; one-hot batches + intercept): ranks as inv_rank's, results as the oracle's
	rng = np.random.default_rng(9)
	ng, n = 300, 2500
	dt = (rng.normal(size=(ng, n)) * rng.uniform(0.5, 2, (ng, 1)) - 9).astype(np.float32)
	w, wt = np.exp(0.25 * rng.normal(size=n)), rng.uniform(0, 1.5, ng)
	wt[::40] = 0
	batch = rng.integers(0, 3, n)
	for dc in (np.vstack([rng.normal(size=(2, n)), np.ones((1, n))]), np.vstack([np.eye(3)[batch].T, np.ones((1, n)), rng.normal(size=(1, n))]),
			   np.vstack([rng.normal(size=(1, n))] * 2 + [np.ones((1, n))] * 2 + [rng.normal(size=(4, n))])):
		got = norm.normvar(dt, dc, w, wt)
		ref = oracle.normvar(dt.astype(np.float64), dc, w, wt)
		assert close(got[0], ref[0], 1e-6, 1e-3) and np.abs(got[0] - ref[0]).max() < 1e-9 and np.array_equal(got[1], ref[1])  # (values of order 1: measured 4e-12)
		assert np.array_equal(eng._normvar_ranks.cpu().numpy(), ranks_by_inv_rank(dt.astype(np.float64), dc, w, wt))
		monkeypatch.setenv('NRM_NORMVAR', 'host')  # the Gram-launch form with the host's pseudo-inverses: the same numbers
		host = norm.normvar(dt, dc, w, wt)
		monkeypatch.delenv('NRM_NORMVAR')
		assert np.abs(got[0] - host[0]).max() < 1e-9  # (the two forms sum the moments in different orders: measured 1e-11 on values of order 1)
	with pytest.raises(AssertionError):  # covariates that are all zero: inv_rank keeps every (zero) singular value and divides by it; the reference's
		norm.normvar(dt, np.zeros((2, n)), w, wt)  # assertion on its result fires (norm.py:160,286) -- here the flag of the kernel that wrote the values
	# the resident chain: device tensors in, device tensors between the steps
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	dt[:40] += (0.8 * rng.normal(size=(1, n))).astype(np.float32)  # a co-expressed module, so that the network is not empty
	d_dt = torch.from_numpy(dt).cuda()
	dtn, dcn = norm.normvar(d_dt, dc, w, wt, device_out=True)
	assert dtn.is_cuda and dtn.dtype == torch.float64 and isinstance(dcn, np.ndarray)
	p, dot, var = norm.coex(dtn, dcn, device_out=True)
	assert p.is_cuda
	net = binnet(p, 0.05)
	assert net.is_cuda and net.dtype == torch.bool
	h_dtn, h_dcn = norm.normvar(dt, dc, w, wt)
	assert np.array_equal(dtn.cpu().numpy(), h_dtn)
	hp, hdot, hvar = norm.coex(h_dtn, h_dcn)
	assert np.array_equal(p.cpu().numpy(), hp) and np.array_equal(net.cpu().numpy(), binnet(hp, 0.05)) and net.any()
	po = oracle.coex(oracle.normvar(dt.astype(np.float64), dc, w, wt)[0], h_dcn)[0]
	assert p_close(hp, po, 1e-5)


def test_single5_with_a_mask(golden):
	"""association_tests(dx, None, dc, single=5, mask=...) (association.py:579-728,969-980, "under development" upstream: the one function of the
	reference's association module without a counterpart until round 5): the Gram matrix of [dx; dc] on the device, one regression per target in
	closed form (the reference: one pseudo-inverse per allowed pair), the per-pair algorithm where a target's set is rank deficient.  Golden G15 --
	the reference's own outputs, incl. its per-block variance of x and a repeated covariate row -- and a larger seeded case against the oracle."""
	from normalisr_amd.association import association_tests
	g = golden('G15_single5')
	for name, kw in (('a', dict(lowmem=False)), ('b', dict(return_dot=False)), ('c', dict(bsx=5, bsy=4, lowmem=False)), ('d', dict(dimreduce=2))):
		p, d, a, vx, vy = association_tests(g['dx'], None, g['dc'], single=5, mask=g['mask'], **kw)
		assert p_close(p, g[name + '_p']) and close(d, g[name + '_dot'], 1e-6, 1e-12) and close(vx, g[name + '_vx'], 1e-9, 1e-12) and close(vy, g[name + '_vy'], 1e-9, 1e-12)
		assert (p[~g['mask']] == 1).all() and (d[~g['mask']] == 0).all() and (vy[~g['mask']] == 0).all()
		if name + '_alpha' in g.files:
			assert close(a, g[name + '_alpha'], 1e-6, 1e-9)
		else:
			assert a is None
	# a sparse prior network over 300 variables, full-rank sets (closed form for every target), fp32 rows
	rng = np.random.default_rng(150)
	nx, n = 300, 3000
	f = rng.normal(size=(6, n))
	dx = (rng.normal(size=(nx, n)) + 0.6 * rng.normal(size=(nx, 6)) @ f).astype(np.float32)
	dc = np.vstack([rng.normal(size=(2, n)), np.ones((1, n))])
	mask = rng.random((nx, nx)) < 0.03
	np.fill_diagonal(mask, False)
	got = association_tests(dx, None, dc, single=5, mask=mask, lowmem=False, return_dot=False)
	ref = oracle.association_tests(dx.astype(np.float64), None, dc, single=5, mask=mask, lowmem=False, return_dot=False)
	assert got[0].dtype == np.float32 and got[3].shape == (nx, nx)
	ok = mask & (ref[0] > 1e-30)
	assert relerr(got[0][ok], ref[0][ok]) < 2e-5 and close(got[1], ref[1], 2e-5, 1e-6) and close(got[3], ref[3], 2e-6, 1e-9) and close(got[4], ref[4], 2e-6, 1e-9)
	assert close(got[2], ref[2], 2e-4, 1e-5) and (got[0][~mask] == 1).all()
	with pytest.raises(KeyError):
		association_tests(dx, None, dc, single=5)
