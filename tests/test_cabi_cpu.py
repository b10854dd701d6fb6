"""CPU-only checks of the boundary: the C-ABI library loads, exports what include/normalisr_hip.h
declares, its host-side p-value plan is right, and the Python mirror validates arguments like the
reference does.  No kernels are launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
from conftest import ROOT, relerr
from normalisr_amd import _lib


def test_library_exports_declared_symbols():
	hdr = open(os.path.join(ROOT, 'include', 'normalisr_hip.h')).read()
	declared = set(re.findall(r'\b(nrm_[a-z0-9_]+)\s*\(', hdr))
	assert len(declared) >= 10
	lib = ctypes.CDLL(_lib.LIB_PATH)
	for name in declared:
		assert hasattr(lib, name), name
	assert declared == set(_lib.exported_symbols())
	assert _lib.load().nrm_version() >= 100


def _plan(dof):
	plan = _lib.PvaluePlan()
	_lib.check(_lib.load().nrm_pvalue_plan_init(ctypes.byref(plan), float(dof)))
	return plan


def test_pvalue_plan_matches_oracle():
	from scipy.special import erfcx  # numpy stand-in for the device formula, host-side check only
	for dof in (16., 37., 297., 9996., 99979., 499996.):
		plan = _plan(dof)
		assert plan.a == dof / 2 and plan.umax == 1.5
		u = np.concatenate([[0.], np.geomspace(1e-12, 1.5, 200)])
		u = u[plan.alpha * u < 700]
		r2 = -np.expm1(-u)
		x = 1 - r2
		w = 1 - x
		uu = -np.log1p(-w)
		z = plan.alpha * uu
		poly = np.polyval(np.array(plan.coef[:])[::-1], uu)
		p = np.exp(-z) * (erfcx(np.sqrt(z)) + np.sqrt(z) * poly)
		ref = oracle.pvalues(r2, dof)
		assert relerr(p, ref) < 2e-12, dof
	small = _plan(5.)
	assert small.umax == 0.  # continued fraction only
	lnf = small.ln_front
	from math import lgamma, log, pi
	assert abs(lnf - (lgamma(3.0) - lgamma(2.5) - 0.5 * log(pi))) < 1e-14
	with pytest.raises(ValueError):
		_plan(0.)


def test_argument_validation_before_device():
	from normalisr_amd.association import association_tests, inv_rank
	x = np.zeros((3, 10))
	c = np.ones((1, 10))
	with pytest.raises(ValueError):
		association_tests(x, None, c, single=7)
	with pytest.raises(ValueError):
		association_tests(x, np.zeros((2, 9)), c)
	with pytest.raises(ValueError):
		association_tests(x[:, :2], None, c[:, :2])  # n <= rank + 1
	with pytest.raises(NotImplementedError):
		association_tests(x, None, c, single=1)  # dy=None with single=1 (association.py:912)
	with pytest.raises(KeyError):
		association_tests(x, None, c, single=5)  # single=5 needs mask= (association.py:969: ka0.pop('mask'))
	with pytest.raises(AssertionError):
		association_tests(x, None, c, single=5, mask=np.ones((2, 2), dtype=bool))  # mask.shape == (n_x, n_y) (association.py:970)
	with pytest.raises(TypeError):
		association_tests(x, None, c, bogus=1)
	with pytest.raises(ValueError):
		inv_rank(np.zeros((2, 3)))
	with pytest.raises(ValueError):
		inv_rank(np.eye(2), tol=0)
	with pytest.raises(ValueError):
		inv_rank(np.array([[1., np.nan], [np.nan, 1.]]))


def test_inv_rank_matches_golden(golden):
	from normalisr_amd.association import inv_rank
	g = golden('G4_invrank')
	for i in range(int(g['ncase'])):
		mi, r = inv_rank(g['m{}'.format(i)])
		assert r == int(g['r{}'.format(i)]) and isinstance(r, int)
		ref = g['mi{}'.format(i)]
		assert np.abs(mi - ref).max() <= 1e-9 * np.abs(ref).max()
	m3 = np.stack([g['m0'], g['m0'] * 2])
	mi, r = inv_rank(m3)
	assert mi.shape == m3.shape and (r == int(g['r0'])).all()


def test_missing_device_fails_loudly():
	import torch
	if torch.cuda.is_available():
		pytest.skip('GPU present')
	import normalisr_amd.normalisr as norm
	with pytest.raises(RuntimeError):
		norm.coex(np.random.default_rng(0).normal(size=(4, 30)), np.ones((1, 30)))


def test_cli_matrix_io_text_and_binary(tmp_path):
	"""run.file_read_tsv / file_write_tsv: reference text format ('%.8G', gzip by suffix, 1-D -> (1,n)) and the .npy extension."""
	from normalisr_amd import run
	a = np.random.default_rng(0).normal(size=(3, 5))
	for name in ('a.tsv', 'a.tsv.gz'):
		f = str(tmp_path / name)
		run.file_write_tsv(f, a)
		b = run.file_read_tsv(f)
		assert b.shape == a.shape and np.allclose(a, b, rtol=1e-7)
	f = str(tmp_path / 'v.tsv')
	run.file_write_tsv(f, a[0])
	assert run.file_read_tsv(f).shape == (5, 1) or run.file_read_tsv(f).shape == (1, 5)
	f = str(tmp_path / 'a.npy')
	run.file_write_tsv(f, a.astype(np.float32))
	b = run.file_read_tsv(f)
	assert b.dtype == np.float32 and np.array_equal(b, a.astype(np.float32))
	f = str(tmp_path / 'v.npy')
	run.file_write_tsv(f, a[0])
	assert run.file_read_tsv(f).shape == (1, 5)
	# CLI parser: same sub-commands / flags as the reference for the association path
	from normalisr_amd.__main__ import build_parser
	p = build_parser()
	ns = vars(p.parse_args(['de', 'g', 'e', 'c', 'pv', 'lfc', '-m', 'covariate', '-n', '2', '-b', '100', '-d', '1', '--vard_out', 'x']))
	assert ns['cmd'] == 'de' and ns['method'] == 'covariate' and ns['nth'] == 2 and ns['bs'] == 100 and ns['dimr'] == 1 and ns['vard_out'] == 'x'
	ns = vars(p.parse_args(['coex', 'e', 'c', 'pv', '--dot_out', 'd']))
	assert ns['cmd'] == 'coex' and ns['nth'] == 0 and ns['dot_out'] == 'd' and ns['var_out'] is None
	ns = vars(p.parse_args(['binnet', 'pv', 'net', '0.05']))
	assert ns['cmd'] == 'binnet' and ns['qcut'] == 0.05


def test_host_logic_under_address_and_ub_sanitizers(tmp_path):
	"""The library's host-only logic -- the Gram kernels' persistent schedule (every k-unit of every tile covered exactly once,
	slabs inside the workspace, bands tiling the problem), the scratch pool of the whole-problem entry, the P-value plan -- built by
	g++ with -fsanitize=address,undefined and run on the CPU (csrc/nrm_host_logic.h is free of HIP headers for this purpose; GPU
	sanitizers are not available on the pool)."""
	import shutil
	import subprocess
	gxx = shutil.which('g++')
	if gxx is None:
		pytest.skip('no g++')
	exe = str(tmp_path / 'host_sanitize')
	src = os.path.join(ROOT, 'tests', 'host', 'host_sanitize.cpp')
	r = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-o', exe, src],
					   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
	if r.returncode != 0 and ('asan' in r.stdout or 'ubsan' in r.stdout or 'sanitize' in r.stdout):
		pytest.skip('sanitizer runtimes not installed: ' + r.stdout[-200:])
	assert r.returncode == 0, r.stdout
	r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
	assert r.returncode == 0 and 'host logic ok' in r.stdout, r.stdout[-2000:]


def _sanitize_build_and_run(tmp_path, hip_source, harness, token):
	import shutil
	import subprocess
	gxx = shutil.which('g++')
	hip_inc = '/opt/rocm/include'
	if gxx is None or not os.path.exists(os.path.join(hip_inc, 'hip', 'hip_runtime.h')):
		pytest.skip('no g++ / HIP headers')
	exe = str(tmp_path / 'sanitize_exe')
	r = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-D__HIP_PLATFORM_AMD__', '-I' + hip_inc, '-pthread', '-w',
						'-x', 'c++', os.path.join(ROOT, 'normalisr_amd', 'csrc', hip_source), os.path.join(ROOT, 'tests', 'host', harness), '-o', exe],
					   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
	if r.returncode != 0 and ('asan' in r.stdout or 'ubsan' in r.stdout or 'sanitize' in r.stdout):
		pytest.skip('sanitizer runtimes not installed: ' + r.stdout[-200:])
	assert r.returncode == 0, r.stdout[-3000:]
	r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
	assert r.returncode == 0 and token in r.stdout, r.stdout[-3000:]


def test_small_numerics_under_address_and_ub_sanitizers(tmp_path):
	"""csrc/nrm_small_pinv.hip (host code): the threaded stack of small pseudo-inverses, the one-pass minimum / maximum / NaN count behind the reference's result
	assertions, the small eigenvalue routine -- odd counts, every thread count, exact-size buffers, under ASan + UBSan (tests/host/pinv_sanitize.cpp)."""
	_sanitize_build_and_run(tmp_path, 'nrm_small_pinv.hip', 'pinv_sanitize.cpp', 'small numerics ok')


def test_text_parser_and_printer_under_address_and_ub_sanitizers(tmp_path):
	"""The command line's text parser / printer (csrc/nrm_tsv.hip: host code) reads what users hand it: built by g++ with -fsanitize=address,undefined beside a
	harness (tests/host/tsv_sanitize.cpp) that round-trips random matrices of every magnitude through '%.8G' / '%i' text in 1 .. 7 threads and feeds it hostile text
	(random bytes from the alphabet of numbers, valid text truncated at every length) from exact-size heap buffers."""
	import shutil
	import subprocess
	gxx = shutil.which('g++')
	hip_inc = '/opt/rocm/include'
	if gxx is None or not os.path.exists(os.path.join(hip_inc, 'hip', 'hip_runtime.h')):
		pytest.skip('no g++ / HIP headers')
	exe = str(tmp_path / 'tsv_sanitize')
	r = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-D__HIP_PLATFORM_AMD__', '-I' + hip_inc, '-pthread', '-w',
						'-x', 'c++', os.path.join(ROOT, 'normalisr_amd', 'csrc', 'nrm_tsv.hip'), os.path.join(ROOT, 'tests', 'host', 'tsv_sanitize.cpp'), '-o', exe],
					   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
	if r.returncode != 0 and ('asan' in r.stdout or 'ubsan' in r.stdout or 'sanitize' in r.stdout):
		pytest.skip('sanitizer runtimes not installed: ' + r.stdout[-200:])
	assert r.returncode == 0, r.stdout[-3000:]
	r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
	assert r.returncode == 0 and 'text io ok' in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_host_mirror_rows(dtype):
	"""nrm_host_mirror_rows (host code of the library, no GPU): h[0:a, a:b] = h[a:b, 0:a]^T, any thread count, ragged tiles, nothing
	else touched -- what the numpy-out coex path does to the rows that arrived over PCIe (association.py:1049-1057 mirrors on the host
	too)."""
	lib = _lib.load()
	rng = np.random.default_rng(5)
	n = 301
	for threads in (1, 3, 0):
		for a, b in ((0, 40), (40, 40), (33, 97), (128, 301), (300, 301)):
			h = rng.normal(size=(n, n + 3)).astype(dtype)[:, :n]  # (a pitch larger than the row)
			want = h.copy()
			want[0:a, a:b] = want[a:b, 0:a].T
			_lib.check(lib.nrm_host_mirror_rows(h.ctypes.data, h.strides[0], h.itemsize, a, b, threads))
			assert np.array_equal(h, want), (threads, a, b)
	with pytest.raises(ValueError):
		_lib.check(lib.nrm_host_mirror_rows(h.ctypes.data, 8, h.itemsize, 3, 9, 1))


def test_pvalue_plans_of_a_screen_are_cheap():
	"""A single=1 screen asks for one P-value plan per grouping (nrm_pvalue_plan_init_many): 1000 of them took 25 ms with powl() in the
	inner loops; the tables of half-integer products are built once now.  The plans still satisfy G12-level accuracy (checked by the
	parity tests); here: the time, and that a plan does not depend on what was asked before."""
	import time
	from normalisr_amd import _lib
	lib = _lib.load()
	dof = np.ascontiguousarray(np.arange(30, 30 + 4000, dtype=np.float64))
	out = np.zeros((dof.size, 24))
	lib.nrm_pvalue_plan_init_many(dof.ctypes.data, 1, out.ctypes.data, 24)
	dt = []
	for _ in range(5):  # (the best of five in this thread's CPU time: a busy box must not fail a parity suite)
		t0 = time.thread_time()
		assert lib.nrm_pvalue_plan_init_many(dof.ctypes.data, dof.size, out.ctypes.data, 24) == 0
		dt.append(time.thread_time() - t0)
	assert min(dt) < 0.05, dt  # 12 us per plan at most (25 us each with powl() in the loops)
	again = np.zeros((1, 24))
	lib.nrm_pvalue_plan_init_many(dof[1234:].ctypes.data, 1, again.ctypes.data, 24)
	assert np.array_equal(again[0], out[1234])


@pytest.mark.parametrize('nc', [1, 2, 5, 8, 12])
def test_small_pinv_matches_inv_rank(nc, monkeypatch):
	"""The library's threaded Jacobi pseudo-inverse of a stack of small symmetric matrices (csrc/nrm_small_pinv.hip; single=1: one matrix per
	grouping, normvar: one per gene) against inv_rank (LAPACK SVD, association.py:4-134): the same integer ranks -- rank-deficient
	matrices (a covariate twice), tiny and huge scales included -- and the same pseudo-inverses to 1e-12."""
	from normalisr_amd.association import inv_rank, small_pinv
	rng = np.random.default_rng(nc)
	a = rng.normal(size=(300, nc, 40))
	m = a @ a.swapaxes(1, 2)
	if nc >= 2:
		for g in (7, 8, 9):  # the last covariate a copy of the first: rank nc - 1
			m[g, :, nc - 1], m[g, nc - 1, :] = m[g, :, 0], m[g, 0, :]
			m[g, nc - 1, nc - 1] = m[g, 0, 0]
		m[11] *= 1e-30
		m[12] *= 1e30
	inv, rk = small_pinv(m)
	ref, rr = inv_rank(m)
	assert np.array_equal(rk, rr) and (nc < 2 or (rk[7:10] == nc - 1).all())
	err = np.abs(inv - ref).max(axis=(1, 2)) / np.abs(ref).max(axis=(1, 2))
	assert err.max() < 1e-12
	assert np.array_equal(inv, inv.swapaxes(1, 2))
	monkeypatch.setenv('NRM_SMALL_SVD', 'lapack')
	inv2, rk2 = small_pinv(m)
	assert np.array_equal(inv2, ref) and np.array_equal(rk2, rr)


def test_result_assertions_from_minima_and_maxima():
	"""de._finite_within(a, lo, hi) == np.isfinite(a).all() and (a >= lo).all() and (a <= hi).all() (the reference's assertions on its
	results, de.py:124-131), taken from the minimum and the maximum: NaN and infinities anywhere, bounds on either side, empty arrays."""
	from normalisr_amd.de import _finite_within
	rng = np.random.default_rng(0)
	base = rng.random((50, 70)).astype(np.float32)
	cases = [base, base.astype(np.float64), np.zeros((0, 5)), np.ones((3, 3))]
	for bad in (np.nan, np.inf, -np.inf, -1e-9, 1 + 1e-6):
		a = base.astype(np.float64).copy()
		a[17, 33] = bad
		cases.append(a)
	for a in cases:
		for lo, hi in ((None, None), (0, None), (0, 1), (None, 1)):
			want = bool(np.isfinite(a).all() and (lo is None or (a >= lo).all()) and (hi is None or (a <= hi).all()))
			assert _finite_within(a, lo, hi) == want, (a.shape, lo, hi)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_host_minmax_in_one_threaded_pass(dtype):
	"""nrm_host_minmax (minimum, maximum, NaN count of a host array, dealt to host threads) against numpy on arrays large enough for several
	threads, with NaNs and infinities planted at thread boundaries and at the ends; and through de._finite_within, which uses it from 2^18
	elements on."""
	from normalisr_amd import _lib
	from normalisr_amd.de import _finite_within
	lib = _lib.load()
	rng = np.random.default_rng(3)
	a = rng.standard_normal(3 * (1 << 20) + 17).astype(dtype)
	out = np.empty(3)
	code = _lib.NRM_F64 if dtype == np.float64 else _lib.NRM_F32
	for threads in (0, 1, 3, 16):
		assert lib.nrm_host_minmax(a.ctypes.data, code, a.size, threads, out.ctypes.data) == 0
		assert out[0] == a.min() and out[1] == a.max() and out[2] == 0
	assert _finite_within(a) and not _finite_within(a, 0) and _finite_within(a, float(a.min()), float(a.max()))
	for where in (0, a.size - 1, a.size // 3, (1 << 20), (1 << 20) - 1):
		for bad, nans, fin in ((np.nan, 1, False), (np.inf, 0, False), (-np.inf, 0, False)):
			b = a.copy()
			b[where] = bad
			lib.nrm_host_minmax(b.ctypes.data, code, b.size, 0, out.ctypes.data)
			assert out[2] == nans and _finite_within(b) == fin
			if bad == np.inf:
				assert out[1] == np.inf
	lib.nrm_host_minmax(a.ctypes.data, code, 0, 0, out.ctypes.data)
	assert out[0] == np.inf and out[1] == -np.inf and out[2] == 0


def test_host_entry_routing_and_result_arrays_without_a_gpu(monkeypatch):
	"""The pieces of the torch-free route that are host logic: NRM_HOST_ENTRY / the command line's preference (and =0 overriding it), and result arrays that
	fall back to plain numpy memory when no page-locked block can be had (no GPU here: hipHostMalloc fails)."""
	from normalisr_amd import _lib, association
	monkeypatch.delenv('NRM_HOST_ENTRY', raising=False)
	assert not _lib.host_entry_preferred()
	prev = _lib.prefer_host_entry(True)
	try:
		assert _lib.host_entry_preferred()
		monkeypatch.setenv('NRM_HOST_ENTRY', '0')
		assert not _lib.host_entry_preferred()
	finally:
		_lib.prefer_host_entry(prev)
	monkeypatch.setenv('NRM_HOST_ENTRY', '1')
	assert _lib.host_entry_preferred()
	a = association._result((700, 800), np.float32)
	assert a.shape == (700, 800) and a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
	a[:] = 1.0
	assert float(a.sum()) == 700 * 800



def test_device_pvalue_plans_match_the_host_plans():
	"""single=1 builds one P-value plan per grouping ON THE DEVICE (k_s1_group_info; round 6: no host inside a resident step) with csrc/nrm_pvalue_plan.h,
	the long-double plan of nrm_pvalue_plan_init restated in double arithmetic.  The same code compiled for the host against the long-double plans: a, alpha,
	ln_front, umax identical; the polynomial sum_j coef_j u^j (a correction of ~1e-7 u beside erfcx in the P-value) to 1e-15 of its terms, over dof 1 ... 2e6."""
	from normalisr_amd import _lib
	lib = _lib.load()
	dof = np.ascontiguousarray(np.concatenate([np.arange(1, 60, 0.5), np.geomspace(60, 2e6, 3000)]))
	a, b = np.zeros((dof.size, 24)), np.zeros((dof.size, 24))
	assert lib.nrm_pvalue_plan_init_many(dof.ctypes.data, dof.size, a.ctypes.data, 24) == 0
	assert lib.nrm_pvalue_plan_fill_many(dof.ctypes.data, dof.size, b.ctypes.data, 24) == 0
	assert np.array_equal(a[:, [0, 1, 3]], b[:, [0, 1, 3]]) and np.abs(a[:, 2] - b[:, 2]).max() <= 1e-15 * np.abs(a[:, 2]).max()
	assert (a[dof < 16, 3] == 0).all() and (a[dof >= 16, 3] == 1.5).all()
	pw = np.array([0.01, 0.2877, 1.5])[None, :] ** np.arange(20)[:, None]
	pa, pb, pabs = a[:, 4:] @ pw, b[:, 4:] @ pw, np.abs(a[:, 4:]) @ pw
	fast = dof >= 16
	assert (np.abs(pa - pb)[fast] <= 1e-15 * pabs[fast]).all()
	bad = np.array([0.0])
	assert lib.nrm_pvalue_plan_fill_many(bad.ctypes.data, 1, b.ctypes.data, 24) == _lib.NRM_E_ARG
