"""Linear association testing on MI355X -- host-side mirror of the reference's association module.

Same function names, argument meaning, defaults, return tuples and error classes as
/root/reference/src/normalisr/association.py (v1.0.0); the arithmetic runs in HIP kernels through the
C ABI (include/normalisr_hip.h).  The reference tiles the pair space into 500x500 blocks and maps them
over a thread pool (association.py:890-909,997); here one call sees the whole matrices, residualises
every row once, and the device schedules 128x128 tiles itself, so bsx/bsy/nth are accepted for
compatibility but do not change the result (the reference's own results vary ~4e-14 with tile size).
"""
import logging
import os

import numpy as np

from . import _lib, _opts
from . import engine as _engine


def _randomized_inverse(mat, tol, mpc, qr, ka):
	"""Truncated pseudo-inverse through scikit-learn's randomized SVD with random_state=0 -- the third-party routine the
	reference calls for method='sklearn' (association.py:81-98: start from min(mpc, n) components, or n; without a cap grow
	the number of components until the smallest one kept falls under tol * largest; count those above the threshold)."""
	try:
		from sklearn.utils.extmath import randomized_svd
	except ImportError as e:  # same dependency as the reference for this option
		raise RuntimeError("inv_rank(method='sklearn') / mpc > 0 on matrices larger than mpc needs scikit-learn: {}".format(e))
	n = mat.shape[-1]
	k = min(mpc, n) if mpc > 0 else n
	opts = dict(ka, random_state=0)
	if qr >= 1:
		opts['power_iteration_normalizer'] = 'QR'
	if qr > 1:
		opts['n_iter'] = qr
	while True:
		_, s, vh = randomized_svd(mat, k, **opts)
		if k == n or s[-1] <= tol * s[0] or mpc > 0:
			break
		k += min(k, n - k)
	r = int(k - np.searchsorted(s[::-1], tol * s[0]))
	if mpc > 0:
		r = min(r, mpc)
	return np.matmul(vh[:r].T / s[:r], vh[:r]).T, r


def inv_rank(m, tol=1E-8, method='auto', logger=None, mpc=0, qr=0, **ka):
	"""Pseudo-inverse and rank of symmetric matrices by truncated SVD (reference association.py:4-134).

	Singular values below tol*largest count as zero; the integer rank is the number kept (capped by mpc > 0).  Runs on
	the host in fp64: the matrices are (n_cov, n_cov) and the rank must be bit-exact.  method 'scipy' = exact LAPACK SVD;
	'sklearn' = scikit-learn's randomized SVD (random_state=0, qr as in the reference); 'auto' picks the exact route
	unless mpc > 0 and the matrix is larger than mpc (association.py:52-63).
	"""
	if logger is None:
		logger = logging
	m = np.asarray(m)
	if m.ndim <= 1 or m.shape[-1] != m.shape[-2]:
		raise ValueError('Wrong shape for m.')
	if tol <= 0:
		raise ValueError('tol must be positive.')
	if qr < 0 or int(qr) != qr:
		raise ValueError('qr must be non-negative integer.')
	if method not in ('auto', 'scipy', 'sklearn'):
		raise ValueError('Unknown method {}'.format(method))
	n = m.shape[-1]
	if method == 'auto':
		if m.ndim > 2 and mpc > 0:
			raise NotImplementedError('No current method supports >2 dimensions with mpc>0.')
		method = 'scipy' if (n <= mpc or mpc == 0) else 'sklearn'
	if m.ndim > 2 and method == 'sklearn':
		raise NotImplementedError('Not supporting >2 dimensions for method=sklearn.')
	if m.ndim > 2 and mpc > 0:
		raise NotImplementedError('Not supporting >2 dimensions for mpc>0.')
	if not np.isfinite(m).all():
		raise ValueError('array must not contain infs or NaNs')  # scipy.linalg.svd(check_finite=True) in the reference
	flat = m.reshape((-1, n, n)).astype(np.float64, copy=False)
	inv = np.empty_like(flat)
	ranks = np.empty(flat.shape[0], dtype=int)
	warned = False
	todo = range(flat.shape[0])
	if method != 'sklearn' and flat.shape[0] > 8 and n <= 16:  # (measured: at 20 x 20 the stacked calls are no faster, at 40 x 40 slower)
		# a stack of small matrices (single=1: one per grouping): numpy's stacked SVD and matmul run the same LAPACK / BLAS routines per
		# matrix as the loop below, without a thousand trips through Python; matrices of equal rank share one stacked product
		try:
			_, s_all, vh_all = np.linalg.svd(flat)
			r_all = (s_all >= tol * s_all[:, :1]).sum(axis=1)
			for r in np.unique(r_all):
				g = np.nonzero(r_all == r)[0]
				v = vh_all[g][:, :r]
				inv[g] = np.swapaxes(np.matmul(np.swapaxes(v, 1, 2) / s_all[g][:, None, :r], v), 1, 2)
			ranks[:] = r_all
			todo = ()
		except np.linalg.LinAlgError:
			pass  # some matrix needs the gesvd fallback: one by one
	for i in todo:
		mat = flat[i]
		if method == 'sklearn':
			inv[i], ranks[i] = _randomized_inverse(mat, tol, mpc, qr, ka)
			continue
		try:
			_, s, vh = np.linalg.svd(mat)
		except np.linalg.LinAlgError:
			# the divide-and-conquer driver did not converge: LAPACK's gesvd, as the reference does (association.py:70-76,111-119)
			if not warned:
				logger.warning("Default SVD failed. Falling back to option lapack_driver='gesvd'. Expecting much slower computation.")
				warned = True
			from scipy.linalg import svd as _svd  # (scipy: a dependency of the reference for this very call)
			_, s, vh = _svd(mat, lapack_driver='gesvd')
		r = int(n - np.searchsorted(s[::-1], tol * s[0]))
		if mpc > 0:
			r = min(r, mpc)
		inv[i] = np.matmul(vh[:r].T / s[:r], vh[:r]).T
		ranks[i] = r
	inv = inv.astype(m.dtype if m.dtype in (np.float32, np.float64) else np.float64, copy=False)
	if m.ndim == 2:
		return inv[0], int(ranks[0])
	return inv.reshape(m.shape), ranks.reshape(m.shape[:-2])


def small_pinv(m, tol=1E-8):
	"""inv_rank for a stack of small symmetric matrices (count, n, n) fp64 -- one per grouping in single=1, one per gene in normvar --
	by the same rule (singular values below tol x the largest count as zero, association.py:77-80) through the library's threaded Jacobi
	iteration (csrc/nrm_small_pinv.hip): numpy's stacked SVD is one LAPACK call per matrix under the GIL, 3.2 us each.  Ranks as LAPACK's
	except for a singular value within rounding of the threshold; inverses to ~1e-14.  NRM_SMALL_SVD=lapack, matrices larger than
	12 x 12 or a stack of fewer than 64: inv_rank itself."""
	import os
	m = np.ascontiguousarray(m, dtype=np.float64)
	if m.ndim != 3 or m.shape[1] != m.shape[2]:
		raise ValueError('Wrong shape for m.')
	count, n = m.shape[0], m.shape[1]
	if _opts.debug('small_svd', 'native') == 'lapack' or n > 12 or count < 64 or n == 0:
		return inv_rank(m, tol=tol)
	if not np.isfinite(m).all():
		raise ValueError('array must not contain infs or NaNs')
	from . import _lib
	inv = np.empty_like(m)
	ranks = np.empty(count, dtype=np.int64)
	_lib.check(_lib.load().nrm_small_pinv(m.ctypes.data, count, n, float(tol), inv.ctypes.data, ranks.ctypes.data, 0))
	return inv, ranks


def _check_dimreduce(dimreduce):
	if np.ndim(dimreduce) != 0:
		d = np.unique(np.asarray(dimreduce))
		if d.size != 1:
			raise NotImplementedError('Per-gene dimreduce arrays are not supported for single=0 (the reference crashes on them too).')
		dimreduce = d[0]
	if int(dimreduce) != dimreduce:
		raise ValueError('dimreduce must be an integer.')
	return int(dimreduce)


def _check_block_args(dx, dy, dc, dci, dcr, dimreduce):
	"""Argument validation of association.py:199-216, same exception classes."""
	if dx.ndim != 2 or dy.ndim != 2 or dc.ndim != 2:
		raise ValueError('Incorrect dx/dy/dc size.')
	n = dx.shape[1]
	if dy.shape[1] != n or dc.shape[1] != n:
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	nc = dc.shape[0]
	if nc == 0:
		logging.warning('No covariate dc input.')
	elif dci is None or np.shape(dci) != (nc, nc):
		raise ValueError('Unmatching dci dimensions.')
	if dcr < 0:
		raise ValueError('Negative dcr detected.')
	if dcr > nc:
		raise ValueError('dcr higher than covariate dimension.')
	if n <= dcr + dimreduce + 1:
		raise ValueError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')


def association_test_1(vx, vy, dx, dy, dc, dci, dcr, dimreduce=0, lowmem=False, return_stats=False):
	"""One (x-block, y-block) tile on the device; same contract as association.py:137-260.

	Returns [vx, vy, pv, gamma, alpha|None, var_x, var_y]; with return_stats=True two extra entries:
	Pearson r and the t statistic of every pair (north-star quantities the reference does not return).
	"""
	dx, dy, dc = np.asarray(dx), np.asarray(dy), np.asarray(dc)
	dimreduce = _check_dimreduce(dimreduce)
	_check_block_args(dx, dy, dc, dci, dcr, dimreduce)
	out_dtype = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	nc = dc.shape[0]
	eng = _engine.get_engine()
	res = eng.association_single0(_engine.as_input(dx), _engine.as_input(dy), np.asarray(dc, dtype=np.float64),
								  np.zeros((nc, nc)) if dci is None else np.asarray(dci, dtype=np.float64), int(dcr),
								  dimreduce, return_dot=False, want_alpha=not lowmem, out_dtype=out_dtype,
								  want_rt=return_stats)
	ans = [vx, vy, res['p'], res['stat'], res['alpha'], res['varx'], res['vary']]
	if return_stats:
		ans += [res['r'], res['t']]
	return ans


def _auto_batchsize(bsx, bsy, itemsizex, itemsizey, itemsizec, nc, ns, samexy, maxx=500, maxy=500, sizemax=2**30):
	"""Tile sizes the reference would use (association.py:731-758).  Only informational here: the
	device path is tile-invariant."""
	if bsx == 0:
		bsx = min(int((sizemax - itemsizec * nc * ns) // (2 * itemsizex * ns)), maxx)
	if bsy == 0 or samexy:
		bsy = bsx if samexy else min(int((sizemax - itemsizec * nc * ns) // (2 * itemsizey * ns)), maxy)
	return bsx, bsy


def _prepare_covariates(dc):
	"""inv_rank(dc @ dc.T) on the host in fp64 iff there is a non-zero covariate (association.py:899-903).
	All-zero covariates are treated as rank 0 (the reference crashes there)."""
	dc64 = np.asarray(dc, dtype=np.float64)
	nc = dc64.shape[0]
	if nc > 0 and (dc64 != 0).any():
		dci, dcr = inv_rank(np.matmul(dc64, dc64.T))
	else:
		dci, dcr = np.zeros((nc, nc)), 0
	return dc64, dci, dcr


def _have_torch():
	try:
		import torch  # noqa: F401
		return True
	except ImportError:
		return False


_pool = None


def _result(shape, dtype):
	"""A result array of the host-entry route: from the recycled page-locked pool the torch engine uses (engine.PinnedPool: ctypes and the library only) --
	a fresh numpy array of 100 MB costs its page faults and a page-lock on EVERY call (configs[1] through the entry: 24 ms per call against 11) --, or plain
	numpy memory when the pool is exhausted or disabled (NRM_PINNED_POOL_MB)."""
	global _pool
	if _pool is None:
		_pool = _engine.PinnedPool(_lib.load())
	a = _pool.empty(shape, dtype)
	return np.empty(shape, dtype=dtype) if a is None else a


def _single0_host_entry(dx, dy, dc64, dci, dcr, dimreduce, return_dot, want_alpha, out_dtype, want_rt):
	"""single=0 through nrm_association_tests_host (include/normalisr_hip.h): the C entry a maintainer of the reference would bind --
	host buffers in, host buffers out, uploads, kernels, the integer engine's guard and its fp64 rerun inside the library.  Used
	when torch cannot be imported: the package then needs numpy and the library only."""
	import ctypes
	lib = _lib.load()
	samexy = dy is None
	nx, n = dx.shape
	ny = nx if samexy else dy.shape[0]
	nc = dc64.shape[0]
	odt = np.dtype(out_dtype)
	code = lambda a: _lib.NRM_F64 if a.dtype == np.float64 else _lib.NRM_F32
	vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
	p, stat = _result((nx, ny), odt), _result((nx, ny), odt)
	alpha = _result((nx, ny, nc), odt) if (want_alpha and not samexy) else None
	varx = None if samexy else np.empty(nx, dtype=odt)
	vary = np.empty(ny, dtype=odt)
	r = _result((nx, ny), odt) if want_rt else None
	t = _result((nx, ny), odt) if want_rt else None
	dc64 = np.ascontiguousarray(dc64)
	dci = np.ascontiguousarray(dci, dtype=np.float64)
	_lib.check(lib.nrm_association_tests_host(vp(dx), code(dx), nx, vp(dy), 0 if samexy else code(dy), 0 if samexy else ny, vp(dc64), _lib.NRM_F64, nc, n,
											  vp(dci), int(dcr), int(dimreduce), 1 if (samexy or return_dot) else 0, vp(p), vp(stat), vp(alpha), vp(varx), vp(vary),
											  vp(r), vp(t), _lib.NRM_F64 if odt == np.float64 else _lib.NRM_F32))
	return dict(p=p, stat=stat, alpha=alpha, varx=varx, vary=vary, r=r, t=t, dof=n - 1 - dcr - dimreduce)


def _use_host_entry():
	"""True when the call goes to the library's whole-problem entries (the command line's default; a process without torch) -- and then the
	GPU they run on has been chosen as every other route chooses it: `device=` / engine.use_device of this thread, else NORMALISR_DEVICE, else GPU 0.
	nrm_set_device validates the index (ValueError for a GPU that is not there), binds this thread and the entries' helper threads, and releases
	what the library cached for another device."""
	if not (_lib.host_entry_preferred() or not _have_torch()):
		return False
	dev = getattr(_engine._selected, 'device', None)
	if dev is None and os.environ.get('NORMALISR_DEVICE', '') != '':
		try:
			dev = int(os.environ['NORMALISR_DEVICE'])
		except ValueError:
			raise ValueError('NORMALISR_DEVICE must be a GPU index, not {!r}'.format(os.environ['NORMALISR_DEVICE']))
	_lib.check(_lib.load().nrm_set_device(0 if dev is None else int(dev)))
	return True


def _single14_host_entry(single, dx, dy, dc, lowmem, return_dot, ka):
	"""single=1 / single=4 through nrm_association_tests_single1_host / _single4_host (include/normalisr_hip.h): host buffers in and out, no torch --
	what `normalisr de -m single|covariate` needs in a process that has numpy and the library only.  NotImplementedError (NRM_E_UNSUPPORTED) for the
	calls the entries do not cover (per-gene dimreduce, mpc / method / qr, rank-deficient designs, negative entries, dy=None): with torch present the
	caller then takes the package's device paths."""
	import ctypes
	ka = dict(ka)
	dimreduce = ka.pop('dimreduce', 0)
	tol = ka.pop('tol', 1E-8) if single == 4 else 1E-8
	if single == 1:
		ka.pop('chunk', None)  # (the package's grouping chunk: no meaning here)
	if single == 4 and (ka.pop('method', 'auto') != 'auto' or ka.pop('mpc', 0) != 0 or ka.pop('qr', 0) != 0):
		raise NotImplementedError('single=4 host entry: inv_rank options other than tol follow the per-grouping algorithm (needs the package\'s device path)')
	if ka:
		raise TypeError("association_test_{}() got an unexpected keyword argument '{}'".format(2 if single == 1 else 4, next(iter(ka))))
	if dy is None or np.ndim(dimreduce) != 0:
		raise NotImplementedError('single={} host entry: dy=None and per-gene dimreduce need the package\'s device path'.format(single))
	if int(dimreduce) != dimreduce or dimreduce < 0:
		raise ValueError('dimreduce must be a non-negative integer.')
	nx, n = dx.shape
	ny, nc = dy.shape[0], dc.shape[0]
	if dc.shape[1] != n or dy.shape[1] != n:
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if nx == 0 or ny == 0 or n == 0:
		raise ValueError('Dimensions in na==0 detected.')
	if nc == 0:
		logging.warning('No covariate dc input.')
	odt = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	dx, dy = _engine.as_input(dx), _engine.as_input(dy)
	dc64, dci, dcr = _prepare_covariates(dc)
	dc64 = np.ascontiguousarray(dc64)
	dci = np.ascontiguousarray(dci, dtype=np.float64)
	lib = _lib.load()
	code = lambda a: _lib.NRM_F64 if a.dtype == np.float64 else _lib.NRM_F32
	vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
	p, stat, vary = (_result((nx, ny), odt) for _ in range(3))
	varx = np.empty(nx, dtype=odt)
	alpha = None if lowmem else _result((nx, ny, nc), odt)
	ocode = _lib.NRM_F64 if odt == np.float64 else _lib.NRM_F32
	if single == 1:
		_lib.check(lib.nrm_association_tests_single1_host(vp(dx), code(dx), nx, vp(dy), code(dy), ny, vp(dc64), _lib.NRM_F64, nc, n, int(dimreduce), 1 if return_dot else 0,
														  vp(p), vp(stat), vp(alpha), vp(varx), vp(vary), ocode))
	else:
		_lib.check(lib.nrm_association_tests_single4_host(vp(dx), code(dx), nx, vp(dy), code(dy), ny, vp(dc64), _lib.NRM_F64, nc, n, vp(dci), int(dcr), int(dimreduce),
														  1 if return_dot else 0, float(tol), vp(p), vp(stat), vp(alpha), vp(varx), vp(vary), ocode))
	return (p, stat, alpha, varx, vary)


def _single14_without_torch(single, dx, dy, dc, lowmem, return_dot, ka):
	"""What is left of single=1 / single=4 when the whole-problem entries answered NRM_E_UNSUPPORTED and torch cannot be imported: one dimreduce per gene
	(association.py:449,558: the entries once per distinct value, on the genes that share it), and for single=4 the reference's per-grouping algorithm on
	device-computed Gram matrices (single4.association_tests_single4_hostlib: rank-deficient designs, mpc / method / qr, dy=None)."""
	dimreduce = ka.get('dimreduce', 0)
	if dy is not None and np.ndim(dimreduce) != 0:
		dimreduce = np.asarray(dimreduce)
		if single == 1:
			dimreduce = dimreduce.reshape(-1)  # ((ny, 1) broadcasts against the (ny, nx) blocks of association_test_2 too: association.py:405)
		if dimreduce.shape != (dy.shape[0], ):
			raise ValueError('dimreduce must be an integer or have one entry per row of dy.')
		if (dimreduce != dimreduce.astype(np.int64)).any():
			raise ValueError('dimreduce must be an integer.')
		try:
			out = None
			for d in np.unique(dimreduce):
				sel = np.nonzero(dimreduce == d)[0]
				part = _single14_host_entry(single, dx, np.ascontiguousarray(dy[sel]), dc, lowmem, return_dot, dict(ka, dimreduce=int(d)))
				if out is None:
					shape = lambda v: v.shape[:1] + (dy.shape[0], ) + v.shape[2:]
					out = [None if v is None else (np.array(v) if v.ndim == 1 else np.empty(shape(v), dtype=v.dtype)) for v in part]
				for o, v in zip(out, part):
					if v is not None and v.ndim > 1:
						o[:, sel] = v
			return tuple(out)
		except NotImplementedError:
			if single != 4:
				raise
	if single != 4:
		raise NotImplementedError('single=1: this call needs the package\'s device path (torch)')
	from .single4 import association_tests_single4_hostlib
	return association_tests_single4_hostlib(dx, dy, dc, lowmem=lowmem, return_dot=return_dot, **ka)


def association_tests(dx, dy, dc, bsx=0, bsy=0, nth=1, lowmem=True, return_dot=True, single=0, bs4=500,
					  return_stats=False, device_out=False, **ka):
	"""All-pairs association tests between rows of dx and dy (or dx with itself when dy is None).

	Same contract as association.py:761-1093: returns (P-values, dot|gamma, alpha|None, varx|None, vary).
	single=0 runs on the device; single=1 (per-grouping cell subsets), single=4 (other X as covariates) and single=5 (single=4 under a mask of
	allowed pairs, mask=...) run the closed-form device paths in normalisr_amd.single1 / .single4 / .single5.  bsx, bsy, nth, bs4 are accepted and ignored (see module docstring).
	With return_stats=True a sixth element {'r':..., 't':..., 'dof':...} is appended.  With device_out=True (single=0 only)
	the two (n_x, n_y) matrices are returned as torch tensors resident in HBM (e.g. to feed binnet without crossing PCIe).
	device=<index> runs the call on that GPU instead of the current one (also engine.use_device, NORMALISR_DEVICE).
	"""
	device = ka.pop('device', None)
	if device is not None:
		with _engine.use_device(device):
			return association_tests(dx, dy, dc, bsx=bsx, bsy=bsy, nth=nth, lowmem=lowmem, return_dot=return_dot, single=single, bs4=bs4,
									 return_stats=return_stats, device_out=device_out, **ka)
	bs = ka.pop('bs', None)  # the reference's docstring promises `bs` (coex.py:38) but crashes on it
	if bs is not None and not bsx and not bsy:
		bsx = bsy = bs
	for v in (bsx, bsy, bs4):
		if v < 0 or int(v) != v:
			raise ValueError('Batch sizes must be non-negative integers.')
	if single not in {0, 1, 4, 5}:
		raise ValueError('Unknown value single={}'.format(single))
	# dx / dy may be torch CUDA tensors already in HBM (what normvar(..., device_out=True) returns; a resident pipeline normvar -> coex -> binnet)
	is_dev = lambda a: hasattr(a, 'is_cuda') and a.is_cuda
	if not is_dev(dx):
		dx = np.asarray(dx)
	dc = np.asarray(dc)
	samexy = dy is None
	if not samexy and not is_dev(dy):
		dy = np.asarray(dy)
	if dx.ndim != 2 or dc.ndim != 2 or (not samexy and dy.ndim != 2):
		raise ValueError('Incorrect dx/dy/dc size.')
	if single in (1, 4) and _use_host_entry() and not return_stats and not device_out and not is_dev(dx) and not (dy is not None and is_dev(dy)):
		# no torch in this process (or asked for): the library's whole-problem entries for `de -m single|covariate`
		try:
			return _single14_host_entry(single, dx, dy, dc, lowmem, return_dot, ka)
		except NotImplementedError:
			if not _have_torch():
				return _single14_without_torch(single, dx, dy, dc, lowmem, return_dot, ka)
	if single == 1:
		if samexy:
			raise NotImplementedError('dy=None with single=1')
		from .single1 import association_tests_single1
		return association_tests_single1(dx, dy, dc, lowmem=lowmem, return_dot=return_dot, return_stats=return_stats, **ka)
	if single == 5:
		from .single5 import association_tests_single5
		if return_stats:
			raise NotImplementedError('return_stats is only available for single=0.')
		if 'mask' not in ka:
			raise KeyError('mask')  # association.py:969: ka0.pop('mask')
		return association_tests_single5(dx, dy, dc, ka.pop('mask'), bsx=bsx, bsy=bsy, lowmem=lowmem, return_dot=return_dot, **ka)
	if single == 4:
		from .single4 import association_tests_single4
		return association_tests_single4(dx, dy, dc, lowmem=lowmem, return_dot=return_dot, return_stats=return_stats, **ka)

	dimreduce = _check_dimreduce(ka.pop('dimreduce', 0))
	if ka:
		raise TypeError("association_test_1() got an unexpected keyword argument '{}'".format(next(iter(ka))))
	ref_y = dx if samexy else dy
	n = dx.shape[1]
	if ref_y.shape[1] != n or dc.shape[1] != n:
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if dx.shape[0] == 0 or ref_y.shape[0] == 0:
		raise AssertionError('No association test to perform.')  # assert len(ans0) > 0, association.py:998
	if is_dev(ref_y):
		out_dtype = np.dtype(np.float32 if 'float32' in str(ref_y.dtype) else np.float64)
	else:
		out_dtype = ref_y.dtype if ref_y.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	nc = dc.shape[0]
	if nc == 0:
		logging.warning('No covariate dc input.')
	dc64, dci, dcr = _prepare_covariates(dc)
	if n <= dcr + dimreduce + 1:
		raise ValueError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	if samexy and not lowmem:
		raise NotImplementedError('alpha for dy=None is not meaningful in the reference (symmetrised) and is not provided.')
	if _use_host_entry():
		# no torch in this process (or asked for): the library's own whole-problem entry, numpy buffers in and out
		if device_out:
			raise RuntimeError('device_out=True returns torch tensors: torch is needed for it.')
		res = _single0_host_entry(_engine.as_input(dx), None if samexy else _engine.as_input(dy), dc64, dci, dcr, dimreduce, return_dot,
								  not lowmem, out_dtype, return_stats)
	else:
		prep = lambda a: a if is_dev(a) else _engine.as_input(a)
		res = _engine.get_engine().association_single0(prep(dx), None if samexy else prep(dy), dc64, dci, dcr,
								  dimreduce, return_dot=return_dot, want_alpha=not lowmem, out_dtype=out_dtype,
								  want_rt=return_stats, device_out=device_out and not return_stats)
	stat = res['stat']
	if device_out and samexy and not return_dot:
		raise NotImplementedError('device_out with dy=None and return_dot=False')
	if samexy and not return_dot:
		# association.py:1059-1061: covariance back to coefficient, row-wise by the row's variance
		stat = (stat.T / res['vary']).T.astype(out_dtype, copy=False)
	ans = (res['p'], stat, res['alpha'], res['varx'], res['vary'])
	if return_stats:
		ans += (dict(r=res['r'], t=res['t'], dof=res['dof']), )
	return ans


assert __name__ != "__main__"
