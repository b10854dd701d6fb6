"""numpy restatement of the reference's association path -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Follows /root/reference/src/normalisr/ (v1.0.0), function by function, in fp64 numpy; the p-value
arithmetic (scipy.stats.beta.cdf at association.py:249) is the C restatement in normalisr_oracle.c.
Pinned against golden vectors generated from the reference itself (tests/golden/*.npz, made by
tests/golden/make_golden.py); see tests/test_oracle.py.

Not imported by the product path.  Allowed users: tests/, __graft_entry__.smoke(), bench.py cpu_baseline.
"""
import ctypes
import itertools
import logging
import os
import subprocess
from multiprocessing.pool import ThreadPool

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libnrm_oracle.so')
_lib = None


def ensure_built():
	"""Compile oracle/normalisr_oracle.c with gcc (oracle/Makefile) if needed and load it."""
	global _lib
	if _lib is not None:
		return _lib
	src = os.path.join(_HERE, 'normalisr_oracle.c')
	if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
		subprocess.run(['make', '-s', '-C', _HERE], check=True)
	lib = ctypes.CDLL(_SO)
	dp = ctypes.POINTER(ctypes.c_double)
	lib.nrm_oracle_pvalues.argtypes = [dp, ctypes.c_size_t, ctypes.c_double, dp]
	lib.nrm_oracle_pvalues.restype = None
	lib.nrm_oracle_beta_cdf_half.argtypes = [ctypes.c_double, ctypes.c_double]
	lib.nrm_oracle_beta_cdf_half.restype = ctypes.c_double
	lib.nrm_oracle_lngamma_ratio_half.argtypes = [ctypes.c_double]
	lib.nrm_oracle_lngamma_ratio_half.restype = ctypes.c_double
	lib.nrm_oracle_block.argtypes = [dp, dp, dp, dp] + [ctypes.c_long] * 6 + [dp] * 5
	lib.nrm_oracle_block.restype = None
	_lib = lib
	return lib


def _dp(a):
	return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def pvalues(r2, dof):
	"""beta.cdf(1-r2, dof/2, 0.5) elementwise (association.py:249)."""
	lib = ensure_built()
	r2 = np.ascontiguousarray(r2, dtype=np.float64)
	out = np.empty_like(r2)
	lib.nrm_oracle_pvalues(_dp(r2), r2.size, float(dof), _dp(out))
	return out


def beta_cdf_half(x, a):
	return ensure_built().nrm_oracle_beta_cdf_half(float(x), float(a))


def inv_rank(m, tol=1E-8, method='auto', mpc=0, qr=0):
	"""Truncated-SVD pseudo-inverse and integer rank of a symmetric PSD matrix.

	association.py:67-80 (2-D, method='scipy'): singular values below tol*largest are dropped;
	rank = number kept (at most mpc when mpc > 0, :78-79); inverse = (Vh[:r].T / s[:r]) @ Vh[:r], transposed.
	method='sklearn', or 'auto' with mpc > 0 on a matrix larger than mpc (:52-63): the reference calls scikit-learn's
	randomized_svd (third-party, absent from /root/reference; the version used for the fixtures is in tests/golden/meta.json)
	with random_state=0, starting from min(mpc, n) components (n without a cap) and -- only without a cap -- doubling them
	until the smallest kept singular value falls below the threshold (:81-98).
	"""
	m = np.asarray(m, dtype=np.float64)
	if m.ndim != 2 or m.shape[0] != m.shape[1]:
		raise ValueError('Wrong shape for m.')
	if tol <= 0:
		raise ValueError('tol must be positive.')
	n = m.shape[0]
	if method == 'auto':
		method = 'scipy' if (n <= mpc or mpc == 0) else 'sklearn'
	if method == 'scipy':
		u, s, vh = np.linalg.svd(m)
		r = int(n - np.searchsorted(s[::-1], tol * s[0]))
	else:
		from sklearn.utils.extmath import randomized_svd
		k = min(mpc, n) if mpc > 0 else n
		opt = dict(random_state=0)
		if qr >= 1:
			opt['power_iteration_normalizer'] = 'QR'
		if qr > 1:
			opt['n_iter'] = qr
		while True:
			u, s, vh = randomized_svd(m, k, **opt)
			if k == n or s[-1] <= tol * s[0] or mpc > 0:
				break
			k += min(k, n - k)
		r = int(k - np.searchsorted(s[::-1], tol * s[0]))
	if mpc > 0:
		r = min(r, mpc)
	mi = np.matmul(vh[:r].T / s[:r], vh[:r]).T
	return mi, r


def association_test_1(vx, vy, dx, dy, dc, dci, dcr, dimreduce=0, lowmem=False):
	"""One (x-block, y-block) tile: association.py:137-260."""
	if dx.ndim != 2 or dy.ndim != 2 or dc.ndim != 2:
		raise ValueError('Incorrect dx/dy/dc size.')
	n = dx.shape[1]
	if dy.shape[1] != n or dc.shape[1] != n:
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	nc = dc.shape[0]
	if nc == 0:
		logging.warning('No covariate dc input.')
	elif dci.shape != (nc, nc):
		raise ValueError('Unmatching dci dimensions.')
	if dcr < 0:
		raise ValueError('Negative dcr detected.')
	if dcr > nc:
		raise ValueError('dcr higher than covariate dimension.')
	if n <= dcr + dimreduce + 1:
		raise ValueError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	nx, ny = dx.shape[0], dy.shape[0]
	rx, ry = dx, dy
	if dcr > 0:
		bx = np.matmul(dci, np.matmul(dc, dx.T)).T  # :226
		by = np.matmul(dci, np.matmul(dc, dy.T)).T  # :227
		rx = dx - np.matmul(bx, dc)  # :228
		ry = dy - np.matmul(by, dc)  # :229
	varx = (rx**2).mean(axis=1)  # :230
	varx[varx == 0] = 1
	vary = (ry**2).mean(axis=1)
	vary[vary == 0] = 1
	gam = (np.matmul(ry, rx.T) / (n * varx)).T  # :234
	r2 = ((gam**2).T * varx).T / vary  # :235
	if lowmem:
		alpha = None
	elif dcr > 0:
		alpha = by[None, :, :] - gam[:, :, None] * bx[:, None, :]  # :238-243
	else:
		alpha = np.zeros((nx, ny, nc), dtype=dx.dtype)
	assert (r2 >= 0).all() and (r2 <= 1 + 1E-8).all()  # :248
	p = pvalues(r2, n - 1 - dcr - dimreduce).astype(r2.dtype, copy=False)  # :249
	assert np.isfinite(p).all() and np.isfinite(gam).all()
	return [vx, vy, p, gam, alpha, varx, vary]


def _auto_batchsize(bsx, bsy, isx, isy, isc, nc, ns, samexy, maxx=500, maxy=500, sizemax=2**30):
	"""association.py:731-758."""
	if bsx == 0:
		bsx = min(int((sizemax - isc * nc * ns) // (2 * isx * ns)), maxx)
	if bsy == 0 or samexy:
		bsy = bsx if samexy else min(int((sizemax - isc * nc * ns) // (2 * isy * ns)), maxy)
	return bsx, bsy


def _pool_map(nth, tasks):
	"""parallel.py:12-74 with dummy=True (thread pool); nth==0 -> cpu_count, nth==1 -> serial."""
	if nth == 0:
		nth = os.cpu_count()
	if nth == 1:
		return [f(*a, **k) for f, a, k in tasks]
	with ThreadPool(nth) as pool:
		return pool.map(lambda t: t[0](*t[1], **t[2]), tasks)


def _test4_block(vx, vy, prod, prody, prodyy, na, dimreduce=0, lowmem=False, **ka):
	"""association.py:421-576: per tested x, every other row of [dx;dc] is a covariate; pseudo-inverse of their Gram matrix
	(ka: inv_rank options tol/method/mpc/qr, :527-528); Schur-style partial products.  prody None = dx tested against itself
	(:489-498): pairs x < y only, neither of the two a covariate (:506-510,:524-525).  dimreduce: int or one value per y (:558)."""
	nx, ny, nc, n, lenx = na
	samexy = prody is None
	if samexy:
		prody = prod[:, vy:vy + ny]
		prodyy = prod[np.arange(ny) + vy, np.arange(ny) + vy]
	p = np.zeros((lenx, ny))
	vxo = np.zeros((lenx, ))
	vyo = np.zeros((lenx, ny))
	gam = np.zeros((lenx, ny))
	alpha = None if lowmem else np.zeros((lenx, ny, nc))
	rank = np.zeros((lenx, ny), dtype=int)
	if samexy:
		todo = [(i, [j]) for i in range(lenx) for j in range(ny) if i + vx < j + vy]
	else:
		todo = [(i, np.arange(ny)) for i in range(lenx)]
	for i, ys in todo:
		t0 = [k for k in range(nx + nc) if k != vx + i and not (samexy and k == vy + ys[0])]  # :523-525
		if len(t0) > 0:
			t1i, r = inv_rank(prod[np.ix_(t0, t0)], **ka)  # :527-528
		else:
			r = 0
		rank[i, ys] = r
		if r == 0:
			dxx = prod[vx + i, vx + i] / n
			dyy = prodyy[ys] / n
			dxy = prody[vx + i, ys] / n
		else:
			ccx = np.matmul(prod[[vx + i], t0], t1i)
			dxx = (prod[vx + i, vx + i] - float(np.matmul(ccx, prod[t0, [vx + i]]))) / n  # :539-540
			ccy = np.matmul(prody[t0][:, ys].T, t1i)
			dyy = (prodyy[ys] - (ccy.T * prody[t0][:, ys]).sum(axis=0)) / n  # :542
			dxy = (prody[vx + i, ys] - np.matmul(ccy, prod[t0, [vx + i]]).ravel()) / n  # :543-544
		if dxx == 0:
			dxx = 1
		vxo[i] = dxx
		vyo[i, ys] = dyy
		gam[i, ys] = dxy / dxx
		if (not lowmem) and r > 0:
			alpha[i, ys] = (ccy[:, -nc:] - gam[i, ys][:, None] * ccx[-nc:]) if nc > 0 else 0
		p[i, ys] = (dxy**2) / (dxx * dyy)
	assert (p >= 0).all() and (p <= 1 + 1E-8).all()
	dof = n - 1 - rank - dimreduce  # :558 (an array dimreduce broadcasts over the y axis)
	if (dof <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	for d in np.unique(dof):
		sel = dof == d
		p[sel] = pvalues(p[sel], d)
	return [vx, vy, p, gam, alpha, None if samexy else vxo, vyo]


def _test5_block(vx, vy, prod, prody, prodyy, na, mask, dimreduce=0, lowmem=False, **ka):
	"""association.py:579-728 ("under development" upstream): like association_test_4, but only the pairs (x, y) the mask allows are tested, and
	the covariates of a pair are the OTHER x's the mask allows for this y plus the real covariates (:671-676).  Restated with the reference's
	quirk: the variance of x is written to its whole ROW of the (lenx, ny) block (:699), so after the loop a row holds the value of the
	last allowed pair of that row in the block."""
	nx, ny, nc, n, lenx = na
	p = np.zeros((lenx, ny))
	vxo = np.zeros((lenx, ny))
	vyo = np.zeros((lenx, ny))
	gam = np.zeros((lenx, ny))
	alpha = None if lowmem else np.zeros((lenx, ny, nc))
	rank = np.zeros((lenx, ny), dtype=int)
	it = np.array(np.nonzero(mask)).T  # row by row (:664-666)
	it = it[(it[:, 0] >= vx) & (it[:, 0] < vx + lenx)]
	it[:, 0] -= vx
	for i, j in it:
		t0 = [k for k in list(np.nonzero(mask[:, j])[0]) + list(np.arange(nc) + nx) if k != vx + i]  # :669-676
		if len(t0) > 0:
			t1i, r = inv_rank(prod[np.ix_(t0, t0)], **ka)
		else:
			r = 0
		rank[i, j] = r
		if r == 0:
			dxx, dyy, dxy = prod[vx + i, vx + i] / n, prodyy[j] / n, prody[vx + i, j] / n
		else:
			ccx = np.matmul(prod[[vx + i], t0], t1i)
			dxx = (prod[vx + i, vx + i] - float(np.matmul(ccx, prod[t0, [vx + i]]))) / n
			ccy = np.matmul(prody[t0, j], t1i)
			dyy = (prodyy[j] - np.matmul(ccy, prody[t0, j])) / n
			dxy = (prody[vx + i, j] - np.matmul(ccy, prod[t0, vx + i])) / n
		if dxx == 0:
			dxx = 1
		vxo[i] = dxx  # (the whole row: :699)
		vyo[i, j] = dyy
		gam[i, j] = dxy / dxx
		if (not lowmem) and r > 0:
			alpha[i, j] = (ccy[-nc:] - gam[i, j] * ccx.ravel()[-nc:]) if nc > 0 else 0
		p[i, j] = (dxy**2) / (dxx * dyy)
	assert (p >= 0).all() and (p <= 1 + 1E-8).all()
	dof = n - 1 - rank - dimreduce
	if (dof <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	for d in np.unique(dof):
		sel = dof == d
		p[sel] = pvalues(p[sel], d)
	return [vx, vy, p, gam, alpha, vxo, vyo]


def _test2_block(vx, vy, dx, dy, dc, sselectx, dimreduce=0, lowmem=False):
	"""association.py:263-390: like association_test_1 but every x uses its own subset of cells."""
	nx, n = dx.shape
	ny, nc = dy.shape[0], dc.shape[0]
	p = np.zeros((nx, ny))
	vxo = np.zeros((nx, ))
	vyo = np.zeros((nx, ny))
	gam = np.zeros((nx, ny))
	alpha = None if lowmem else np.zeros((nx, ny, nc))
	rank = np.zeros((nx, ny), dtype=int)
	for i in range(nx):
		t1 = np.nonzero(sselectx[i])[0]  # :341
		ns = len(t1)
		if len(np.unique(dx[i, t1])) < 2:
			continue
		x1 = dx[i, t1].astype(np.float64)
		y1 = dy[:, t1].astype(np.float64)
		r = 0
		if nc > 0:
			c1 = dc[:, t1].astype(np.float64)
			ci, r = inv_rank(np.matmul(c1, c1.T))  # :350-351
		rank[i] = r
		if r > 0:
			ccx = np.matmul(ci, np.matmul(c1, x1.T)).T
			ccy = np.matmul(ci, np.matmul(c1, y1.T)).T
			x1 = x1 - np.matmul(ccx, c1)
			y1 = y1 - np.matmul(ccy, c1)
		v = (x1**2).mean()
		if v == 0:
			v = 1
		vxo[i] = v
		vyo[i] = (y1**2).mean(axis=1)
		gam[i] = np.matmul(x1, y1.T).ravel() / (ns * v)  # :367
		if (not lowmem) and r > 0:
			alpha[i] = ccy - gam[i][:, None] * ccx.ravel()
		p[i] = (gam[i]**2) * v / vyo[i]
	assert (p >= 0).all() and (p <= 1 + 1E-8).all()
	dof = (sselectx.sum(axis=1) - 1 - rank.T - dimreduce).T  # :374
	if (dof <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	for i in range(nx):  # :377 is elementwise in dof: one value per (grouping, gene) when dimreduce is a (n_y, 1) column
		for d in np.unique(dof[i]):
			cols = dof[i] == d
			p[i, cols] = pvalues(p[i, cols], d)
	return [vx, vy, p, gam, alpha, vxo, vyo]


def association_tests(dx, dy, dc, bsx=0, bsy=0, nth=1, lowmem=True, return_dot=True, single=0, **ka):
	"""association.py:761-1093 for single=0 (any dy), single=1, single=4 and single=5 (mask=...: the targets are the first ny rows of [dx; dc], as the
	reference takes them from its Gram matrix whatever dy holds, :969-980)."""
	samexy = dy is None
	if samexy:
		dy = dx
	nx, ns = dx.shape
	ny = dy.shape[0]
	nc = dc.shape[0]
	if single in (0, 1):
		bsx, bsy = _auto_batchsize(bsx, bsy, dx.dtype.itemsize, dy.dtype.itemsize, dc.dtype.itemsize, nc, ns, samexy)
	elif single == 4:
		bsx, bsy = _auto_batchsize(bsx, bsy, dx.dtype.itemsize, dy.dtype.itemsize, dc.dtype.itemsize, nc, ns, samexy, maxx=10, maxy=500000)
	elif single == 5:
		bsx, bsy = _auto_batchsize(bsx, bsy, dx.dtype.itemsize, dy.dtype.itemsize, dc.dtype.itemsize, nc, ns, samexy, maxx=500000, maxy=10)
	else:
		raise ValueError('Unknown value single={}'.format(single))
	tiles = itertools.product([(a, min(a + bsx, nx)) for a in range(0, nx, bsx)],
							  [(b, min(b + bsy, ny)) for b in range(0, ny, bsy)])
	if samexy and single != 5:
		tiles = [t for t in tiles if t[0][0] <= t[1][0]]  # :893-894
	ka0 = dict(ka, lowmem=lowmem)
	if single == 0:
		if nc > 0 and (dc != 0).any():
			dci, dcr = inv_rank(np.matmul(dc, dc.T))  # :899-900
		else:
			dci, dcr = np.zeros((nc, nc)), 0  # reference sets dci=None and crashes for nc>0 (Q11)
		tasks = [(association_test_1, (x[0], y[0], dx[x[0]:x[1]], dy[y[0]:y[1]], dc, dci, dcr), ka0) for x, y in tiles]
	elif single == 1:
		if samexy:
			raise NotImplementedError('dy=None with single=1')  # :912
		assert dx.max() == 1  # :914
		sel = dx == dx.sum(axis=0)  # :915-916: a cell counts for x_i iff no OTHER grouping is present in it
		for i in range(nx):
			assert len(np.unique(dx[i, sel[i]])) > 1  # :917-918
		tasks = [(_test2_block, (x[0], y[0], dx[x[0]:x[1]], dy[y[0]:y[1]], dc, sel[x[0]:x[1]]), ka0) for x, y in tiles]
	else:
		t1 = np.concatenate([dx, dc], axis=0).astype(np.float64)  # :935
		prod = np.matmul(t1, t1.T)
		dr = ka0.pop('dimreduce', 0)
		drs = (lambda y: dr[y[0]:y[1]]) if np.ndim(dr) else (lambda y: dr)
		if single == 5:  # :969-980
			mask = np.asarray(ka0.pop('mask'))
			assert mask.shape == (nx, ny)
			dg = np.diag(prod)
			tasks = [(_test5_block, (x[0], y[0], prod, prod[:, y[0]:y[1]], dg[y[0]:y[1]], [nx, y[1] - y[0], nc, ns, x[1] - x[0]], mask[:, y[0]:y[1]]),
					  dict(ka0, dimreduce=drs(y))) for x, y in tiles]
		elif samexy:  # :974-980: the Gram matrix alone; association_test_4 takes its y products from it
			tasks = [(_test4_block, (x[0], y[0], prod, None, None, [nx, y[1] - y[0], nc, ns, x[1] - x[0]]), dict(ka0, dimreduce=drs(y)))
					 for x, y in tiles]
		else:
			prody = np.matmul(t1, dy.T.astype(np.float64))
			prodyy = (dy.astype(np.float64)**2).sum(axis=1)
			tasks = [(_test4_block, (x[0], y[0], prod, prody[:, y[0]:y[1]], prodyy[y[0]:y[1]], [nx, y[1] - y[0], nc, ns, x[1] - x[0]]),
					  dict(ka0, dimreduce=drs(y))) for x, y in tiles]
	res = _pool_map(nth, tasks)
	assert len(res) > 0
	p = np.ones((nx, ny), dtype=dy.dtype)  # :1005
	dot = np.zeros((nx, ny), dtype=dy.dtype)
	alpha = None if lowmem else np.zeros((nx, ny, nc), dtype=dy.dtype)
	varx = np.zeros((nx, ny), dtype=dy.dtype) if single == 5 else (None if samexy else np.zeros((nx, ), dtype=dy.dtype))
	vary = np.zeros((ny, ) if single == 0 else (nx, ny), dtype=dy.dtype)
	for r in res:
		i, j = r[0], r[1]
		p[i:i + r[2].shape[0], j:j + r[2].shape[1]] = r[2]
		dot[i:i + r[3].shape[0], j:j + r[3].shape[1]] = r[3]
		if not lowmem:
			alpha[i:i + r[4].shape[0], j:j + r[4].shape[1]] = r[4]
		if single == 5:
			varx[i:i + r[5].shape[0], j:j + r[5].shape[1]] = r[5]
		elif not samexy:
			varx[i:i + r[5].shape[0]] = r[5]
		if single == 0:
			vary[j:j + r[6].shape[0]] = r[6]
		else:
			vary[i:i + r[6].shape[0], j:j + r[6].shape[1]] = r[6]
	if single == 5:
		if return_dot:
			dot = dot * varx  # :1045-1046
	elif samexy:
		dot = (dot.T * vary).T if single == 0 else dot * vary  # :1039-1042  coefficient -> covariance x~_i.x~_j/n
		p = np.triu(p, 1)
		p = p + p.T  # :1050-1051  diagonals exactly 0
		if single == 4:  # :1052-1055
			vary = np.triu(vary, 1)
			vary = vary + vary.T
			vary[np.arange(ny), np.arange(ny)] = 1
		dot = np.triu(dot, 1)
		dot = dot + dot.T
		if not return_dot:
			dot = (dot.T / vary).T if single == 0 else dot / vary  # :1059-1064
	elif return_dot:
		dot = (dot.T * varx).T  # :1048
	assert np.isfinite(p).all() and np.isfinite(dot).all() and np.isfinite(vary).all()
	return (p, dot, alpha, varx, vary)


def de(dg, dt, dc, bs=0, **ka):
	"""de.py:4-132: drop single-valued grouping rows, test, re-inflate with p=1, gamma=0, var=0."""
	gid = np.array([len(np.unique(x)) > 1 for x in dg], dtype=bool)  # de.py:93
	p, g, a, vg, vt = association_tests(dg[gid], dt, dc, bsx=bs, bsy=bs, return_dot=False, **ka)
	ng, nt, nc = dg.shape[0], dt.shape[0], dc.shape[0]
	P = np.ones((ng, nt), dtype=dt.dtype)
	P[gid] = p
	G = np.zeros((ng, nt), dtype=dt.dtype)
	G[gid] = g
	A = None
	if a is not None:
		A = np.zeros((ng, nt, nc), dtype=dt.dtype)
		A[gid] = a
	VG = np.zeros((ng, ), dtype=dt.dtype)
	VG[gid] = vg
	VT = np.zeros((ng, nt), dtype=dt.dtype)
	VT[gid] = vt  # (ny,) broadcasts per kept row for single=0 (Q5)
	return (P, G, A, VG, VT)


def coex(dt, dc, **ka):
	"""coex.py:46-48."""
	r = association_tests(dt, None, dc, **ka)
	return (r[0], r[1], r[4])


def pearson_r_t(dot, varx, vary, dof):
	"""Derived north-star quantities the reference does not return:
	r = dot/sqrt(var_i var_j) (coex.py:34), t = sign(r) sqrt(dof r^2/(1-r^2))."""
	r = dot / np.sqrt(np.outer(varx, vary))
	r2 = np.minimum(r * r, 1.0)
	with np.errstate(divide='ignore'):
		t = np.sign(r) * np.sqrt(dof * r2 / (1.0 - r2))
	return r, t


def block_plain_c(dx, dy, dc, dci, dcr, dimreduce=0):
	"""Plain-C-loop restatement of one tile (nrm_oracle_block) for tiny cases."""
	lib = ensure_built()
	f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
	dx, dy, dc, dci = f(dx), f(dy), f(dc), f(dci)
	nx, n = dx.shape
	ny, nc = dy.shape[0], dc.shape[0]
	p = np.empty((nx, ny))
	g = np.empty((nx, ny))
	vx = np.empty(nx)
	vy = np.empty(ny)
	work = np.empty((nx + ny) * n + (nx + ny) * max(nc, 1))
	lib.nrm_oracle_block(_dp(dx), _dp(dy), _dp(dc), _dp(dci), nx, ny, nc, n, int(dcr), int(dimreduce), _dp(p), _dp(g), _dp(vx), _dp(vy), _dp(work))
	return p, g, vx, vy


def bh(pv, weight=None):
	"""Benjamini-Hochberg q-values with ties and optional weights, restated step by step from binnet.py:77-131 (not from normalisr_amd/binnet.py, whose
	vectorised form this checks): shrink to the unique p-values (:113), add up each one's weights in the order of the entries (:117-119), cumulative
	weight fraction (:122-123), p / fraction with non-finite -> 1 and clipping to [0, 1] (:124-127), running minimum from the largest p-value down
	(:128-129), back to the entries (:132).  Arithmetic in pv.dtype, as the reference's arrays are."""
	pv = np.asarray(pv)
	assert pv.ndim == 1 and pv.size > 0  # :103
	assert np.isfinite(pv).all() and pv.min() >= 0 and pv.max() <= 1  # :104
	n0 = pv.size
	pv2, ids = np.unique(pv, return_inverse=True)  # :113
	n = pv2.size
	w = np.zeros(n, dtype=pv.dtype)
	if weight is None:
		# n0 additions of 1.0 (:106-107,:118-119): an exact count in either dtype (fewer than 2^24 entries), so counting is the same arithmetic
		assert n0 < (1 << 24)
		w += np.bincount(ids, minlength=n).astype(pv.dtype)
	else:
		weight = np.asarray(weight)
		assert weight.shape == pv.shape  # :109
		assert np.isfinite(weight).all() and weight.min() >= 0 and weight.max() > 0  # :110
		for xi in range(n0):  # :118-119
			w[ids[xi]] += weight[xi]
	w = np.cumsum(w)  # :122
	w /= w[-1]  # :123
	with np.errstate(divide='ignore', invalid='ignore'):
		pv2 = pv2 / w  # :124
	pv2[~np.isfinite(pv2)] = 1  # :125
	pv2 = np.min([pv2, np.repeat(1, n)], axis=0)  # :126
	pv2 = np.max([pv2, np.repeat(0, n)], axis=0)  # :127
	for xi in range(n - 2, -1, -1):  # :128-129
		pv2[xi] = min(pv2[xi], pv2[xi + 1])
	ans = pv2[ids].astype(pv.dtype, copy=False)  # :132
	assert ans.shape == pv.shape
	return ans


def binnet(net, qcut):
	"""binnet.py:134-173: per-row BH q-values over the off-diagonal entries, thresholded at qcut; diagonal False."""
	net = np.asarray(net)
	assert net.ndim == 2 and np.isfinite(net).all() and net.min() >= 0 and net.max() <= 1
	nt = net.shape[0]
	if net.shape[1] != nt or nt <= 1:
		raise ValueError('Wrong shape of net or namet.')
	if qcut <= 0 or qcut >= 1:
		raise ValueError('Q-value cutoff must be between 0 and 1.')
	out = np.zeros((nt, nt), dtype=bool)
	off = ~np.eye(nt, dtype=bool)
	for i in range(nt):
		q = bh(net[i, off[i]])
		out[i, off[i]] = q <= qcut
	if out.sum() == 0:
		raise RuntimeError('Empty binary network.')
	return out


def normvar1(dt, dc, w2=None):
	"""norm.py:131-163: remove covariates from every row; with w2 each row g uses its own covariates dc * w2[g]."""
	if w2 is not None:
		return np.concatenate([normvar1(dt[[g]], dc * w2[g]) for g in range(dt.shape[0])], axis=0)
	mi, r = inv_rank(np.matmul(dc, dc.T))
	if r <= 0:
		raise RuntimeError('Zero-rank covariates found.')
	return dt - np.matmul(dc.T, np.matmul(mi, np.matmul(dc, dt.T))).T


def normvar(dt, dc, w, wt, dextra=None, cat=1, keepvar=True, normmean=False, **ka):
	"""norm.py:166-289: variance normalisation. Gene g is multiplied by w**wt[g], covariates dc * w**wt[g] are
	removed per gene, the variance is optionally restored, covariates (and dextra) are scaled by w."""
	if any(x.ndim != 2 for x in (dt, dc)):
		raise ValueError('dt and dc should have 2 dimensions.')
	if any(x.ndim != 1 for x in (w, wt)):
		raise ValueError('w and wt should have 1 dimension.')
	nt, ns = dt.shape
	if dc.shape[0] == 0:
		raise ValueError('No covariates.')
	if dc.shape[1] != ns or w.shape[0] != ns or wt.shape[0] != nt:
		raise ValueError('Unmatched gene or cell counts.')
	if w.min() <= 0:
		raise ValueError('w must be positive.')
	if wt.min() < 0:
		raise ValueError('wt must be non-negative.')
	w2 = (np.repeat([w], nt, axis=0).T**wt).T  # :244
	w2[wt == 0] = 1
	dt = dt * w2
	if keepvar:
		dv = np.sqrt(((dt.T - dt.mean(axis=1))**2).mean(axis=0))  # :248-249
	dtn = normvar1(dt, dc, w2=w2)
	if keepvar:
		dv2 = np.sqrt((dtn**2).mean(axis=1))
		dtn = (dtn.T * ((dv / dv2)**wt)).T  # :259
	if cat == 2:
		dcn = dc * w
	elif cat in (0, 1):
		dcn = dc.copy()
		t0 = ((dc != 0) & (dc != 1)).any(axis=1)
		if cat == 1:
			t0 |= (dc == 1).all(axis=1)  # the intercept is scaled too
		dcn[t0] = dc[t0] * w
	else:
		raise ValueError('Invalid cat value.')
	if normmean:
		dtn = normvar1(dtn, dcn)
	ans = [dtn, dcn]
	if dextra is not None:
		ans.append(dextra * w)
	return ans
