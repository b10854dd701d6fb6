"""single=5: association tests between rows of dx under a mask of allowed (x, y) pairs (a prior network: which X may affect which Y).

Reference: association.py:969-980 forms the Gram matrix of A = [dx; dc] and association_test_5 (:579-728, "under development" upstream) then
loops over the allowed pairs: the covariates of pair (x, y) are the OTHER rows the mask allows for target y plus the real covariates, one
pseudo-inverse of their Gram matrix per pair.  The targets are the first ny rows of A: the reference takes their products from the Gram
matrix itself (tprod[:, y0:y1], :974-977), so dy only says how many there are; dy=None (the rows of dx against each other) is the use.

Here: the one contraction that grows with the cells, A A^T, runs on the device (K1-free: raw rows on the fp64 Gram kernel, as single=4 with
dy=None); what follows is small algebra per TARGET on the host.  For a target y with allowed set S_y, every pair (x, y), x in S_y, asks for the
partial regression of y on x given S_y \\ {x} and the covariates -- the coefficient of x in ONE multiple regression of y on T = S_y + covariates
(Frisch-Waugh, as in single4.py): with M = (A A^T)[T, T], N = M^-1, B = N (A A^T)[T, y],
    gamma_xy = B_x,  varx = 1 / (n N_xx),  vary_xy = (RSS_y + B_x^2 / N_xx) / n,  R^2 = gamma^2 varx / vary,  rank = |T| - 1,
one inverse per target instead of one SVD per pair.  Valid while M passes the reference's rank test (singular values >= tol x the largest,
association.py:77); targets whose set does not -- a repeated covariate, collinear rows -- follow the reference's per-pair algorithm with
inv_rank, on the same device-computed Gram matrix.  The reference's quirk is kept: the variance of x is written to x's whole row of a block of
targets (:699), so varx[x, block] is the value of the last allowed pair of x in that block, and it depends on the tiling (bsx, bsy)."""
import logging

import numpy as np

from . import _lib
from . import engine as _engine
from .association import inv_rank


def association_tests_single5(dx, dy, dc, mask, bsx=0, bsy=0, lowmem=True, return_dot=True, dimreduce=0, **ka):
	"""Device path of association_tests(..., single=5, mask=mask); returns (p, gamma|dot, alpha|None, varx (n_x, n_y), vary (n_x, n_y))."""
	from .single4 import _pvalues_grouped
	dx, dc, mask = np.asarray(dx), np.asarray(dc), np.asarray(mask)
	nx, n = dx.shape
	nc = dc.shape[0]
	ny = nx if dy is None else np.asarray(dy).shape[0]
	if dc.shape[1] != n or (dy is not None and np.asarray(dy).shape[1] != n):
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if mask.shape != (nx, ny):
		raise AssertionError('mask must have shape (n_x, n_y) (association.py:970)')
	if ny > nx + nc:
		raise IndexError('single=5 takes its targets from the rows of [dx; dc] (association.py:974-977): n_y <= n_x + n_cov')
	if nx == 0 or ny == 0 or n == 0:
		raise ValueError('Dimensions in na==0 detected.')
	if nc == 0:
		logging.warning('No covariate dc input.')
	mask = mask != 0
	out_dtype = (dx if dy is None else np.asarray(dy)).dtype
	out_dtype = out_dtype if out_dtype in (np.float32, np.float64) else np.dtype(np.float64)
	tol = ka.get('tol', 1E-8)
	closed_ok = ka.get('mpc', 0) == 0 and ka.get('method', 'auto') in ('auto', 'scipy')
	m = nx + nc
	eng = _engine.get_engine()
	with eng.lock:
		torch = eng.torch
		mp, kp = _engine._round_up(m, _lib.ROW_TILE), _engine._round_up(n, _lib.K_TILE)
		with torch.cuda.device(eng.device):
			a_dev = eng.zeros((mp, kp), torch.float64)
			eng.copy_rows(a_dev, eng.upload(_engine.as_input(dx)).to(torch.float64) if dx.dtype != np.float64 else eng.upload(_engine.as_input(dx)))
			if nc:
				eng.copy_rows(a_dev[nx:], eng.upload(np.asarray(dc, dtype=np.float64)))
			ra = _engine.Residualized(m, n, a_dev, None, None)
			prod = eng.gram(ra, ra, True)[:m, :m].cpu().numpy()  # A A^T, tiles on / above the diagonal (association.py:936-950)
		prod = np.triu(prod) + np.triu(prod, 1).T
		# the reference's tiles of targets (its x blocks cover every x: maxx = 500 000): what varx's rows are filled by
		from .association import _auto_batchsize
		bsx_, bsy_ = _auto_batchsize(bsx, bsy, dx.dtype.itemsize, (dx if dy is None else np.asarray(dy)).dtype.itemsize, dc.dtype.itemsize, nc, n, dy is None, maxx=500000,
									 maxy=10)
		p = np.ones((nx, ny))
		gam = np.zeros((nx, ny))
		vx = np.zeros((nx, ny))
		vy = np.zeros((nx, ny))
		r2 = np.zeros((nx, ny))
		rank = np.zeros((nx, ny), dtype=np.int64)
		alpha = None if lowmem else np.zeros((nx, ny, nc))
		last = np.full((nx, ny), np.nan)  # dxx of every tested pair: the blocks' rows are filled from it below
		cov = list(range(nx, m))
		with _engine.host_blas():
			for y in range(ny):
				xs = np.nonzero(mask[:, y])[0]
				if xs.size == 0:
					continue
				t = list(xs) + cov
				mt = prod[np.ix_(t, t)]
				ev = np.linalg.eigvalsh(mt) if closed_ok else None
				# (a target that is allowed for itself -- a mask with a diagonal entry -- makes its own regression degenerate: left to the per-pair algorithm, whatever the reference does with it)
				if closed_ok and ev[-1] > 0 and ev[0] >= tol * ev[-1] * (1 + 1e-6) and len(t) > 1 and not (y < nx and mask[y, y]):
					ninv = np.linalg.inv(mt)
					ninv = 0.5 * (ninv + ninv.T)
					py = prod[t, y]
					b = ninv @ py
					rss = prod[y, y] - py @ b
					k = xs.size
					d = np.diag(ninv)[:k]
					dxx = 1.0 / (n * d)
					dyy = (rss + b[:k]**2 / d) / n
					gam[xs, y] = b[:k]
					last[xs, y] = dxx
					vy[xs, y] = dyy
					r2[xs, y] = b[:k]**2 * dxx / dyy
					rank[xs, y] = len(t) - 1
					if alpha is not None and nc:
						alpha[xs, y] = b[k:][None, :]  # the covariates' coefficients of the full regression: the same for every x of the target (:705-706 in closed form)
					continue
				for x in xs:  # the reference's per-pair algorithm (:668-708)
					t0 = [k for k in t if k != x]
					r = 0
					if t0:
						t1i, r = inv_rank(prod[np.ix_(t0, t0)], **ka)
					rank[x, y] = r
					if r == 0:
						dxx, dyy, dxy = prod[x, x] / n, prod[y, y] / n, prod[x, y] / n
					else:
						ccx = prod[x, t0] @ t1i
						dxx = (prod[x, x] - ccx @ prod[t0, x]) / n
						ccy = prod[t0, y] @ t1i
						dyy = (prod[y, y] - ccy @ prod[t0, y]) / n
						dxy = (prod[x, y] - ccy @ prod[t0, x]) / n
					if dxx == 0:
						dxx = 1
					gam[x, y] = dxy / dxx
					last[x, y] = dxx
					vy[x, y] = dyy
					r2[x, y] = dxy**2 / (dxx * dyy)
					if alpha is not None and r > 0 and nc:
						alpha[x, y] = ccy[-nc:] - gam[x, y] * ccx[-nc:]
		if not ((r2 >= 0).all() and (r2 <= 1 + 1E-8).all()):
			raise AssertionError('R^2 out of range (association.py:711)')
		dof = n - 1 - rank - np.asarray(dimreduce)
		if (dof <= 0).any():
			raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
		p = _pvalues_grouped(eng, r2, dof)  # (pairs the mask does not allow: R^2 = 0 -> P = 1, as the reference leaves them)
		# varx: inside every block of targets a row holds the variance of the row's LAST allowed pair of the block (:699)
		for x0 in range(0, nx, bsx_):
			for y0 in range(0, ny, bsy_):
				blk = last[x0:x0 + bsx_, y0:y0 + bsy_]
				has = ~np.isnan(blk)
				idx = np.where(has.any(axis=1), blk.shape[1] - 1 - np.argmax(has[:, ::-1], axis=1), -1)
				rows = np.nonzero(idx >= 0)[0]
				vx[x0 + rows, y0:y0 + bsy_] = blk[rows, idx[rows]][:, None]
	stat = gam * vx if return_dot else gam  # association.py:1045-1046
	if not (np.isfinite(p).all() and np.isfinite(stat).all() and np.isfinite(vy).all() and np.isfinite(vx).all()):
		raise AssertionError('non-finite results (association.py:1078-1079)')
	cast = lambda v: None if v is None else v.astype(out_dtype, copy=False)
	return (cast(p), cast(stat), cast(alpha), cast(vx), cast(vy))


assert __name__ != "__main__"
