"""single=4: association tests that treat every other grouping as a covariate (high-MOI CRISPR screens).

Reference: association.py:926-980 builds the Gram matrices of A = [dx; dc] and association_test_4
(:421-576) then, for EVERY tested x_i, pseudo-inverts the Gram matrix of all other rows (one
(nx+nc-1)^2 SVD per grouping) to get partial variances.  For a full-rank A A^T those partial quantities are
exactly the pieces of ONE multiple regression of each gene on all rows of A (Frisch-Waugh):

    M = A A^T,  N = M^-1,  B = N (A Y^T)
    gamma_iy = B_iy                          (association.py:550)
    varx_i   = 1 / (n N_ii)                  (association.py:539-540, Schur complement)
    vary_iy  = (RSS_y + B_iy^2 / N_ii) / n   (association.py:542)
    R2_iy    = gamma^2 varx / vary           (association.py:554)
    alpha_iy = B[nx:, y]                     (association.py:551-553)
    dof      = n - 1 - (nx + nc - 1) - dimreduce   (association.py:558, rank = nx+nc-1)

so the device does three Gram contractions on the fp64 matrix cores (A A^T, Y A^T, (Y A^T) N) and one sweep.
If A A^T is rank deficient with respect to the reference's threshold (singular values < tol * largest,
association.py:77) the per-grouping ranks differ and the reference's own algorithm is followed on the host,
using the device-computed Gram matrices (slow path, same results).
"""
import logging

import numpy as np

from . import _lib
from . import engine as _engine



def _round_up(v, m):
	return (v + m - 1) // m * m


def _pvalues_grouped(eng, r2, dof):
	"""p = I_{1-R^2}(dof/2, 1/2) on the device for an R^2 matrix whose dof varies per entry (one launch per distinct dof:
	the p-value plan is per dof, association.py:558-563)."""
	torch = eng.torch
	dof = np.broadcast_to(dof, r2.shape)
	p = np.empty(r2.shape)
	for d in np.unique(dof):
		sel = dof == d
		d_r2 = eng.upload(np.ascontiguousarray(r2[sel]))
		d_p = torch.empty_like(d_r2)
		_lib.check(eng.lib.nrm_pvalues_from_r2(d_r2.data_ptr(), d_r2.numel(), float(d), d_p.data_ptr(), eng._stream()))
		p[sel] = d_p.cpu().numpy()
	return p


def _per_grouping_host(prod, prodyT, yy, nx, nc, n, dimreduce, lowmem, eng, ka):
	"""The reference's per-grouping algorithm (association.py:521-563) on the device-computed Gram matrices, for the cases
	its closed form does not cover: rank-deficient A A^T, or a pseudo-inverse truncated with mpc.  ka: inv_rank options."""
	from .association import inv_rank
	ny, m = prodyT.shape[0], nx + nc
	gam = np.zeros((nx, ny))
	vx = np.zeros(nx)
	vy = np.zeros((nx, ny))
	r2 = np.zeros((nx, ny))
	alpha = None if lowmem else np.zeros((nx, ny, nc))
	ranks = np.zeros(nx, dtype=int)
	for i in range(nx):
		t0 = [k for k in range(m) if k != i]
		r = 0
		if t0:
			t1i, r = inv_rank(prod[np.ix_(t0, t0)], **ka)
		ranks[i] = r
		if r == 0:
			dxx, dyy, dxy = prod[i, i] / n, yy / n, prodyT[:, i] / n
		else:
			ccx = prod[i, t0] @ t1i
			dxx = (prod[i, i] - ccx @ prod[t0, i]) / n
			ccy = prodyT[:, t0] @ t1i
			dyy = (yy - (ccy * prodyT[:, t0]).sum(axis=1)) / n
			dxy = (prodyT[:, i] - ccy @ prod[t0, i]) / n
		if dxx == 0:
			dxx = 1
		vx[i], vy[i], gam[i] = dxx, dyy, dxy / dxx
		if alpha is not None and r > 0 and nc > 0:
			alpha[i] = ccy[:, -nc:] - gam[i][:, None] * ccx[-nc:]
		r2[i] = dxy**2 / (dxx * dyy)
	if not ((r2 >= 0).all() and (r2 <= 1 + 1E-8).all()):
		raise AssertionError('R^2 out of range (association.py:557)')
	dof = (n - 1 - ranks)[:, None] - np.asarray(dimreduce)  # (nx, 1) - scalar or (ny,)  (association.py:558)
	if (dof <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	return _pvalues_grouped(eng, r2, dof), gam, alpha, vx, vy


def _pairwise_host(prod, nx, nc, n, dimreduce, eng, ka):
	"""dy=None, cases without a closed form: the reference's loop over pairs x < y (association.py:506-556), every other row
	of [dx; dc] a covariate (one pseudo-inverse per pair)."""
	from .association import inv_rank
	m = nx + nc
	gam = np.zeros((nx, nx))
	vy = np.zeros((nx, nx))
	r2 = np.zeros((nx, nx))
	ranks = np.zeros((nx, nx), dtype=int)
	for i in range(nx):
		for j in range(i + 1, nx):
			t0 = [k for k in range(m) if k != i and k != j]
			r = 0
			if t0:
				t1i, r = inv_rank(prod[np.ix_(t0, t0)], **ka)
			ranks[i, j] = r
			if r == 0:
				dxx, dyy, dxy = prod[i, i] / n, prod[j, j] / n, prod[i, j] / n
			else:
				ccx = prod[i, t0] @ t1i
				dxx = (prod[i, i] - ccx @ prod[t0, i]) / n
				ccy = prod[j, t0] @ t1i
				dyy = (prod[j, j] - ccy @ prod[t0, j]) / n
				dxy = (prod[i, j] - ccy @ prod[t0, i]) / n
			if dxx == 0:
				dxx = 1
			vy[i, j], gam[i, j], r2[i, j] = dyy, dxy / dxx, dxy**2 / (dxx * dyy)
	return gam, vy, r2, ranks


def _single4_samexy(dx, dc, lowmem, return_dot, dimreduce, ka, eng, out_dtype):
	"""association_tests(dx, None, dc, single=4): partial association of every pair of rows given ALL other rows and the
	covariates (association.py:489-498,506-510,1037-1068).  Full-rank A A^T (A = [dx; dc], N its inverse): the pair (i, j)
	conditioned on the rest has covariance inv([[N_ii, N_ij], [N_ij, N_jj]]) / n, so
	    gamma_ij = -N_ij / N_jj,   vary_ij = N_ii / (n det),   R^2_ij = N_ij^2 / (N_ii N_jj),   det = N_ii N_jj - N_ij^2,
	rank = nx + nc - 2 for every pair.  The reference then forms dot = gamma * vary for i < j, mirrors p, dot and vary
	(diagonal of vary = 1) and divides dot by vary again when return_dot is False."""
	if not lowmem:
		raise NotImplementedError('alpha for dy=None is not meaningful in the reference (symmetrised) and is not provided.')
	torch = eng.torch
	nx, n = dx.shape
	nc = dc.shape[0]
	m = nx + nc
	from .engine import Residualized
	mp, kp = _engine._round_up(m, _lib.ROW_TILE), _engine._round_up(n, _lib.K_TILE)
	with torch.cuda.device(eng.device):
		a_dev = eng.zeros((mp, kp), torch.float64)
		a_dev[:nx, :n] = eng.upload(_engine.as_input(dx))
		if nc:
			a_dev[nx:m, :n] = eng.upload(np.asarray(dc, dtype=np.float64))
		ra = Residualized(m, n, a_dev, None, None)
		prod = eng.gram(ra, ra, True)[:m, :m].cpu().numpy()
	prod = np.triu(prod) + np.triu(prod, 1).T
	tol, mpc = ka.get('tol', 1E-8), ka.get('mpc', 0)
	with _engine.host_blas():
		ev = np.linalg.eigvalsh(prod)
		closed = mpc == 0 and ka.get('method', 'auto') in ('auto', 'scipy') and ev[-1] > 0 and ev[0] >= tol * ev[-1] * (1 + 1e-6)
		if closed:
			ninv = _spd_inverse(prod)
			d = np.diag(ninv)[:nx]
			nij = ninv[:nx, :nx]
			det = np.outer(d, d) - nij**2
			with np.errstate(divide='ignore', invalid='ignore'):
				gam = -nij / d[None, :]
				vy = d[:, None] / (n * det)
				r2 = nij**2 / np.outer(d, d)
			ranks = np.full((nx, nx), m - 2)
		else:
			logging.info('single=4, dy=None: no closed form (rank-deficient A A^T or truncated inverse); following the per-pair algorithm on the host.')
			gam, vy, r2, ranks = _pairwise_host(prod, nx, nc, n, dimreduce, eng, ka)
	up = np.triu(np.ones((nx, nx), dtype=bool), 1)
	r2 = np.where(up, r2, 0.0)
	if not ((r2 >= 0).all() and (r2 <= 1 + 1E-8).all()):
		raise AssertionError('R^2 out of range (association.py:557)')
	dof = n - 1 - ranks - np.asarray(dimreduce)  # per-gene dimreduce broadcasts over the columns (association.py:558)
	if (np.where(up, dof, 1) <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	p = np.where(up, _pvalues_grouped(eng, r2, np.where(up, dof, 1)), 0.0)
	vy = np.where(up, vy, 0.0)
	dot = np.where(up, gam, 0.0) * vy  # association.py:1041-1042
	p, dot, vy = p + p.T, dot + dot.T, vy + vy.T  # :1049-1057
	vy[np.arange(nx), np.arange(nx)] = 1
	if not return_dot:
		dot = dot / vy  # :1063-1064
	if not (np.isfinite(p).all() and np.isfinite(dot).all() and np.isfinite(vy).all()):
		raise AssertionError('non-finite results (association.py:1078-1079)')
	cast = lambda v: v.astype(out_dtype, copy=False)
	return (cast(p), cast(dot), None, None, cast(vy))


def _spd_inverse(m):
	"""Inverse of a symmetric positive definite matrix (its conditioning was checked by the caller): Cholesky factor,
	triangular inverse, L^-T L^-1 -- an order of magnitude cheaper than the eigendecomposition for the ~1000 x 1000
	matrices of a CRISPR screen (numpy only: the GPU box need not have scipy)."""
	try:
		li = np.linalg.inv(np.linalg.cholesky(m))
		inv = li.T @ li
		return 0.5 * (inv + inv.T)
	except np.linalg.LinAlgError:
		w, v = np.linalg.eigh(m)
		return (v / w) @ v.T


def association_tests_single4(dx, dy, dc, lowmem=True, return_dot=True, return_stats=False, dimreduce=0, tol=1E-8,
							  method='auto', mpc=0, qr=0, **ka):
	"""Device path of association_tests(..., single=4); returns (p, gamma|dot, alpha|None, varx, vary) with vary of shape
	(n_x, n_y) as the reference does for single=4.  dy=None tests every pair of rows of dx given all the others
	(_single4_samexy).  dimreduce may be an int or one value per row of dy (association.py:449,558); tol / method / mpc /
	qr go to inv_rank as in the reference (:527-528) -- a truncated inverse (mpc > 0) has no closed form and follows the
	reference's per-grouping algorithm on the device-computed Gram matrices."""
	if ka:
		raise TypeError("association_test_4() got an unexpected keyword argument '{}'".format(next(iter(ka))))
	if return_stats:
		raise NotImplementedError('return_stats is only available for single=0.')
	dx, dc = np.asarray(dx), np.asarray(dc)
	nx, n = dx.shape
	nc = dc.shape[0]
	ny = nx if dy is None else np.asarray(dy).shape[0]
	if np.ndim(dimreduce) != 0:
		dimreduce = np.asarray(dimreduce)
		if dimreduce.shape != (ny, ):
			raise ValueError('dimreduce must be an integer or have one entry per row of dy.')
		if (dimreduce != dimreduce.astype(np.int64)).any():
			raise ValueError('dimreduce must be an integer.')
		dimreduce = dimreduce.astype(np.int64)
		if (dimreduce == dimreduce[0]).all():
			dimreduce = int(dimreduce[0])
	else:
		if int(dimreduce) != dimreduce:
			raise ValueError('dimreduce must be an integer.')
		dimreduce = int(dimreduce)
	if dc.shape[1] != n or (dy is not None and np.asarray(dy).shape[1] != n):
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if nx == 0 or ny == 0 or n == 0:
		raise ValueError('Dimensions in na==0 detected.')
	if nc == 0:
		logging.warning('No covariate dc input.')
	ik = dict(tol=tol, method=method, mpc=mpc, qr=qr)  # inv_rank options (association.py:527-528)
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		torch = eng.torch
		if dy is None:
			out_dtype = dx.dtype if dx.dtype in (np.float32, np.float64) else np.dtype(np.float64)
			return _single4_samexy(dx, dc, lowmem, return_dot, dimreduce, ik, eng, out_dtype)
		dy = np.asarray(dy)
		out_dtype = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
		m = nx + nc
		# A = [X; C] (association.py:935) is stacked on the device: X travels in its own dtype and is widened there
		from .engine import Residualized
		mp, kp = _engine._round_up(m, _lib.ROW_TILE), _engine._round_up(n, _lib.K_TILE)
		with torch.cuda.device(eng.device):
			a_dev = eng.zeros((mp, kp), torch.float64)
			a_dev[:nx, :n] = eng.upload(_engine.as_input(dx))
			if nc:
				a_dev[nx:m, :n] = eng.upload(np.asarray(dc, dtype=np.float64))
		ra = Residualized(m, n, a_dev, None, None)
		ry = eng.residualize(_engine.as_input(dy), None, None, 0)  # fp64 padded copy of Y and sum y^2 (association.py:968)
		prod_d = eng.gram(ra, ra, True)  # A A^T, tiles on/above the diagonal (association.py:936-950)
		prod = prod_d[:m, :m].cpu().numpy()
		prod = np.triu(prod) + np.triu(prod, 1).T
		# Y A^T (association.py:952-967, transposed).  It is the K-operand of the next contraction (K = m_pad columns), and K2
		# leaves 16-column sub-blocks that are pure padding unwritten: start from zeros so that no stale NaN/Inf bit pattern
		# of the allocator can reach 0 * NaN there
		with torch.cuda.device(eng.device):
			prodyT_d = eng.zeros((ry.rows_pad, mp), torch.float64)
		eng.gram(ry, ra, False, dot=prodyT_d)
		# the spectrum decides whether the closed form applies; the inverse it needs is taken beside it on a second host thread (LAPACK
		# releases the GIL; both take ~20 ms at 1000 groupings) and thrown away when it does not
		spec = None
		with _engine.host_blas():
			if mpc == 0 and method in ('auto', 'scipy'):
				from concurrent.futures import ThreadPoolExecutor

				def _try_inverse():
					try:
						return _spd_inverse(prod)
					except Exception:  # not positive definite: the spectrum will say so
						return None
				with ThreadPoolExecutor(1) as ex:
					fut = ex.submit(_try_inverse)
					ev = np.linalg.eigvalsh(prod)
					spec = fut.result()
			else:
				ev = np.linalg.eigvalsh(prod)
		closed = mpc == 0 and method in ('auto', 'scipy') and ev[-1] > 0 and ev[0] >= tol * ev[-1] * (1 + 1e-6)
		if not closed:
			logging.info('single=4: no closed form (rank-deficient A A^T or truncated inverse); following the per-grouping algorithm on the host.')
			with _engine.host_blas():
				p, gam, alpha, vx, vy = _per_grouping_host(prod, prodyT_d[:ny, :m].cpu().numpy(), ry.ss[:ny].cpu().numpy(), nx, nc, n,
														   dimreduce, lowmem, eng, ik)
			stat = (gam.T * vx).T if return_dot else gam
			cast = lambda v: None if v is None else v.astype(out_dtype, copy=False)
			return (cast(p), cast(stat), cast(alpha), cast(vx), cast(vy))
		if n <= m + np.max(dimreduce):
			raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
		dr_groups = [int(dimreduce)] if np.ndim(dimreduce) == 0 else [int(v) for v in np.unique(dimreduce)]
		if spec is not None:
			ninv = spec
		else:
			with _engine.host_blas():
				ninv = _spd_inverse(prod)  # N = M^-1 (symmetric)
		dxx = 1.0 / (n * np.diag(ninv)[:nx])
		n_pad = np.zeros((mp, mp))
		n_pad[:m, :m] = ninv
		with torch.cuda.device(eng.device):
			d_n = eng.upload(n_pad)
			pt = Residualized(ny, mp, prodyT_d, None, None)  # (ny_pad, m_pad): K dimension = rows of A, zero padded
			bt_d = eng.gram(pt, Residualized(m, mp, d_n, None, None), False)  # Bt = (Y A^T) N
			tdt = torch.float64 if out_dtype == np.float64 else torch.float32
			p = torch.empty((nx, ny), dtype=tdt, device=eng.device)
			stat = torch.empty((nx, ny), dtype=tdt, device=eng.device)
			vary = torch.empty((nx, ny), dtype=tdt, device=eng.device)
			work = torch.empty((ny, ), dtype=torch.float64, device=eng.device)
			flags = eng.zeros((2, ), torch.int32)
			d_dxx = eng.upload(dxx)
			code = _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32
			p_host = None
			# dof = n - 1 - (m - 1) - dimreduce (association.py:558): uniform, or one sweep per distinct per-gene dimreduce value
			# (gamma and vary do not depend on it; the P-value columns of each group are kept)
			for gi, dr in enumerate(dr_groups):
				_lib.check(eng.lib.nrm_single4_sweep(bt_d.data_ptr(), prodyT_d.data_ptr(), bt_d.stride(0), ry.ss.data_ptr(), d_dxx.data_ptr(),
													 nx, ny, m, n, float(n - m - dr), 1 if return_dot else 0, p.data_ptr(), stat.data_ptr(),
													 vary.data_ptr(), code, ny, work.data_ptr(), flags.data_ptr(), eng._stream()))
				if len(dr_groups) > 1:
					if p_host is None:
						p_host = np.empty((nx, ny), dtype=out_dtype)
					cols = np.nonzero(dimreduce == dr)[0]
					p_host[:, cols] = eng.download(p)[:, cols]
			eng.check_flags(flags)
			alpha = None
			if not lowmem:
				b_cov = bt_d[:ny, nx:m].cpu().numpy().astype(out_dtype)  # (ny, nc): identical for every grouping
				alpha = np.broadcast_to(b_cov[None, :, :], (nx, ny, nc)).copy()
			vx = dxx.copy()
			vx[vx == 0] = 1
			return (eng.download(p) if p_host is None else p_host, eng.download(stat), alpha, vx.astype(out_dtype), eng.download(vary))
