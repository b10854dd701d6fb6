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


def _inv_rank_sym(m, tol):
	"""Truncated pseudo-inverse and rank of a symmetric PSD matrix, rule of association.py:77-80."""
	from .association import inv_rank
	return inv_rank(m, tol=tol)


def _per_grouping_host(prod, prodyT, yy, nx, nc, n, dimreduce, tol, lowmem, eng, out_dtype):
	"""Rank-deficient case: the reference's per-grouping algorithm (association.py:521-563) on the Gram matrices."""
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
			t1i, r = _inv_rank_sym(prod[np.ix_(t0, t0)], tol)
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
	dof = n - 1 - ranks - dimreduce
	if (dof <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	torch = eng.torch
	p = np.empty((nx, ny))
	for d in np.unique(dof):
		rows = np.nonzero(dof == d)[0]
		d_r2 = eng.upload(np.ascontiguousarray(r2[rows]))
		d_p = torch.empty_like(d_r2)
		_lib.check(eng.lib.nrm_pvalues_from_r2(d_r2.data_ptr(), d_r2.numel(), float(d), d_p.data_ptr(), eng._stream()))
		p[rows] = d_p.cpu().numpy()
	return p, gam, alpha, vx, vy


def _spd_inverse(m):
	"""Inverse of a symmetric positive definite matrix (its conditioning was checked by the caller): Cholesky, an order
	of magnitude cheaper than the eigendecomposition for the ~1000 x 1000 matrices of a CRISPR screen."""
	from scipy.linalg import cho_factor, cho_solve, LinAlgError
	try:
		c = cho_factor(m, lower=True, check_finite=False)
		inv = cho_solve(c, np.eye(m.shape[0]), check_finite=False)
		return 0.5 * (inv + inv.T)
	except LinAlgError:
		w, v = np.linalg.eigh(m)
		return (v / w) @ v.T


def association_tests_single4(dx, dy, dc, lowmem=True, return_dot=True, return_stats=False, dimreduce=0, tol=1E-8,
							  method='auto', mpc=0, qr=0, **ka):
	"""Device path of association_tests(..., single=4) for dy is not None; returns (p, gamma|dot, alpha|None, varx, vary)
	with vary of shape (n_x, n_y) as the reference does for single=4."""
	if ka:
		raise TypeError("association_test_4() got an unexpected keyword argument '{}'".format(next(iter(ka))))
	if dy is None:
		raise NotImplementedError('single=4 with dy=None (pairwise competition among genes) is not on the device path.')
	if return_stats:
		raise NotImplementedError('return_stats is only available for single=0.')
	if mpc:
		raise NotImplementedError('mpc (principal-component truncation of the covariates) is not on the device path.')
	if np.ndim(dimreduce) != 0:
		d = np.unique(np.asarray(dimreduce))
		if d.size != 1:
			raise NotImplementedError('Per-gene dimreduce arrays are not supported on the device path.')
		dimreduce = d[0]
	dimreduce = int(dimreduce)
	dx, dy, dc = np.asarray(dx), np.asarray(dy), np.asarray(dc)
	nx, n = dx.shape
	ny, nc = dy.shape[0], dc.shape[0]
	if dy.shape[1] != n or dc.shape[1] != n:
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if nx == 0 or ny == 0 or n == 0:
		raise ValueError('Dimensions in na==0 detected.')
	if nc == 0:
		logging.warning('No covariate dc input.')
	out_dtype = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	m = nx + nc
	eng = _engine.get_engine()
	torch = eng.torch
	# A = [X; C] (association.py:935) is stacked on the device: X travels in its own dtype and is widened there
	from .engine import Residualized
	mp, kp = _engine._round_up(m, _lib.ROW_TILE), _engine._round_up(n, _lib.K_TILE)
	with torch.cuda.device(eng.device):
		a_dev = torch.zeros((mp, kp), dtype=torch.float64, device=eng.device)
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
		prodyT_d = torch.zeros((ry.rows_pad, mp), dtype=torch.float64, device=eng.device)
	eng.gram(ry, ra, False, dot=prodyT_d)
	with _engine.host_blas():
		ev = np.linalg.eigvalsh(prod)
	full_rank = ev[-1] > 0 and ev[0] >= tol * ev[-1] * (1 + 1e-6)
	if not full_rank:
		logging.info('single=4: A A^T is rank deficient; following the per-grouping algorithm on the host.')
		with _engine.host_blas():
			p, gam, alpha, vx, vy = _per_grouping_host(prod, prodyT_d[:ny, :m].cpu().numpy(), ry.ss[:ny].cpu().numpy(), nx, nc, n,
													   dimreduce, tol, lowmem, eng, out_dtype)
		stat = (gam.T * vx).T if return_dot else gam
		cast = lambda v: None if v is None else v.astype(out_dtype, copy=False)
		return (cast(p), cast(stat), cast(alpha), cast(vx), cast(vy))
	if n <= m + dimreduce:
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	dof = n - m - dimreduce
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
		flags = torch.zeros(2, dtype=torch.int32, device=eng.device)
		d_dxx = eng.upload(dxx)
		_lib.check(eng.lib.nrm_single4_sweep(bt_d.data_ptr(), prodyT_d.data_ptr(), bt_d.stride(0), ry.ss.data_ptr(), d_dxx.data_ptr(),
											 nx, ny, m, n, float(dof), 1 if return_dot else 0, p.data_ptr(), stat.data_ptr(),
											 vary.data_ptr(), _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32, ny,
											 work.data_ptr(), flags.data_ptr(), eng._stream()))
		eng.check_flags(flags)
		alpha = None
		if not lowmem:
			b_cov = bt_d[:ny, nx:m].cpu().numpy().astype(out_dtype)  # (ny, nc): identical for every grouping
			alpha = np.broadcast_to(b_cov[None, :, :], (nx, ny, nc)).copy()
		vx = dxx.copy()
		vx[vx == 0] = 1
		return (eng.download(p), eng.download(stat), alpha, vx.astype(out_dtype), eng.download(vary))
