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
import os

import numpy as np

from . import _lib, _opts
from . import engine as _engine



def _round_up(v, m):
	return (v + m - 1) // m * m


def _pvalues_grouped(eng, r2, dof):
	"""p = I_{1-R^2}(dof/2, 1/2) on the device for an R^2 matrix whose dof varies per entry (one launch per distinct dof:
	the p-value plan is per dof, association.py:558-563)."""
	dof = np.broadcast_to(dof, r2.shape)
	p = np.empty(r2.shape)
	if eng is None:  # no torch in this process: the same device function through the library's host entry
		for d in np.unique(dof):
			sel = dof == d
			v = np.ascontiguousarray(r2[sel], dtype=np.float64)
			out = np.empty_like(v)
			_lib.check(_lib.load().nrm_pvalues_host(v.ctypes.data, v.size, float(d), out.ctypes.data))
			p[sel] = out
		return p
	torch = eng.torch
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


def _gram_host(a, b=None, want_ss=False):
	"""A B^T (B = A when b is None) over the cells on the fp64 Gram kernel through nrm_gram_host: numpy in, numpy out, no torch (association.py:936-968's
	numpy.matmul products).  want_ss: also the sums of squares of the rows of the second operand."""
	a = np.ascontiguousarray(_engine.as_input(a))
	b = None if b is None else np.ascontiguousarray(_engine.as_input(b))
	code = lambda v: _lib.NRM_F64 if v.dtype == np.float64 else _lib.NRM_F32
	rb = a.shape[0] if b is None else b.shape[0]
	out = np.empty((a.shape[0], rb))
	ss = np.empty(rb) if want_ss else None
	_lib.check(_lib.load().nrm_gram_host(a.ctypes.data, code(a), a.shape[0], None if b is None else b.ctypes.data, 0 if b is None else code(b), rb, a.shape[1],
										 out.ctypes.data, None, None if ss is None else ss.ctypes.data))
	return (out, ss) if want_ss else out


def _stack_design(dx, dc):
	"""A = [dx; dc] in fp64 (association.py:935)."""
	return np.concatenate([np.asarray(dx, dtype=np.float64), np.asarray(dc, dtype=np.float64)], axis=0)


def _single4_samexy(dx, dc, lowmem, return_dot, dimreduce, ka, eng, out_dtype):
	"""association_tests(dx, None, dc, single=4): partial association of every pair of rows given ALL other rows and the
	covariates (association.py:489-498,506-510,1037-1068).  Full-rank A A^T (A = [dx; dc], N its inverse): the pair (i, j)
	conditioned on the rest has covariance inv([[N_ii, N_ij], [N_ij, N_jj]]) / n, so
	    gamma_ij = -N_ij / N_jj,   vary_ij = N_ii / (n det),   R^2_ij = N_ij^2 / (N_ii N_jj),   det = N_ii N_jj - N_ij^2,
	rank = nx + nc - 2 for every pair.  The reference then forms dot = gamma * vary for i < j, mirrors p, dot and vary
	(diagonal of vary = 1) and divides dot by vary again when return_dot is False."""
	if not lowmem:
		raise NotImplementedError('alpha for dy=None is not meaningful in the reference (symmetrised) and is not provided.')
	nx, n = dx.shape
	nc = dc.shape[0]
	m = nx + nc
	if eng is None:
		prod = _gram_host(_stack_design(dx, dc))
	else:
		torch = eng.torch
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


def _is_dev(a):
	return hasattr(a, 'is_cuda') and a.is_cuda


class _Marks:
	"""NRM_DEBUG=s4_trace=1 (s1_trace=1 from single1.py): wall-clock of the phases of a call, each closed by a device synchronisation (profiling aid)."""

	def __init__(self, eng, key='s4_trace', label='single=4'):
		import time
		self.on = _opts.debug(key, '') == '1'
		self.label = label
		self.eng, self.clock, self.rows = eng, time.perf_counter, []
		self.t = self.clock()

	def __call__(self, name):
		if self.on:
			self.eng.torch.cuda.synchronize(self.eng.device)
			t = self.clock()
			self.rows.append((name, 1e3 * (t - self.t)))
			self.t = t

	def report(self):
		if self.on:
			logging.warning(self.label + ' phases (ms): ' + ', '.join('%s %.2f' % r for r in self.rows))


def _spd_inverse_device(eng, m_d, nx, ss):
	"""Inverse of the symmetric positive definite nx x nx matrix in the upper tiles of m_d (a symmetric K2 product, padded to row
	tiles) WITHOUT leaving the device: Newton-Schulz iteration X <- X (2 I - M X) from X = I / ||M||_1 on the fp64 matrix cores (two
	1024^3 products per step; the iterates are polynomials in M, so every product is a K2 call A B^T).  Quadratic convergence once
	||I - M X|| < 1: log2(cond) + ~6 steps -- 7 for the nearly orthogonal residual rows of a gRNA screen, against 25 - 90 ms for a
	LAPACK inverse of a 1000 x 1000 matrix on the host.  The steps between the products (symmetrising, the start, ||I - M X||_F, a
	transpose, 2 X - X T, what the host needs of the result) are kernels of the library (csrc/nrm_single4.hip: nrm_spd_*).
	ss: |x~_i|^2 of the design rows (for kappa).  Returns (the padded device inverse, zero in the padding; ||M||_1; small = the inverse's
	diagonal, sum_j |N_ij| |x~_j|, sum_j |N_ij| as a (3, nx) numpy array) or None when the iteration has not converged in 60 steps (the
	caller falls back to the host)."""
	from .engine import Residualized
	torch = eng.torch
	nxp = m_d.shape[0]
	lib, st = eng.lib, eng._stream()
	with torch.cuda.device(eng.device):
		mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=eng.device)
		mp, t, tt, xt, x = mk(nxp, nxp), mk(nxp, nxp), mk(nxp, nxp), mk(nxp, nxp), mk(nxp, nxp)
		scal, work, res = mk(2), mk(max(2 * nxp, (nxp // 32)**2)), mk(1)
		_lib.check(lib.nrm_spd_prepare(m_d.data_ptr(), m_d.stride(0), nx, nxp, mp.data_ptr(), scal.data_ptr(), work.data_ptr(), st))
		scale = float(scal.cpu().numpy()[0])  # ||M||_1 >= lambda_max
		if not np.isfinite(scale) or scale <= 0:
			return None
		rm = Residualized(nxp, nxp, mp, None, None)
		rx_, rtt = Residualized(nxp, nxp, x, None, None), Residualized(nxp, nxp, tt, None, None)
		# Two starts.  X = diag(1 / M_ii) first: the residual rows of a gRNA screen are nearly orthogonal, I - M X then has a spectral radius of
		# ~0.3 and five steps suffice; it need not converge for every design (it does iff 2 diag(M) - M is positive definite), so the residual
		# is looked at after two steps and, unless it is falling, the iteration starts over from X = I / ||M||_1, which always converges
		# (log2(cond) + ~6 steps: 8 for the same matrices).  NRM_S4_START=norm: the second start only.
		done = False
		for start in (('diagonal', 'norm') if _opts.debug('s4_start', 'diagonal') != 'norm' else ('norm', )):
			_lib.check(lib.nrm_spd_start(mp.data_ptr(), nxp, 1 if start == 'diagonal' else 0, scal.data_ptr(), x.data_ptr(), st))
			look_from = 2 if start == 'diagonal' else 4
			diverged = False
			for it in range(60):
				eng.gram(rm, rx_, False, dot=t)  # T = M X   (X symmetric)
				# X M X = X T.  K2 forms A B^T, and T^T = X M equals T only while X commutes with M -- true from the norm start (every iterate a
				# polynomial in M), not from the diagonal one: the transpose is taken explicitly (with ||I - M X||_F of the X going into this step)
				_lib.check(lib.nrm_spd_transpose_residual(t.data_ptr(), nxp, tt.data_ptr(), res.data_ptr(), work.data_ptr(), st))
				r = float(res.cpu().numpy()[0]) if it >= look_from else np.inf
				if np.isnan(r) or (start == 'diagonal' and it == look_from and not r < 1.0):
					diverged = True
					break
				eng.gram(rx_, rtt, False, dot=xt)
				_lib.check(lib.nrm_spd_update(x.data_ptr(), xt.data_ptr(), nxp * nxp, st))
				if r < 1e-7:  # the step just taken squares it: below the rounding floor
					done = True
					eng._s4_inverse = (start, it + 1)  # (Single4Plan: the same steps again without looking at the residual in between)
					break
			if done:
				break
			if start == 'norm' and diverged:
				return None
		if not done:
			return None
		small = mk(3, nx)
		_lib.check(lib.nrm_spd_finish(x.data_ptr(), nx, nxp, ss.data_ptr(), t.data_ptr(), small.data_ptr(), st))
	return t, scale, small.cpu().numpy()


def _spd_inverse_fixed(eng, m_d, nx, ss, start, iters):
	"""_spd_inverse_device without the host: the start and the number of steps the eager call needed on this design, no look at the residual in
	between (the residual of the iterate that went into the LAST step is left in `res` for the caller to check afterwards: < 1e-7 means the step taken
	squared it below the rounding floor, exactly the eager criterion).  Returns (N~ padded, scal = [||M~||_1, ...] on the device, small (3, nx), res)."""
	from .engine import Residualized
	torch = eng.torch
	nxp = m_d.shape[0]
	lib, st = eng.lib, eng._stream()
	mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=eng.device)
	mp, t, tt, xt, x = mk(nxp, nxp), mk(nxp, nxp), mk(nxp, nxp), mk(nxp, nxp), mk(nxp, nxp)
	scal, work, res, small = mk(2), mk(max(2 * nxp, (nxp // 32)**2)), mk(1), mk(3, nx)
	_lib.check(lib.nrm_spd_prepare(m_d.data_ptr(), m_d.stride(0), nx, nxp, mp.data_ptr(), scal.data_ptr(), work.data_ptr(), st))
	rm = Residualized(nxp, nxp, mp, None, None)
	rx_, rtt = Residualized(nxp, nxp, x, None, None), Residualized(nxp, nxp, tt, None, None)
	_lib.check(lib.nrm_spd_start(mp.data_ptr(), nxp, 1 if start == 'diagonal' else 0, scal.data_ptr(), x.data_ptr(), st))
	for _ in range(iters):
		eng.gram(rm, rx_, False, dot=t)
		_lib.check(lib.nrm_spd_transpose_residual(t.data_ptr(), nxp, tt.data_ptr(), res.data_ptr(), work.data_ptr(), st))
		eng.gram(rx_, rtt, False, dot=xt)
		_lib.check(lib.nrm_spd_update(x.data_ptr(), xt.data_ptr(), nxp * nxp, st))
	_lib.check(lib.nrm_spd_finish(x.data_ptr(), nx, nxp, ss.data_ptr(), t.data_ptr(), small.data_ptr(), st))
	return t, scal, small, res


class Single4Plan:
	"""A resident single=4 screen (`normalisr de -m covariate` on inputs that stay in HBM) with nothing of a step on the host (round 6).

	The FIRST step is the public call, association_tests_single4(device_out=True): it decides everything that depends on the design and the
	covariates alone -- whether the closed form applies (the reference's rank test on A A^T through the norm certificate, association.py:77), whether
	the design is sparse enough for the entry lists, which start the Newton-Schulz inverse of M~ needs and how many steps.  When that call took the
	sparse-design closed form with a scalar dimreduce and nothing was handed back (no row near the span of the covariates), every later step on the
	same, unwritten tensors is that call's device work and nothing else --

	    k_design_stats, k_de_sparse on the design rows (M~), nrm_spd_* + 2 K2' products per Newton-Schulz step (the count the first call needed),
	    k_s4_design_scalars, k_de_sparse on the genes (Y~ X~^T), K2' (B = G N~), k_s4_rss + k_s4_sweep

	-- captured as ONE HIP graph: no read-back, no upload, no allocation, no launch gap.  What the eager call looks at between its kernels is
	counted on the device instead and checked in results(): rows too close to the span of the covariates, a diagonal of N~ that is not positive, a
	last Newton-Schulz residual above the eager criterion, non-finite results, R^2 out of range -- any of them and the step is redone by the public
	call (which has the fallbacks).  Anything else (dense design, guard-carrying integer engine, rank-deficient design, per-gene dimreduce) keeps
	calling the public function every step."""

	def __init__(self, dx, dy, dc, dimreduce=0, lowmem=True, return_dot=True, eng=None):
		eng = self.eng = eng or _engine.get_engine()
		self.dx, self.dy = dx, dy
		self.dc = np.ascontiguousarray(np.asarray(dc, dtype=np.float64))
		self.dimreduce, self.lowmem, self.return_dot = dimreduce, lowmem, return_dot
		self.out = None
		self.lean = None  # decided by the first step
		self.fallbacks = 0
		from .distributed import StepGraph
		self._graph = StepGraph(eng.torch)

	def _public(self):
		return association_tests_single4(self.dx, self.dy, self.dc, lowmem=self.lowmem, return_dot=self.return_dot, dimreduce=self.dimreduce, device_out=True)

	def _first(self):
		eng = self.eng
		eng._s4_inverse = eng._s4_path = None
		out = self._public()
		path, inv = getattr(eng, '_s4_path', None), getattr(eng, '_s4_inverse', None)
		ok = (path == 'sparse closed form' and inv is not None and _is_dev(self.dx) and _is_dev(self.dy) and np.ndim(self.dimreduce) == 0 and self.lowmem
			  and _opts.debug('s4_plan', 'lean') != 'public')
		if ok:
			from .association import inv_rank
			from . import de_sparse
			torch = eng.torch
			nc = self.dc.shape[0]
			self.start, self.iters = inv
			with torch.cuda.device(eng.device):
				dci, self.dcr = inv_rank(self.dc @ self.dc.T) if nc else (np.zeros((0, 0)), 0)
				self.d_c, self.d_dci = eng.covariates(self.dc, dci) if nc else (None, None)
				self.lists = de_sparse.lists_for(eng, self.dx)
				self.flags = eng.zeros((8, ), torch.int32)
				self._side = torch.cuda.Stream(device=eng.device)
			self.versions = (self.dx._version, self.dy._version)
		self.lean = bool(ok)
		return out

	def _launch(self):
		from .engine import Residualized
		from . import de_sparse
		eng, torch = self.eng, self.eng.torch
		d_x, d_y, lists, d_c, d_dci, dcr = self.dx, self.dy, self.lists, self.d_c, self.d_dci, self.dcr
		nx, n = d_x.shape
		ny, nc = d_y.shape[0], self.dc.shape[0]
		m = nx + nc
		flags = self.flags  # [0] non-finite, [1] R^2 out of range (the sweep), [2] rows near the span of the covariates (k_design_stats, k_de_sparse), [5] diagonal of N~
		rx = de_sparse.design_stats(eng, lists, d_c, d_dci, dcr, nx, nc, flags)
		nxp = _engine._round_up(nx, _lib.ROW_TILE)
		# The design side (M~ from the design's own entries, its Newton-Schulz inverse: two dozen small launches, the 1024^3 products on 64 of the 256 CUs)
		# and the gene side (the one pass over the expression matrix) need nothing of each other until B = G N~: a fork and a join of the captured graph --
		# the design side runs on a second stream beside the gather kernel instead of in front of it.
		main = torch.cuda.current_stream(eng.device)
		forked, joined = torch.cuda.Event(), torch.cuda.Event()
		forked.record(main)
		with torch.cuda.stream(self._side):
			self._side.wait_event(forked)
			mt_d, _, _, _ = de_sparse.products(eng, lists, d_x, d_c, d_dci, dcr, rx.coef, nx, nx, n, nc, False, False, flags)
			d_n, scal, small, res = _spd_inverse_fixed(eng, mt_d, nx, rx.ss, self.start, self.iters)
			dxx = torch.empty(nx, dtype=torch.float64, device=eng.device)
			varx = torch.empty(nx, dtype=torch.float64, device=eng.device)
			_lib.check(eng.lib.nrm_single4_design_scalars(small.data_ptr(), nx, n, dxx.data_ptr(), varx.data_ptr(), flags.data_ptr(), eng._stream()))
			joined.record(self._side)
		g_d, ssy, _, _ = de_sparse.products(eng, lists, d_y, d_c, d_dci, dcr, rx.coef, nx, ny, n, nc, False, True, flags)
		main.wait_event(joined)  # (what the second stream made is used on this one from here on; every such tensor outlives the step's launches)
		bt_d = eng.gram(Residualized(ny, nxp, g_d, None, None), Residualized(nx, nxp, d_n, None, None), False)  # B^T = (Y~ X~^T) N~
		tdt = d_y.dtype if d_y.dtype in (torch.float32, torch.float64) else torch.float64
		p, stat, vary = (torch.empty((nx, ny), dtype=tdt, device=eng.device) for _ in range(3))
		work = torch.empty((ny, ), dtype=torch.float64, device=eng.device)
		code = _lib.NRM_F64 if tdt == torch.float64 else _lib.NRM_F32
		with _engine._Span(eng, 'sweep'):
			_lib.check(eng.lib.nrm_single4_sweep(bt_d.data_ptr(), g_d.data_ptr(), bt_d.stride(0), ssy.data_ptr(), dxx.data_ptr(), nx, ny, nx, n, float(n - m - int(self.dimreduce)),
												 1 if self.return_dot else 0, p.data_ptr(), stat.data_ptr(), vary.data_ptr(), code, ny, work.data_ptr(), flags.data_ptr(), eng._stream()))
		return p, stat, varx, vary, res

	def step(self, timed=False):
		eng = self.eng
		with eng.lock, eng.torch.cuda.device(eng.device):
			if self.lean is None:
				self.out = self._first()
				return
			if not self.lean or (self.dx._version, self.dy._version) != self.versions:
				self.out = None
				self.out = self._public()
				if self.lean:  # (inputs written to since the first step: decided anew)
					self.lean = None
				return
			if eng.trace is not None:
				p, stat, varx, vary, self.res = self._launch()
			else:
				p, stat, varx, vary, self.res = self._graph.run(self._launch)
			self.out = ((p, stat, None, varx, vary), None)
			eng.last_guard = dict(hits=0, worst=0.0, fallback=False)

	def check(self):
		"""True when the lean steps since the last look stand; otherwise the last step has been redone by the public call (self.out replaced)."""
		if not self.lean:
			return True
		eng = self.eng
		with eng.lock, eng.torch.cuda.device(eng.device):
			f = self.flags.cpu().numpy()
			r = float(self.res.cpu()[0]) if getattr(self, 'res', None) is not None else 0.0
			self.flags.zero_()
			if not (f[0] or f[1] or f[2] or f[5]) and r < 1e-7:
				return True
			logging.info('single=4 plan: a lean step did not stand (flags %s, last Newton-Schulz residual %.3g); redone by the public call', f.tolist(), r)
			self.fallbacks += 1
			self.lean = False
			self.out = None
			self.out = self._public()
			return False

	def results(self, device_out=False):
		"""(p, gamma|dot, alpha|None, varx (n_x,), vary (n_x, n_y)) of the last step."""
		self.check()
		out = self.out
		if isinstance(out, tuple) and len(out) == 2 and isinstance(out[0], tuple):
			out = out[0]
		p, stat, alpha, varx, vary = out
		od = np.dtype(np.float32 if str(p.dtype) in ('torch.float32', 'float32') else np.float64)
		if device_out:
			return out
		eng = self.eng
		dl = lambda t: eng.download(t) if _is_dev(t) else t
		vx = varx.cpu().numpy().astype(od) if _is_dev(varx) else varx
		return (dl(p), dl(stat), alpha, vx, dl(vary))


def _check_arguments(dx, dy, dc, dimreduce):
	"""The argument checks of association_tests(single=4) (association.py:449-470,761-930), shared by the device path and the torch-free one."""
	dx, dc = dx if _is_dev(dx) else np.asarray(dx), np.asarray(dc)
	if dy is not None and not _is_dev(dy):
		dy = np.asarray(dy)
	nx, n = dx.shape
	nc = dc.shape[0]
	ny = nx if dy is None else dy.shape[0]
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
	if dc.shape[1] != n or (dy is not None and dy.shape[1] != n):
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if nx == 0 or ny == 0 or n == 0:
		raise ValueError('Dimensions in na==0 detected.')
	if nc == 0:
		logging.warning('No covariate dc input.')
	return dx, dy, dc, dimreduce


def association_tests_single4_hostlib(dx, dy, dc, lowmem=True, return_dot=True, dimreduce=0, tol=1E-8, method='auto', mpc=0, qr=0):
	"""single=4 in a process WITHOUT torch, for the calls nrm_association_tests_single4_host leaves (NRM_E_UNSUPPORTED): a rank-deficient A A^T, a truncated
	or differently computed pseudo-inverse (mpc / method / qr), dy=None.  The reference's own per-grouping algorithm (association.py:421-576; per pair for
	dy=None) on Gram matrices the fp64 Gram kernel computes (nrm_gram_host), the small pseudo-inverses in numpy, the P-values by the device function
	(nrm_pvalues_host).  Same results as association_tests_single4 takes on its slow path; numpy arrays in and out."""
	dx, dy, dc, dimreduce = _check_arguments(dx, dy, dc, dimreduce)
	ik = dict(tol=tol, method=method, mpc=mpc, qr=qr)
	nx, n = dx.shape
	nc = dc.shape[0]
	if dy is None:
		out_dtype = dx.dtype if dx.dtype in (np.float32, np.float64) else np.dtype(np.float64)
		return _single4_samexy(dx, dc, lowmem, return_dot, dimreduce, ik, None, out_dtype)
	out_dtype = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	logging.info('single=4: no closed form (rank-deficient A A^T or truncated inverse); following the per-grouping algorithm on the host.')
	a = _stack_design(dx, dc)
	prod = _gram_host(a)
	prod = np.triu(prod) + np.triu(prod, 1).T
	prody, yy = _gram_host(a, dy, want_ss=True)  # (m, ny): A Y^T and sum y^2 (association.py:952-968)
	with _engine.host_blas():
		p, gam, alpha, vx, vy = _per_grouping_host(prod, np.ascontiguousarray(prody.T), yy, nx, nc, n, dimreduce, lowmem, None, ik)
	stat = (gam.T * vx).T if return_dot else gam
	cast = lambda v: None if v is None else v.astype(out_dtype, copy=False)
	return (cast(p), cast(stat), cast(alpha), cast(vx), cast(vy))


def association_tests_single4(dx, dy, dc, lowmem=True, return_dot=True, return_stats=False, dimreduce=0, tol=1E-8,
							  method='auto', mpc=0, qr=0, device_out=False, **ka):
	"""Device path of association_tests(..., single=4); returns (p, gamma|dot, alpha|None, varx, vary) with vary of shape
	(n_x, n_y) as the reference does for single=4.  dy=None tests every pair of rows of dx given all the others
	(_single4_samexy).  dimreduce may be an int or one value per row of dy (association.py:449,558); tol / method / mpc /
	qr go to inv_rank as in the reference (:527-528) -- a truncated inverse (mpc > 0) has no closed form and follows the
	reference's per-grouping algorithm on the device-computed Gram matrices.
	dx / dy may be torch CUDA tensors already in HBM (a resident screen: bench.py, distributed.de); device_out=True leaves p, the
	statistic and vary there too."""
	if ka:
		raise TypeError("association_test_4() got an unexpected keyword argument '{}'".format(next(iter(ka))))
	if return_stats:
		raise NotImplementedError('return_stats is only available for single=0.')
	dx, dy, dc, dimreduce = _check_arguments(dx, dy, dc, dimreduce)
	nx, n = dx.shape
	nc = dc.shape[0]
	ny = nx if dy is None else dy.shape[0]
	ik = dict(tol=tol, method=method, mpc=mpc, qr=qr)  # inv_rank options (association.py:527-528)
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		torch = eng.torch
		if dy is None:
			if _is_dev(dx):
				dx = dx.cpu().numpy()
			out_dtype = dx.dtype if dx.dtype in (np.float32, np.float64) else np.dtype(np.float64)
			return _single4_samexy(dx, dc, lowmem, return_dot, dimreduce, ik, eng, out_dtype)
		if _is_dev(dy):
			out_dtype = np.dtype(np.float32 if dy.dtype == torch.float32 else np.float64)
		else:
			out_dtype = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
		m = nx + nc
		mark = _Marks(eng)
		from .engine import Residualized
		mp, kp = _engine._round_up(m, _lib.ROW_TILE), _engine._round_up(n, _lib.K_TILE)
		eng._s4_path = None
		with torch.cuda.device(eng.device):
			d_x = dx if _is_dev(dx) else eng.upload(_engine.as_input(dx))
		may_close = mpc == 0 and method in ('auto', 'scipy')
		# The closed form applies when A A^T (A = [X; C], association.py:935) passes the reference's own rank threshold -- every singular
		# value >= tol x the largest (association.py:77).  It runs first; the norms of what it computes anyway (M~, its inverse, the
		# covariate block) usually settle the question (_surely_full_rank).  Only when they do not is A A^T formed and its spectrum taken
		# (21 ms at 1000 groupings), and the closed form's results thrown away if that says the design is rank deficient.
		res, err, closed, dcr = None, None, False, -1
		if may_close:
			from .association import inv_rank
			dc64 = np.asarray(dc, dtype=np.float64)
			mcc = dc64 @ dc64.T
			dci, dcr = inv_rank(mcc, tol=tol) if nc and (dc64 != 0).any() else (np.zeros((nc, nc)), 0)
			if dcr == nc:  # (a rank-deficient C C^T is a principal block of A A^T: no closed form then)
				try:
					res, cert = _closed_form(eng, d_x, dy, dc64, dci, dcr, dimreduce, lowmem, return_dot, out_dtype, device_out=device_out, mark=mark)
					closed = _surely_full_rank(cert, mcc, dci, tol)
					mark('rank certificate')
				except (AssertionError, RuntimeError, np.linalg.LinAlgError) as e:  # raised for good only if the closed form applies
					err = e
		prod = None
		if not closed:
			with torch.cuda.device(eng.device):
				a_dev = eng.zeros((mp, kp), torch.float64)
				a_dev[:nx, :n] = d_x
				if nc:
					a_dev[nx:m, :n] = eng.upload(np.asarray(dc, dtype=np.float64))
			ra = Residualized(m, n, a_dev, None, None)
			prod_d = eng.gram(ra, ra, True)  # A A^T, tiles on/above the diagonal (association.py:936-950)
			prod = prod_d[:m, :m].cpu().numpy()
			prod = np.triu(prod) + np.triu(prod, 1).T
			mark('A A^T')
			if may_close and dcr == nc and (res is not None or err is not None):  # (a rank-deficient C C^T is a principal block of A A^T)
				with _engine.host_blas():
					ev = np.linalg.eigvalsh(prod)
				closed = ev[-1] > 0 and ev[0] >= tol * ev[-1] * (1 + 1e-6)
		mark.report()
		if closed:
			if err is not None:
				raise err
			return res
		del res
		eng._s4_path = None  # (no closed form: nothing for a Single4Plan to replay)
		if _is_dev(dy):
			dy = dy.cpu().numpy()
		logging.info('single=4: no closed form (rank-deficient A A^T or truncated inverse); following the per-grouping algorithm on the host.')
		ry = eng.residualize(_engine.as_input(dy), None, None, 0)  # fp64 padded copy of Y and sum y^2 (association.py:968)
		# Y A^T (association.py:952-967, transposed).  K2 leaves 16-column sub-blocks that are pure padding unwritten: start from zeros
		with torch.cuda.device(eng.device):
			prodyT_d = eng.zeros((ry.rows_pad, mp), torch.float64)
		eng.gram(ry, ra, False, dot=prodyT_d)
		with _engine.host_blas():
			p, gam, alpha, vx, vy = _per_grouping_host(prod, prodyT_d[:ny, :m].cpu().numpy(), ry.ss[:ny].cpu().numpy(), nx, nc, n,
													   dimreduce, lowmem, eng, ik)
		stat = (gam.T * vx).T if return_dot else gam
		cast = lambda v: None if v is None else v.astype(out_dtype, copy=False)
		return (cast(p), cast(stat), cast(alpha), cast(vx), cast(vy))


def _surely_full_rank(cert, mcc, mcc_inv, tol):
	"""True when A A^T certainly passes the reference's rank test -- every singular value >= tol x the largest (association.py:77) --
	judged from norms the closed form has at hand instead of the spectrum.  With M = [[Mxx, Mxc], [Mcx, Mcc]], M~ = Mxx - Mxc Mcc^-1 Mcx
	its Schur complement, N~ = M~^-1 and W = Mcc^-1 Mcx = b^T (the design rows' OLS coefficients):
	    lambda_max(M) <= ||Mxx|| + ||Mcc|| <= ||M~||_1 + ||a||_F^2 ||Mcc^-1|| + ||Mcc||,       a = Mxc = b Mcc,
	    1 / lambda_min(M) = ||M^-1|| <= ||N~||_1 (1 + ||W||_F)^2 + ||Mcc^-1||     (M^-1 = [[N~, -N~ W^T], [-W N~, Mcc^-1 + W N~ W^T]]).
	Asks for twice the margin; False means "form A A^T and take its eigenvalues"."""
	norm_mt, norm_ninv, bx = cert
	nc = mcc.shape[0]
	if nc:
		ev = np.linalg.eigvalsh(mcc)
		if not ev[0] > 0:
			return False
		a = np.stack([(bx * mcc[:, c][None, :]).sum(axis=1) for c in range(nc)], axis=1)  # b Mcc, element-wise (no BLAS on small operands)
		lam_max = norm_mt + float((a * a).sum()) / ev[0] + ev[-1]
		inv_norm = norm_ninv * (1.0 + float(np.sqrt((bx * bx).sum())))**2 + 1.0 / ev[0]
	else:
		lam_max, inv_norm = norm_mt, norm_ninv
	return bool(np.isfinite(lam_max) and np.isfinite(inv_norm) and lam_max > 0 and inv_norm > 0 and 1.0 / (inv_norm * lam_max) >= 2.0 * tol)


def _closed_form(eng, d_x, dy, dc64, dci, dcr, dimreduce, lowmem, return_dot, out_dtype, force_f64=False, device_out=False, mark=lambda name: None):
	"""The closed form for a full-rank design, on rows residualised against the covariates (Frisch-Waugh): with X~, Y~ the residuals,
	M~ = X~ X~^T is the Schur complement of C C^T in A A^T, so N~ = M~^-1 is the design block of (A A^T)^-1 and
	    B = (Y~ X~^T) N~,   varx_i = 1 / (n N~_ii),   RSS_y = |y~|^2 - sum_j (Y~ X~^T)_yj B_yj,   alpha_y = b_y - B_y b_x
	are the quantities of the module header.  That puts the one large contraction, Y~ X~^T (1.5e15 flop-equivalent at BASELINE
	configs[3]), on K1's digit planes and the integer Gram engine exactly as in single=0 (with its row records, exact mean correction
	and a guard: nrm_single4_sweep_guarded; hits -> this function again on the fp64 Gram kernel); the small M~ is taken on the fp64
	matrix cores, its inverse on the host while the device works on the genes."""
	from .engine import Residualized, GuardHit
	torch = eng.torch
	nx, n = d_x.shape
	ny, nc = dy.shape[0], dc64.shape[0]
	m = nx + nc
	if n <= m + np.max(dimreduce):
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	ns = 0 if force_f64 else eng.gram_slices(n)
	with torch.cuda.device(eng.device):
		d_c, d_dci = eng.covariates(dc64, dci) if nc else (None, None)
		d_y = dy if _is_dev(dy) else eng.upload(_engine.as_input(dy))
		if not (eng.k1_quantises(d_x, d_c) and eng.k1_quantises(d_y, d_c)):
			ns = 0  # (rows K1 cannot quantise itself -- not 16-byte aligned: fp64 all the way)
		# A design matrix with few entries (gRNA incidence): Y~ X~^T = Y X^T - (Y C^T) b_x^T needs of every expression row only its values
		# at those entries -- the raw rows read once (csrc/nrm_de_sparse.hip), fp64 sums of a few hundred terms: no digit planes, no guard
		lists = None
		if not force_f64:
			from . import de_sparse
			if de_sparse.candidate(eng, d_x, d_y, dc64, False):
				lists = de_sparse.lists_for(eng, d_x)
				lists = lists if lists.ok else None
		if lists is not None:
			ns = 0
		if lists is not None and _opts.debug('s4_sparse_m', '1') != '0':
			# M~ = X~ X~^T is the same product with the design rows in the place of the expression rows: x~_i . x~_j = x_i . x_j - (x_j C^T) . b_i
			# -- the design's own statistics from its entries and one more run of the sparse-design kernels over the 200 MB design matrix,
			# instead of K1's fp64 residuals of it (400 MB) and a 1e11-flop product of them
			from . import de_sparse
			sp_flags = eng.new_flags()  # [2]: rows -- design rows here, expression rows below -- too close to the span of the covariates for these differences
			rx = de_sparse.design_stats(eng, lists, d_c, d_dci, dcr, nx, nc, sp_flags)
			nxp = _engine._round_up(nx, _lib.ROW_TILE)
			mark('design rows')
			mt_d, _, _, _ = de_sparse.products(eng, lists, d_x, d_c, d_dci, dcr, rx.coef, nx, nx, n, nc, False, False, sp_flags)  # (rows i, columns j; its upper triangle is used)
		else:
			rx = eng.residualize(d_x, d_c, d_dci, dcr, want_coef=bool(nc), nslices=ns, keep_fp64=True)
			nxp = rx.rows_pad
			mark('K1 design')
			mt_d = eng.gram(rx, rx, True)  # M~ = X~ X~^T (fp64 kernel, tiles on / above the diagonal)
		# N~ = M~^-1 on the device (Newton-Schulz on the fp64 matrix cores); the host's LAPACK only if that does not converge
		inv = _spd_inverse_device(eng, mt_d, nx, rx.ss) if _opts.debug('s4_inverse', 'device') != 'host' else None
		if inv is None:
			mt = mt_d[:nx, :nx].cpu().numpy()
			mt = np.triu(mt) + np.triu(mt, 1).T
			with _engine.host_blas():
				ninv = _spd_inverse(mt)
			n_pad = np.zeros((nxp, nxp))
			n_pad[:nx, :nx] = ninv
			d_n, norm_mt = eng.upload(n_pad), float(np.abs(mt).sum(axis=0).max())
			na = np.abs(ninv)
			small = np.stack([np.diag(ninv), (na * np.sqrt(rx.ss[:nx].cpu().numpy())[None, :]).sum(axis=1), na.sum(axis=0)])
		else:
			# what the host needs of N~ -- its diagonal, kappa (see S4Guard in csrc/nrm_single4.hip) and ||N~||_1 -- was reduced on the device:
			# three vectors travel instead of the 8 MB matrix
			d_n, norm_mt, small = inv
		d, norm_ninv = small[0].copy(), float(small[2].max())
		mark('M~ and its inverse')
		if not (np.isfinite(small).all() and (d > 0).all()):
			raise np.linalg.LinAlgError('design rows are linearly dependent given the covariates')
		kappa = small[1] / np.sqrt(d)
		dxx = 1.0 / (n * d)
		# the genes: K1 and the large contraction
		if lists is not None:
			from . import de_sparse
			if _opts.debug('s4_sparse_m', '1') == '0':
				sp_flags = eng.new_flags()
			g_d, ssy, coefy, _ = de_sparse.products(eng, lists, d_y, d_c, d_dci, dcr, rx.coef, nx, ny, n, nc, not lowmem, True, sp_flags)
			if int(sp_flags[2]) > 0:  # rows all but inside the span of the covariates: K1's two sweeps and the fp64 Gram kernel for this call
				eng._s4_path = None
				logging.info('single=4: %d rows (design or expression) too close to the span of the covariates for the sparse-design products; fp64 Gram kernel', int(sp_flags[2]))
				return _closed_form(eng, d_x, dy, dc64, dci, dcr, dimreduce, lowmem, return_dot, out_dtype, force_f64=True, device_out=device_out)
			ry = Residualized(ny, n, None, ssy, coefy, shape=(g_d.shape[0], _engine._round_up(n, _lib.K_TILE)))
			mark('Y~ X~^T (sparse design: expression rows read once)')
			eng._s4_path = 'sparse closed form' if _opts.debug('s4_sparse_m', '1') != '0' and inv is not None else None  # (Single4Plan)
		else:
			ry = eng.residualize(d_y, d_c, d_dci, dcr, want_coef=not lowmem, nslices=ns, keep_fp64=not ns)
			mark('K1 genes')
			g_d = eng.zeros((ry.rows_pad, nxp), torch.float64)  # (K2 leaves pure-padding sub-blocks unwritten)
			with _engine._Span(eng, 'gram_yx'):
				eng._gram(ry, rx, False, g_d, None, ns)
			if ns:
				_lib.check(eng.lib.nrm_gram_i8_fix_dot(g_d.data_ptr(), g_d.stride(0), ry.fix.data_ptr(), rx.fix.data_ptr(), ny, nx, ns, n, eng._stream()))
			mark('Y~ X~^T')
		bt_d = eng.gram(Residualized(ny, nxp, g_d, None, None), Residualized(nx, nxp, d_n, None, None), False)  # B^T = (Y~ X~^T) N~, K = design rows
		mark('B')
		tdt = torch.float64 if out_dtype == np.float64 else torch.float32
		p = torch.empty((nx, ny), dtype=tdt, device=eng.device)
		stat = torch.empty((nx, ny), dtype=tdt, device=eng.device)
		vary = torch.empty((nx, ny), dtype=tdt, device=eng.device)
		work = torch.empty((ny, ), dtype=torch.float64, device=eng.device)
		flags = eng.new_flags()
		d_dxx = eng.upload(dxx)
		code = _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32
		mark('result buffers')
		guard = ()
		if ns:
			fx = rx.fix[:nx].cpu().numpy()
			d_kappa = eng.upload(kappa)
			gene_hits = eng.zeros((ny, ), torch.int32)
			guard = (ry.fix.data_ptr(), d_kappa.data_ptr(), float(fx[:, 5].max()), float(fx[:, 6].max()), int(ns), float(eng.guard_tol), gene_hits.data_ptr())
			mark('kappa')
		dr_groups = [int(dimreduce)] if np.ndim(dimreduce) == 0 else [int(v) for v in np.unique(dimreduce)]
		p_host = None
		# dof = n - 1 - (m - 1) - dimreduce (association.py:558): uniform, or one sweep per distinct per-gene dimreduce value
		# (gamma and vary do not depend on it; the P-value columns of each group are kept)
		for dr in dr_groups:
			args = (bt_d.data_ptr(), g_d.data_ptr(), bt_d.stride(0), ry.ss.data_ptr(), d_dxx.data_ptr(), nx, ny, nx, n, float(n - m - dr),
					1 if return_dot else 0, p.data_ptr(), stat.data_ptr(), vary.data_ptr(), code, ny, work.data_ptr(), flags.data_ptr())
			with _engine._Span(eng, 'sweep'):
				if guard and eng.guard_tol > 0:
					_lib.check(eng.lib.nrm_single4_sweep_guarded(*args, *guard, eng._stream()))
				else:
					_lib.check(eng.lib.nrm_single4_sweep(*args, eng._stream()))
			if len(dr_groups) > 1:
				if p_host is None:
					p_host = np.empty((nx, ny), dtype=out_dtype)
				cols = np.nonzero(dimreduce == dr)[0]
				p_host[:, cols] = eng.download(p)[:, cols]
		mark('sweep')
		try:
			eng.check_flags(flags)
		except GuardHit as g:  # the integer engine could not certify every P-value
			idx = torch.nonzero(gene_hits).flatten() if len(dr_groups) == 1 else None
			if idx is None or idx.numel() == 0 or idx.numel() > max(64, ny // 8):
				# many genes (or per-gene dimreduce groups): the whole call once more on the fp64 Gram kernel
				logging.info('single=4: %s; redone on the fp64 Gram kernel', g)
				out = _closed_form(eng, d_x, dy, dc64, dci, dcr, dimreduce, lowmem, return_dot, out_dtype, force_f64=True, device_out=device_out)
				eng.last_guard = dict(hits=g.hits, worst=g.worst, fallback=True)
				return out
			# A screen always has a few strongly associated genes, and those are the pairs a bound on the P-value's RELATIVE change is
			# hardest on: only the genes with such a pair get their products again, from fp64 residuals on the fp64 Gram kernel (the
			# design side -- X~ in fp64, M~, N~ -- is the same), and their columns of the results are replaced.
			k = int(idx.numel())
			logging.info('single=4: %s; the %d genes concerned redone on the fp64 Gram kernel', g, k)
			ry_s = eng.residualize(d_y.index_select(0, idx), d_c, d_dci, dcr, want_coef=not lowmem, nslices=0, keep_fp64=True)
			g_s = eng.zeros((ry_s.rows_pad, nxp), torch.float64)
			eng.gram(ry_s, rx, False, dot=g_s)
			bt_s = eng.gram(Residualized(k, nxp, g_s, None, None), Residualized(nx, nxp, d_n, None, None), False)
			p_s, stat_s, vary_s = (torch.empty((nx, k), dtype=tdt, device=eng.device) for _ in range(3))
			work_s = torch.empty((k, ), dtype=torch.float64, device=eng.device)
			flags_s = eng.zeros((2, ), torch.int32)
			_lib.check(eng.lib.nrm_single4_sweep(bt_s.data_ptr(), g_s.data_ptr(), bt_s.stride(0), ry_s.ss.data_ptr(), d_dxx.data_ptr(), nx, k, nx, n,
												 float(n - m - dr_groups[0]), 1 if return_dot else 0, p_s.data_ptr(), stat_s.data_ptr(), vary_s.data_ptr(), code, k,
												 work_s.data_ptr(), flags_s.data_ptr(), eng._stream()))
			eng.check_flags(flags_s)
			for whole, part in ((p, p_s), (stat, stat_s), (vary, vary_s)):
				whole.index_copy_(1, idx, part)
			bt_d.index_copy_(0, idx, bt_s[:k])  # (alpha below is formed from B and the genes' OLS coefficients)
			if not lowmem and nc:
				ry.coef.index_copy_(0, idx, ry_s.coef[:k])
			eng.last_guard = dict(hits=g.hits, worst=g.worst, fallback=True, genes_redone=k)
		alpha = None
		if not lowmem:
			# alpha_y = b_y - B_y b_x (nc values per gene, the same for every grouping: association.py:551-553 in the closed form)
			b_cov = ry.coef[:ny].cpu().numpy()
			if nx and nc:
				cxt = eng.zeros((_engine._round_up(nc, _lib.ROW_TILE), nxp), torch.float64)
				cxt[:nc, :nx] = rx.coef[:nx].T
				bb = eng.gram(Residualized(ny, nxp, bt_d, None, None), Residualized(nc, nxp, cxt, None, None), False)
				b_cov = b_cov - bb[:ny, :nc].cpu().numpy()
			alpha = np.broadcast_to(b_cov.astype(out_dtype)[None, :, :], (nx, ny, nc)).copy()
		vx = dxx.copy()
		vx[vx == 0] = 1
		mark('flags')
		cert = (norm_mt, norm_ninv, rx.coef[:nx].cpu().numpy() if nc else np.zeros((nx, 0)))
		if device_out and p_host is None:
			return (p, stat, alpha, vx.astype(out_dtype), vary), cert
		out = (eng.download(p) if p_host is None else p_host, eng.download(stat), alpha, vx.astype(out_dtype), eng.download(vary))
		mark('downloads')
		return out, cert
