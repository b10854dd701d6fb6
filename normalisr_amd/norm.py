"""Variance normalisation (mirror of the reference's norm.normvar / normvar1, norm.py:131-289) on the device.

normvar multiplies gene g by w**wt[g] and removes the covariates dc * w**wt[g] from it -- a different small OLS
per gene.  The reference loops over genes (one (n_cov, n_cov) SVD and two skinny matmuls each); here the per-gene
Gram matrices and moment vectors are rows of two Gram contractions on the fp64 matrix cores,
    M_g = sum_k e_gk^2 C_k C_k^T = (U P^T)_g        a_g = sum_k e_gk^2 y_gk C_k = (V C^T)_g ,
with U = e^2, V = e^2 y, P = the pairwise products of covariate rows; the pseudo-inverses (integer ranks) are a
batched SVD on the host, and two HBM-bound element-wise kernels (csrc/nrm_normvar.hip) do the rest.
Only normvar / normvar1 are provided from the reference's norm module (normcov, compute_var are out of scope)."""
import os

import numpy as np

from . import _lib, _opts
from . import engine as _engine
from ._lib import ROW_TILE, K_TILE
from .association import inv_rank
from .de import _finite_within  # (np.isfinite(a).all() from the array's minimum and maximum: one pass each, no temporary)


def _round_up(v, m):
	return (v + m - 1) // m * m


def normvar1(dt, dc, w2=None):
	"""Remove covariates from every row of dt (norm.py:131-163).  w2 (n_gene, n_cell): row g uses dc * w2[g]."""
	dt, dc = np.asarray(dt), np.asarray(dc)
	if w2 is not None:
		return _normvar1_weighted(dt, dc, np.asarray(w2))
	dc64 = np.asarray(dc, dtype=np.float64)
	mi, r = inv_rank(np.matmul(dc64, dc64.T))
	if r <= 0:
		raise RuntimeError('Zero-rank covariates found.')
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		d_c, d_mi = eng.covariates(dc64, mi)
		res = eng.residualize(_engine.as_input(dt), d_c, d_mi, r)
		out = eng.download(res.data[:dt.shape[0], :dt.shape[1]].contiguous())
		assert _finite_within(out)
		return out.astype(np.result_type(dt.dtype, dc.dtype, np.float32), copy=False)


def _normvar1_weighted(dt, dc, w2, tol=1E-8):
	"""normvar1 with w2 (norm.py:150-153): gene g against dc * w2[g].  The per-gene Gram matrices sum_k w2_gk^2 C_k C_k^T and moment
	vectors sum_k w2_gk y_gk C_k are rows of two contractions on the fp64 matrix cores, the pseudo-inverses (integer ranks) a batched
	SVD on the host as in normvar, the residuals one element-wise kernel (nrm_normvar_apply_w2)."""
	nt, ns = dt.shape
	nc = dc.shape[0]
	if w2.shape != (nt, ns):
		raise ValueError('w2 must have the shape of dt.')
	if nc == 0 or nc > 63:
		raise NotImplementedError('normvar1 on the device takes 1 to 63 covariates.')
	out_dtype = np.dtype(np.float32) if np.result_type(dt.dtype, dc.dtype, w2.dtype, np.float32) == np.float32 else np.dtype(np.float64)
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		torch = eng.torch
		iu = np.triu_indices(nc)
		npair = len(iu[0])
		with torch.cuda.device(eng.device):
			y = eng.upload(_engine.as_input(dt))
			d_w = eng.upload(np.asarray(w2, dtype=np.float64))
			d_c = eng.upload(np.asarray(dc, dtype=np.float64))
			rp, kp = _round_up(nt, ROW_TILE), _round_up(ns, K_TILE)
			u = torch.zeros((rp, kp), dtype=torch.float64, device=eng.device)
			v = torch.zeros((rp, kp), dtype=torch.float64, device=eng.device)
			u[:nt, :ns] = d_w * d_w
			v[:nt, :ns] = d_w * y.to(torch.float64)
			pr = torch.zeros((_round_up(npair, ROW_TILE), kp), dtype=torch.float64, device=eng.device)
			pr[:npair, :ns] = d_c[torch.as_tensor(iu[0], device=eng.device)] * d_c[torch.as_tensor(iu[1], device=eng.device)]
			cp = torch.zeros((_round_up(nc, ROW_TILE), kp), dtype=torch.float64, device=eng.device)
			cp[:nc, :ns] = d_c
			R = _engine.Residualized
			gm = eng.gram(R(nt, ns, u, None, None), R(npair, ns, pr, None, None), False)[:nt, :npair].cpu().numpy()
			ga = eng.gram(R(nt, ns, v, None, None), R(nc, ns, cp, None, None), False)[:nt, :nc].cpu().numpy()
			m = np.zeros((nt, nc, nc))
			m[:, iu[0], iu[1]] = gm
			m[:, iu[1], iu[0]] = gm
			from .association import small_pinv
			mi, rk = small_pinv(m, tol)  # per-gene pseudo-inverse by the rank rule of inv_rank (association.py:77), threaded in the library
			if (np.asarray(rk) <= 0).any():
				raise RuntimeError('Zero-rank covariates found.')
			b = np.einsum('gcd,gd->gc', mi, ga)  # b_g = M_g^+ a_g
			tdt = torch.float64 if out_dtype == np.float64 else torch.float32
			out = torch.empty((nt, ns), dtype=tdt, device=eng.device)
			d_b = eng.upload(b)
			_lib.check(eng.lib.nrm_normvar_apply_w2(y.data_ptr(), _lib.NRM_F64 if y.dtype == torch.float64 else _lib.NRM_F32, nt, ns, y.stride(0), d_w.data_ptr(),
													d_w.stride(0), d_c.data_ptr(), nc, d_c.stride(0), d_b.data_ptr(), out.data_ptr(),
													_lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32, ns, eng._stream()))
			dtn = eng.download(out)
		assert _finite_within(dtn)
		return dtn


def _is_dev(a):
	return hasattr(a, 'is_cuda') and a.is_cuda


def _scaled_covariates(dc, w, cat):
	"""Continuous covariate rows (and, cat=1, the intercept; cat=2: every row) are scaled by w (norm.py:261-273)."""
	if cat == 2:
		return dc * w
	dcn = dc.copy()
	t0 = ((dc != 0) & (dc != 1)).any(axis=1)
	if cat == 1:
		t0 |= (dc == 1).all(axis=1)
	dcn = dcn.astype(np.result_type(dc.dtype, w.dtype), copy=False)
	dcn[t0] = dc[t0] * w
	return dcn


def _normvar_host_entry(dt, dc, w, wt, dextra, cat, keepvar, tol, out_dtype):
	"""normvar through nrm_normvar_host (include/normalisr_hip.h): host buffers in and out, no torch -- what `normalisr normvar` needs in a process that
	has numpy and the library only.  NotImplementedError beyond the entry's covariate count (the caller then takes the package's Gram-launch form)."""
	import ctypes
	lib = _lib.load()
	y = _engine.as_input(dt)
	nt, ns = y.shape
	nc = dc.shape[0]
	c64 = np.ascontiguousarray(dc, dtype=np.float64)
	lnw = np.log(np.asarray(w, dtype=np.float64))
	wt64 = np.ascontiguousarray(wt, dtype=np.float64)
	from .association import _result
	out = _result((nt, ns), out_dtype)  # (page-locked, recycled: 400 MB of fresh numpy memory per call cost more than the kernels)
	zero = ctypes.c_int64(0)
	vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
	code = lambda a: _lib.NRM_F64 if a.dtype == np.float64 else _lib.NRM_F32
	_lib.check(lib.nrm_normvar_host(vp(y), code(y), nt, ns, vp(lnw), vp(wt64), vp(c64), nc, float(tol), 1 if keepvar else 0, vp(out), code(out), ctypes.byref(zero)))
	if zero.value:
		raise RuntimeError('Zero-rank covariates found.')
	dcn = _scaled_covariates(dc, w, cat)
	assert _finite_within(out) and _finite_within(dcn)
	ans = [out, dcn]
	if dextra is not None:
		dextran = dextra * w
		assert _finite_within(dextran)
		ans.append(dextran)
	return ans


def normvar(dt, dc, w, wt, dextra=None, cat=1, nth=1, bs=500, keepvar=True, normmean=False, tol=1E-8, device_out=False):
	"""Mean and variance normalisation, same contract as reference norm.py:166-289: returns [dtn, dcn] (+ [dextran]).
	nth and bs are accepted for compatibility and ignored.
	dt may be a torch CUDA tensor already in HBM; device_out=True leaves dtn there (a torch tensor) -- what coex / de / binnet take next
	(examples/GSE123139/code/cmd_coex.sh:38-46 chains the three through files): nothing of the expression matrix crosses PCIe.  With up to 8
	covariates the whole computation stays on the device (csrc/nrm_normvar.hip: per-gene moments in one pass, pseudo-inverses by a thread per
	gene with the host's Jacobi code, one more pass writes the result); more covariates take the Gram-launch form with the host's batched
	pseudo-inverses."""
	if not _is_dev(dt):
		dt = np.asarray(dt)
	dc, w, wt = np.asarray(dc), np.asarray(w), np.asarray(wt)
	if any(x.ndim != 2 for x in (dt, dc)):
		raise ValueError('dt and dc should have 2 dimensions.')
	if any(x.ndim != 1 for x in (w, wt)):
		raise ValueError('w and wt should have 1 dimension.')
	nt, ns = dt.shape
	nc = dc.shape[0]
	if nc == 0:
		raise ValueError('No covariates.')
	if dc.shape[1] != ns or w.shape[0] != ns or wt.shape[0] != nt:
		raise ValueError('Unmatched gene or cell counts.')
	if dextra is not None:
		dextra = np.asarray(dextra)
		if dextra.ndim != 2 or dextra.shape[0] == 0 or dextra.shape[1] != ns:
			raise ValueError('Unmatched shape or size for dextra.')
	if w.min() <= 0:
		raise ValueError('w must be positive.')
	if wt.min() < 0:
		raise ValueError('wt must be non-negative.')
	if cat not in (0, 1, 2):
		raise ValueError('Invalid cat value.')
	if nc > 63:
		raise NotImplementedError('normvar on the device supports at most 63 covariates.')
	dt_dtype = np.dtype(str(dt.dtype).replace('torch.', '')) if _is_dev(dt) else dt.dtype
	out_dtype = np.result_type(dt_dtype, dc.dtype, w.dtype, wt.dtype, np.float32)
	out_dtype = np.dtype(np.float32) if out_dtype == np.float32 else np.dtype(np.float64)
	from .association import _use_host_entry
	if not _is_dev(dt) and not device_out and not normmean and _use_host_entry():
		# no torch in this process (or the command line / NRM_HOST_ENTRY=1): the library's whole-problem entry, numpy buffers in and out
		try:
			return _normvar_host_entry(dt, dc, w, wt, dextra, cat, keepvar, tol, out_dtype)
		except NotImplementedError:
			from .association import _have_torch
			if not _have_torch():
				raise
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		torch = eng.torch
		c64 = np.asarray(dc, dtype=np.float64)
		npair = nc * (nc + 1) // 2
		iu = np.triu_indices(nc)
		with torch.cuda.device(eng.device):
			y = dt if _is_dev(dt) else eng.upload(_engine.as_input(dt))
			if y.dtype not in (torch.float32, torch.float64):
				y = y.to(torch.float64)
			if y.stride(1) != 1:
				y = y.contiguous()
			ycode = _lib.NRM_F64 if y.dtype == torch.float64 else _lib.NRM_F32
			d_lnw = eng.upload(np.log(np.asarray(w, dtype=np.float64)))
			d_wt = eng.upload(np.asarray(wt, dtype=np.float64))
			d_c = eng.upload(c64)
			tdt = torch.float64 if out_dtype == np.float64 else torch.float32
			flags = eng.zeros((4, ), torch.int32)
			on_device = nc <= int(eng.lib.nrm_normvar_device_covariates()) and _opts.debug('normvar', 'device') != 'host'
			if on_device:
				# everything on the device: one pass sums the per-gene moments, a thread per gene solves its small OLS, one pass writes the result
				mom = torch.empty((nt, npair + nc + 2), dtype=torch.float64, device=eng.device)
				d_b = torch.empty((nt, nc), dtype=torch.float64, device=eng.device)
				d_scale = torch.empty((nt, ), dtype=torch.float64, device=eng.device)
				d_rank = torch.empty((nt, ), dtype=torch.int64, device=eng.device)
				with _engine._Span(eng, 'normvar_solve'):
					_lib.check(eng.lib.nrm_normvar_solve(y.data_ptr(), ycode, nt, ns, y.stride(0), d_lnw.data_ptr(), d_wt.data_ptr(), d_c.data_ptr(), nc, d_c.stride(0), float(tol),
														 1 if keepvar else 0, mom.data_ptr(), d_b.data_ptr(), d_scale.data_ptr(), d_rank.data_ptr(), flags.data_ptr(), eng._stream()))
				out = torch.empty((nt, ns), dtype=tdt, device=eng.device)
				with _engine._Span(eng, 'normvar_apply'):
					_lib.check(eng.lib.nrm_normvar_apply(y.data_ptr(), ycode, nt, ns, y.stride(0), d_lnw.data_ptr(), d_wt.data_ptr(), d_c.data_ptr(), nc, d_c.stride(0),
														 d_b.data_ptr(), d_scale.data_ptr(), out.data_ptr(), _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32, ns,
														 flags.data_ptr(), eng._stream()))
				eng._normvar_ranks = d_rank  # (tests: the integer ranks of the last call)
			if not on_device:
				rp, kp = _round_up(nt, ROW_TILE), _round_up(ns, K_TILE)
				u = torch.empty((rp, kp), dtype=torch.float64, device=eng.device)
				v = torch.empty((rp, kp), dtype=torch.float64, device=eng.device)
				s1 = torch.empty((rp, ), dtype=torch.float64, device=eng.device)
				s2 = torch.empty((rp, ), dtype=torch.float64, device=eng.device)
				_lib.check(eng.lib.nrm_normvar_weights(y.data_ptr(), ycode, nt, ns, y.stride(0), d_lnw.data_ptr(), d_wt.data_ptr(), u.data_ptr(),
													   v.data_ptr(), kp, rp, s1.data_ptr(), s2.data_ptr(), eng._stream()))
				# operands of the two Gram contractions: P = pairwise products of covariate rows, C itself
				pr = torch.zeros((_round_up(npair, ROW_TILE), kp), dtype=torch.float64, device=eng.device)
				pr[:npair, :ns] = d_c[torch.as_tensor(iu[0], device=eng.device)] * d_c[torch.as_tensor(iu[1], device=eng.device)]
				cp = torch.zeros((_round_up(nc, ROW_TILE), kp), dtype=torch.float64, device=eng.device)
				cp[:nc, :ns] = d_c
				R = _engine.Residualized
				gm = eng.gram(R(nt, ns, u, None, None), R(npair, ns, pr, None, None), False)[:nt, :npair].cpu().numpy()
				ga = eng.gram(R(nt, ns, v, None, None), R(nc, ns, cp, None, None), False)[:nt, :nc].cpu().numpy()
				# per-gene pseudo-inverse on the host: batched SVD, rank rule of inv_rank (association.py:77)
				m = np.zeros((nt, nc, nc))
				m[:, iu[0], iu[1]] = gm
				m[:, iu[1], iu[0]] = gm
				from .association import small_pinv
				mi, rk = small_pinv(m, tol)  # per-gene pseudo-inverse by the rank rule of inv_rank (association.py:77), threaded in the library
				if (np.asarray(rk) <= 0).any():
					raise RuntimeError('Zero-rank covariates found.')
				b = np.einsum('gcd,gd->gc', mi, ga)  # b_g = M_g^+ a_g
				scale = np.ones(nt)
				if keepvar:
					mean = s1[:nt].cpu().numpy() / ns
					dv = np.sqrt(np.maximum(s2[:nt].cpu().numpy() / ns - mean * mean, 0.0))  # norm.py:248-249
					dv2 = np.sqrt(np.maximum(s2[:nt].cpu().numpy() - np.einsum('gc,gc->g', ga, b), 0.0) / ns)  # |y' - P y'|^2 = |y'|^2 - a.b
					with np.errstate(divide='ignore', invalid='ignore'):
						scale = (dv / dv2)**np.asarray(wt, dtype=np.float64)  # norm.py:259
				out = torch.empty((nt, ns), dtype=tdt, device=eng.device)
				d_b, d_scale = eng.upload(b), eng.upload(scale)
				_lib.check(eng.lib.nrm_normvar_apply(y.data_ptr(), ycode, nt, ns, y.stride(0), d_lnw.data_ptr(), d_wt.data_ptr(), d_c.data_ptr(), nc,
													 d_c.stride(0), d_b.data_ptr(), d_scale.data_ptr(), out.data_ptr(),
													 _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32, ns, flags.data_ptr(), eng._stream()))
			# covariates: continuous rows (and the intercept for cat=1) are scaled by w (norm.py:261-273)
			dcn = _scaled_covariates(dc, w, cat)
			if normmean:
				dcn64 = np.asarray(dcn, dtype=np.float64)
				mi, r = inv_rank(np.matmul(dcn64, dcn64.T))
				if r <= 0:
					raise RuntimeError('Zero-rank covariates found.')
				cov = eng.covariates(dcn64, mi)
				res = eng.residualize(out, cov[0], cov[1], r)
				out = res.data[:nt, :ns].to(tdt).contiguous()
			f = flags.cpu().numpy()
			if f[0]:
				raise RuntimeError('Zero-rank covariates found.')
			assert not f[1]  # np.isfinite(dtn).all() (norm.py:286): counted by the kernel that wrote the values
			dtn = out if device_out else eng.download(out)
		assert (device_out or _finite_within(dtn)) and _finite_within(dcn)
		ans = [dtn, dcn]
		if dextra is not None:
			dextran = dextra * w
			assert _finite_within(dextran)
			ans.append(dextran)
		return ans



class NormvarPlan:
	"""normvar on an expression matrix RESIDENT in HBM, step after step (the matrix rewritten in place between steps: the pipeline normvar -> coex -> binnet of
	examples/GSE123139/code/cmd_coex.sh:38-46 over batches of one shape).  What a call of normvar does on the host for such a step -- log w, three uploads, five
	allocations, the scaled covariates, the flags read back -- is done ONCE here; a step is the three kernels of csrc/nrm_normvar.hip (moments, a lane per gene
	for its small system, the result pass) on the same buffers, one HIP graph from the second step on.  Same results as normvar(dt, ..., device_out=True), bit for bit
	(the same kernels on the same inputs).  check() reads the counters of the last step and raises what normvar raises (norm.py:160,286).
	Up to nrm_normvar_device_covariates() covariates and normmean=False; otherwise every step is the public call."""

	def __init__(self, dt, dc, w, wt, cat=1, keepvar=True, tol=1E-8, eng=None):
		from .distributed import StepGraph
		if not _is_dev(dt):
			raise ValueError('NormvarPlan takes an expression matrix resident in HBM (a torch CUDA tensor); normvar() is the call for host arrays.')
		self.eng = eng = eng or _engine.get_engine(dt.device.index)
		torch = eng.torch
		self.dt, self.dc, self.w, self.wt = dt, np.asarray(dc), np.asarray(w), np.asarray(wt)
		self.cat, self.keepvar, self.tol = cat, keepvar, tol
		nt, ns = dt.shape
		nc = self.dc.shape[0]
		first = normvar(dt, self.dc, self.w, self.wt, cat=cat, keepvar=keepvar, tol=tol, device_out=True)  # (every argument check and error of the public call, once)
		self.dcn = first[1]
		self.lean = nc <= int(eng.lib.nrm_normvar_device_covariates()) and _opts.debug('normvar', 'device') != 'host' and dt.dtype in (torch.float32, torch.float64) and dt.stride(1) == 1
		self.out = first[0]
		self._graph = StepGraph(torch)
		if self.lean:
			with eng.lock, torch.cuda.device(eng.device):
				npair = nc * (nc + 1) // 2
				self._lnw = eng.upload(np.log(np.asarray(self.w, dtype=np.float64)))
				self._wt = eng.upload(np.asarray(self.wt, dtype=np.float64))
				self._c = eng.upload(np.asarray(self.dc, dtype=np.float64))
				self._mom = torch.empty((nt, npair + nc + 2), dtype=torch.float64, device=eng.device)
				self._b = torch.empty((nt, nc), dtype=torch.float64, device=eng.device)
				self._scale = torch.empty((nt, ), dtype=torch.float64, device=eng.device)
				self._rank = torch.empty((nt, ), dtype=torch.int64, device=eng.device)
				self._flags = eng.zeros((4, ), torch.int32)

	def _launch(self):
		eng, y, out = self.eng, self.dt, self.out
		nt, ns = y.shape
		nc = self.dc.shape[0]
		ycode = _lib.NRM_F64 if y.dtype == eng.torch.float64 else _lib.NRM_F32
		ocode = _lib.NRM_F64 if out.dtype == eng.torch.float64 else _lib.NRM_F32
		_lib.check(eng.lib.nrm_normvar_solve(y.data_ptr(), ycode, nt, ns, y.stride(0), self._lnw.data_ptr(), self._wt.data_ptr(), self._c.data_ptr(), nc, self._c.stride(0), float(self.tol),
											 1 if self.keepvar else 0, self._mom.data_ptr(), self._b.data_ptr(), self._scale.data_ptr(), self._rank.data_ptr(), self._flags.data_ptr(), eng._stream()))
		_lib.check(eng.lib.nrm_normvar_apply(y.data_ptr(), ycode, nt, ns, y.stride(0), self._lnw.data_ptr(), self._wt.data_ptr(), self._c.data_ptr(), nc, self._c.stride(0),
											 self._b.data_ptr(), self._scale.data_ptr(), out.data_ptr(), ocode, ns, self._flags.data_ptr(), eng._stream()))

	def step(self, timed=False):
		"""One pass over the matrix as it stands in HBM now; the result in self.out (the same tensor every step), the scaled covariates in self.dcn."""
		eng = self.eng
		with eng.lock, eng.torch.cuda.device(eng.device):
			if not self.lean:
				self.out, self.dcn = normvar(self.dt, self.dc, self.w, self.wt, cat=self.cat, keepvar=self.keepvar, tol=self.tol, device_out=True)[:2]
			else:
				self._graph.run(self._launch)
		return self.out

	def check(self):
		"""The counters of the steps since the last check (one small read-back): RuntimeError / AssertionError as normvar raises them."""
		if not self.lean:
			return True
		eng = self.eng
		with eng.lock, eng.torch.cuda.device(eng.device):
			f = self._flags.cpu().numpy()
			self._flags.zero_()
		if f[0]:
			raise RuntimeError('Zero-rank covariates found.')
		assert not f[1]  # np.isfinite(dtn).all() (norm.py:286): counted by the kernel that wrote the values
		return True

	def results(self):
		"""[dtn, dcn] as normvar returns them (dtn downloaded)."""
		self.check()
		return [self.eng.download(self.out), self.dcn]


assert __name__ != "__main__"
