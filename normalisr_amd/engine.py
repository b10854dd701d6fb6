"""Device orchestration of the association hot path: numpy in -> HBM -> HIP kernels (through the C ABI
of libnormalisr_hip.so) -> numpy out.  torch is used only as plumbing: device memory, the current HIP
stream and (in normalisr_amd.distributed) RCCL collectives.  There is no CPU fallback here.

Data layout in HBM:
  residualised rows   fp64 [rows_pad][k_pad]   rows_pad % 128 == 0, k_pad % 16 == 0, zero padded
  sums of squares     fp64 [rows_pad]
  dot                 fp64 [m_pad][n_pad]
  outputs p / stat    out dtype [nx][ny]
"""
import functools
import os
import threading

import numpy as np

from . import _lib, _opts
from ._lib import NRM_F32, NRM_F64, ROW_TILE, K_TILE, FIX_STRIDE


def _torch():
	import torch
	if not torch.cuda.is_available():
		raise RuntimeError('normalisr_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback.')
	return torch


def _round_up(v, m):
	return (v + m - 1) // m * m


def _code(dtype):
	return NRM_F64 if np.dtype(dtype) == np.float64 else NRM_F32


def as_input(a):
	"""C-contiguous fp32/fp64 view or copy of a 2-D array (ints and other floats -> fp64)."""
	a = np.asarray(a)
	if a.dtype not in (np.float32, np.float64):
		a = a.astype(np.float64)
	return np.ascontiguousarray(a)


def host_blas(limit=8):
	"""Context for the small host LAPACK calls of the path (inverses / eigenvalues of matrices of a few hundred to a
	thousand rows): on a many-core host an unbounded BLAS thread pool makes them 5-10x slower, so cap it."""
	try:
		from threadpoolctl import threadpool_limits
		return threadpool_limits(limits=limit)
	except ImportError:
		import contextlib
		return contextlib.nullcontext()


class PinnedPool:
	"""Page-locked host blocks behind the numpy result arrays of the numpy-in / numpy-out calls.  A fresh 200 MB result needs its
	pages faulted in and registered before a device-to-host copy can run at the PCIe rate (4-9 ms: longer than the whole C2
	computation); a block handed back when its array is garbage-collected is reused by the next call of the same size at no cost.
	The pool is bounded (NRM_PINNED_POOL_MB, default 4096; 0 disables it): beyond the bound results fall back to ordinary numpy
	memory that is page-locked in place for the duration of the call.  Arrays handed to the caller keep their block page-locked
	for as long as the caller holds them (up to the bound in total)."""

	def __init__(self, lib):
		self.lib = lib
		self.limit = int(float(os.environ.get('NRM_PINNED_POOL_MB', '4096')) * (1 << 20))
		self.free = {}   # capacity -> [pointers]
		self.total = 0   # bytes allocated (handed out or free)
		self.lock = threading.RLock()  # re-entrant: _give runs from weakref.finalize, possibly inside a GC pass triggered under the lock

	def _give(self, ptr, cap):
		with self.lock:
			self.free.setdefault(cap, []).append(ptr)

	def empty(self, shape, dtype):
		"""A C-contiguous numpy array on page-locked memory, or None when the pool is exhausted / disabled."""
		import ctypes
		import weakref
		dtype = np.dtype(dtype)
		nbytes = int(np.prod(shape)) * dtype.itemsize
		if nbytes < (1 << 20) or self.limit <= 0:
			return None
		cap = (nbytes + (1 << 21) - 1) >> 21 << 21
		with self.lock:
			ptr = self.free[cap].pop() if self.free.get(cap) else None
			if ptr is None:
				if self.total + cap > self.limit:  # drop idle blocks of other sizes before giving up
					for c in list(self.free):
						while self.free[c] and self.total + cap > self.limit:
							self.lib.nrm_host_free(self.free[c].pop())
							self.total -= c
				if self.total + cap > self.limit:
					return None
				out = ctypes.c_void_p()
				if self.lib.nrm_host_alloc(ctypes.byref(out), cap) != 0:
					return None
				ptr = out.value
				self.total += cap
		buf = (ctypes.c_char * nbytes).from_address(ptr)
		weakref.finalize(buf, self._give, ptr, cap)  # the array (and every view of it) keeps `buf` alive
		return np.frombuffer(buf, dtype=dtype).reshape(shape)


class _Span:
	"""HIP events around one kernel group on the launch stream while engine.trace is a list (bench.py's per-kernel split of steps
	that are otherwise timed as a whole); nothing when it is None."""

	def __init__(self, eng, name):
		self.eng, self.name = eng, name

	def __enter__(self):
		if self.eng.trace is not None:
			self.e0 = self.eng.torch.cuda.Event(enable_timing=True)
			self.e0.record(self.eng.torch.cuda.current_stream(self.eng.device))
		return self

	def __exit__(self, *exc):
		if self.eng.trace is not None:
			e1 = self.eng.torch.cuda.Event(enable_timing=True)
			e1.record(self.eng.torch.cuda.current_stream(self.eng.device))
			self.eng.trace.append((self.name, self.e0, e1))
		return False


def serialised(method):
	"""Engine method that runs with the engine's lock held.  One Engine per device is shared by every thread of the process --
	its Gram / skinny / K1 scratch buffers, copy streams and pinned pool are shared state -- so calls from several Python threads
	(the reference calls its block kernel from a thread pool: association.py:895,997, parallel.py:47-52; ctypes releases the GIL
	inside every launch) take turns per device; the lock is re-entrant because the composite methods call the simple ones."""

	@functools.wraps(method)
	def locked(self, *a, **ka):
		with self.lock:
			return method(self, *a, **ka)

	return locked


class GuardHit(Exception):
	"""The integer Gram engine's accuracy guard (csrc/nrm_fix.h) could not certify every P-value of a call: the caller redoes it on
	the fp64 Gram kernel.  Internal: never reaches the user."""

	def __init__(self, hits, worst):
		Exception.__init__(self, '{} pairs not certified (largest error estimate {:.3g})'.format(hits, worst))
		self.hits, self.worst = hits, worst


class Residualized:
	"""Residualised rows resident in HBM -- as fp64 (`data`) and / or as the fixed-point digit planes of the integer Gram
	engine (`_quant`, with the row records `fix` K3 needs for them: csrc/nrm_fix.h) -- plus their sums of squares and OLS
	coefficients."""

	def __init__(self, rows, n, data, ss, coef, shape=None):
		self.rows, self.n = rows, n
		self.data, self.ss, self.coef = data, ss, coef
		self.rows_pad, self.k_pad = data.shape if shape is None else shape
		self.cks = None  # chunked digit planes: k-steps (32 cells) per chunk; _quant[0] is then the list of chunk operands
		self.fix = None  # (rows_pad, FIX_STRIDE) fp64 row records written next to the digit planes


class Engine:
	def __init__(self, device=None):
		torch = _torch()
		self.torch = torch
		self.lib = _lib.load()
		self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
		self.lock = threading.RLock()
		self._tls = threading.local()
		self._gram_work = None
		self._skinny_ws = None
		self._copy = None
		self.pool = PinnedPool(self.lib)
		self._cmax = {}
		self._force_f64 = False
		self.trace = None  # a list collects (name, start event, end event) of K1 / K2 / K3 launches (see _Span)
		# the integer engine's accuracy guard: largest relative change of a P-value it may cause (NRM_I8_GUARD_TOL, 0 = no guard)
		self.guard_tol = float(os.environ.get('NRM_I8_GUARD_TOL', '2.5e-7'))
		self.last_guard = dict(hits=0, worst=0.0, fallback=False)

	@property
	def last_guard(self):
		"""Verdict of the accuracy guard for the calling thread's last call on this engine."""
		return getattr(self._tls, 'guard', None) or dict(hits=0, worst=0.0, fallback=False)

	@last_guard.setter
	def last_guard(self, v):
		self._tls.guard = v

	def _stream(self):
		return self.torch.cuda.current_stream(self.device).cuda_stream

	@serialised
	def upload(self, a, dtype=None):
		"""numpy array -> device tensor.  Arrays of half a GB and more go through the library's staged copy (csrc/nrm_upload.hip: host threads
		fill page-locked blocks beside the DMA: 38 -> 54 GB/s for the 3 GB expression matrix of BASELINE configs[3]; at 200 MB the runtime's
		own pageable copy already runs at 56 GB/s and stays); NRM_UPLOAD=torch: torch's copy whatever the size."""
		a = np.ascontiguousarray(a)
		if dtype is None and a.nbytes >= (512 << 20) and str(a.dtype) in self._NPT and _opts.debug('upload', 'staged') != 'torch':
			torch = self.torch
			with torch.cuda.device(self.device):
				t = torch.empty(a.shape, dtype=getattr(torch, self._NPT[str(a.dtype)]), device=self.device)
				_lib.check(self.lib.nrm_upload(a.ctypes.data, t.data_ptr(), a.nbytes, 0, self._stream()))
			return t
		t = self.torch.from_numpy(a)
		if dtype is not None:
			t = t.to(dtype)
		return t.to(self.device, non_blocking=False)

	_NPT = {'float32': 'float32', 'float64': 'float64', 'int32': 'int32', 'int64': 'int64', 'uint8': 'uint8', 'int16': 'int16', 'bool': 'bool'}

	_NP = {'torch.float32': np.float32, 'torch.float64': np.float64, 'torch.bool': np.bool_, 'torch.int32': np.int32,
		   'torch.uint8': np.uint8, 'torch.int64': np.int64}

	def host_pin(self, a):
		_lib.check(self.lib.nrm_host_pin(a.ctypes.data, a.nbytes, 0))

	def host_unpin(self, a):
		_lib.check(self.lib.nrm_host_unpin(a.ctypes.data))

	@serialised
	def download(self, t):
		"""Device tensor -> a fresh numpy array.  Large results are copied into the array while it is page-locked (a
		pageable D2H copy runs at a tenth of the PCIe rate); the lock is dropped before returning."""
		if t.numel() * t.element_size() < (1 << 20) or str(t.dtype) not in self._NP:
			return t.cpu().numpy()
		torch = self.torch
		t = t.contiguous()
		out = self.pool.empty(tuple(t.shape), self._NP[str(t.dtype)])
		if out is not None:  # recycled page-locked block
			with torch.cuda.device(self.device):
				_lib.check(self.lib.nrm_copy_to_host(out.ctypes.data, t.data_ptr(), out.nbytes, self._stream()))
				torch.cuda.current_stream(self.device).synchronize()
			return out
		out = np.empty(tuple(t.shape), dtype=self._NP[str(t.dtype)])
		try:
			self.host_pin(out)
		except RuntimeError:  # e.g. a locked-memory limit: plain pageable copy
			return t.cpu().numpy()
		try:
			with torch.cuda.device(self.device):
				_lib.check(self.lib.nrm_copy_to_host(out.ctypes.data, t.data_ptr(), out.nbytes, self._stream()))
				torch.cuda.current_stream(self.device).synchronize()
		finally:
			self.host_unpin(out)
		return out

	@serialised
	def zeros(self, shape, dtype):
		"""Zero-filled device buffer (allocation by torch, the fill through the C ABI on the launch stream)."""
		t = self.torch.empty(shape, dtype=dtype, device=self.device)
		_lib.check(self.lib.nrm_fill_zero(t.data_ptr(), t.numel() * t.element_size(), self._stream()))
		return t

	@serialised
	def copy_rows(self, dst, src):
		"""dst[:rows, :cols] = src for 2-D device tensors of one dtype with unit column stride (strided rows allowed)."""
		rows, cols = src.shape
		assert dst.dtype == src.dtype and dst.stride(1) == 1 and src.stride(1) == 1 and dst.shape[0] >= rows and dst.shape[1] >= cols
		es = src.element_size()
		_lib.check(self.lib.nrm_copy_rows(dst.data_ptr(), dst.stride(0) * es, src.data_ptr(), src.stride(0) * es, cols * es, rows, self._stream()))

	@serialised
	def download_into(self, t, out):
		"""Device tensor -> an existing C-contiguous numpy array (e.g. this rank's rows of a result matrix shared between the
		ranks of a node): page-locked in place for the copy when possible, plain pageable copy otherwise."""
		assert out.flags['C_CONTIGUOUS'] and out.nbytes == t.numel() * t.element_size(), 'download_into: shape/dtype mismatch'
		if out.nbytes == 0:
			return out
		torch = self.torch
		t = t.contiguous()
		pinned = False
		if out.nbytes >= (1 << 20):
			try:
				self.host_pin(out)
				pinned = True
			except RuntimeError:  # locked-memory limit, or a mapping that cannot be registered
				pinned = False
		try:
			with torch.cuda.device(self.device):
				_lib.check(self.lib.nrm_copy_to_host(out.ctypes.data, t.data_ptr(), out.nbytes, self._stream()))
				torch.cuda.current_stream(self.device).synchronize()
		finally:
			if pinned:
				self.host_unpin(out)
		return out

	@serialised
	def covariates(self, dc, dci):
		"""fp64 covariates and pseudo-inverse on the device (replicated; tiny)."""
		torch = self.torch
		nc = dc.shape[0]
		if nc == 0:
			return None, None
		dc64 = np.asarray(dc, dtype=np.float64)
		d_c = self.upload(dc64)
		# largest |C_c| of every covariate row: lets K1 bound the residuals it quantises without sweeping them for their maximum
		self._cmax = {k: v for k, v in self._cmax.items() if v[0]() is not None}
		import weakref
		const = np.nonzero((dc64 == dc64[:, :1]).all(axis=1) & (dc64[:, 0] != 0))[0]  # constant rows (the intercept)
		self._cmax[d_c.data_ptr()] = (weakref.ref(d_c), self.upload(np.abs(dc64).max(axis=1)), int(const[0]) if const.size else -1,
									 float(dc64[const[0], 0]) if const.size else 0.0)
		return d_c, self.upload(np.asarray(dci, dtype=np.float64).reshape(nc, nc))

	def constant_row(self, d_c):
		"""(index, value) of a constant covariate row of device covariates uploaded through covariates(), or (-1, 0)."""
		e = None if d_c is None else self._cmax.get(d_c.data_ptr())
		return (e[2], e[3]) if e is not None and e[0]() is d_c else (-1, 0.0)

	def cmax_ptr(self, d_c):
		e = None if d_c is None else self._cmax.get(d_c.data_ptr())
		return e[1].data_ptr() if e is not None and e[0]() is d_c else 0

	@serialised
	def residualize(self, x, d_c, d_dci, rank, want_coef=False, rows_pad=None, nslices=0, keep_fp64=True, into=None):
		"""K1 on a host (numpy) or device (torch) matrix of shape (rows, n).  nslices = 5 / 6: also (keep_fp64=False: only) the
		fixed-point digit planes of the integer Gram engine, written by K1 itself.
		into: a Residualized of an earlier call with the same arguments whose buffers may be overwritten (a resident plan's previous
		step): no GB-sized allocation per step -- the caching allocator serves those from its free blocks most of the time, but a
		free 11 GB block that was split for a 100 MB request in between costs the next step a 50 ms hipMalloc (seen as an
		intermittent 5 - 40x slower "K1" of the configs[4] slice in bench.py)."""
		if isinstance(x, np.ndarray):
			x = self.upload(as_input(x))
		with _Span(self, 'residualize'):
			return self._residualize(x, d_c, d_dci, rank, want_coef, rows_pad, nslices, keep_fp64, into)

	@staticmethod
	def _reusable(into, rows, n, rp, kp, nslices, plane_bytes, chunked):
		q = getattr(into, '_quant', None)
		if into is None or q is None or into.data is not None or into.coef is not None or into.fix is None:
			return False
		planes = q[0]
		if chunked != isinstance(planes, list):
			return False
		have = sum(int(t.numel()) for t in planes) if chunked else int(planes.numel())
		return (into.rows, into.n, into.rows_pad, into.k_pad, q[2]) == (rows, n, rp, kp, nslices) and have == plane_bytes

	def _residualize(self, x, d_c, d_dci, rank, want_coef, rows_pad, nslices, keep_fp64, into=None):
		torch = self.torch
		with torch.cuda.device(self.device):
			if isinstance(x, np.ndarray):
				x = self.upload(as_input(x))
			rows, n = x.shape
			nc = 0 if d_c is None else d_c.shape[0]
			rp = _round_up(max(rows, 1), ROW_TILE) if rows_pad is None else rows_pad
			kp = _round_up(n, K_TILE)
			esz = x.element_size()
			fused = bool(nslices) and rp % ROW_TILE == 0 and self.k1_quantises(x, d_c)
			if fused:
				coef = self.zeros((rows, nc), torch.float64) if want_coef else None
				out = torch.empty((rp, kp), dtype=torch.float64, device=self.device) if keep_fp64 else None
				pb = int(self.lib.nrm_quant_bytes(rp, kp, nslices))
				if not keep_fp64 and not want_coef and self._reusable(into, rows, n, rp, kp, nslices, pb, False):
					ss, planes, exps, fix = into.ss, into._quant[0], into._quant[1], into.fix
				else:
					ss = torch.empty((rp, ), dtype=torch.float64, device=self.device)
					planes = torch.empty((pb, ), dtype=torch.uint8, device=self.device)
					exps = torch.empty((rp, ), dtype=torch.int32, device=self.device)
					fix = torch.empty((rp, FIX_STRIDE), dtype=torch.float64, device=self.device)
				_lib.check(self.lib.nrm_residualize_q(
					x.data_ptr(), NRM_F64 if x.dtype == torch.float64 else NRM_F32, rows, n, x.stride(0),
					0 if d_c is None else d_c.data_ptr(), nc, 0 if d_c is None else d_c.stride(0),
					0 if d_dci is None else d_dci.data_ptr(), int(rank), 0 if out is None else out.data_ptr(), kp, rp, ss.data_ptr(),
					0 if coef is None else coef.data_ptr(), nslices, planes.data_ptr(), exps.data_ptr(), 0, self.cmax_ptr(d_c), fix.data_ptr(), self._stream()))
				r = Residualized(rows, n, out, ss, coef, shape=(rp, kp))
				r._quant = (planes, exps, nslices)
				r.fix = fix
				return r
			out = torch.empty((rp, kp), dtype=torch.float64, device=self.device)
			ss = torch.empty((rp, ), dtype=torch.float64, device=self.device)
			coef = self.zeros((rows, nc), torch.float64) if want_coef else None
			_lib.check(self.lib.nrm_residualize(
				x.data_ptr(), NRM_F64 if x.dtype == torch.float64 else NRM_F32, rows, n, x.stride(0),
				0 if d_c is None else d_c.data_ptr(), nc, 0 if d_c is None else d_c.stride(0),
				0 if d_dci is None else d_dci.data_ptr(), int(rank),
				out.data_ptr(), kp, rp, ss.data_ptr(), 0 if coef is None else coef.data_ptr(), self._stream()))
		return Residualized(rows, n, out, ss, coef)

	@staticmethod
	def k1_quantises(x, d_c=None):
		"""K1 writes the digit planes itself when the rows (and covariates) are 16-byte aligned."""
		return x.stride(1) == 1 and (x.stride(0) * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0 and (
			d_c is None or ((d_c.stride(0) * 8) % 16 == 0 and d_c.data_ptr() % 16 == 0))

	@serialised
	def residualize_chunked(self, x, d_c, d_dci, rank, rows_pad, nslices, chunks, into=None):
		"""K1 with the digit planes cut along the cells into (at most) `chunks` operands of equal size that share the row
		exponents (nrm_residualize_q_chunked): what the sharded coex path sends to the other GPUs piece by piece.
		into: see residualize."""
		torch = self.torch
		with torch.cuda.device(self.device):
			rows, n = x.shape
			nc = 0 if d_c is None else d_c.shape[0]
			kp = _round_up(n, K_TILE)
			nks = (kp + 31) // 32
			cks = (nks + max(1, chunks) - 1) // max(1, chunks)
			nchunks = (nks + cks - 1) // cks
			cb = int(self.lib.nrm_quant_bytes(rows_pad, 32 * cks, nslices))
			if self._reusable(into, rows, n, rows_pad, kp, nslices, nchunks * cb, True) and into.cks == cks and getattr(into, '_planes', None) is not None:
				planes, exps, ss, fix = into._planes, into._quant[1], into.ss, into.fix
			else:
				planes = torch.empty((nchunks * cb, ), dtype=torch.uint8, device=self.device)
				exps = torch.empty((rows_pad, ), dtype=torch.int32, device=self.device)
				ss = torch.empty((rows_pad, ), dtype=torch.float64, device=self.device)
				fix = torch.empty((rows_pad, FIX_STRIDE), dtype=torch.float64, device=self.device)
			_lib.check(self.lib.nrm_residualize_q_chunked(
				x.data_ptr(), NRM_F64 if x.dtype == torch.float64 else NRM_F32, rows, n, x.stride(0),
				0 if d_c is None else d_c.data_ptr(), nc, 0 if d_c is None else d_c.stride(0),
				0 if d_dci is None else d_dci.data_ptr(), int(rank), rows_pad, ss.data_ptr(), nslices, planes.data_ptr(), exps.data_ptr(),
				cks, self.cmax_ptr(d_c), fix.data_ptr(), self._stream()))
		r = Residualized(rows, n, None, ss, None, shape=(rows_pad, kp))
		r._quant = ([planes[c * cb:(c + 1) * cb] for c in range(nchunks)], exps, nslices)
		r._planes = planes  # (the one buffer the chunk operands are views of)
		r.cks = cks
		r.fix = fix
		return r

	@serialised
	def gram_chunk(self, a, b, symmetric, chunk, dot, accumulate):
		"""K2 over one cell chunk of chunked operands (residualize_chunked): dot (+)= a_chunk @ b_chunk.T, exact per chunk."""
		torch = self.torch
		with torch.cuda.device(self.device):
			if dot is None:
				dot = torch.empty((a.rows_pad, b.rows_pad), dtype=torch.float64, device=self.device)
			if self._gram_work is None:
				self._gram_work = torch.empty((int(self.lib.nrm_gram_workspace_bytes()) // 8, ), dtype=torch.float64, device=self.device)
			qa, qb = a._quant, b._quant
			assert a.cks is not None and a.cks == b.cks and qa[2] == qb[2]
			pitch = lambda q: q[3] if len(q) > 3 else 0
			_lib.check(self.lib.nrm_gram_i8_chunk(qa[0][chunk].data_ptr(), qa[1].data_ptr(), pitch(qa), qb[0][chunk].data_ptr(), qb[1].data_ptr(), pitch(qb),
												  a.rows_pad, b.rows_pad, 32 * a.cks, qa[2], dot.data_ptr(), dot.stride(0), 1 if symmetric else 0,
												  int(a.rows), int(b.rows), 1 if accumulate else 0, 0, 0, 0, 1, self._gram_work.data_ptr(), self._stream()))
		return dot

	@serialised
	def gram_chunk_blocks(self, a, g_chunk, g_exps, first, count, chunk, dot, accumulate):
		"""The same against `count` consecutive blocks (cyclically from block `first`) of a gathered buffer in ONE launch:
		g_chunk (world, chunk bytes) holds every rank's digit planes of this cell chunk, g_exps (world, rows_pad) their row
		exponents.  dot: (a.rows_pad, count * a.rows_pad), block j in columns [j rows_pad, (j + 1) rows_pad)."""
		torch = self.torch
		with torch.cuda.device(self.device):
			world, rp = g_exps.shape
			if dot is None:
				dot = torch.empty((a.rows_pad, count * rp), dtype=torch.float64, device=self.device)
			if self._gram_work is None:
				self._gram_work = torch.empty((int(self.lib.nrm_gram_workspace_bytes()) // 8, ), dtype=torch.float64, device=self.device)
			qa = a._quant
			pitch = lambda q: q[3] if len(q) > 3 else 0
			_lib.check(self.lib.nrm_gram_i8_chunk(qa[0][chunk].data_ptr(), qa[1].data_ptr(), pitch(qa), g_chunk.data_ptr(), g_exps.data_ptr(), 0,
												  a.rows_pad, count * rp, 32 * a.cks, qa[2], dot.data_ptr(), dot.stride(0), 0, int(a.rows), 0,
												  1 if accumulate else 0, rp, g_chunk.stride(0) * g_chunk.element_size(), int(first), int(world),
												  self._gram_work.data_ptr(), self._stream()))
		return dot

	I8_MIN_CELLS = 2048
	I8_MAX_CELLS = 1 << 22  # (K1's digit statistics are summed in 32 bits across 16 lanes)

	def gram_slices(self, n_cells):
		"""Digit slices of the integer Gram engine for the association path: NRM_GRAM=i8 (default, 6 slices = 46-bit fixed point),
		i8x5 (5 slices = 38 bits, faster), f64 (the fp64 matrix-core kernel).  Below I8_MIN_CELLS cells the fp64 kernel is used
		anyway: the problem is small, and the integer engine's error in Pearson r (its dropped low-order digit products,
		~2e-15 at 10 000 cells) grows as 1 / sqrt(n_cells).  0 as well while a call is being redone after the accuracy guard fired."""
		mode = os.environ.get('NRM_GRAM', 'i8')
		if mode not in ('i8', 'i8x5', 'f64'):
			raise ValueError('NRM_GRAM must be i8, i8x5 or f64')
		if self._force_f64:
			return 0
		return {'i8': 6, 'i8x5': 5, 'f64': 0}[mode] if self.I8_MIN_CELLS <= n_cells < self.I8_MAX_CELLS else 0

	def fix_args(self, rx, ry):
		"""(digit planes, row records of the x rows, of the y rows, guard tolerance) for the sweep of a dot product the integer engine
		made from rx and ry; zeros for the fp64 Gram kernels."""
		fx, fy = getattr(rx, 'fix', None), getattr(ry, 'fix', None)
		if fx is None or fy is None or _opts.debug('i8_fix', '1') == '0':  # (the switch exists for the tests that show what the records are for)
			return 0, 0, 0, 0.0
		return int(rx._quant[2]), fx.data_ptr(), fy.data_ptr(), float(self.guard_tol)

	def new_flags(self):
		"""int32[4] device counters of a call: non-finite, R^2 > 1 + 1e-8 (association.py:248,252), guard hits, largest guard estimate."""
		return self.zeros((4, ), self.torch.int32)

	@serialised
	def quantized(self, r, nslices):
		"""Fixed-point digit planes and row exponents of residualised rows (cached on the Residualized object; written by K1
		itself when it could, see residualize)."""
		torch = self.torch
		q = getattr(r, '_quant', None)
		if q is None or q[2] != nslices:
			assert r.data is not None, 'no fp64 residuals to quantise'  # (keep_fp64=False rows carry their digit planes)
			with torch.cuda.device(self.device):
				planes = torch.empty((int(self.lib.nrm_quant_bytes(r.rows_pad, r.k_pad, nslices)), ), dtype=torch.uint8, device=self.device)
				exps = torch.empty((r.rows_pad, ), dtype=torch.int32, device=self.device)
				fix = torch.empty((r.rows_pad, FIX_STRIDE), dtype=torch.float64, device=self.device)
				_lib.check(self.lib.nrm_quantize_rows(r.data.data_ptr(), r.rows_pad, r.k_pad, r.data.stride(0), nslices, planes.data_ptr(),
													  exps.data_ptr(), fix.data_ptr(), int(r.n), self._stream()))
			q = (planes, exps, nslices)
			r._quant = q
			r.fix = fix
		return q

	def row_block(self, r, lo, hi, rows=None):
		"""Rows [lo, hi) (multiples of ROW_TILE) of residualised rows as an operand of their own: a view, nothing is copied."""
		assert lo % ROW_TILE == 0 and hi % ROW_TILE == 0 and 0 <= lo < hi <= r.rows_pad
		sub = Residualized(max(0, min(hi, r.rows) - lo) if rows is None else rows, r.n, None if r.data is None else r.data[lo:hi],
						   None if r.ss is None else r.ss[lo:hi], None, shape=(hi - lo, r.k_pad))
		q = getattr(r, '_quant', None)
		if q is not None:
			nks = (r.k_pad + 31) // 32 if r.cks is None else r.cks
			dense = (r.rows_pad // 32) * nks * 1024
			first = (lo // 32) * nks * 1024
			sub._quant = (q[0][first:] if r.cks is None else [t[first:] for t in q[0]], q[1][lo:hi], q[2], q[3] if len(q) > 3 else dense)
			sub.cks = r.cks
			sub.fix = None if r.fix is None else r.fix[lo:hi]
		return sub

	@serialised
	def gram(self, a, b, symmetric, dot=None, rows=None, nslices=0):
		"""K2: dot[m_pad, n_pad] = a.data @ b.data.T.  nslices = 0: fp64 matrix cores (nrm_gram.hip); 5 / 6: the exact
		fixed-point engine on the int8 matrix cores (nrm_gram_i8.hip), used by the association path for expression-like rows.
		rows=(row0, row1): only that band of dot."""
		with _Span(self, 'gram'):
			return self._gram(a, b, symmetric, dot, rows, nslices)

	def _gram(self, a, b, symmetric, dot, rows, nslices):
		torch = self.torch
		with torch.cuda.device(self.device):
			if dot is None:
				dot = torch.empty((a.rows_pad, b.rows_pad), dtype=torch.float64, device=self.device)
			if self._gram_work is None:
				self._gram_work = torch.empty((int(self.lib.nrm_gram_workspace_bytes()) // 8, ), dtype=torch.float64, device=self.device)
			row0, row1 = (0, a.rows_pad) if rows is None else rows
			if nslices:
				qa = self.quantized(a, nslices)
				qb = qa if b is a else self.quantized(b, nslices)
				pitch = lambda q: q[3] if len(q) > 3 else 0  # plane pitch of a row block of a larger quantised matrix (0 = dense)
				_lib.check(self.lib.nrm_gram_i8_band(qa[0].data_ptr(), qa[1].data_ptr(), pitch(qa), qb[0].data_ptr(), qb[1].data_ptr(), pitch(qb),
													 a.rows_pad, b.rows_pad, a.k_pad, nslices, dot.data_ptr(), dot.stride(0), 1 if symmetric else 0,
													 int(a.rows), int(b.rows), int(row0), int(row1), self._gram_work.data_ptr(), self._stream()))
				return dot
			_lib.check(self.lib.nrm_gram_f64_band(a.data.data_ptr(), b.data.data_ptr(), a.rows_pad, b.rows_pad, a.k_pad,
												  a.data.stride(0), b.data.stride(0), dot.data_ptr(), dot.stride(0),
												  1 if symmetric else 0, int(a.rows), int(b.rows), int(row0), int(row1),
												  self._gram_work.data_ptr(), self._stream()))
		return dot

	BAND = 8 * ROW_TILE  # rows per band of the pipelined path (one row of K2's 8x8 super-blocks)

	def banded_ok(self, nx, ny, out_dtype):
		"""Pipeline K2 -> K3 -> copy-out by row bands when there is more than one band and the results are large enough
		for the PCIe leg to matter (NRM_PIPELINE=0 switches it off)."""
		if _opts.debug('pipeline', '1') == '0':
			return False
		return nx > self.BAND and 2 * nx * ny * np.dtype(out_dtype).itemsize >= (16 << 20)

	CHUNK_BYTES = 256 << 20

	def chunked_ok(self, dy):
		"""de whose expression matrix still sits on the host and is large: upload it in row chunks on a second stream
		so that K1/K2/K3 of chunk c run while chunk c+1 crosses PCIe (NRM_PIPELINE=0 switches it off)."""
		return (isinstance(dy, np.ndarray) and dy.nbytes >= 2 * self.CHUNK_BYTES and _opts.debug('pipeline', '1') != '0')

	@serialised
	def association_de_chunked(self, dx, dy, dc, dci, rank, dof, stat_kind, out_dtype, cov=None):
		"""General de path with the expression rows streamed from the host: every chunk of genes is an independent
		problem against the same residualised design rows (association.py:890-909: the reference's tiles are independent
		in the same way), so the H2D leg -- the longest leg of a one-shot call -- hides the kernels."""
		torch = self.torch
		dy = as_input(dy)
		nx, n = dx.shape
		ny = dy.shape[0]
		tdt = torch.float64 if np.dtype(out_dtype) == np.float64 else torch.float32
		rows = max(ROW_TILE, (self.CHUNK_BYTES // (n * dy.itemsize)) // ROW_TILE * ROW_TILE)
		with torch.cuda.device(self.device):
			main = torch.cuda.current_stream(self.device)
			if self._copy is None:
				self._copy = torch.cuda.Stream(device=self.device)
			d_c, d_dci = self.covariates(dc, dci) if cov is None else cov
			rx = self.residualize(dx, d_c, d_dci, rank, nslices=self.gram_slices(n))
			p = torch.empty((nx, ny), dtype=tdt, device=self.device)
			stat = torch.empty((nx, ny), dtype=tdt, device=self.device)
			ssy = torch.empty((ny, ), dtype=torch.float64, device=self.device)
			flags = self.new_flags()
			esz = p.element_size()
			for a in range(0, ny, rows):
				b = min(ny, a + rows)
				with torch.cuda.stream(self._copy):
					yc = torch.from_numpy(dy[a:b]).to(self.device)  # host blocks here; the GPU is busy with the previous chunk
				arrived = torch.cuda.Event()
				arrived.record(self._copy)
				main.wait_event(arrived)
				yc.record_stream(main)
				ry = self.residualize(yc, d_c, d_dci, rank, nslices=self.gram_slices(n), keep_fp64=False)
				dot = self.gram(rx, ry, False, nslices=self.gram_slices(n))
				_lib.check(self.lib.nrm_assoc_sweep(dot.data_ptr(), dot.stride(0), rx.ss.data_ptr(), ry.ss.data_ptr(), nx, b - a, int(n),
													float(dof), 0, int(stat_kind), p.data_ptr() + a * esz, stat.data_ptr() + a * esz, 0, 0,
													_code(out_dtype), ny, flags.data_ptr(), *self.fix_args(rx, ry), self._stream()))
				_lib.check(self.lib.nrm_copy_rows(ssy.data_ptr() + a * 8, 8 * (b - a), ry.ss.data_ptr(), 8 * (b - a), 8 * (b - a), 1, self._stream()))
			self.check_flags(flags)
			return dict(p=self.download(p), stat=self.download(stat), alpha=None, varx=self.variances(rx.ss, nx, n, out_dtype),
						vary=self.variances(ssy, ny, n, out_dtype), dof=dof)

	@serialised
	def start_host_results(self, nx, ny, out_dtype, bands=None):
		"""Result arrays p and stat on the host, being page-locked by a helper thread (overlaps K1 and the first band of K2).
		bands: row cuts [0, ..., nx] -- the rows are then locked band by band in that order and host['ready'] counts the bands
		done, so that copies into the first bands can start while the later ones are still being faulted in and registered."""
		odt = np.dtype(out_dtype)
		cuts = [0, nx] if bands is None else list(bands)
		pp, ps = self.pool.empty((nx, ny), odt), self.pool.empty((nx, ny), odt)
		if pp is not None and ps is not None:  # recycled page-locked blocks: nothing to fault in or register
			res = dict(p=pp, stat=ps, pinned=[], error=[], ready=len(cuts), thread=threading.Thread(target=lambda: None))
			res['thread'].start()
			return res
		del pp, ps
		res = dict(p=np.empty((nx, ny), dtype=odt), stat=np.empty((nx, ny), dtype=odt), pinned=[], error=[], ready=0)

		def lock_pages():
			try:
				for a, b in zip(cuts[:-1], cuts[1:]):
					for k in ('p', 'stat'):
						part = res[k][a:b]
						self.host_pin(part)
						res['pinned'].append(part)
					res['ready'] += 1
			except Exception as e:  # re-raised by the consumer
				res['error'].append(e)
		res['thread'] = threading.Thread(target=lock_pages)
		res['thread'].start()
		return res

	@serialised
	def finish_host_results(self, host):
		host['thread'].join()
		self.torch.cuda.synchronize(self.device)
		for a in host['pinned']:
			self.host_unpin(a)
		host['pinned'] = []

	@serialised
	def association_banded(self, rx, ry, samexy, nx, ny, n, dof, stat_kind, out_dtype, host):
		"""K2 + K3 band by band on the compute stream while finished bands of p and stat travel to the page-locked result
		arrays on a copy stream.  Same kernels as the one-launch path; K2's split of the cells between workgroups depends
		on the tiles of a launch, so dot may differ from it in the last bits (each path is bitwise reproducible)."""
		torch = self.torch
		tdt = torch.float64 if np.dtype(out_dtype) == np.float64 else torch.float32
		cuts = list(range(0, nx, self.BAND)) + [nx]
		with torch.cuda.device(self.device):
			main = torch.cuda.current_stream(self.device)
			if self._copy is None:
				self._copy = torch.cuda.Stream(device=self.device)
			odt = np.dtype(out_dtype)
			hp, hs, th = host['p'], host['stat'], host['thread']
			try:
				dot = torch.empty((rx.rows_pad, ry.rows_pad), dtype=torch.float64, device=self.device)
				p = torch.empty((nx, ny), dtype=tdt, device=self.device)
				stat = torch.empty((nx, ny), dtype=tdt, device=self.device)
				flags = self.new_flags()
				done = []
				for a, b in zip(cuts[:-1], cuts[1:]):
					self.gram(rx, ry, samexy, dot=dot, rows=(a, rx.rows_pad if b == nx else b), nslices=self.gram_slices(n))
					_lib.check(self.lib.nrm_assoc_sweep_band(dot.data_ptr(), dot.stride(0), rx.ss.data_ptr(), ry.ss.data_ptr(), nx, ny, int(n),
															 float(dof), 1 if samexy else 0, int(stat_kind), p.data_ptr(), stat.data_ptr(), 0, 0,
															 _code(out_dtype), max(ny, 1), flags.data_ptr(), a, b, *self.fix_args(rx, ry), self._stream()))
					ev = torch.cuda.Event()
					ev.record(main)
					done.append(ev)
				th.join()
				if host['error']:  # e.g. a locked-memory limit: the copies below still work, at the pageable rate
					import logging
					logging.warning('normalisr_amd: result arrays could not be page-locked (%s); copying out unpinned.', host['error'][0])
				row = ny * odt.itemsize
				for (a, b), ev in zip(zip(cuts[:-1], cuts[1:]), done):
					self._copy.wait_event(ev)
					for h, d in ((hp, p), (hs, stat)):
						_lib.check(self.lib.nrm_copy_to_host(h.ctypes.data + a * row, d.data_ptr() + a * row, (b - a) * row, self._copy.cuda_stream))
				self._copy.synchronize()
			finally:
				self.finish_host_results(host)
			self.check_flags(flags)
		return hp, hs

	def coex_pipelined_ok(self, dx, dc, n):
		"""coex whose expression matrix still sits on the host, large enough for PCIe to matter, on the integer engine with rows K1
		can quantise itself (16-byte aligned): upload, kernels and copy-out overlap chunk by chunk (NRM_PIPELINE=0 switches it off)."""
		return (isinstance(dx, np.ndarray) and _opts.debug('pipeline', '1') != '0' and self.gram_slices(n) > 0 and dx.shape[0] > self.BAND
				and dx.nbytes >= (32 << 20) and (dx.shape[1] * dx.itemsize) % 16 == 0 and (dc.shape[0] == 0 or (dc.shape[1] * 8) % 16 == 0))

	@serialised
	def association_coex_pipelined(self, dx, dc, dci, rank, dimreduce, out_dtype, cov=None):
		"""norm.coex, numpy in -> numpy out, with the three PCIe / compute legs overlapped.  The gene rows travel to the GPU in
		chunks of BAND rows on a copy stream; as soon as chunk c = rows [a, b) has landed, K1 residualises and quantises it, K2
		contracts it with itself (symmetric) and with all earlier rows [0, a) (a rectangle), K3 turns both into P-values --
		writing the rectangle's pairs at (i, j) AND (j, i) -- and everything chunk c completes (rows a..b up to column b, columns
		a..b of the rows above) leaves on a second copy stream into the page-locked result arrays while chunk c + 1 is still
		arriving.  The reference fills its result arrays tile by tile the same way (association.py:997-1034,1049-1057)."""
		torch = self.torch
		ng, n = dx.shape
		dof = n - 1 - rank - dimreduce
		ns = self.gram_slices(n)
		odt = np.dtype(out_dtype)
		tdt = torch.float64 if odt == np.float64 else torch.float32
		code = _code(out_dtype)
		esz = odt.itemsize
		mp, kp = _round_up(ng, ROW_TILE), _round_up(n, K_TILE)
		nks = (kp + 31) // 32
		plane = (mp // 32) * nks * 1024
		cuts = list(range(0, ng, self.BAND)) + [ng]
		import time
		trace = [] if _opts.debug('trace') else None
		mark = (lambda what: trace.append((what, time.perf_counter()))) if trace is not None else (lambda what: None)
		mark('start')
		with torch.cuda.device(self.device):
			main = torch.cuda.current_stream(self.device)
			if self._copy is None:
				self._copy = torch.cuda.Stream(device=self.device)
			if getattr(self, '_copy_out', None) is None:
				self._copy_out = torch.cuda.Stream(device=self.device)
			d_c, d_dci = self.covariates(dc, dci) if cov is None else cov
			nc = 0 if d_c is None else d_c.shape[0]
			planes = torch.empty((plane * ns, ), dtype=torch.uint8, device=self.device)
			exps = torch.empty((mp, ), dtype=torch.int32, device=self.device)
			ss = torch.empty((mp, ), dtype=torch.float64, device=self.device)
			fixt = torch.empty((mp, FIX_STRIDE), dtype=torch.float64, device=self.device)
			p = torch.empty((ng, ng), dtype=tdt, device=self.device)
			stat = torch.empty((ng, ng), dtype=tdt, device=self.device)
			flags = self.new_flags()
			whole = Residualized(ng, n, None, ss, None, shape=(mp, kp))
			whole._quant = (planes, exps, ns)
			whole.fix = fixt
			host = None
			pending = []
			row = ng * esz
			# Half of P and of the covariance is the mirror image of the other half: with NRM_HOST_MIRROR=1 only the rows a..b up to column b
			# cross PCIe and a helper thread mirrors each block above the diagonal as soon as it has landed (nrm_host_mirror_rows: bitwise
			# symmetric by construction, like the reference's own triu + transpose, association.py:1049-1057).  Opt-in, because it measured
			# SLOWER on the MI355X box (configs[1]: 11.6 ms against 7.7): the copies of all chunks but the last are hidden behind the uploads
			# anyway, so halving them shortens the tail by 0.5 ms only (copied out at 6.9 instead of 7.4 ms), while 32 host threads transpose
			# the 100 MB at ~16 GB/s -- below the 27 GB/s of the rectangular copy-out they replace.
			host_mirror = _opts.debug('host_mirror', '0') == '1'
			import queue
			mirror_q, mirror_err = queue.Queue(), []

			def mirror_worker():
				while True:
					item = mirror_q.get()
					if item is None:
						return
					ev, ma, mb = item
					try:
						ev.synchronize()
						for h in (host['p'], host['stat']):
							_lib.check(self.lib.nrm_host_mirror_rows(h.ctypes.data, row, esz, ma, mb, 0))
					except Exception as e:  # noqa: BLE001 -- re-raised by the caller
						mirror_err.append(e)
			mirror_thread = threading.Thread(target=mirror_worker) if host_mirror else None
			if mirror_thread is not None:
				mirror_thread.start()

			def ship(block):
				"""Queue the copy-out of every finished chunk whose rows of the result arrays are page-locked by now (all of them, waiting
				for the locks, when block is set): rows a..b up to column b, and columns a..b of the rows above (the mirrored halves)."""
				while pending:
					ci, a, b, done = pending[0]
					while host['ready'] <= ci and host['thread'].is_alive() and block:
						time.sleep(2e-5)
					if host['ready'] <= ci and not (block and not host['thread'].is_alive()):
						return
					if host['error'] and not host.get('warned'):
						import logging
						logging.warning('normalisr_amd: result arrays could not be page-locked (%s); copying out unpinned.', host['error'][0])
						host['warned'] = True
					pending.pop(0)
					self._copy_out.wait_event(done)
					for h, d in ((host['p'], p), (host['stat'], stat)):
						_lib.check(self.lib.nrm_copy_rect_to_host(h.ctypes.data + a * row, row, d.data_ptr() + a * row, row, b * esz, b - a, self._copy_out.cuda_stream))
						if not host_mirror:  # the mirrored halves travel too: columns a..b of the rows above
							for u, v in zip(cuts[:ci], cuts[1:ci + 1]):  # (one copy per band of rows: a copy must stay inside one page-locked range)
								_lib.check(self.lib.nrm_copy_rect_to_host(h.ctypes.data + u * row + a * esz, row, d.data_ptr() + u * row + a * esz, row, (b - a) * esz, v - u,
																		  self._copy_out.cuda_stream))
					if host_mirror and a > 0:  # ... or are made on the host from what has arrived, while later rows are in flight
						arrived_out = torch.cuda.Event()
						arrived_out.record(self._copy_out)
						mirror_q.put((arrived_out, a, b))
			mark('buffers')
			try:
				for ci, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
					with torch.cuda.stream(self._copy):
						xc = torch.from_numpy(dx[a:b]).to(self.device)  # the host blocks here while the GPU works on the previous chunk
					mark('upload %d returned' % ci)
					arrived = torch.cuda.Event()
					arrived.record(self._copy)
					main.wait_event(arrived)
					xc.record_stream(main)
					if host is None:  # page-lock the result arrays from a helper thread, started after the first upload (see association_single0)
						host = self.start_host_results(ng, ng, out_dtype, bands=cuts)
					rpc = _round_up(b - a, ROW_TILE)
					_lib.check(self.lib.nrm_residualize_q(
						xc.data_ptr(), NRM_F64 if xc.dtype == torch.float64 else NRM_F32, b - a, n, xc.stride(0),
						0 if d_c is None else d_c.data_ptr(), nc, 0 if d_c is None else d_c.stride(0), 0 if d_dci is None else d_dci.data_ptr(), int(rank),
						0, kp, rpc, ss.data_ptr() + a * 8, 0, ns, planes.data_ptr() + (a // 32) * nks * 1024, exps.data_ptr() + a * 4, plane, self.cmax_ptr(d_c),
						fixt.data_ptr() + a * FIX_STRIDE * 8, self._stream()))
					blk = self.row_block(whole, a, a + rpc, rows=b - a)
					dot = self.gram(blk, blk, True, nslices=ns)
					_lib.check(self.lib.nrm_assoc_sweep(dot.data_ptr(), dot.stride(0), blk.ss.data_ptr(), blk.ss.data_ptr(), b - a, b - a, int(n), float(dof), 1, 0,
														p.data_ptr() + (a * ng + a) * esz, stat.data_ptr() + (a * ng + a) * esz, 0, 0, code, ng, flags.data_ptr(),
														*self.fix_args(blk, blk), self._stream()))
					if a > 0:
						prev = self.row_block(whole, 0, a, rows=a)
						dot2 = self.gram(blk, prev, False, nslices=ns)
						_lib.check(self.lib.nrm_assoc_sweep_mirror(dot2.data_ptr(), dot2.stride(0), blk.ss.data_ptr(), ss.data_ptr(), b - a, a, int(n), float(dof),
																   p.data_ptr(), stat.data_ptr(), code, ng, a, 0, flags.data_ptr(), *self.fix_args(blk, prev), self._stream()))
					done = torch.cuda.Event()
					done.record(main)
					pending.append((ci, a, b, done))
					mark('launched %d' % ci)
					ship(False)
				ship(True)
				mark('queued')
				self._copy_out.synchronize()
				mark('copied out')
			finally:
				if mirror_thread is not None:
					mirror_q.put(None)
					mirror_thread.join()
					mark('mirrored')
				if host is not None:
					self.finish_host_results(host)
			if mirror_err:
				raise mirror_err[0]
			mark('unpinned')
			self.check_flags(flags)
			res = dict(p=host['p'], stat=host['stat'], alpha=None, varx=None, vary=self.variances(ss, ng, n, out_dtype), dof=dof)
			mark('done')
			if trace:
				t0 = trace[0][1]
				print('coex pipeline trace (ms): ' + ', '.join('%s %.2f' % (w, (t - t0) * 1e3) for w, t in trace))
			return res

	@serialised
	def coex_blocks_resident(self, block_fn, n_genes, n, dc, dci, rank, dimreduce, out_dtype, block_rows=3840, timings=None):
		"""coex of a matrix that is too large to sit in HBM next to its own digit planes (BASELINE configs[4]: 30 000 genes x 500 000
		cells fp64 = 120 GB in, 90 GB of planes) on ONE GPU: the gene rows arrive block by block -- block_fn(lo, hi) returns rows
		[lo, hi) as a device tensor, which is dropped as soon as K1 has residualised and quantised it into the resident planes -- then
		K2 and K3 run band by band over the whole symmetric problem (the reference walks the same pair space tile by tile,
		association.py:890-909).  Results stay in HBM: dict(p, stat = covariance, ss, flags).  block_rows: a multiple of 128."""
		torch = self.torch
		ns = self.gram_slices(n)
		assert ns, 'coex_blocks_resident runs on the integer engine (2048 <= cells < 2^22)'
		assert block_rows % ROW_TILE == 0
		dof = n - 1 - rank - dimreduce
		odt = np.dtype(out_dtype)
		tdt = torch.float64 if odt == np.float64 else torch.float32
		mp, kp = _round_up(n_genes, ROW_TILE), _round_up(n, K_TILE)
		nks = (kp + 31) // 32
		plane = (mp // 32) * nks * 1024
		with torch.cuda.device(self.device):
			ev = lambda: torch.cuda.Event(enable_timing=True)
			marks = []

			def mark(name):
				if timings is not None:
					e = ev()
					e.record()
					marks.append((name, e))
			d_c, d_dci = self.covariates(dc, dci)
			nc = 0 if d_c is None else d_c.shape[0]
			planes = torch.empty((plane * ns, ), dtype=torch.uint8, device=self.device)
			exps = torch.empty((mp, ), dtype=torch.int32, device=self.device)
			ss = torch.empty((mp, ), dtype=torch.float64, device=self.device)
			fixt = torch.empty((mp, FIX_STRIDE), dtype=torch.float64, device=self.device)
			whole = Residualized(n_genes, n, None, ss, None, shape=(mp, kp))
			whole._quant = (planes, exps, ns)
			whole.fix = fixt
			mark('start')
			for a in range(0, n_genes, block_rows):
				b = min(n_genes, a + block_rows)
				x = block_fn(a, b)
				assert tuple(x.shape) == (b - a, n) and self.k1_quantises(x, d_c)
				rpc = _round_up(b - a, ROW_TILE)
				with _Span(self, 'residualize'):
					_lib.check(self.lib.nrm_residualize_q(
						x.data_ptr(), NRM_F64 if x.dtype == torch.float64 else NRM_F32, b - a, n, x.stride(0),
						0 if d_c is None else d_c.data_ptr(), nc, 0 if d_c is None else d_c.stride(0), 0 if d_dci is None else d_dci.data_ptr(), int(rank),
						0, kp, rpc, ss.data_ptr() + a * 8, 0, ns, planes.data_ptr() + (a // 32) * nks * 1024, exps.data_ptr() + a * 4, plane, self.cmax_ptr(d_c),
						fixt.data_ptr() + a * FIX_STRIDE * 8, self._stream()))
				torch.cuda.current_stream(self.device).synchronize()  # the block is released before the next one is made
				del x
			mark('residualised')
			dot = torch.empty((mp, mp), dtype=torch.float64, device=self.device)
			p = torch.empty((n_genes, n_genes), dtype=tdt, device=self.device)
			stat = torch.empty((n_genes, n_genes), dtype=tdt, device=self.device)
			flags = self.new_flags()
			cuts = list(range(0, n_genes, self.BAND)) + [n_genes]
			for a, b in zip(cuts[:-1], cuts[1:]):
				self.gram(whole, whole, True, dot=dot, rows=(a, mp if b == n_genes else b), nslices=ns)
				with _Span(self, 'sweep'):
					_lib.check(self.lib.nrm_assoc_sweep_band(dot.data_ptr(), dot.stride(0), ss.data_ptr(), ss.data_ptr(), n_genes, n_genes, int(n), float(dof), 1, 0,
															 p.data_ptr(), stat.data_ptr(), 0, 0, _code(out_dtype), n_genes, flags.data_ptr(), a, b,
															 *self.fix_args(whole, whole), self._stream()))
			mark('swept')
			if timings is not None:
				torch.cuda.synchronize(self.device)
				for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
					timings[n1] = e0.elapsed_time(e1)
			return dict(p=p, stat=stat, ss=ss, flags=flags, dof=dof, dot=dot)

	@serialised
	def sweep(self, dot, ssx, ssy, nx, ny, n_cells, dof, symmetric, stat_kind, out_dtype, want_rt=False, flags=None, fix=None):
		"""K3: p, stat (covariance or gamma) and optionally Pearson r and t for every pair.  fix = fix_args(rx, ry) when dot came
		from the integer engine."""
		torch = self.torch
		tdt = torch.float64 if np.dtype(out_dtype) == np.float64 else torch.float32
		with torch.cuda.device(self.device), _Span(self, 'sweep'):
			p = torch.empty((nx, ny), dtype=tdt, device=self.device)
			stat = torch.empty((nx, ny), dtype=tdt, device=self.device)
			r = torch.empty((nx, ny), dtype=tdt, device=self.device) if want_rt else None
			t = torch.empty((nx, ny), dtype=tdt, device=self.device) if want_rt else None
			if flags is None:
				flags = self.new_flags()
			_lib.check(self.lib.nrm_assoc_sweep(dot.data_ptr(), dot.stride(0), ssx.data_ptr(), ssy.data_ptr(), nx, ny,
												int(n_cells), float(dof), 1 if symmetric else 0, int(stat_kind),
												p.data_ptr(), stat.data_ptr(), 0 if r is None else r.data_ptr(),
												0 if t is None else t.data_ptr(), _code(out_dtype), max(ny, 1),
												flags.data_ptr(), *(fix if fix is not None else (0, 0, 0, 0.0)), self._stream()))
		return p, stat, r, t, flags

	@serialised
	def alpha(self, stat, stat_kind, ssx, n_cells, bx, by, nc):
		"""alpha = by - gamma bx (association.py:238-243) from K3's statistic: gamma (stat_kind 1) or the covariance
		(stat_kind 0, return_dot) -- the reference derives alpha from gamma either way (:1044-1048 only rescales)."""
		torch = self.torch
		nx, ny = stat.shape
		with torch.cuda.device(self.device):
			out = torch.empty((nx, ny, nc), dtype=stat.dtype, device=self.device)
			code = NRM_F64 if stat.dtype == torch.float64 else NRM_F32
			_lib.check(self.lib.nrm_alpha(stat.data_ptr(), code, stat.stride(0), int(stat_kind), ssx.data_ptr(), int(n_cells),
										  bx.data_ptr(), by.data_ptr(), nx, ny, nc, out.data_ptr(), code, self._stream()))
		return out

	@serialised
	def check_flags(self, flags):
		"""The reference's assertions (association.py:248,252) on a call's device counters, then the verdict of the integer engine's
		accuracy guard: pairs it could not certify raise GuardHit, which association_single0 (and the sharded drivers) answer by
		redoing the call on the fp64 Gram kernel."""
		f = flags.cpu().numpy()
		if f[0] or f[1]:
			raise AssertionError('association results failed the reference assertions (association.py:248,252): '
								 '{} tiles with non-finite values, {} tiles with R^2 > 1+1e-8'.format(int(f[0]), int(f[1])))
		if f.shape[0] >= 4:
			worst = float(f[3:4].view(np.float32)[0])
			self.last_guard = dict(hits=int(f[2]), worst=worst, fallback=False)
			if f[2] > 0:
				raise GuardHit(int(f[2]), worst)

	@serialised
	def variances(self, ss, count, n_cells, out_dtype):
		"""ss/n with the variance 0 -> 1 substitution (association.py:230-233)."""
		v = ss[:count].cpu().numpy() / float(n_cells)
		v[v == 0] = 1
		return v.astype(out_dtype, copy=False)

	def _skinny_work(self):
		if self._skinny_ws is None:
			self._skinny_ws = self.torch.empty((int(self.lib.nrm_gram_skinny_workspace_bytes()) // 8, ), dtype=self.torch.float64, device=self.device)
		return self._skinny_ws

	def _rows_padded16(self, a):
		"""Device copy of a (rows, n) matrix whose rows are readable and zero up to a multiple of 16 cells (K2s streams
		16-cell slabs without bounds checks).  No copy when n is already a multiple of 16 and the array is on the device."""
		torch = self.torch
		rows, n = a.shape
		n16 = _round_up(n, 16)
		if isinstance(a, np.ndarray):
			t = torch.from_numpy(np.ascontiguousarray(a))
			if n16 == n:
				return t.to(self.device)
			buf = self.zeros((rows, n16), t.dtype)
			self.copy_rows(buf, t.to(self.device))
			return buf[:, :n]
		if n16 == n and a.stride(1) == 1 and (a.stride(0) * a.element_size()) % 16 == 0 and a.data_ptr() % 16 == 0:
			return a
		buf = self.zeros((rows, n16), a.dtype)
		self.copy_rows(buf, a if a.stride(1) == 1 else a.contiguous())
		return buf[:, :n]

	def de_streaming_ok(self, dx, dy, dc):
		"""The streaming path (K2s) applies to de with few design rows and 16-byte aligned expression rows."""
		mode = _opts.debug('de_path', 'auto')
		if dy is None or mode == 'general':
			return False
		ok = dx.shape[0] + dc.shape[0] <= 32
		if mode == 'streaming' and not ok:
			raise ValueError('NRM_DE_PATH=streaming needs nx + nc <= 32')
		return ok

	@serialised
	def association_de_streaming(self, dx, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt=False, cov=None,
								 resident=False, state=None):
		"""de with nx + nc <= 32: stream the raw expression rows once (HBM-bound), see csrc/nrm_gram_skinny.hip.
		state: a dict owned by the caller that keeps the buffers worth keeping between calls on the same covariates (the permuted
		covariates, the Z buffer).  A DePlan passes its own, so that the pointers its captured HIP graph holds live exactly as long
		as the plan, whatever else runs on this engine in between; one-shot calls keep nothing."""
		if state is None:
			state = {}
		torch = self.torch
		nx, n = dx.shape
		ny, nc = dy.shape[0], dc.shape[0]
		dof = n - 1 - rank - dimreduce
		tdt = torch.float64 if np.dtype(out_dtype) == np.float64 else torch.float32
		with torch.cuda.device(self.device):
			d_c, d_dci = self.covariates(dc, dci) if cov is None else cov
			# A constant covariate row (the intercept) leaves Z: its product with every expression row is a plain sum, taken on the
			# vector ALU by the streaming kernel (column 31 of G) instead of occupying a matrix-core row group.  The covariates are
			# reordered so that it comes last; dci is permuted with them (b' = P b: no rank assumption), alpha is put back in order.
			ci, cval = self.constant_row(d_c)
			if _opts.debug('const_row', '1') == '0' or nc + nx > 31 + (ci >= 0):
				ci, cval = -1, 0.0
			perm = None
			if ci >= 0:
				perm = [c for c in range(nc) if c != ci] + [ci]
				pc = state.get('cperm')
				if pc is None or pc[0] is not d_c:  # (kept with the covariates: no upload inside a step that is replayed as a graph)
					pc = (d_c, d_c[perm].contiguous(),
						  self.upload(np.ascontiguousarray(np.asarray(dci, dtype=np.float64).reshape(nc, nc)[np.ix_(perm, perm)])))
					state['cperm'] = pc
				d_cz, d_dciz = pc[1], pc[2]
			else:
				d_cz, d_dciz = d_c, d_dci
			ncz = nc - (1 if ci >= 0 else 0)  # covariate rows that stay in Z
			# design rows: a = x C^T through the streaming Gram (all CUs), then x~ = x - (a dci) C spread along the cells
			k32 = _round_up(n, 128)
			# Z = [C; X~; 0], stacked on the device through the C ABI.  The covariate rows do not change between calls on the same
			# covariates (a DePlan's steps): the buffer is kept and only the rows past the covariates are rewritten.
			zc = state.get('z')
			if zc is not None and zc[0] is d_c and d_c is not None and zc[1].shape[1] == k32 and zc[2] == ci:
				z = zc[1]
				_lib.check(self.lib.nrm_fill_zero(z[ncz:].data_ptr(), (31 - ncz) * k32 * 8, self._stream()))
			else:
				z = self.zeros((32, k32), torch.float64)
				if ncz:
					self.copy_rows(z, d_cz[:ncz])
				if nc:
					state['z'] = (d_c, z, ci)
			xd = self._rows_padded16(as_input(dx) if isinstance(dx, np.ndarray) else dx)
			xcode = NRM_F64 if xd.dtype == torch.float64 else NRM_F32
			gx = torch.empty((256, 32), dtype=torch.float64, device=self.device)
			active = rank > 0 and nc > 0
			if active:  # a = x C^T (against the covariates in Z's order, the constant one last: column 31)
				xc_work = torch.empty((int(self.lib.nrm_design_products_workspace_doubles(nx, n)), ), dtype=torch.float64, device=self.device)
				_lib.check(self.lib.nrm_design_products(xd.data_ptr(), xcode, nx, n, xd.stride(0), d_cz.data_ptr(), nc, d_cz.stride(0), gx.data_ptr(),
														xc_work.data_ptr(), 1 if ci >= 0 else 0, self._stream()))
			xt = z[ncz:ncz + nx]  # the residualised design rows are written straight into their rows of Z (zero padded up to k32)
			rw_work = torch.empty((64 * ((k32 + 1023) // 1024), ), dtype=torch.float64, device=self.device)
			ssx = torch.empty((ROW_TILE, ), dtype=torch.float64, device=self.device)
			coefx = self.zeros((nx, nc), torch.float64) if want_alpha else None
			_lib.check(self.lib.nrm_residualize_wide(xd.data_ptr(), xcode, nx, n, xd.stride(0), 0 if d_cz is None else d_cz.data_ptr(), nc,
													 0 if d_cz is None else d_cz.stride(0), gx.data_ptr(), 0 if d_dciz is None else d_dciz.data_ptr(),
													 int(rank), xt.data_ptr(), k32, ssx.data_ptr(), 0 if coefx is None else coefx.data_ptr(),
													 rw_work.data_ptr(), 1 if ci >= 0 else 0, self._stream()))
			rx = Residualized(nx, n, xt, ssx, coefx)
			y = self._rows_padded16(dy)
			ycode = NRM_F64 if y.dtype == torch.float64 else NRM_F32
			ny_pad = _round_up(ny, 256)
			g = torch.empty((ny_pad, 32), dtype=torch.float64, device=self.device)
			ssraw = torch.empty((ny_pad, ), dtype=torch.float64, device=self.device)
			with _Span(self, 'gram'):
				_lib.check(self.lib.nrm_gram_skinny(y.data_ptr(), ycode, ny, n, y.stride(0), z.data_ptr(), k32, k32, g.data_ptr(), ssraw.data_ptr(),
													ny_pad, ncz + nx, float(cval), self._skinny_work().data_ptr(), self._stream()))
			p = torch.empty((nx, ny), dtype=tdt, device=self.device)
			stat = torch.empty((nx, ny), dtype=tdt, device=self.device)
			r = torch.empty((nx, ny), dtype=tdt, device=self.device) if want_rt else None
			t = torch.empty((nx, ny), dtype=tdt, device=self.device) if want_rt else None
			ssy = torch.empty((ny_pad, ), dtype=torch.float64, device=self.device)
			by = self.zeros((ny, nc), torch.float64) if (want_alpha and nc) else None
			flags = self.new_flags()
			stat_kind = 0 if return_dot else 1
			_lib.check(self.lib.nrm_de_small_sweep(g.data_ptr(), ssraw.data_ptr(), 0 if d_dciz is None else d_dciz.data_ptr(), nc, int(rank),
												   rx.ss.data_ptr(), nx, ny, n, float(dof), stat_kind, p.data_ptr(), stat.data_ptr(),
												   0 if r is None else r.data_ptr(), 0 if t is None else t.data_ptr(), _code(out_dtype),
												   ny, ssy.data_ptr(), 0 if by is None else by.data_ptr(), flags.data_ptr(), 1 if ci >= 0 else 0, self._stream()))
			alpha = None
			if want_alpha:
				if nc > 0:
					alpha = self.alpha(stat, stat_kind, rx.ss, n, rx.coef, by, nc).cpu().numpy()
					if perm is not None:  # coefficients came out in the reordered covariate order
						back = np.empty(nc, dtype=np.int64)
						back[perm] = np.arange(nc)
						alpha = np.ascontiguousarray(alpha[..., back])
				else:
					alpha = np.zeros((nx, ny, nc), dtype=out_dtype)
			if resident:  # resident pipeline: nothing leaves the device, the caller checks `flags` when it reads the results
				return dict(p=p, stat=stat, alpha=alpha, ssx=rx.ss, ssy=ssy, flags=flags, dof=dof)
			self.check_flags(flags)
			res = dict(p=self.download(p), stat=self.download(stat), alpha=alpha, varx=self.variances(rx.ss, nx, n, out_dtype),
					   vary=self.variances(ssy, ny, n, out_dtype), dof=dof)
			if want_rt:
				res['r'] = r.cpu().numpy()
				res['t'] = t.cpu().numpy()
		return res

	@serialised
	def association_single0(self, dx, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt=False, cov=None,
							device_out=False, resident=False, state=None):
		"""Whole-problem single=0 path on one device.  dy None -> coex (symmetric).
		cov: optional (d_c, d_dci) already on the device (repeated calls with the same covariates).
		A call whose P-values the integer engine's guard cannot certify (GuardHit) is redone on the fp64 Gram kernel; resident calls
		leave that to whoever reads the flags (DePlan.results)."""
		import logging
		import time
		t0 = time.perf_counter()
		self._tls.path = None
		try:
			res = self._association_single0(dx, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, cov, device_out, resident, state)
		except GuardHit as g:
			logging.info('normalisr_amd: integer Gram engine: %s; redoing the call on the fp64 matrix cores.', g)
			with self.forced_f64():
				res = self._association_single0(dx, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, cov, device_out, resident, state)
			self.last_guard = dict(hits=g.hits, worst=g.worst, fallback=True)
		if not resident and logging.getLogger().isEnabledFor(logging.INFO):
			# which engine ran, how fast, what the guard said (the reference logs its batch decisions: association.py:745-757, run.py:274-276)
			el = time.perf_counter() - t0
			nx, n = dx.shape
			tests = nx * (nx - 1) // 2 if dy is None else nx * dy.shape[0]
			g = self.last_guard
			logging.info('normalisr_amd: %s, %d tests over %d cells on %s in %.1f ms (%.3g tests/s incl. transfers); %s', 'coex' if dy is None else 'de', tests, n,
						 getattr(self._tls, 'path', None) or 'the device', 1e3 * el, tests / max(el, 1e-9),
						 'fp64 matrix cores, no guard needed' if (g['hits'] == 0 and g['worst'] == 0 and not g['fallback']) else
						 ('guard: %d pairs not certified by the integer engine (largest estimate %.2g): redone on the fp64 matrix cores' % (g['hits'], g['worst']) if g['fallback']
						  else 'guard: every P-value certified by the integer engine (largest relative error bound %.2g, budget %.2g)' % (g['worst'], self.guard_tol)))
		return res

	def forced_f64(self):
		"""Context: every Gram product of this engine goes to the fp64 kernel (the redo after a GuardHit)."""
		import contextlib

		@contextlib.contextmanager
		def ctx():
			prev, self._force_f64 = self._force_f64, True
			try:
				yield
			finally:
				self._force_f64 = prev
		return ctx()

	def _association_single0(self, dx, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, cov, device_out, resident, state=None):
		samexy = dy is None
		if self.de_streaming_ok(dx, dy, dc):
			self._tls.path = 'the streaming de kernel (fp64 matrix cores, raw rows read once)'
			return self.association_de_streaming(dx, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, cov,
												 resident=resident and not (want_alpha or want_rt), state=state)
		from . import de_sparse
		if de_sparse.candidate(self, dx, dy, dc, samexy):
			# a design matrix with few entries (gRNA incidence): the expression rows are read once, raw (csrc/nrm_de_sparse.hip)
			with self.torch.cuda.device(self.device):
				if isinstance(dx, np.ndarray):
					dx = self.upload(as_input(dx))  # (kept for the dense path below if the matrix turns out not to be sparse)
				# a resident caller's lists live in its state (kept as long as the plan, whatever else runs on this engine in between); they belong to
				# one tensor in one state: torch counts in-place writes in ._version (permutation nulls on the same buffer get new lists)
				lists = state.get('sparse') if state is not None else None
				if lists is None or lists[0] is not dx or lists[1] != dx._version:
					lists = (dx, dx._version, de_sparse.lists_for(self, dx))
					if state is not None:
						state['sparse'] = lists
			if lists[2].ok and not (state is not None and state.get('sparse_refused') is dx):
				self._tls.path = 'the sparse-design kernel (expression rows read once, %d design entries)' % lists[2].nnz
				return self._association_de_sparse(dx, lists[2], dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, cov, device_out, resident)
		nx, n = dx.shape
		ny = nx if samexy else dy.shape[0]
		nc = dc.shape[0]
		dof = n - 1 - rank - dimreduce
		stat_kind = 0 if (samexy or return_dot) else 1
		eng_name = (lambda k: 'the integer Gram engine (%d-bit fixed point on the int8 matrix cores)' % (8 * k - 2) if k else 'the fp64 Gram kernel')(self.gram_slices(n))
		if samexy and not (device_out or resident or want_rt or want_alpha) and self.coex_pipelined_ok(dx, dc, n):
			self._tls.path = eng_name + ', rows uploaded / results returned chunk by chunk beside the kernels'
			return self.association_coex_pipelined(dx, dc, dci, rank, dimreduce, out_dtype, cov)
		if not (samexy or device_out or resident or want_rt or want_alpha) and self.chunked_ok(dy):
			self._tls.path = eng_name + ', expression rows uploaded in chunks beside the kernels'
			return self.association_de_chunked(dx, dy, dc, dci, rank, dof, stat_kind, out_dtype, cov)
		self._tls.path = eng_name
		d_c, d_dci = self.covariates(dc, dci) if cov is None else cov
		ns = self.gram_slices(n)  # integer engine: K1 writes the digit planes itself and the fp64 residuals are never stored
		# a resident caller's state (a DePlan) lends the previous step's K1 outputs to be overwritten: no GB-sized allocation per step
		keep = resident and state is not None and state.get('keep', False)
		rx = self.residualize(dx, d_c, d_dci, rank, want_coef=want_alpha, nslices=ns, keep_fp64=not ns, into=state.get('rx') if keep else None)
		ry = rx if samexy else self.residualize(dy, d_c, d_dci, rank, want_coef=want_alpha, nslices=ns, keep_fp64=not ns, into=state.get('ry') if keep else None)
		if keep:
			state['rx'], state['ry'] = rx, ry
		host = None
		if not (device_out or resident or want_rt or want_alpha) and self.banded_ok(nx, ny, out_dtype):
			# the result arrays are page-locked by a helper thread while K1/K2 run.  Started only now, after the uploads: a
			# hipHostRegister racing a pageable H2D copy (right after the previous call's hipHostUnregister) was measured
			# to stall that copy by ~20 ms -- twice the whole C2 call
			host = self.start_host_results(nx, ny, out_dtype)
		if host is not None:
			p, stat = self.association_banded(rx, ry, samexy, nx, ny, n, dof, stat_kind, out_dtype, host)
			return dict(p=p, stat=stat, alpha=None, varx=None if samexy else self.variances(rx.ss, nx, n, out_dtype),
						vary=self.variances(ry.ss, ny, n, out_dtype), dof=dof)
		dot = self.gram(rx, ry, samexy, nslices=ns)
		p, stat, r, t, flags = self.sweep(dot, rx.ss, ry.ss, nx, ny, n, dof, samexy, stat_kind, out_dtype, want_rt, fix=self.fix_args(rx, ry))
		alpha = None
		if want_alpha:
			if nc > 0 and not samexy:
				alpha = self.download(self.alpha(stat, stat_kind, rx.ss, n, rx.coef, ry.coef, nc))
			else:
				alpha = np.zeros((nx, ny, nc), dtype=out_dtype)
		if resident and not (want_alpha or want_rt):  # resident pipeline (see association_de_streaming)
			return dict(p=p, stat=stat, alpha=alpha, ssx=None if samexy else rx.ss, ssy=ry.ss, flags=flags, dof=dof)
		self.check_flags(flags)
		keep = (lambda t: t) if device_out else self.download  # device_out: p / stat stay in HBM (torch tensors) for a device pipeline
		res = dict(p=keep(p), stat=keep(stat), alpha=alpha,
				   varx=None if samexy else self.variances(rx.ss, nx, n, out_dtype),
				   vary=self.variances(ry.ss, ny, n, out_dtype), dof=dof)
		if want_rt:
			res['r'] = self.download(r)
			res['t'] = self.download(t)
		return res


	def _association_de_sparse(self, d_x, lists, dy, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, cov, device_out, resident):
		"""single=0 de through de_sparse.run: same results dictionary as the dense path of _association_single0."""
		from . import de_sparse
		nx, n = d_x.shape
		ny, nc = dy.shape[0], dc.shape[0]
		dof = n - 1 - rank - dimreduce
		stat_kind = 0 if return_dot else 1
		with self.torch.cuda.device(self.device):
			d_c, d_dci = self.covariates(dc, dci) if cov is None else cov
			flags = self.new_flags()
			dot, rx, ssy, coefy = de_sparse.run(self, d_x, lists, dy, d_c, d_dci, rank, nx, ny, n, nc, want_alpha, flags)
			p, stat, r, t, flags = self.sweep(dot, rx.ss, ssy, nx, ny, n, dof, False, stat_kind, out_dtype, want_rt, flags=flags)
			alpha = None
			if want_alpha:
				alpha = self.download(self.alpha(stat, stat_kind, rx.ss, n, rx.coef, coefy, nc)) if nc > 0 else np.zeros((nx, ny, nc), dtype=out_dtype)
			self.last_guard = dict(hits=0, worst=0.0, fallback=False)  # (fp64 sums of a few hundred terms: nothing to certify)
			if resident and not (want_alpha or want_rt):
				return dict(p=p, stat=stat, alpha=alpha, ssx=rx.ss, ssy=ssy, flags=flags, dof=dof)
			redo = self._sparse_flagged_rows(flags, ssy, rx.yraw, ny)
			if redo is None:
				self.check_flags(flags)  # (raises GuardHit for what the rows below do not cover: the whole call is then redone on the dense path)
			keep = (lambda v: v) if device_out else self.download
			vary = self.variances(ssy, ny, n, out_dtype)
			r_h, t_h = (self.download(r), self.download(t)) if want_rt else (None, None)
			if redo is not None:
				# A few expression rows too close to the span of the covariates for the differences the kernel takes (csrc/nrm_de_sparse.hip): only THOSE
				# rows are redone on K1's two sweeps and the fp64 Gram kernel and their columns replaced -- one high-mean, low-variance gene in
				# 15 000 costs a 1000 x 1 problem, not the whole call on the dense path
				import logging
				logging.info('normalisr_amd: %d of %d expression rows too close to the span of the covariates for the sparse-design kernel: those rows redone on the fp64 matrix cores.',
							 redo.numel(), ny)
				rows_h = redo.cpu().numpy()
				sub = dy[rows_h] if isinstance(dy, np.ndarray) else dy.index_select(0, redo)
				with self.forced_f64():
					part = self._association_single0(d_x, sub, dc, dci, rank, dimreduce, return_dot, want_alpha, out_dtype, want_rt, (d_c, d_dci), True, False)
				as_dev = lambda v: v if hasattr(v, 'is_cuda') else self.torch.from_numpy(np.ascontiguousarray(v)).to(self.device)  # (the streaming de kernel answers in numpy)
				p[:, redo] = as_dev(part['p'])
				stat[:, redo] = as_dev(part['stat'])
				vary[rows_h] = part['vary']
				if alpha is not None:
					alpha[:, rows_h] = part['alpha']
				if want_rt:
					r_h[:, rows_h], t_h[:, rows_h] = part['r'], part['t']
				self.last_guard = dict(hits=int(redo.numel()), worst=0.0, fallback=True)
			res = dict(p=keep(p), stat=keep(stat), alpha=alpha, varx=self.variances(rx.ss, nx, n, out_dtype), vary=vary, dof=dof)
			if want_rt:
				res['r'], res['t'] = r_h, t_h
			return res

	def _sparse_flagged_rows(self, flags, ssy, yraw, ny):
		"""After a sparse-design call: the expression rows to redo on the dense path (device indices), or None -- nothing flagged, or a verdict the rows
		do not explain (a design row flagged, the reference's assertions hit, more than an eighth of the rows): check_flags then speaks for the whole call."""
		from . import de_sparse
		f = flags.cpu().numpy()
		if f[2] <= 0 or f[0] or f[1]:
			return None
		rows = de_sparse.flagged_rows(self, ssy, yraw, ny)
		if rows.numel() != int(f[2]) or rows.numel() > max(1, ny // 8):
			return None
		return rows


_engines = {}
_selected = threading.local()


class use_device:
	"""Context manager: every entry point called inside runs on GPU `device` (an index) instead of the current one.
	association_tests / coex / de / binnet / normvar take it as their `device=` keyword; NORMALISR_DEVICE sets the process
	default.  (The reference has no devices; SURVEY section 5 lists this as the one build-only option.)"""

	def __init__(self, device):
		self.device = None if device is None else int(device)

	def __enter__(self):
		self.prev = getattr(_selected, 'device', None)
		if self.device is not None:
			_selected.device = self.device
		return self

	def __exit__(self, *exc):
		_selected.device = self.prev
		return False


def get_engine(device=None):
	torch = _torch()
	if device is None:
		device = getattr(_selected, 'device', None)
	if device is None and os.environ.get('NORMALISR_DEVICE', '') != '':
		device = int(os.environ['NORMALISR_DEVICE'])
	idx = torch.cuda.current_device() if device is None else int(device)
	if not 0 <= idx < torch.cuda.device_count():
		raise ValueError('GPU {} requested, {} visible'.format(idx, torch.cuda.device_count()))
	if idx not in _engines:
		_engines[idx] = Engine(idx)
	return _engines[idx]
