"""single=1: each grouping is tested only on cells that carry no OTHER grouping (low-MOI CRISPR screens).

Reference: association.py:911-925 builds sselectx[i,k] = (dx[i,k] == sum_j dx[j,k]) and association_test_2
(:263-390) loops over groupings: subset the cells, pseudo-invert the subset's covariate Gram matrix, residualise
x_i and every gene on the subset, R^2, p with dof_i = ns_i - 1 - rank_i - dimreduce.

Device formulation: all per-(grouping, gene) quantities are bilinear in the gene's expression row, so the whole
loop collapses into two Gram contractions on the fp64 matrix cores (K2),

    G  = Y  W^T,   W_i = [1_Si * C (nc rows); 1_Si * x_i]      and      G2 = (Y*Y) S^T,  S_i = 1_Si,

a tiny host step per grouping (inv_rank of C_S C_S^T: the rank is an integer and stays on the host) and one
sweep kernel (csrc/nrm_single1.hip).  Groupings are processed in chunks to bound the size of W.
"""
import logging
import os

import numpy as np

from . import _lib, _opts
from . import engine as _engine
from ._lib import ROW_TILE
from .association import inv_rank, small_pinv


def _round_up(v, m):
	return (v + m - 1) // m * m


def _is_dev(a):
	return hasattr(a, 'is_cuda') and a.is_cuda


def association_tests_single1(dx, dy, dc, lowmem=True, return_dot=True, return_stats=False, dimreduce=0, chunk=256, device_out=False, **ka):
	"""Device path of association_tests(..., single=1); returns (p, gamma|dot, alpha|None, varx (n_x,), vary (n_x,n_y)).
	dx / dy may be torch CUDA tensors already in HBM (a resident screen: bench.py); device_out=True leaves the large results there."""
	if ka:
		raise TypeError("association_test_2() got an unexpected keyword argument '{}'".format(next(iter(ka))))
	if dy is None:
		raise NotImplementedError('dy=None with single=1')  # association.py:912
	if return_stats:
		raise NotImplementedError('return_stats is only available for single=0.')
	if np.ndim(dimreduce) != 0:
		# One value per row of dy (the reference subtracts it from the degrees of freedom of every (grouping, gene) pair, association.py:374 -- its own
		# broadcast takes a (n_y, 1) column): gamma and the variances do not depend on it, so the call is repeated per DISTINCT value and every gene keeps
		# the P-values of its own.
		per_gene = np.asarray(dimreduce).reshape(-1)
		if per_gene.size != np.shape(dy)[0]:
			raise ValueError('dimreduce must be an integer or have one entry per row of dy.')
		if (per_gene != per_gene.astype(np.int64)).any():
			raise ValueError('dimreduce must be an integer.')
		vals = np.unique(per_gene.astype(np.int64))
		if vals.size > 1:
			if device_out:
				raise NotImplementedError('device_out with one dimreduce per gene')
			out = None
			for v in vals:
				r = association_tests_single1(dx, dy, dc, lowmem=lowmem, return_dot=return_dot, dimreduce=int(v), chunk=chunk)
				if out is None:
					out = [None if a is None else np.array(a) for a in r]
				else:
					cols = per_gene == v
					out[0][:, cols] = r[0][:, cols]
			return tuple(out)
		dimreduce = vals[0]
	dimreduce = int(dimreduce)
	dx, dy, dc = (dx if _is_dev(dx) else np.asarray(dx)), (dy if _is_dev(dy) else np.asarray(dy)), np.asarray(dc)
	nx, n = dx.shape
	ny, nc = dy.shape[0], dc.shape[0]
	if dy.shape[1] != n or dc.shape[1] != n:
		raise ValueError('Unmatching dx/dy/dc dimensions.')
	if nc == 0:
		logging.warning('No covariate dc input.')
	chunk = max(1, min(chunk, 8192 // (nc + 1)))  # bounds the masked-row operand W (chunk * (nc + 1) rows) for many covariates
	c64 = np.asarray(dc, dtype=np.float64)
	if _is_dev(dy):
		out_dtype = np.dtype(np.float32 if str(dy.dtype) == 'torch.float32' else np.float64)
	else:
		out_dtype = dy.dtype if dy.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	eng = _engine.get_engine()
	with eng.lock:  # one call at a time per device (engine scratch, streams and guard state are shared)
		torch = eng.torch
		tdt = torch.float64 if out_dtype == np.float64 else torch.float32
		nw = nc + 1
		with torch.cuda.device(eng.device):
			# cell selection on the device (association.py:914-918): the design matrix travels once, in its own dtype
			d_dx = dx if _is_dev(dx) else eng.upload(_engine.as_input(dx))
			lists = None
			if nc <= 32 and _opts.debug('single1', 'sparse') != 'dense':
				# the design's entries listed by the library (csrc/nrm_design_lists.hip: one pass counts them and says what they are like, a
				# second writes them row by row) -- unless more than a quarter of the matrix is set, which no design of this method is
				from . import de_sparse
				lists = de_sparse.Lists(eng, d_dx, ell=False, max_density=0.25)
				b = lists.bits
				assert (b & _lib.DESIGN_HAS1) and not (b & (_lib.DESIGN_GT1 | _lib.DESIGN_NAN))  # dx.max() == 1 (association.py:914)
				if lists.ok and not (b & _lib.DESIGN_NEG):
					# entries >= 0: the selection follows from the LIST of the design's entries (a cell is selected for grouping i when i is its
					# only entry, and for every grouping when it has none) -- no (groupings x cells) selection matrix, no passes over one
					if nc <= int(_S1_DEVICE_NC) and _opts.debug('single1_stats', 'device') != 'host':
						plan = Single1Plan(d_dx, dy, c64, dimreduce=dimreduce, lowmem=lowmem, return_dot=return_dot, lists=lists, eng=eng)
						plan.step()
						return plan.results(device_out=device_out)
					return _sparse(eng, lists, dy, c64, nx, ny, n, nc, dimreduce, lowmem, return_dot, out_dtype, tdt, device_out)
			else:
				assert float(torch.amax(d_dx)) == 1  # association.py:914
			sel = d_dx == torch.sum(d_dx, dim=0, dtype=torch.float64)  # association.py:915-916
			big = torch.finfo(d_dx.dtype).max
			lo = torch.where(sel, d_dx, big).amin(dim=1)
			hi = torch.where(sel, d_dx, -big).amax(dim=1)
			assert bool((hi > lo).all())  # >1 distinct value among the selected cells (:917-918)
			del lo, hi
			ns = sel.sum(dim=1).cpu().numpy().astype(np.float64)
			ry = eng.residualize(dy if _is_dev(dy) else _engine.as_input(dy), None, None, 0)  # fp64 padded copy of Y
			y2 = Residualized_sq(ry, eng)
			d_c = eng.upload(c64) if nc else None
			p = torch.empty((nx, ny), dtype=tdt, device=eng.device)
			stat = torch.empty((nx, ny), dtype=tdt, device=eng.device)
			vary = torch.empty((nx, ny), dtype=tdt, device=eng.device)
			alpha = None if lowmem else torch.zeros((nx, ny, nc), dtype=tdt, device=eng.device)
			flags = torch.zeros(2, dtype=torch.int32, device=eng.device)
			varx = np.empty(nx)
			pitch = 26 + nc + nc * nc
			kp = ry.k_pad
			if nc:
				cp = eng.zeros((_round_up(nc, ROW_TILE), kp), torch.float64)
				eng.copy_rows(cp, d_c)
				cpad = _engine.Residualized(nc, n, cp, None, None)
			for i0 in range(0, nx, chunk):
				i1 = min(nx, i0 + chunk)
				m = i1 - i0
				d_sel = sel[i0:i1].to(torch.float64)  # (m, n)
				d_x = d_dx[i0:i1].to(torch.float64)
				wrows = _round_up(m * nw, ROW_TILE)
				w = torch.zeros((wrows, kp), dtype=torch.float64, device=eng.device)
				wv = w[:m * nw].view(m, nw, kp)
				if nc:
					wv[:, :nc, :n] = d_sel[:, None, :] * d_c[None, :, :]
				wv[:, nc, :n] = d_sel * d_x
				srows = _round_up(m, ROW_TILE)
				s = torch.zeros((srows, kp), dtype=torch.float64, device=eng.device)
				s[:m, :n] = d_sel
				W = _engine.Residualized(m * nw, n, w, None, None)
				S = _engine.Residualized(m, n, s, None, None)
				g = eng.gram(ry, W, False)   # (ny_pad, wrows): y . (1_S C), y . (1_S x)
				g2 = eng.gram(y2, S, False)  # (ny_pad, srows): |y_S|^2
				# grouping-side statistics: M_i = C_S C_S^T, xC_i = C_S x_S, xx_i = |x_S|^2 (tiny; W against [C; x] rows)
				xx = (wv[:, nc, :n] * d_x).sum(dim=1).cpu().numpy()
				info = np.zeros((m, pitch))
				rk = np.zeros(m, dtype=np.int64)
				if nc:
					# W against the covariate rows on the fp64 Gram kernel (rounds 3-4: two torch.einsum calls -- rocBLAS behind torch on a product path)
					gc = eng.gram(W, cpad, False)[:m * nw, :nc].cpu().numpy().reshape(m, nw, nc)
					mc, xc = np.ascontiguousarray(gc[:, :nc, :]), np.ascontiguousarray(gc[:, nc, :])  # (m, nc, nc), (m, nc)
					mi, rk = small_pinv(mc)  # association.py:350-351, all groupings of the chunk
					mi[rk == 0] = 0
					ccx = np.einsum('icd,id->ic', mi, xc)
					info[:, 26:26 + nc] = ccx
					info[:, 26 + nc:] = mi.reshape(m, nc * nc)
					xx = xx - np.einsum('ic,ic->i', xc, ccx)
				vxx = xx / ns[i0:i1]
				vxx[vxx == 0] = 1  # association.py:362-364
				varx[i0:i1] = vxx
				dof = ns[i0:i1] - 1 - rk - dimreduce
				if (dof <= 0).any():
					raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
				info[:, 0], info[:, 1] = ns[i0:i1], vxx
				dof = np.ascontiguousarray(dof, dtype=np.float64)
				_lib.check(eng.lib.nrm_pvalue_plan_init_many(dof.ctypes.data, m, info.ctypes.data + 16, pitch))
				d_info = eng.upload(info)
				code = _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32
				_lib.check(eng.lib.nrm_single1_sweep(g.data_ptr(), g.stride(0), g2.data_ptr(), g2.stride(0), d_info.data_ptr(), pitch, nc, m, ny,
													 1 if return_dot else 0, p[i0:i1].data_ptr(), stat[i0:i1].data_ptr(), vary[i0:i1].data_ptr(),
													 0 if alpha is None else alpha[i0:i1].data_ptr(), code, ny, flags.data_ptr(), eng._stream()))
			eng.check_flags(flags)
			return (eng.download(p), eng.download(stat), None if alpha is None else eng.download(alpha), varx.astype(out_dtype),
					eng.download(vary))


_S1_DEVICE_NC = 8  # covariates up to which the groupings' statistics are finished on the device (csrc/nrm_single1.hip: S1_GS_NC)


def _lists_for(eng, d_dx):
	"""The design's CSR lists, kept for the next call on the SAME device tensor as long as it has not been written to (as de_sparse.lists_for
	does for single=0 / single=4: a resident screen pays for its lists once; bench.py reports a cold call beside the resident step)."""
	import weakref
	from . import de_sparse
	hit = getattr(eng, '_s1_lists', None)
	if hit is not None and hit[0]() is d_dx and hit[1] == d_dx._version:
		return hit[2]
	lists = de_sparse.Lists(eng, d_dx, ell=False, max_density=0.25)
	eng._s1_lists = (weakref.ref(d_dx), d_dx._version, lists)
	return lists


class Single1Plan:
	"""single=1 for a design with entries >= 0 and at most 8 covariates, with NOTHING of a step on the host (round 6; rounds 3-5 finished the
	groupings' statistics there: five read-backs, 1000 pseudo-inverses, 1000 P-value plans and an upload per call, and a step took 2 ms on one
	box and 18 on another).  A step is seven launches on the engine's stream --

	    nrm_single1_select       the cell selection from the design's entry lists (association.py:914-918)
	    nrm_single1_group_stats  each grouping's sums over its own cells (a wave per grouping)
	    nrm_single1_stream       the expression matrix read ONCE: sums over the shared cells, the values at the groupings' own cells
	    nrm_single1_group_info   per grouping: pseudo-inverse + integer rank (Jacobi, the rule of inv_rank), ccx, vx, dof, the P-value plan (:350-374)
	    nrm_single1_cells        the sweep

	-- into buffers the plan owns (no allocation inside a step), replayed as ONE HIP graph from the third step on; what the reference asserts or
	raises on the way (:917-918, dof <= 0, non-finite results, R^2 > 1) is counted on the device and looked at once, in results().
	dx / dy: torch CUDA tensors resident in HBM (numpy arrays are uploaded once); dc: host array."""

	def __init__(self, dx, dy, dc, dimreduce=0, lowmem=True, return_dot=True, lists=None, eng=None):
		eng = self.eng = eng or _engine.get_engine()
		self._c64 = np.ascontiguousarray(np.asarray(dc, dtype=np.float64))
		self.dimreduce, self.lowmem, self.return_dot = int(dimreduce), lowmem, return_dot
		with eng.lock, eng.torch.cuda.device(eng.device):
			self.d_dx = dx if _is_dev(dx) else eng.upload(_engine.as_input(np.asarray(dx)))
			self.d_y = dy if _is_dev(dy) else eng.upload(_engine.as_input(np.asarray(dy)))
		self._build(lists)

	def _build(self, lists=None):
		"""Everything that depends on the design tensor as it is NOW: its entry lists, the size of the stream kernel's transposed output, the step's buffers,
		a fresh graph.  Run at construction, and again by step() when the design tensor has been written to in place since (torch counts that in ._version)."""
		eng = self.eng
		torch = eng.torch
		dev = eng.device
		c64 = self._c64
		with eng.lock, torch.cuda.device(dev):
			nx, n = self.d_dx.shape
			ny, nc = self.d_y.shape[0], c64.shape[0]
			if self.d_y.shape[1] != n or c64.shape[1] != n:
				raise ValueError('Unmatching dx/dy/dc dimensions.')
			if nc > _S1_DEVICE_NC:
				raise NotImplementedError('Single1Plan: at most {} covariates (association_tests_single1 takes more)'.format(_S1_DEVICE_NC))
			self.nx, self.ny, self.n, self.nc = nx, ny, n, nc
			lowmem = self.lowmem
			self._dx_version = self.d_dx._version
			self.out_dtype = np.dtype(np.float32 if self.d_y.dtype == torch.float32 else np.float64)
			tdt = torch.float64 if self.out_dtype == np.float64 else torch.float32
			lists = self.lists = lists if lists is not None else _lists_for(eng, self.d_dx)
			b = lists.bits
			assert (b & _lib.DESIGN_HAS1) and not (b & (_lib.DESIGN_GT1 | _lib.DESIGN_NAN))  # dx.max() == 1 (association.py:914)
			if not lists.ok or (b & _lib.DESIGN_NEG):
				raise NotImplementedError('Single1Plan: a design with entries >= 0 of which at most a quarter are set (association_tests_single1 takes the others)')
			nnz = self.nnz = lists.nnz
			f64 = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
			self.d_c = eng.upload(c64) if nc else None
			self.cnt = torch.empty(n, dtype=torch.int32, device=dev)
			self.code = torch.empty(n, dtype=torch.int32, device=dev)
			self.seg = torch.empty(nx + 1, dtype=torch.int64, device=dev)
			self.idx = torch.empty(nnz, dtype=torch.int64, device=dev)
			self.xe = f64(nnz)
			self.ce = f64(nnz, nc) if nc else None
			self.rowinfo = f64(nx, 3)
			self.sel_info = torch.empty(8, dtype=torch.int64, device=dev)
			gb = int(eng.lib.nrm_single1_select_gram_blocks())
			self.gpart = f64(1, gb, 64) if nc else None
			self.gs = f64(nx, nc * (nc + 1) // 2 + nc + 1)
			self.ldye = _round_up(ny, 8)  # (a multiple of a 128-byte line instead: measured, no difference -- 1.74 ms either way on one box)
			# One row of YE per cell that carries exactly one grouping: the selection is run once here and that count read back (the only read-back of the
			# plan, at construction: it depends on the design alone).  By the design's entry count instead -- no read-back at all -- the buffer of a design
			# with a quarter of its entries set would be tens of GB.
			ptr0 = lambda t: 0 if t is None else t.data_ptr()
			_lib.check(eng.lib.nrm_single1_select(lists.row_ptr.data_ptr(), lists.cells.data_ptr(), ptr0(lists.row_vals), nx, n, nnz, ptr0(self.d_c), n, nc, self.cnt.data_ptr(),
												  self.code.data_ptr(), self.seg.data_ptr(), self.idx.data_ptr(), self.xe.data_ptr(), ptr0(self.ce), self.rowinfo.data_ptr(),
												  0, self.sel_info.data_ptr(), eng._stream()))
			self.n_kept = int(self.sel_info.cpu()[4])
			self.ye = torch.empty((max(self.n_kept, 1), self.ldye), dtype=self.d_y.dtype, device=dev)
			self.common = f64(nc + 1, ny)
			self.pitch = 26 + nc + nc * nc
			self.info = f64(nx, self.pitch)
			self.varx = f64(nx)
			self.p, self.stat, self.vary = (torch.empty((nx, ny), dtype=tdt, device=dev) for _ in range(3))
			self.alpha = None if lowmem else eng.zeros((nx, ny, nc), tdt)
			self.flags = eng.zeros((8, ), torch.int32)
			self._side = torch.cuda.Stream(device=dev)
		from .distributed import StepGraph
		self._graph = StepGraph(torch)

	def _launch(self):
		eng, lib, L = self.eng, self.eng.lib, self.lists
		nx, ny, n, nc = self.nx, self.ny, self.n, self.nc
		ptr = lambda t: 0 if t is None else t.data_ptr()
		st = eng._stream()
		d_y = self.d_y
		ycode = _lib.NRM_F64 if d_y.dtype == eng.torch.float64 else _lib.NRM_F32
		torch = eng.torch
		with _engine._Span(eng, 's1_select'):
			_lib.check(lib.nrm_single1_select(L.row_ptr.data_ptr(), L.cells.data_ptr(), ptr(L.row_vals), nx, n, self.nnz, ptr(self.d_c), n, nc, self.cnt.data_ptr(),
											  self.code.data_ptr(), self.seg.data_ptr(), self.idx.data_ptr(), self.xe.data_ptr(), ptr(self.ce), self.rowinfo.data_ptr(),
											  0, self.sel_info.data_ptr(), st))  # (the shared cells' Gram matrix: on the second stream, below)
		# The groupings' own statistics (a wave, then a LANE per grouping: 1000 lanes of Jacobi rotations and P-value plans, 70 us on 16 CUs) need the
		# selection only, and the stream kernel needs nothing of them: they run beside it on a second stream (a fork and a join of the captured graph)
		# instead of in front of it.
		main = torch.cuda.current_stream(eng.device)
		forked, joined = torch.cuda.Event(), torch.cuda.Event()
		forked.record(main)
		with torch.cuda.stream(self._side):
			self._side.wait_event(forked)
			ss = self._side.cuda_stream
			if nc:
				_lib.check(lib.nrm_single1_common_gram(self.cnt.data_ptr(), n, self.d_c.data_ptr(), n, nc, self.gpart.data_ptr(), ss))
			_lib.check(lib.nrm_single1_group_stats(self.seg.data_ptr(), self.idx.data_ptr(), self.xe.data_ptr(), ptr(self.d_c), n, nc, nx, self.gs.data_ptr(), ss))
			_lib.check(lib.nrm_single1_group_info(self.gs.data_ptr(), ptr(self.gpart), self.rowinfo.data_ptr(), self.sel_info.data_ptr(), nc, nx, self.dimreduce,
												  self.info.data_ptr(), self.pitch, self.varx.data_ptr(), self.flags.data_ptr(), ss))
			joined.record(self._side)
		with _engine._Span(eng, 's1_stream'):
			_lib.check(lib.nrm_single1_stream(d_y.data_ptr(), ycode, d_y.stride(0), ptr(self.d_c), n, nc, self.code.data_ptr(), n, ny, self.common.data_ptr(),
											  self.ye.data_ptr(), self.ldye, st))
		main.wait_event(joined)
		code_o = _lib.NRM_F64 if self.out_dtype == np.float64 else _lib.NRM_F32
		with _engine._Span(eng, 's1_cells'):
			_lib.check(lib.nrm_single1_cells(self.ye.data_ptr(), ycode, self.ldye, ptr(self.ce), self.xe.data_ptr(), self.seg.data_ptr(), self.common.data_ptr(),
											 self.info.data_ptr(), self.pitch, nc, nx, ny, 1 if self.return_dot else 0, self.p.data_ptr(), self.stat.data_ptr(),
											 self.vary.data_ptr(), ptr(self.alpha), code_o, ny, self.flags.data_ptr(), st))

	def step(self, timed=False):
		"""One pass over the screen; results stay in HBM (results() takes them)."""
		eng = self.eng
		with eng.lock, eng.torch.cuda.device(eng.device):
			if self.d_dx._version != self._dx_version:  # the design written to in place: its lists, the buffer sizes and the captured graph belong to the old values
				self._build()
			if eng.trace is not None:  # (bench.py's per-kernel split: events between the launches, not capturable)
				self._launch()
			else:
				self._graph.run(self._launch)

	def cells_kept(self):
		"""Cells that carry exactly one grouping (the rows of YE the stream kernel writes)."""
		return self.n_kept

	def check(self):
		"""The reference's assertions and errors, from the counters of the steps since the last look."""
		f = self.flags.cpu().numpy()
		self.flags.zero_()
		if f[2]:
			raise AssertionError('{} groupings take a single value on the cells selected for them (association.py:917-918)'.format(int(f[2])))
		if f[4]:
			raise ValueError('array must not contain infs or NaNs')
		if f[3]:
			raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
		if f[0] or f[1]:
			raise AssertionError('association results failed the reference assertions (association.py:248,252): '
								 '{} tiles with non-finite values, {} tiles with R^2 > 1+1e-8'.format(int(f[0]), int(f[1])))

	def results(self, device_out=False):
		"""(p, gamma|dot, alpha|None, varx (n_x,), vary (n_x, n_y)) of the last step -- numpy arrays, or with device_out=True the plan's own
		device tensors (overwritten by the next step)."""
		eng = self.eng
		with eng.lock, eng.torch.cuda.device(eng.device):
			self.check()
			if device_out:
				return (self.p, self.stat, self.alpha, self.varx.to(self.p.dtype), self.vary)
			return (eng.download(self.p), eng.download(self.stat), None if self.alpha is None else eng.download(self.alpha),
					self.varx.cpu().numpy().astype(self.out_dtype), eng.download(self.vary))


def _segment_sums(v, starts, counts):
	"""Sums of v (..., cells) over consecutive segments of the last axis; empty segments give 0 (np.add.reduceat would not)."""
	out = np.zeros(v.shape[:-1] + (len(counts), ))
	nz = counts > 0
	if nz.any():
		out[..., nz] = np.add.reduceat(v, starts[nz], axis=-1)
	return out


def _sparse(eng, lists, dy, c64, nx, ny, n, nc, dimreduce, lowmem, return_dot, out_dtype, tdt, device_out=False):
	"""single=1 for a design with entries >= 0 (csrc/nrm_single1.hip, second half): the cells every grouping shares (all of dx is 0)
	are summed once per gene, each grouping adds its own few cells inside the sweep; no masked Gram contraction, no loop over chunks
	of groupings, no transposed copy of the expression matrix (the stream kernel reads it once, where it lies).  The cell selection
	(association.py:914-918) comes from the design's entry lists by kernels of the library (nrm_single1_select); the statistics of the
	groupings themselves (M_i = C_S C_S^T, C_S x_S, |x_S|^2: association.py:350-364) are finished on the host WHILE the stream kernel runs."""
	torch = eng.torch
	from .single4 import _Marks
	mark = _Marks(eng, 's1_trace', 'single=1')
	mark('entry lists')
	dev = eng.device
	nnz = lists.nnz
	d_c = eng.upload(c64) if nc else None
	cnt = torch.empty(n, dtype=torch.int32, device=dev)
	code = torch.empty(n, dtype=torch.int32, device=dev)
	d_seg = torch.empty(nx + 1, dtype=torch.int64, device=dev)
	idx_e = torch.empty(nnz, dtype=torch.int64, device=dev)
	xe_d = torch.empty(nnz, dtype=torch.float64, device=dev)
	d_ce = torch.empty((nnz, nc), dtype=torch.float64, device=dev) if nc else None
	rowinfo = torch.empty((nx, 3), dtype=torch.float64, device=dev)
	info = torch.empty(8, dtype=torch.int64, device=dev)
	gb = int(eng.lib.nrm_single1_select_gram_blocks())
	nb = (nc + 7) // 8
	gpart = torch.empty((nb * (nb + 1) // 2, gb, 64), dtype=torch.float64, device=dev) if nc else None
	ptr = lambda t: 0 if t is None else t.data_ptr()
	_lib.check(eng.lib.nrm_single1_select(lists.row_ptr.data_ptr(), lists.cells.data_ptr(), ptr(lists.row_vals), nx, n, nnz, ptr(d_c), n, nc, cnt.data_ptr(),
										  code.data_ptr(), d_seg.data_ptr(), idx_e.data_ptr(), xe_d.data_ptr(), ptr(d_ce), rowinfo.data_ptr(), ptr(gpart), info.data_ptr(),
										  eng._stream()))
	# The groupings' own sums by a wave each (k_s1_group_stats, <= 8 covariates) are queued at once, and so is the stream kernel when its output can be
	# sized without the host (at most one row of YE per design entry; beyond 4 GB of that the kept count is waited for): what the host needs of the
	# selection then comes down on the copy stream BESIDE the stream kernel -- downloaded in front of it (round 4) the GPU sat idle for three read-backs.
	on_device = nc <= 8
	gs_d = None
	if on_device:
		npair = nc * (nc + 1) // 2
		gs_d = torch.empty((nx, npair + nc + 1), dtype=torch.float64, device=dev)
		_lib.check(eng.lib.nrm_single1_group_stats(d_seg.data_ptr(), idx_e.data_ptr(), xe_d.data_ptr(), ptr(d_c), n, nc, nx, gs_d.data_ptr(), eng._stream()))
	d_y = dy if _is_dev(dy) else eng.upload(_engine.as_input(dy))
	ldye = _round_up(ny, 8)
	ycode = _lib.NRM_F64 if d_y.dtype == torch.float64 else _lib.NRM_F32
	common = torch.empty((nc + 1, ny), dtype=torch.float64, device=eng.device)
	early = on_device and nnz * ldye * d_y.element_size() <= (4 << 30)

	def stream_kernel(rows_ye):
		ye_ = torch.empty((max(rows_ye, 1), ldye), dtype=d_y.dtype, device=eng.device)
		with _engine._Span(eng, 's1_stream'):
			_lib.check(eng.lib.nrm_single1_stream(d_y.data_ptr(), ycode, d_y.stride(0), 0 if d_c is None else d_c.data_ptr(), n, nc, code.data_ptr(), n, ny,
												  common.data_ptr(), ye_.data_ptr(), ldye, eng._stream()))
		return ye_
	ye = None
	if early:
		main = torch.cuda.current_stream(dev)
		ready = torch.cuda.Event()
		ready.record(main)
		if eng._copy is None:
			eng._copy = torch.cuda.Stream(device=dev)
		ye = stream_kernel(nnz)
		with torch.cuda.stream(eng._copy):
			eng._copy.wait_event(ready)
			hosts = []
			for t in (info, rowinfo, gpart, gs_d):
				if t is None:
					hosts.append(None)
					continue
				h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
				h.copy_(t, non_blocking=True)
				t.record_stream(eng._copy)
				hosts.append(h)
			arrived = torch.cuda.Event()
			arrived.record(eng._copy)
		arrived.synchronize()
		h_info, h_rows = hosts[0].numpy(), hosts[1].numpy()
		hp_early, gs_early = (None if hosts[2] is None else hosts[2].numpy()), hosts[3].numpy()
	else:
		# what the host needs of the cell order (before the stream kernel is queued: a download behind it would wait for it)
		h_info, h_rows = info.cpu().numpy(), rowinfo.cpu().numpy()
	n_common, n_e = int(h_info[3]), int(h_info[4])
	ns = n_common + h_rows[:, 0]
	# > 1 distinct value among a grouping's selected cells (:917-918): 0 on the shared cells, if there are any, and its values on its own
	vlo, vhi = h_rows[:, 1], h_rows[:, 2]
	if n_common > 0:
		vlo, vhi = np.minimum(vlo, 0.0), np.maximum(vhi, 0.0)
	assert bool((vhi > vlo).all())
	mark('cell order')
	if nc:  # covariate Gram of the shared cells: the kernel's partial sums added up in a fixed order (no BLAS, on either side)
		hp = hp_early if early else gpart.cpu().numpy()
		mcc = np.zeros((nb * 8, nb * 8))
		q = 0
		for bi in range(nb):
			for bj in range(bi, nb):
				blk = np.zeros(64)
				for g in range(gb):
					blk += hp[q, g]
				mcc[bi * 8:bi * 8 + 8, bj * 8:bj * 8 + 8] = blk.reshape(8, 8)
				mcc[bj * 8:bj * 8 + 8, bi * 8:bi * 8 + 8] = blk.reshape(8, 8).T
				q += 1
		mcc = np.ascontiguousarray(mcc[:nc, :nc])
	if on_device:  # (more covariates: numpy segment sums on the host)
		gs = gs_early if early else gs_d.cpu().numpy()
	else:
		counts = h_rows[:, 0].astype(np.int64)
		seg = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
		idx_h = idx_e.cpu().numpy()[:n_e]
		xe = xe_d.cpu().numpy()[:n_e]
	# the device's share: it needs nothing of the host's statistics and runs while they are taken
	if ye is None:
		ye = stream_kernel(n_e)
	eng.s1_cells_kept = n_e  # (bench.py: the bytes the stream kernel writes)
	# grouping-side statistics on the host
	pitch = 26 + nc + nc * nc
	info = np.zeros((nx, pitch))
	rk = np.zeros(nx, dtype=np.int64)
	mark('downloads, stream kernel')
	if on_device:
		xx = gs[:, -1].copy()
		if nc:
			iu = np.triu_indices(nc)
			mc = np.empty((nx, nc, nc))
			mc[:, iu[0], iu[1]] = gs[:, :npair]
			mc[:, iu[1], iu[0]] = gs[:, :npair]
			mc += mcc[None]
			xc = gs[:, npair:npair + nc]
	else:
		starts = seg[:-1]
		xx = _segment_sums(xe * xe, starts, counts)
		ce = c64[:, idx_h]  # (nc, cells of the E_i in the order of the groupings)
		mc = mcc[None] + np.moveaxis(_segment_sums(ce[:, None, :] * ce[None, :, :], starts, counts), -1, 0)
		xc = _segment_sums(ce * xe, starts, counts).T  # (nx, nc)
	if nc:
		mark('host sums')
		mi, rk = small_pinv(mc)  # association.py:350-351
		mi[rk == 0] = 0
		mark('inv_rank')
		ccx = np.einsum('icd,id->ic', mi, xc)
		info[:, 26:26 + nc] = ccx
		info[:, 26 + nc:] = mi.reshape(nx, nc * nc)
		xx = xx - np.einsum('ic,ic->i', xc, ccx)
	vxx = xx / ns
	vxx[vxx == 0] = 1  # association.py:362-364
	dof = ns - 1 - rk - dimreduce
	if (dof <= 0).any():
		raise RuntimeError('Insufficient number of cells: must be greater than degrees of freedom removed + covariate + 1.')
	info[:, 0], info[:, 1] = ns, vxx
	dof = np.ascontiguousarray(dof, dtype=np.float64)
	_lib.check(eng.lib.nrm_pvalue_plan_init_many(dof.ctypes.data, nx, info.ctypes.data + 16, pitch))
	mark('p-value plans')
	p = torch.empty((nx, ny), dtype=tdt, device=eng.device)
	stat = torch.empty((nx, ny), dtype=tdt, device=eng.device)
	vary = torch.empty((nx, ny), dtype=tdt, device=eng.device)
	alpha = None if lowmem else eng.zeros((nx, ny, nc), tdt)
	flags = eng.zeros((2, ), torch.int32)
	d_info = eng.upload(info)
	code_o = _lib.NRM_F64 if out_dtype == np.float64 else _lib.NRM_F32
	with _engine._Span(eng, 's1_cells'):
		_lib.check(eng.lib.nrm_single1_cells(ye.data_ptr(), ycode, ldye, 0 if d_ce is None else d_ce.data_ptr(), xe_d.data_ptr(), d_seg.data_ptr(),
											 common.data_ptr(), d_info.data_ptr(), pitch, nc, nx, ny, 1 if return_dot else 0, p.data_ptr(), stat.data_ptr(),
											 vary.data_ptr(), 0 if alpha is None else alpha.data_ptr(), code_o, ny, flags.data_ptr(), eng._stream()))
	eng.check_flags(flags)
	mark('sweep')
	mark.report()
	if device_out:
		return (p, stat, alpha, vxx.astype(out_dtype), vary)
	return (eng.download(p), eng.download(stat), None if alpha is None else eng.download(alpha), vxx.astype(out_dtype), eng.download(vary))


def Residualized_sq(ry, eng):
	"""Element-wise square of the padded fp64 expression matrix (operand of the |y_S|^2 contraction)."""
	return _engine.Residualized(ry.rows, ry.n, ry.data * ry.data, None, None)
