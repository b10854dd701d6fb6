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
		d = np.unique(np.asarray(dimreduce))
		if d.size != 1:
			raise NotImplementedError('Per-gene dimreduce arrays are not supported on the device path.')
		dimreduce = d[0]
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
