"""de with a sparse design matrix (a CRISPR screen's gRNA incidence): csrc/nrm_de_sparse.hip.

The reference multiplies residualised design and expression rows densely (association.py:224-235).  Because a residual is orthogonal to
the covariates, y~ . x~ = y . x - (y C^T) . b_x: a design row with few cells set needs the expression values at those cells only.  This
module decides whether a call qualifies, has the library turn the design matrix into the kernel's ELL lists (csrc/nrm_design_lists.hip),
takes the design rows' own statistics from those entries, runs the one-pass kernels on the raw expression rows, and K3 as always."""
import os

import numpy as np

from . import _lib, _opts
from ._lib import ROW_TILE

MAX_DENSITY = 1.0 / 16  # (break-even against K1 + the integer Gram engine is near one entry in ten)


def _round_up(v, m):
	return (v + m - 1) // m * m


def candidate(eng, dx, dy, dc, samexy):
	"""Cheap conditions that need no look at the data."""
	mode = os.environ.get('NRM_DE_SPARSE', '1')
	if samexy or dy is None or mode == '0' or eng._force_f64:  # (_force_f64: the call is being redone after a guard hit)
		return False
	nx, n = dx.shape
	if mode == 'force':  # (tests: small shapes through this path)
		return dc.shape[0] <= int(eng.lib.nrm_de_sparse_max_covariates())
	return nx >= 32 and dc.shape[0] <= int(eng.lib.nrm_de_sparse_max_covariates()) and dy.shape[0] >= 64 and n >= 2048 and nx * n >= (1 << 22)


class Lists:
	"""The design matrix as the kernels read it, built by the library's own kernels (csrc/nrm_design_lists.hip: two passes over the matrix, a
	counting rank per chunk, two prefix sums -- no torch / rocPRIM kernel, one small device-to-host copy for the sizes):
	  CSR   row_ptr (nx + 1), cells (int32), row_vals (fp64 or None): the entries row by row, cells ascending (design_stats, single=1's selection);
	  ELL   for every chunk of cells and every 64 positions of k_de_sparse's lanes (the design rows are dealt to the positions chunk by chunk,
	        sorted by their number of entries in the chunk), the entries of the 64 rows side by side, padded to the longest (ell=False: not built).
	max_density: the largest share of entries set for which the lists are built at all (ok = False beyond it, and for an empty design)."""

	def __init__(self, eng, d_x, ell=True, max_density=MAX_DENSITY):
		torch = eng.torch
		nx, n = d_x.shape
		if d_x.dtype not in (torch.float32, torch.float64):
			d_x = d_x.to(torch.float64)
		if d_x.stride(1) != 1:
			d_x = d_x.contiguous()
		ch = int(eng.lib.nrm_de_sparse_chunk())
		nslots = _round_up(nx, 64)
		nch = (n + ch - 1) // ch
		self.ngroups = nslots // 64
		i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=eng.device)
		code = _lib.NRM_F64 if d_x.dtype == torch.float64 else _lib.NRM_F32
		st = eng._stream()
		cnt, coff = i32(nch, nslots), i32(nch, nslots)
		info = torch.empty(8, dtype=torch.int64, device=eng.device)
		self.row_ptr = torch.empty(nx + 1, dtype=torch.int64, device=eng.device)
		self.slot2x = i32(nslots)
		sig = pos = w = base = None
		if ell:
			sig, pos, w = i32(nch, nslots), i32(nch, nslots), i32(nch * self.ngroups)
			base = torch.empty(nch * self.ngroups, dtype=torch.int64, device=eng.device)
		ptr = lambda t: 0 if t is None else t.data_ptr()
		_lib.check(eng.lib.nrm_design_count(d_x.data_ptr(), code, nx, n, d_x.stride(0), cnt.data_ptr(), nslots, info.data_ptr(), st))
		_lib.check(eng.lib.nrm_design_plan(cnt.data_ptr(), nx, n, nslots, ptr(sig), ptr(pos), ptr(w), ptr(base), self.row_ptr.data_ptr(), coff.data_ptr(),
										   self.slot2x.data_ptr(), info.data_ptr(), st))
		h = info.cpu().numpy()  # the one synchronisation: the sizes of what follows
		self.nnz, self.padded, self.bits = int(h[0]), int(h[1]), int(h[2])
		self.binary = not (self.bits & _lib.DESIGN_NOTONE)
		self.ok = 0 < self.nnz <= max_density * nx * n
		if not self.ok:
			return
		self.cells = i32(max(self.nnz, 1))
		self.row_vals = None if self.binary else torch.empty(max(self.nnz, 1), dtype=torch.float64, device=eng.device)
		self.ell = self.vals = None
		if ell:
			self.ell = torch.empty(max(self.padded, 8), dtype=torch.int16, device=eng.device)
			self.vals = None if self.binary else torch.empty(max(self.padded, 8), dtype=torch.float64, device=eng.device)
		_lib.check(eng.lib.nrm_design_fill(d_x.data_ptr(), code, nx, n, d_x.stride(0), nslots, ptr(pos), ptr(w), ptr(base), self.row_ptr.data_ptr(), coff.data_ptr(),
										   ptr(self.ell), ptr(self.vals), self.cells.data_ptr(), ptr(self.row_vals), 1 if self.binary else 0, st))
		self.sig, self.base, self.w = sig, base, w


def lists_for(eng, d_x):
	"""Lists(eng, d_x), kept for the next call on the SAME device tensor as long as it has not been written to (torch counts in-place
	writes in ._version; a weak reference: the cache keeps no design matrix alive, and a new tensor at an old address is not the same
	object) -- a resident screen analysed again and again pays for its lists once."""
	import weakref
	hit = getattr(eng, '_sparse_lists', None)
	if hit is not None and hit[0]() is d_x and hit[1] == d_x._version:
		return hit[2]
	lists = Lists(eng, d_x)
	eng._sparse_lists = (weakref.ref(d_x), d_x._version, lists)
	return lists


class DesignRows:
	"""What the sweep and alpha need of the design rows: their sums of squares and coefficients (the fields of engine.Residualized they read)."""

	def __init__(self, ss, coef):
		self.ss, self.coef = ss, coef


def design_stats(eng, lists, d_c, d_dci, rank, nx, nc, flags=None):
	"""|x~_i|^2 and b_i of every design row from its entries (csrc/nrm_de_sparse.hip: k_design_stats) -- K1 would sweep n cells twice
	for rows that have a few hundred entries.  flags: the call's device counters; [2] counts design rows too close to the span of the
	covariates for that difference (the caller redoes such a call on the dense path)."""
	from . import engine as _engine
	torch = eng.torch
	ncu = nc if (rank > 0 and nc > 0) else 0
	ss = eng.zeros((_round_up(nx, ROW_TILE), ), torch.float64)
	coef = eng.zeros((nx, nc), torch.float64) if nc else eng.zeros((nx, 1), torch.float64)[:, :0]  # (covariates of rank 0: the coefficients stay zero)
	with _engine._Span(eng, 'design_stats'):
		_lib.check(eng.lib.nrm_design_stats(lists.row_ptr.data_ptr(), lists.cells.data_ptr(), 0 if lists.row_vals is None else lists.row_vals.data_ptr(),
											d_c.data_ptr() if ncu else 0, d_c.stride(0) if ncu else 0, ncu, d_dci.data_ptr() if ncu else 0, nx, ss.data_ptr(),
											coef.data_ptr() if ncu else 0, 0 if flags is None else flags.data_ptr(), eng._stream()))
	return DesignRows(ss, coef)


def run(eng, d_x, lists, dy, d_c, d_dci, rank, nx, ny, n, nc, want_coef, flags=None):
	"""The design rows' statistics from their entries, the one-pass kernels on the expression rows.
	Returns (dot (nx_pad, ny_pad) fp64 with dot[i, y] = x~_i . y~_y, rx, ssy, coefy); rx.yraw = the raw rows' |y|^2."""
	rx = design_stats(eng, lists, d_c, d_dci, rank, nx, nc, flags)
	dot, ssy, coefy, common = products(eng, lists, dy, d_c, d_dci, rank, rx.coef, nx, ny, n, nc, want_coef, False, flags)
	rx.yraw = common[-1]  # |y|^2 of the raw rows (flagged_rows)
	return dot, rx, ssy, coefy


def flagged_rows(eng, ssy, yraw, ny):
	"""Indices (device, int64) of the expression rows k_de_sparse counted into flags[2]: |y~|^2 < 1e-4 |y|^2 (the kernel's own test, on what it
	stored).  Only looked at after the counter came back non-zero -- the exceptional path, a few torch launches."""
	torch = eng.torch
	s, q = ssy[:ny], yraw[:ny]
	return torch.nonzero(~(s >= 1e-4 * q) & (q > 0)).reshape(-1)


def products(eng, lists, dy, d_c, d_dci, rank, bx, nx, ny, n, nc, want_coef, by_gene, flags=None):
	"""x~_i . y~_y for every design row and expression row from the RAW expression rows (csrc/nrm_de_sparse.hip), |y~|^2 and, on request,
	the expression rows' coefficients b_y.  by_gene: the products as (ny_pad, nx_pad) (single=4 reads them so), else (nx_pad, ny_pad).
	flags: the call's device counters (engine.new_flags); [2] counts rows too close to the span of the covariates for these differences."""
	from . import engine as _engine
	torch = eng.torch
	active = rank > 0 and nc > 0
	d_y = dy if not isinstance(dy, np.ndarray) else eng.upload(_engine.as_input(dy))
	nxp, nyp = _round_up(nx, ROW_TILE), _round_up(ny, ROW_TILE)
	dot = eng.zeros((nyp, nxp), torch.float64) if by_gene else torch.empty((nxp, nyp), dtype=torch.float64, device=eng.device)
	ssy = torch.empty((nyp, ), dtype=torch.float64, device=eng.device)
	ncu = nc if active else 0  # (covariates of rank 0 -- all zero -- leave the rows as they are: association.py:899-903)
	coefy = eng.zeros((ny, nc), torch.float64) if want_coef else None
	ycode = _lib.NRM_F64 if d_y.dtype == torch.float64 else _lib.NRM_F32
	common = torch.empty((ncu + 1, ny), dtype=torch.float64, device=eng.device)
	# The rows' products with the covariates and their sums of squares: inside the gather kernel, on the fp64 matrix cores, from the chunk of rows
	# it holds in LDS anyway -- one pass over the expression matrix -- for up to 8 covariates besides a constant one (csrc/nrm_de_sparse.hip);
	# beyond that (or NRM_DE_SPARSE_SUMS=stream) the stream kernel of single=1 takes them first, every cell "common": a second pass.
	ci, cval = eng.constant_row(d_c) if ncu else (-1, 0.0)
	fused = ncu - (1 if ci >= 0 else 0) <= int(eng.lib.nrm_de_sparse_fused_covariates()) and _opts.debug('de_sparse_sums', 'inside') != 'stream'
	ct = None
	if fused:
		ct = torch.empty((int(eng.lib.nrm_de_sparse_ct_doubles(n, ncu, ci)), ), dtype=torch.float64, device=eng.device)
	else:
		code = getattr(eng, '_all_common', None)
		if code is None or code.numel() < n:
			code = eng._all_common = torch.empty((n, ), dtype=torch.int32, device=eng.device)
			_lib.check(eng.lib.nrm_fill_i32(code.data_ptr(), _lib.NRM_S1_COMMON, n, eng._stream()))
		with _engine._Span(eng, 'row_sums'):
			_lib.check(eng.lib.nrm_single1_stream(d_y.data_ptr(), ycode, d_y.stride(0), d_c.data_ptr() if ncu else 0, d_c.stride(0) if ncu else n, ncu, code.data_ptr(), n, ny,
												  common.data_ptr(), common.data_ptr(), _round_up(ny, 8), eng._stream()))  # (no cell keeps its values: the last buffer is not written)
	with _engine._Span(eng, 'de_sparse'):
		_lib.check(eng.lib.nrm_de_sparse(d_y.data_ptr(), ycode, ny, n, d_y.stride(0), common.data_ptr(), ncu, d_dci.data_ptr() if ncu else 0, lists.ell.data_ptr(),
										 0 if lists.vals is None else lists.vals.data_ptr(), lists.base.data_ptr(), lists.w.data_ptr(), lists.sig.data_ptr(), lists.ngroups,
										 lists.slot2x.data_ptr(), bx.data_ptr() if ncu else 0, max(nc, 1), dot.data_ptr(), dot.stride(0), 1 if by_gene else 0, ssy.data_ptr(),
										 coefy.data_ptr() if (coefy is not None and ncu) else 0, 0 if flags is None else flags.data_ptr(),
										 d_c.data_ptr() if ncu else 0, d_c.stride(0) if ncu else n, ci, float(cval), 0 if ct is None else ct.data_ptr(), eng._stream()))
	return dot, ssy, coefy, common


assert __name__ != "__main__"
