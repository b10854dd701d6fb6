"""de with a sparse design matrix (a CRISPR screen's gRNA incidence): csrc/nrm_de_sparse.hip.

The reference multiplies residualised design and expression rows densely (association.py:224-235).  Because a residual is orthogonal to
the covariates, y~ . x~ = y . x - (y C^T) . b_x: a design row with few cells set needs the expression values at those cells only.  This
module decides whether a call qualifies, turns the design matrix into the kernel's ELL lists (on the device, a dozen torch passes over
its non-zero entries), takes the design rows' own statistics from those entries, runs the one-pass kernels on the raw expression rows, and
K3 as always."""
import os

import numpy as np

from . import _lib
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
	"""The design matrix as the kernel reads it: for every chunk of cells and every 64 positions of the workgroup's lanes (the design rows
	are dealt to the positions chunk by chunk, sorted by their number of entries in the chunk), the entries of the 64 rows side by side,
	padded to the longest."""

	def __init__(self, eng, d_x):
		torch = eng.torch
		nx, n = d_x.shape
		ch = int(eng.lib.nrm_de_sparse_chunk())
		self.nnz = int(torch.count_nonzero(d_x))  # (one cheap pass first: a dense design must not be listed entry by entry to find that out)
		self.ok = 0 < self.nnz <= MAX_DENSITY * nx * n
		if not self.ok:
			return
		nz = torch.nonzero(d_x)  # row-major: by design row, then by cell
		xi, k = nz[:, 0], nz[:, 1]
		vals = d_x[xi, k]
		self.binary = bool((vals == 1).all())
		# the entries row by row (nonzero() lists them so): what the design rows' own statistics are taken from (design_stats)
		self.row_ptr = torch.zeros(nx + 1, dtype=torch.int64, device=eng.device)
		self.row_ptr[1:] = torch.cumsum(torch.bincount(xi, minlength=nx), 0)
		self.cells = k.to(torch.int32).contiguous()
		self.row_vals = None if self.binary else vals.to(torch.float64).contiguous()
		nslots = _round_up(nx, 64)
		self.ngroups = nslots // 64
		self.slot2x = torch.full((nslots, ), -1, dtype=torch.int32, device=eng.device)
		self.slot2x[:nx] = torch.arange(nx, dtype=torch.int32, device=eng.device)  # (slot = design row; the dealing happens per chunk, below)
		nch = (n + ch - 1) // ch
		c = k // ch
		# In every chunk the slots are dealt anew to the positions of the workgroup's lanes, sorted by their number of entries IN that chunk
		# (inside every block of 1024 positions -- one pass of the kernel): the 64 lists a wave walks in step are then equally long, where
		# one dealing for all chunks left a third of the padded entries to the spread between a wave's lists.
		cnt_cs = torch.bincount(c * nslots + xi, minlength=nch * nslots).view(nch, nslots)
		sig = torch.empty((nch, nslots), dtype=torch.int64, device=eng.device)
		for lo in range(0, nslots, 1024):
			hi = min(nslots, lo + 1024)
			sig[:, lo:hi] = torch.argsort(cnt_cs[:, lo:hi], dim=1, descending=True, stable=True) + lo
		pos = torch.empty_like(sig)
		pos.scatter_(1, sig, torch.arange(nslots, device=eng.device).expand(nch, nslots).contiguous())  # pos[c, slot] = its position in chunk c
		self.sig = sig.to(torch.int32).contiguous()
		key = c * nslots + pos[c, xi]  # (chunk, position)
		# Inside a list the order is free.  ds_read_b128 serves a wave in four groups of 16 lanes, and two lanes of a group collide when
		# their records share a bank quad (record index mod 16) without being the same record (MI355X_MICROARCH.md, LDS): every list is
		# ordered by that residue, starting at a residue of its lane's own, so that the 16 lanes of a group walk the residues out of step.
		grp16 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
		rot = np.zeros(64, dtype=np.int64)
		for g16 in grp16:
			for pos16, lane in enumerate(g16):
				rot[lane], rot[lane + 32] = pos16, pos16
		rot = torch.as_tensor(rot, device=eng.device)
		if os.environ.get('NRM_DE_SPARSE_ORDER', 'residue') == 'residue':
			res = (k - c * ch - rot[key % 64]) % 16
			perm = torch.argsort(key * 16 + res, stable=True)
		else:
			perm = torch.argsort(key, stable=True)  # by chunk, then position; cells ascending inside (nonzero() listed them so)
		key_s, k_s = key[perm], k[perm]
		cnt = torch.bincount(key_s, minlength=nch * nslots)
		w = (cnt.view(nch, self.ngroups, 64).max(dim=2).values + 7) // 8 * 8  # longest list of every (chunk, 64 positions), in blocks of 8 entries
		w64 = w.flatten() * 64
		base = torch.cumsum(w64, 0) - w64
		start = torch.cumsum(cnt, 0) - cnt
		j = torch.arange(self.nnz, device=eng.device) - start[key_s]
		pos_e = base[key_s // 64] + ((j // 8) * 64 + key_s % 64) * 8 + j % 8  # 8 consecutive entries of a position side by side: one 16-byte load
		total = int(w64.sum())
		self.ell = torch.full((max(total, 8), ), ch, dtype=torch.int16, device=eng.device)  # padding: the record of zeros
		self.ell[pos_e] = (k_s - (key_s // nslots) * ch).to(torch.int16)
		self.vals = None
		if not self.binary:
			self.vals = torch.zeros((max(total, 8), ), dtype=torch.float64, device=eng.device)
			self.vals[pos_e] = vals[perm].to(torch.float64)
		self.base = base.contiguous()
		self.w = w.flatten().to(torch.int32).contiguous()
		self.padded = total


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


def design_stats(eng, lists, d_c, d_dci, rank, nx, nc):
	"""|x~_i|^2 and b_i of every design row from its entries (csrc/nrm_de_sparse.hip: k_design_stats) -- K1 would sweep n cells twice
	for rows that have a few hundred entries."""
	from . import engine as _engine
	torch = eng.torch
	ncu = nc if (rank > 0 and nc > 0) else 0
	ss = eng.zeros((_round_up(nx, ROW_TILE), ), torch.float64)
	coef = eng.zeros((nx, nc), torch.float64) if nc else eng.zeros((nx, 1), torch.float64)[:, :0]  # (covariates of rank 0: the coefficients stay zero)
	with _engine._Span(eng, 'design_stats'):
		_lib.check(eng.lib.nrm_design_stats(lists.row_ptr.data_ptr(), lists.cells.data_ptr(), 0 if lists.row_vals is None else lists.row_vals.data_ptr(),
											d_c.data_ptr() if ncu else 0, d_c.stride(0) if ncu else 0, ncu, d_dci.data_ptr() if ncu else 0, nx, ss.data_ptr(),
											coef.data_ptr() if ncu else 0, eng._stream()))
	return DesignRows(ss, coef)


def run(eng, d_x, lists, dy, d_c, d_dci, rank, nx, ny, n, nc, want_coef, flags=None):
	"""The design rows' statistics from their entries, the one-pass kernels on the expression rows.
	Returns (dot (nx_pad, ny_pad) fp64 with dot[i, y] = x~_i . y~_y, rx, ssy, coefy)."""
	rx = design_stats(eng, lists, d_c, d_dci, rank, nx, nc)
	dot, ssy, coefy = products(eng, lists, dy, d_c, d_dci, rank, rx.coef, nx, ny, n, nc, want_coef, False, flags)
	return dot, rx, ssy, coefy


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
	# the rows' products with the covariates and their sums of squares: the stream kernel of single=1, every cell "common"
	common = torch.empty((ncu + 1, ny), dtype=torch.float64, device=eng.device)
	code = getattr(eng, '_all_common', None)
	if code is None or code.numel() < n:
		code = eng._all_common = torch.full((n, ), _lib.NRM_S1_COMMON, dtype=torch.int32, device=eng.device)
	with _engine._Span(eng, 'row_sums'):
		_lib.check(eng.lib.nrm_single1_stream(d_y.data_ptr(), ycode, d_y.stride(0), d_c.data_ptr() if ncu else 0, d_c.stride(0) if ncu else n, ncu, code.data_ptr(), n, ny,
											  common.data_ptr(), common.data_ptr(), _round_up(ny, 8), eng._stream()))  # (no cell keeps its values: the last buffer is not written)
	with _engine._Span(eng, 'de_sparse'):
		_lib.check(eng.lib.nrm_de_sparse(d_y.data_ptr(), ycode, ny, n, d_y.stride(0), common.data_ptr(), ncu, d_dci.data_ptr() if ncu else 0, lists.ell.data_ptr(),
										 0 if lists.vals is None else lists.vals.data_ptr(), lists.base.data_ptr(), lists.w.data_ptr(), lists.sig.data_ptr(), lists.ngroups,
										 lists.slot2x.data_ptr(), bx.data_ptr() if ncu else 0, max(nc, 1), dot.data_ptr(), dot.stride(0), 1 if by_gene else 0, ssy.data_ptr(),
										 coefy.data_ptr() if (coefy is not None and ncu) else 0, 0 if flags is None else flags.data_ptr(), eng._stream()))
	return dot, ssy, coefy


assert __name__ != "__main__"
