"""Differential expression: all genes against all groupings (mirror of the reference's de module)."""
import numpy as np

from .association import association_tests


def _varying_rows(dg):
	"""Rows with more than one distinct value (de.py:93 uses len(np.unique(x)) > 1; NaNs compare equal there).
	Column blocks of growing width, only over the rows still undecided: a row is settled by its first value that differs
	from its first entry, so a 0/1 incidence matrix costs far less than one pass."""
	rows, n = dg.shape
	varying = np.zeros(rows, dtype=bool)
	if n < 2:
		return varying
	isf = dg.dtype.kind == 'f'
	todo = np.arange(rows)
	c0, width = 1, 256
	while todo.size and c0 < n:
		c1 = min(n, c0 + width)
		whole = todo.size == rows
		blk = dg[:, c0:c1] if whole else dg[todo, c0:c1]
		first = dg[:, :1] if whole else dg[todo, :1]
		diff = blk != first
		if isf:
			diff &= ~(np.isnan(blk) & np.isnan(first))
		hit = diff.any(axis=1)
		varying[todo[hit]] = True
		todo = todo[~hit]
		c0, width = c1, width * 4
	return varying


def _finite_within(a, lo=None, hi=None):
	"""np.isfinite(a).all() and (a >= lo).all() and (a <= hi).all() (the reference's assertions on its results, de.py:124-131) from the
	array's minimum, maximum and NaN count: one threaded pass in the library for large arrays (numpy's min and max for small ones: a NaN
	anywhere makes both NaN and every comparison False) instead of up to five numpy passes with temporaries (the assertions were 12 ms
	of an 85 ms de call at BASELINE configs[3] size, 35 ms of a 50 ms normvar call)."""
	if a.size == 0:
		return True
	if a.size >= (1 << 18) and a.dtype in (np.float32, np.float64) and a.flags.c_contiguous:
		# one threaded pass in the library (numpy's NaN-propagating min / max of 400 MB of fp64 take 35 ms)
		from . import _lib
		out = np.empty(3)
		_lib.check(_lib.load().nrm_host_minmax(a.ctypes.data, _lib.NRM_F64 if a.dtype == np.float64 else _lib.NRM_F32, a.size, 0, out.ctypes.data))
		if out[2] > 0:
			return False
		mn, mx = float(out[0]), float(out[1])
	else:
		mn, mx = float(a.min()), float(a.max())
	return bool(np.isfinite(mn) and np.isfinite(mx) and (lo is None or mn >= lo) and (hi is None or mx <= hi))


def de(dg, dt, dc, bs=0, **ka):
	"""Differential expression of every gene (rows of dt) against every grouping (rows of dg) with
	covariates dc: Y = gamma*X + alpha*C + eps, H0: gamma = 0.  Same contract as reference de.py:4-132.

	Returns (P-values (n_group,n_gene), gamma (n_group,n_gene), alpha (n_group,n_gene,n_cov)|None,
	varg (n_group,), vart (n_group,n_gene)).  Keyword arguments single, lowmem, nth, dimreduce, tol, ...
	are passed to association_tests.  Single-valued grouping rows are not tested and come back with
	p=1, gamma=0, alpha=0, varg=0, vart=0 (de.py:107-122).
	"""
	dg0 = np.asarray(dg)
	dt = np.asarray(dt)
	dc = np.asarray(dc)
	if dg0.ndim != 2 or dt.ndim != 2 or dc.ndim != 2:
		raise ValueError('Incorrect dx/dy/dc size.')
	gid = _varying_rows(dg0)
	nt, nc, ng0 = dt.shape[0], dc.shape[0], dg0.shape[0]
	odt = dt.dtype if dt.dtype in (np.float32, np.float64) else np.dtype(np.float64)
	allrows = bool(gid.all())
	p, gam, alpha, varg, vart = association_tests(dg0 if allrows else dg0[gid], dt, dc, bsx=bs, bsy=bs, return_dot=False, **ka)
	if allrows and p.dtype == odt:  # nothing to re-inflate (de.py:107-122 is the identity then)
		P, G, A, VG = p, gam, alpha, varg
		VT = vart if np.ndim(vart) == 2 else np.broadcast_to(vart, (ng0, nt)).copy()  # single=0: (n_gene,) for every row
		assert _finite_within(P, 0, 1) and _finite_within(G) and _finite_within(VG, 0) and _finite_within(VT, 0)
		return (P, G, A, VG, VT)
	P = np.ones((ng0, nt), dtype=odt)
	P[gid] = p
	G = np.zeros((ng0, nt), dtype=odt)
	G[gid] = gam
	A = None
	if alpha is not None:
		A = np.zeros((ng0, nt, nc), dtype=odt)
		A[gid] = alpha
	VG = np.zeros((ng0, ), dtype=odt)
	VG[gid] = varg
	VT = np.zeros((ng0, nt), dtype=odt)
	VT[gid] = vart  # (n_gene,) for single=0 broadcasts to every tested row
	assert _finite_within(P, 0, 1) and _finite_within(G) and _finite_within(VG, 0) and _finite_within(VT, 0)
	return (P, G, A, VG, VT)


assert __name__ != "__main__"
