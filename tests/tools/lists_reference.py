"""Test infrastructure: the design-matrix lists of csrc/nrm_de_sparse.hip built with torch ops (the builder rounds 1-4 shipped; the library
now builds them with its own kernels, csrc/nrm_design_lists.hip) and a reader that takes lists apart the way the kernel reads them.  The
GPU tests hold the library's lists to both: every entry of the design exactly once, in its chunk, on its slot; padding points at the record
of zeros; widths are multiples of 8; the dealing of every chunk is the stable descending sort by entries."""
import numpy as np
import torch

MAX_DENSITY = 1.0 / 16
CH = 2048  # nrm_de_sparse_chunk()


def _round_up(v, m):
	return (v + m - 1) // m * m


class _Eng:
	def __init__(self, device):
		self.torch, self.device = torch, device


class ReferenceLists:
	"""The design matrix as the kernel reads it: for every chunk of cells and every 64 positions of the workgroup's lanes (the design rows
	are dealt to the positions chunk by chunk, sorted by their number of entries in the chunk), the entries of the 64 rows side by side,
	padded to the longest."""

	def __init__(self, d_x, order='cells'):
		eng = _Eng(d_x.device)
		torch = eng.torch
		nx, n = d_x.shape
		ch = CH
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
		if order == 'residue':
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


def decode(nx, n, binary, ell, vals, base, w, slot2x, sig, ngroups, ch=CH):
	"""The design matrix the lists stand for (numpy arrays in, (nx, n) fp64 out) and the padded entry count, asserting on the way what the
	kernel relies on: widths in blocks of 8, the dealing a permutation inside every block of 1024 positions, lists sorted by length, no
	cell twice, padding = the record of zeros."""
	nch = (n + ch - 1) // ch
	ng = ngroups
	assert w.shape == (nch * ng, ) and (w % 8 == 0).all() and slot2x.shape == (ng * 64, ) and sig.shape == (nch, ng * 64)
	assert np.array_equal(slot2x[:nx], np.arange(nx)) and (slot2x[nx:] == -1).all()
	for c in range(nch):
		for lo in range(0, ng * 64, 1024):
			hi = min(ng * 64, lo + 1024)
			assert sorted(sig[c, lo:hi].tolist()) == list(range(lo, hi))
	back = np.zeros((nx, n))
	padded = 0
	for c in range(nch):
		for g in range(ng):
			b, wd = int(base[c * ng + g]), int(w[c * ng + g])
			blk = ell[b:b + wd * 64].reshape(wd // 8, 64, 8)  # [block of 8 entries][lane][entry]
			val = None if binary else vals[b:b + wd * 64].reshape(wd // 8, 64, 8)
			padded += wd * 64
			lens = []
			for lane in range(64):
				x = slot2x[sig[c, g * 64 + lane]]
				offs = blk[:, lane, :].ravel()
				real = offs != ch
				lens.append(int(real.sum()))
				assert not real[lens[-1]:].any()  # a list's entries come first, its padding after them
				if x < 0:
					assert not real.any()
					continue
				cells = c * ch + offs[real].astype(np.int64)
				assert cells.size == np.unique(cells).size and (cells < n).all()
				back[x, cells] += 1.0 if binary else val[:, lane, :].ravel()[real]
			assert lens == sorted(lens, reverse=True)  # positions sorted by the number of entries in the chunk
			assert wd == (max(lens) + 7) // 8 * 8
	return back, padded
