"""The lists csrc/nrm_de_sparse.hip reads a sparse design matrix from (normalisr_amd/de_sparse.py: Lists), built here on CPU tensors and
read back the way the kernel reads them: every entry of the design exactly once, in its chunk, on its slot; padding points at the record of
zeros; widths are multiples of 8; a dense design is refused before it is listed."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')


class _CpuEngine:
	def __init__(self):
		from normalisr_amd import _lib
		self.torch, self.device, self.lib = torch, torch.device('cpu'), _lib.load()


@pytest.mark.parametrize('binary,nx,n', [(True, 70, 9000), (False, 33, 4097), (True, 1100, 5000)])
def test_lists_hold_every_entry_once(binary, nx, n):
	from normalisr_amd import de_sparse
	rng = np.random.default_rng(nx)
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	if not binary:
		dx *= rng.uniform(0.5, 2.0, dx.shape)
	dx[3] = 0
	eng = _CpuEngine()
	ch = int(eng.lib.nrm_de_sparse_chunk())
	lst = de_sparse.Lists(eng, torch.as_tensor(dx))
	assert lst.ok and lst.nnz == np.count_nonzero(dx) and lst.binary == binary and (lst.vals is None) == binary
	ell, base, w, slot2x, sig = lst.ell.numpy(), lst.base.numpy(), lst.w.numpy(), lst.slot2x.numpy(), lst.sig.numpy()
	ng = lst.ngroups
	nch = (n + ch - 1) // ch
	assert w.shape == (nch * ng, ) and (w % 8 == 0).all() and slot2x.shape == (ng * 64, ) and sig.shape == (nch, ng * 64)
	assert np.array_equal(slot2x[:nx], np.arange(nx)) and (slot2x[nx:] == -1).all()
	for c in range(nch):  # the dealing of a chunk: a permutation inside every block of 1024 positions (one pass of the kernel)
		for lo in range(0, ng * 64, 1024):
			hi = min(ng * 64, lo + 1024)
			assert sorted(sig[c, lo:hi].tolist()) == list(range(lo, hi))
	back = np.zeros_like(dx)
	padded = 0
	for c in range(nch):
		for g in range(ng):
			b, wd = int(base[c * ng + g]), int(w[c * ng + g])
			blk = ell[b:b + wd * 64].reshape(wd // 8, 64, 8)  # [block of 8 entries][lane][entry]
			val = None if binary else lst.vals.numpy()[b:b + wd * 64].reshape(wd // 8, 64, 8)
			padded += wd * 64
			lens = []
			for lane in range(64):
				x = slot2x[sig[c, g * 64 + lane]]
				offs = blk[:, lane, :].ravel()
				real = offs != ch
				lens.append(int(real.sum()))
				if x < 0:
					assert not real.any()
					continue
				cells = c * ch + offs[real].astype(np.int64)
				assert cells.size == np.unique(cells).size and (cells < n).all()
				back[x, cells] += 1.0 if binary else val[:, lane, :].ravel()[real]
			assert lens == sorted(lens, reverse=True) or g * 64 % 1024 + 64 > 1024  # positions sorted by the number of entries in the chunk
	assert np.array_equal(back, dx)
	assert padded == lst.padded and (nx < 1000 or padded < 1.3 * lst.nnz)  # full groups: little padding beyond the rounding to blocks of 8


def test_a_dense_design_is_refused_before_it_is_listed():
	from normalisr_amd import de_sparse
	eng = _CpuEngine()
	dx = torch.ones((40, 3000), dtype=torch.float64)
	lst = de_sparse.Lists(eng, dx)
	assert not lst.ok and lst.nnz == 40 * 3000 and not hasattr(lst, 'ell')
	assert not de_sparse.Lists(eng, torch.zeros((40, 3000))).ok


def test_lists_are_kept_for_the_same_unmodified_tensor():
	from normalisr_amd import de_sparse
	eng = _CpuEngine()
	dx = torch.as_tensor((np.random.default_rng(2).random((40, 5000)) < 0.02).astype(np.float32))
	a = de_sparse.lists_for(eng, dx)
	assert de_sparse.lists_for(eng, dx) is a
	dx[3, 7] = 1.0  # an in-place write: analysed again
	b = de_sparse.lists_for(eng, dx)
	assert b is not a and b.nnz in (a.nnz, a.nnz + 1)
	c = de_sparse.lists_for(eng, dx.clone())  # another tensor with the same content: its own lists
	assert c is not b and c.nnz == b.nnz
