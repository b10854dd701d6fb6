"""The list format csrc/nrm_de_sparse.hip reads a sparse design matrix from (include/normalisr_hip.h: nrm_de_sparse), pinned on the CPU
by the torch builder kept as test infrastructure (tests/tools/lists_reference.py) and its reader: every entry of the design exactly once, in
its chunk, on its slot; padding points at the record of zeros; widths are multiples of 8.  The library builds the same lists with its
own kernels (csrc/nrm_design_lists.hip); tests/test_gpu_round5.py holds those to this reader and to this builder."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))


@pytest.mark.parametrize('binary,nx,n', [(True, 70, 9000), (False, 33, 4097), (True, 1100, 5000)])
def test_lists_hold_every_entry_once(binary, nx, n):
	import lists_reference as lr
	from normalisr_amd import _lib
	assert int(_lib.load().nrm_de_sparse_chunk()) == lr.CH
	rng = np.random.default_rng(nx)
	dx = (rng.random((nx, n)) < 0.02).astype(np.float64)
	if not binary:
		dx *= rng.uniform(0.5, 2.0, dx.shape)
	dx[3] = 0
	for order in ('cells', 'residue'):
		lst = lr.ReferenceLists(torch.as_tensor(dx), order=order)
		assert lst.ok and lst.nnz == np.count_nonzero(dx) and lst.binary == binary and (lst.vals is None) == binary
		back, padded = lr.decode(nx, n, binary, lst.ell.numpy(), None if binary else lst.vals.numpy(), lst.base.numpy(), lst.w.numpy(), lst.slot2x.numpy(),
								 lst.sig.numpy(), lst.ngroups)
		assert np.array_equal(back, dx)
		assert padded == lst.padded and (nx < 1000 or padded < 1.3 * lst.nnz)  # full groups: little padding beyond the rounding to blocks of 8


def test_a_dense_design_is_refused_before_it_is_listed():
	import lists_reference as lr
	lst = lr.ReferenceLists(torch.ones((40, 3000), dtype=torch.float64))
	assert not lst.ok and lst.nnz == 40 * 3000 and not hasattr(lst, 'ell')
	assert not lr.ReferenceLists(torch.zeros((40, 3000))).ok
