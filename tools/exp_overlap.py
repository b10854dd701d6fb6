"""Experiment: row sums (k_s1_stream) on a second stream beside k_de_sparse -- how much of the 0.64 ms hides?  (Timing only: the gather
kernel reads the sums in its epilogue, so a real implementation needs that epilogue as a kernel of its own.)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import engine as _engine, de_sparse, _lib
from normalisr_amd.association import inv_rank
eng = _engine.get_engine()
nx, ny, n, nc = 1000, 15000, 50000, 5
g = torch.Generator(device='cuda').manual_seed(4)
dy = torch.randn((ny, n), generator=g, device='cuda')
dc = np.vstack([np.random.default_rng(1).normal(size=(nc - 1, n)), np.ones((1, n))])
dx = (torch.rand((nx, n), generator=g, device='cuda') < 0.01).float()
lists = de_sparse.Lists(eng, dx)
dci, rank = inv_rank(dc @ dc.T)
d_c, d_dci = eng.covariates(dc, dci)
rx = de_sparse.design_stats(eng, lists, d_c, d_dci, rank, nx, nc)
common = torch.empty((nc + 1, ny), dtype=torch.float64, device='cuda')
code = torch.full((n, ), -2, dtype=torch.int32, device='cuda')
dot = torch.empty((1024, 15104), dtype=torch.float64, device='cuda')
ssy = torch.empty((15104, ), dtype=torch.float64, device='cuda')
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def sums(stream):
	_lib.check(eng.lib.nrm_single1_stream(dy.data_ptr(), 0, n, d_c.data_ptr(), d_c.stride(0), nc, code.data_ptr(), n, ny, common.data_ptr(), common.data_ptr(), 15000, stream))


def gather(stream):
	_lib.check(eng.lib.nrm_de_sparse(dy.data_ptr(), 0, ny, n, n, common.data_ptr(), nc, d_dci.data_ptr(), lists.ell.data_ptr(), 0, lists.base.data_ptr(), lists.w.data_ptr(),
									 lists.ngroups, lists.slot2x.data_ptr(), rx.coef.data_ptr(), nc, dot.data_ptr(), dot.stride(0), 0, ssy.data_ptr(), 0, 0, stream))


for mode in ('one stream', 'two streams, sums first', 'two streams, gathers first'):
	for it in range(8):
		if it == 3:
			torch.cuda.synchronize()
			t0 = time.perf_counter()
		if mode == 'one stream':
			sums(main.cuda_stream)
			gather(main.cuda_stream)
		else:
			side.wait_stream(main)
			if mode.endswith('sums first'):
				sums(side.cuda_stream)
				gather(main.cuda_stream)
			else:
				gather(main.cuda_stream)
				sums(side.cuda_stream)
			main.wait_stream(side)
	torch.cuda.synchronize()
	print('%s: %.3f ms' % (mode, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
