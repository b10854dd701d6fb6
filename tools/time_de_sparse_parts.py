"""k_de_sparse alone at BASELINE configs[3] size with parts of its work taken away (which part costs what).
python tools/time_de_sparse_parts.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import engine as _engine, de_sparse
from normalisr_amd.association import inv_rank
eng = _engine.get_engine()
nx, ny, n = 1000, 15000, 50000
g = torch.Generator(device='cuda').manual_seed(4)
dy = torch.randn((ny, n), generator=g, device='cuda', dtype=torch.float64 if 'f64' in sys.argv else torch.float32)
for nc, dens in ((5, 0.01), (0, 0.01), (5, 0.00002), (0, 0.00002), (5, 0.002), (5, 0.03)):
	dc = np.vstack([np.random.default_rng(1).normal(size=(max(nc - 1, 0), n)), np.ones((1, n))])[:nc] if nc else np.zeros((0, n))
	dx = (torch.rand((nx, n), generator=g, device='cuda') < dens).float()
	lists = de_sparse.Lists(eng, dx)
	d_c, d_dci = (None, None)
	rank = 0
	if nc:
		dci, rank = inv_rank(dc @ dc.T)
		d_c, d_dci = eng.covariates(dc, dci)
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	for it in range(7):
		eng.trace = [] if it >= 2 else None
		if it == 2:
			acc = 0.0
		de_sparse.run(eng, dx, lists, dy, d_c, d_dci, rank, nx, ny, n, nc, False)
		if it >= 2:
			torch.cuda.synchronize()
			acc += sum(e0.elapsed_time(e1) for name, e0, e1 in eng.trace if name == 'de_sparse')
	eng.trace = None
	print('%d covariates, density %.5f (%d entries, %d padded): k_de_sparse %.3f ms' % (nc, dens, lists.nnz, lists.padded, acc / 5), flush=True)
