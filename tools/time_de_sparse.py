"""de at BASELINE configs[3] size (1000 gRNAs x 15 000 genes x 50 000 cells fp32, 1 % of the design set, 5 covariates), resident: the
sparse-design path against K1 + K2 (NRM_DE_SPARSE=0), per-kernel times from the engine's spans.  python tools/time_de_sparse.py [density]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import engine as _engine
eng = _engine.get_engine()
dens = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
nx, ny, n, nc = 1000, 15000, 50000, 5
g = torch.Generator(device='cuda').manual_seed(4)
dc = torch.cat([torch.randn((nc - 1, n), generator=g, device='cuda'), torch.ones((1, n), device='cuda')]).cpu().numpy().astype(np.float64)
dx = (torch.rand((nx, n), generator=g, device='cuda') < dens).float()
dy = torch.randn((ny, n), generator=g, device='cuda')
from normalisr_amd.association import inv_rank
dci, rank = inv_rank(dc @ dc.T)
for mode in ('1', '0', '1'):
	os.environ['NRM_DE_SPARSE'] = mode
	state = {}
	def step():
		return eng.association_single0(dx, dy, dc, dci, rank, 0, False, False, np.float32, resident=True, state=state)
	r = step()
	try:
		eng.check_flags(r['flags'])
	except Exception as e:  # (experiment builds that skip work on purpose)
		print('flags:', type(e).__name__, str(e)[:100])
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(5):
		r = step()
	torch.cuda.synchronize()
	ms = (time.perf_counter() - t0) / 5 * 1e3
	eng.trace = []
	step(); torch.cuda.synchronize()
	split = {}
	for name, e0, e1 in eng.trace:
		split[name] = split.get(name, 0.0) + e0.elapsed_time(e1)
	eng.trace = None
	print('NRM_DE_SPARSE=%s: %.2f ms per step (same design each step: lists kept)  kernels: %s' % (mode, ms, ', '.join('%s %.2f' % kv for kv in split.items())), flush=True)
	if mode == '1':
		ps = r['p'].clone()
	elif mode == '0':
		pd = r['p']
		ok = pd > 1e-30
		print('largest relative difference of the P-values: %.2e' % float(((ps - pd).abs() / pd)[ok].max()))
t0 = time.perf_counter()
from normalisr_amd import de_sparse
for _ in range(5):
	l = de_sparse.Lists(eng, dx)
torch.cuda.synchronize()
print('building the lists: %.2f ms (%d entries, %d with padding)' % ((time.perf_counter() - t0) / 5 * 1e3, l.nnz, l.padded))
