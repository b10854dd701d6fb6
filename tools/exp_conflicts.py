"""Experiment: what would conflict-free gathers buy k_de_sparse?  Hand-made lists at configs[3] size -- 1000 design rows x 13 chunks,
56 entries per row and chunk -- once with random cell offsets (what a real design gives: 2.4-way bank conflicts on average), once with
offsets whose bank quads differ across the 16 lanes ds_read_b128 serves together."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import engine as _engine, _lib
eng = _engine.get_engine()
lib = eng.lib
nx, ny, n = 1024, 15000, 50000
ch = int(lib.nrm_de_sparse_chunk())
nch, ng, w = (n + ch - 1) // ch, nx // 64, 56
g = torch.Generator(device='cuda').manual_seed(4)
dy = torch.randn((ny, n), generator=g, device='cuda')
groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
rot = np.zeros(64, dtype=np.int64)
for g16 in groups:
	for p, lane in enumerate(g16):
		rot[lane], rot[lane + 32] = p, p
rng = np.random.default_rng(0)
base = torch.arange(nch * ng, dtype=torch.int64, device='cuda') * (w * 64)
wt = torch.full((nch * ng, ), w, dtype=torch.int32, device='cuda')
sig = torch.arange(nx, dtype=torch.int32, device='cuda').repeat(nch, 1).contiguous()
slot2x = torch.arange(nx, dtype=torch.int32, device='cuda')
common = torch.zeros((1, ny), dtype=torch.float64, device='cuda')
dot = torch.empty((nx, 15104), dtype=torch.float64, device='cuda')
ssy = torch.empty((15104, ), dtype=torch.float64, device='cuda')
lastcells = n - (nch - 1) * ch
for name in ('random offsets', 'bank quads distinct within every 16 lanes'):
	off = np.empty((nch * ng, w // 8, 64, 8), dtype=np.int16)
	for j in range(w):
		quad = rng.integers(0, 16, (nch * ng, 64)) if name.startswith('random') else (rot[None, :] + j) % 16
		cells = rng.integers(0, lastcells // 16, (nch * ng, 64)) * 16 + quad
		off[:, j // 8, :, j % 8] = cells
	ell = torch.as_tensor(off.reshape(-1), device='cuda')
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	for it in range(7):
		if it == 2:
			ev[0].record()
		_lib.check(lib.nrm_de_sparse(dy.data_ptr(), 0, ny, n, n, common.data_ptr(), 0, 0, ell.data_ptr(), 0, base.data_ptr(), wt.data_ptr(), sig.data_ptr(), ng, slot2x.data_ptr(),
									 0, 1, dot.data_ptr(), dot.stride(0), 0, ssy.data_ptr(), 0, 0, 0, n, -1, 0.0, 0, eng._stream()))  # (no covariates: the gathers alone, sums from `common`)
	ev[1].record()
	torch.cuda.synchronize()
	print('%-45s %d entries per pass: %.3f ms' % (name, nch * ng * w * 64, ev[0].elapsed_time(ev[1]) / 5), flush=True)
