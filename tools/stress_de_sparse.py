"""Sparse-design path against K1 + the Gram engines at shapes the unit tests do not reach: several passes of 1024 design rows, dozens of
chunks, cell counts off every grid, valued entries, fp64 rows."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd import engine as _engine
from normalisr_amd.association import inv_rank
eng = _engine.get_engine()
for nx, ny, n, nc, dens, dt, valued in ((2500, 2000, 120000, 7, 0.005, torch.float32, False), (1500, 1001, 70001, 3, 0.01, torch.float64, True),
										(4100, 515, 30003, 0, 0.02, torch.float32, True), (1030, 3000, 262144, 5, 0.002, torch.float32, False)):
	g = torch.Generator(device='cuda').manual_seed(nx)
	dc = np.vstack([np.random.default_rng(1).normal(size=(max(nc - 1, 0), n)), np.ones((1, n))])[:nc] if nc else np.zeros((0, n))
	dx = (torch.rand((nx, n), generator=g, device='cuda') < dens).to(dt)
	if valued:
		dx = dx * (0.5 + torch.rand((nx, n), generator=g, device='cuda', dtype=dt))
	dy = torch.randn((ny, n), generator=g, device='cuda', dtype=dt) + 4.0
	dy[:50] += 0.3 * dx[:50]
	dci, rank = inv_rank(dc @ dc.T) if nc else (np.zeros((0, 0)), 0)
	out = {}
	for mode in ('force', '0'):
		os.environ['NRM_DE_SPARSE'] = mode
		r = eng.association_single0(dx, dy, dc, dci, rank, 0, False, True, np.float64, want_rt=True, device_out=True)
		out[mode] = r
	s, d = out['force'], out['0']
	ok = d['p'] > 1e-280
	relp = float(((s['p'] - d['p']).abs() / d['p'])[ok].max())
	dr = float((torch.as_tensor(s['r']) - torch.as_tensor(d['r'])).abs().max())
	da = float(np.abs(s['alpha'] - d['alpha']).max() / max(np.abs(d['alpha']).max(), 1e-300)) if nc else 0.0
	print('%d x %d x %d, %d covariates, %s%s: max rel diff of P %.2e (smallest P %.1e), max |dr| %.2e, alpha %.2e' % (
		nx, ny, n, nc, str(dt).split('.')[-1], ', valued' if valued else '', relp, float(d['p'].min()), dr, da), flush=True)
	assert relp < 1e-6 and dr < 1e-9 and da < 1e-7
print('ok')
