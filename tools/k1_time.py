"""Time K1 with fixed-point output (nrm_residualize_q through the engine) for a shape and several covariate counts.
Usage: k1_time.py rows cells dtype(f32|f64) [nc ...]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from normalisr_amd.engine import get_engine
from normalisr_amd.association import _prepare_covariates
eng = get_engine()
rows, n = int(sys.argv[1]), int(sys.argv[2])
dt = torch.float64 if sys.argv[3] == 'f64' else torch.float32
ncs = [int(a) for a in sys.argv[4:]] or [0, 3, 8]
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn((rows, n), dtype=dt, device='cuda', generator=g)
rp = (rows + 127) // 128 * 128
rng = np.random.default_rng(1)
for nc in ncs:
	dc = np.vstack([rng.normal(size=(max(nc - 1, 0), n)), np.ones((1, n))])[:nc] if nc else np.zeros((0, n))
	dc64, dci, dcr = _prepare_covariates(dc)
	d_c, d_dci = eng.covariates(dc64, dci) if nc else (None, None)
	f = lambda: eng.residualize(x, d_c, d_dci, dcr, rows_pad=rp, nslices=6, keep_fp64=False)
	for _ in range(2):
		f()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(5):
		f()
	e1.record()
	torch.cuda.synchronize()
	ms = e0.elapsed_time(e1) / 5
	esz = x.element_size()
	print('%d x %d %s, %d covariates: %.3f ms  (input %.2f GB: %.2f TB/s counting 2 reads + 6 B/value written)' % (
		rows, n, sys.argv[3], nc, ms, rows * n * esz / 1e9, rows * n * (2 * esz + 6) / ms / 1e9))
