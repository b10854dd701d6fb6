"""BASELINE configs[3] at FULL size (1000 gRNAs x 15 000 genes x 50 000 cells fp32, 5 covariates) numpy in -> numpy out, through the package (torch
engine) and through the library's torch-free whole-problem entries (NRM_HOST_ENTRY=1): wall time per call (the 3.2 GB upload included) and the largest
relative difference of the P-values between the two routes.  python tools/c_entry_c4.py [genes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import normalisr_amd.normalisr as norm
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
nx, n, nc = 1000, 50000, 5
rng = np.random.default_rng(3)
dt = rng.standard_normal((ny, n), dtype=np.float32)
dt += 9
dc = np.vstack([rng.standard_normal((nc - 1, n)), np.ones((1, n))])
dg4 = (rng.random((nx, n), dtype=np.float32) < 0.01).astype(np.float32)   # high MOI: 10 gRNAs per cell
dg1 = (rng.random((nx, n), dtype=np.float32) < 0.001).astype(np.float32)  # low MOI: one gRNA per cell
dt[:20] += 0.4 * dg4[0] + 0.4 * dg1[1]


def timed(f, reps=3):
	f()
	best = 1e9
	for _ in range(reps):
		t0 = time.perf_counter()
		out = f()
		best = min(best, time.perf_counter() - t0)
	return best, out


for name, dg, ka in (('de (single=0)', dg4, {}), ('de -m covariate (single=4)', dg4, dict(single=4)), ('de -m single (single=1)', dg1, dict(single=1))):
	res = {}
	for route in ('package', 'c_entry'):
		os.environ['NRM_HOST_ENTRY'] = '1' if route == 'c_entry' else '0'
		res[route] = timed(lambda: norm.de(dg, dt, dc, **ka))
	p0, p1 = res['package'][1][0].astype(np.float64), res['c_entry'][1][0].astype(np.float64)
	ok = p0 > 1e-30
	print('%-28s package %7.1f ms   C entry %7.1f ms   largest relative difference of P: %.2e   (%d x %d tests)' % (
		name, res['package'][0] * 1e3, res['c_entry'][0] * 1e3, float(np.max(np.abs(p1[ok] / p0[ok] - 1))), nx, ny), flush=True)
